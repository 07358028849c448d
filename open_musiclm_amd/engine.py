"""Forward / backward engine of the TokenConditionedTransformer path on MI355X.

This is the host side of the hot path: it owns the activation buffers, sequences the HIP kernels of
``ops.py`` for the forward and the hand-written backward, and exposes them to autograd through two
``torch.autograd.Function``s (logits path, fused-loss path).  Gradients of parameters are accumulated by the
kernels directly into ``param.grad`` (GEMM epilogue ``C += ...`` / atomics), so there is no autograd
AccumulateGrad pass and, under data parallelism, the flat gradient buffer is ready for ONE all-reduce.

Reference arithmetic being reproduced (file:line into /root/reference/open_musiclm):
  transformer.py:385-424 (trunk), :214-333 (attention), :140-161 (feed-forward), :36-117 (rel-pos bias),
  open_musiclm.py:100-190 (embed / heads), :389-410 (loss).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .hip import require_gpu

ATTN_SCALE = 8.0
DIM_HEAD = 64

# precision modes -> GEMM / attention operand type in HBM.  "fp16": IEEE half operands on the same v_mfma_f32_32x32x16 rate as bf16
# (11 instead of 8 significand bits: meets the 1e-3 logits bar bf16 misses); its 5-bit exponent needs a loss scale in the
# backward (loss_scale below), divided out by the fused optimizer.
# "fp16ff" (round 5): "fp16" whose two ConvFeedForward linears run the FORWARD on hi/lo half planes (three products, omlm_gemm_planes16) with
# h1 / h2 un-rounded in between (omlm_ffmid_fwd_planes): those two GEMMs carry 86-88 % of the fp16 logits-error variance at depth 6 and 24
# (profiles/r05_error_budget.md).  Everything else, and the whole backward (which reads the hi planes), is "fp16".
_PRECISIONS = {"bf16": torch.bfloat16, "bf16x3": torch.float32, "fp16": torch.float16, "fp16ff": torch.float16}


def is_half(precision: str) -> bool:
    """The IEEE-half modes: loss scale in the backward, fp16 weight shadow in the optimizer, fp16 decode kernels."""
    return precision in ("fp16", "fp16ff")
_H16 = (torch.bfloat16, torch.float16)
_WT = os.environ.get("OMLM_WT", "0") == "1"
_FF_SAVE_GH = os.environ.get("OMLM_FF_SAVE_GH", "1") == "1"        # forward keeps the normalised GEGLU output for the backward (bf16 mode)
_WGRAD_GROUP = os.environ.get("OMLM_WGRAD_GROUP", "1") == "1"      # grouped weight-gradient GEMMs (0: one split-K GEMM per weight)
# The rel-pos-bias MLP (transformer.py:36-67: 3 x Linear(dim/2)+SiLU + Linear(heads) on N distances) is ~20 tiny fp32 kernels per
# step, 4..36 workgroups each: ~0.4 ms of pure launch-to-launch latency when they sit in the trunk's stream.  They depend on
# nothing but the weights (forward) / the finished d(table) (backward), so they run on a second HIP stream, forked and joined with
# events (captured as a parallel branch of the micro-step graph): OMLM_RELPOS_ASYNC=1.  OFF since round 4: with the trunk's kernels now
# filling the machine the branch contends for CUs more than it hides (same-box A/B, graph replay: 24.81 ms with it, 24.58 without), and
# a single stream has no cross-stream hazards at all.
_RELPOS_ASYNC = os.environ.get("OMLM_RELPOS_ASYNC", "0") == "1"
# The MLP's 0.3-GFLOP fp32 GEMMs stay on the register-staged fp32 kernel (csrc/gemm.hip gemm_kernel<float>): 42 us either way, and the
# hi/lo plane route (ops.operand_planes: extra buffers + a cache shared with the trunk's stream) buys nothing at this size.
# OMLM_RELPOS_PLANES=1: the plane route (what rounds 2-3 ran).
_RELPOS_PLANES = os.environ.get("OMLM_RELPOS_PLANES", "0") == "1"
# bf16 mode: d(LN output) leaves the input-gradient GEMMs as bf16 (fp32 accumulate, one rounding) instead of fp32 -- it is read once,
# by the LayerNorm backward, and every other GEMM operand of that mode is rounded the same way.  OMLM_BF16_LN_GRAD=0: fp32 as before.
_BF16_LN_GRAD = os.environ.get("OMLM_BF16_LN_GRAD", "1") == "1"
# 16-bit modes: the K/V projection's input gradient leaves its GEMM as a 16-bit tensor and is added to the residual gradient inside the
# attention LayerNorm's backward (ops.layernorm_bwd dres2) instead of by the GEMM's own fp32 read-add-write epilogue (0: that form)
_KV_DGRAD_H16 = os.environ.get("OMLM_KV_DGRAD_H16", "1") == "1"
# 16-bit modes: the attention's l2-norm + learned scale of q / k (transformer.py:265-271) is the epilogue of the projection GEMMs
# (ops.gemm_qknorm): q, k, v leave them in the operand type with the per-(row, head) norms in fp32 -- no fp32 q_raw / kv_raw, no qk_norm
# forward launch, and the backward derives xh = y / scale from the saved operand.  OMLM_QKNORM_FUSED=0: separate kernels on fp32 projections.
_QKNORM_FUSED = os.environ.get("OMLM_QKNORM_FUSED", "1") == "1"
# The rel-pos MLP as one fused forward launch + two backward launches (round 5; Hd = 256 / 512; exact fp32 matrix instruction, every
# gradient element owned by one lane: deterministic) instead of the layer-by-layer path (7 + 14 launches, three register-staged fp32 GEMMs
# each way, split-K atomics in the backward).  Measured same box: forward 93 us, backward 276 us, the step equal within noise (23.60 / 23.74
# vs 23.62 / 23.66 ms): a row-block workgroup has to stream both 1 MB weight matrices through its own L2 -> CU path (~2 MB at ~40 GB/s per
# CU), which bounds it whatever does the arithmetic (VALU FMAs and MFMAs measured the same).  On by default for the launch count and the
# determinism; OMLM_RELPOS_FUSED=0: the layer-by-layer path.
_RELPOS_FUSED = os.environ.get("OMLM_RELPOS_FUSED", "1") == "1"
_LN_COLSUM_GROUP = os.environ.get("OMLM_LN_COLSUM_GROUP", "1") == "1"    # 0: every LayerNorm backward sums its own d(gamma) partial rows


def ff_mx_enabled() -> bool:
    """precision "fp16ff": FF-in / FF-out as one half product + two fp8 correction products at twice the matrix rate (ops.gemm_mx16, round 6) instead
    of three half products (ops.gemm_planes16, round 5).  What the fp8 corrections leave in the logits: profiles/r06_error_budget_fp8corr.md.
    OMLM_FF_MX=0 (read when the weights are prepared): the three-product form."""
    return os.environ.get("OMLM_FF_MX", "1") == "1"
_SIDE_STREAMS: Dict[int, "torch.cuda.Stream"] = {}


def side_stream(dev: torch.device) -> "torch.cuda.Stream":
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(idx)
    if st is None:
        st = _SIDE_STREAMS[idx] = torch.cuda.Stream(device=dev)
    return st


def default_precision() -> str:
    p = os.environ.get("OMLM_PRECISION", "bf16")
    if p not in _PRECISIONS:
        raise ValueError(f"OMLM_PRECISION must be one of {list(_PRECISIONS)}, got {p}")
    return p


def loss_scale_initial() -> float:
    return float(os.environ.get("OMLM_FP16_LOSS_SCALE", "4096"))


def loss_scale_state(model) -> torch.Tensor:
    """The loss-scale block of a precision-"fp16" model: 5 floats on its device {scale, good steps since its last change, skipped
    steps, applied steps, scale the last finished step's gradients carried}.  It lives on the device because the training micro-step is a captured HIP graph: the backward multiplies the
    loss gradient by element 0 with a tensor op (LossFunction.backward), the fused optimizer divides it out, skips a step whose gradient
    norm overflowed, counts the applied steps (its Adam clock) and halves / doubles the scale -- all without a host read
    (FusedAdam.step -> omlm_loss_scale_update; GradScaler's rule: x0.5 on overflow, x2 after $OMLM_FP16_GROWTH_INTERVAL = 2000 good
    steps, within [1, 65536]).  $OMLM_FP16_DYNAMIC=0 keeps the scale at its initial value ($OMLM_FP16_LOSS_SCALE, default 4096: the
    non-target logit gradients of a 28 k-token batch are ~3e-8, half's smallest normal is 6e-5)."""
    dev = model.start_tokens[0].device
    st = model.__dict__.get("_omlm_ls_state")
    if st is None or st.device != dev:
        st = torch.tensor([loss_scale_initial(), 0.0, 0.0, 0.0, loss_scale_initial()], device=dev, dtype=torch.float32)
        model.__dict__["_omlm_ls_state"] = st
    return st


def loss_scale(precision: str, model=None) -> float:
    """Factor the backward of `precision` multiplies the loss gradient by: 1 except for "fp16" (gradients would fall below half's range),
    where it is the CURRENT value of the model's device-side block (reading it synchronises; without a model, or before its first
    optimizer step: the initial value).  Parameter gradients (param.grad, the optimizer's flat buffer) carry that factor until
    FusedAdam.step divides it out inside its kernel; anything else that reads param.grad in this mode divides by this (unscale_grads_)."""
    if not is_half(precision):
        return 1.0
    if model is None or "_omlm_ls_state" not in model.__dict__:
        return loss_scale_initial()
    return float(model.__dict__["_omlm_ls_state"][0].item())


_MODEL_OF: Dict[int, tuple] = {}               # id(parameter) -> (weakref(parameter), weakref(model))


def tag_parameters(model, precision: Optional[str]):
    """Mark the model's parameters with its precision mode: the fused optimizer picks its 16-bit shadow type and loss scale from it.
    The parameter -> model link (the optimizer finds the model's loss-scale block through it) lives in a module-level registry, NOT on
    the Parameter: Parameter.__reduce_ex__ pickles the tensor's __dict__, and a weakref there broke torch.save(model) / pickle /
    spawn-based multiprocessing of every TokenConditionedTransformer."""
    import weakref
    ref = weakref.ref(model)
    for p in model.parameters():
        p._omlm_precision = precision
        key = id(p)
        _MODEL_OF[key] = (weakref.ref(p, lambda _r, k=key: _MODEL_OF.pop(k, None)), ref)


def model_of(p) -> Optional[torch.nn.Module]:
    """The TokenConditionedTransformer that tagged parameter p (None if it is gone or p was never tagged)."""
    ent = _MODEL_OF.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1]()


@torch.no_grad()
def unscale_grads_(model, precision: Optional[str] = None):
    """Divide the loss scale out of every param.grad (fp16 mode, for consumers other than FusedAdam: torch optimizers, clip_grad_norm_,
    gradient logging).  Call once per optimizer step, before anything reads the gradients."""
    s = loss_scale(precision or getattr(model, "precision", None) or default_precision(), model)
    if s != 1.0:
        for p in model.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / s)


def ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def grad_of(p: torch.Tensor) -> torch.Tensor:
    """fp32 gradient buffer of a parameter (allocated zeroed on first use); kernels accumulate into it."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# ------------------------------------------------------------------------------------------------------
# static description of one forward call (positions, row maps) -- cached per (batch, lengths)
# ------------------------------------------------------------------------------------------------------
@dataclass
class SeqLayout:
    B: int
    N: int
    lens: Tuple[int, ...]                  # tokens per sequence (after eos append / last-token drop)
    starts: Tuple[int, ...]                # position of each sequence's start token
    n_out: Tuple[int, ...]                 # logit rows per sample per sequence
    seg: torch.Tensor                      # [N] int32
    posidx: torch.Tensor                   # [N] int32
    start_col: torch.Tensor                # [N] bool: position is a start token
    final_only: bool = False               # AR decode: only the last row of the last sequence is scored -> [B, ldV]
    head_maps: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = field(default_factory=dict)


def build_layout(B: int, lens: Sequence[int], quantizers: Sequence[int], device, final_rows_only: bool = False) -> SeqLayout:
    starts, pos = [], 0
    for l in lens:
        starts.append(pos)
        pos += l + 1
    N = pos
    seg = torch.empty(N, dtype=torch.int32)
    posidx = torch.zeros(N, dtype=torch.int32)
    start_col = torch.zeros(N, dtype=torch.bool)
    for s, (st, l) in enumerate(zip(starts, lens)):
        seg[st: st + l + 1] = s
        start_col[st] = True
        posidx[st + 1: st + l + 1] = torch.arange(l, dtype=torch.int32)
    n_out = [l for l in lens[:-1]] + [lens[-1] + 1]
    lay = SeqLayout(B, N, tuple(lens), tuple(starts), tuple(n_out), seg.to(device), posidx.to(device),
                    start_col.to(device), final_rows_only)
    if final_rows_only:
        s, st, n_s, Q = len(lens) - 1, starts[-1], n_out[-1], quantizers[-1]
        b = torch.arange(B)
        a_map = (b * N + st + n_s - 1).to(torch.int32)
        lay.head_maps[(s, (n_s - 1) % Q)] = (a_map.to(device), b.to(torch.int32).to(device), B)
        return lay
    for s, (st, n_s, Q) in enumerate(zip(starts, n_out, quantizers)):
        for qq in range(Q):
            js = torch.arange(qq, n_s, Q)
            if js.numel() == 0:
                continue
            b = torch.arange(B)[:, None]
            a_map = (b * N + st + js[None, :]).reshape(-1).to(torch.int32)
            c_map = (b * n_s + js[None, :]).reshape(-1).to(torch.int32)
            lay.head_maps[(s, qq)] = (a_map.to(device), c_map.to(device), int(a_map.numel()))
    return lay


# ------------------------------------------------------------------------------------------------------
# operand copies of the weights in the GEMM operand dtype / padded layouts
# ------------------------------------------------------------------------------------------------------
def h16_operand(w: torch.Tensor, T: torch.dtype) -> torch.Tensor:
    """16-bit copy (bf16 / fp16) of a weight as GEMM operand: the optimizer's shadow if it is current and of that type, else a fresh cast."""
    sh = getattr(w, "_omlm_bf16", None)
    if (sh is not None and sh.dtype == T and getattr(w, "_omlm_bf16_version", -1) == w._version and sh.device == w.device):
        return sh
    c = torch.empty(w.shape, dtype=T, device=w.device)
    ops.cast_pad(w.detach(), c, w.numel() // w.shape[-1], w.shape[-1], w.shape[-1], w.shape[-1])
    return c


class PreparedWeights:
    def __init__(self, model, precision: str, with_transposes: bool = False, persistent: bool = True):
        """persistent: the padded FF / tap / gamma images live in per-model buffers that EVERY persistent instance rewrites in place (the
        training step: one instance alive at a time, forward -> backward).  Instances that outlive a step (the no_grad cache below, a
        CachedDecoder) take persistent=False: private images, so a later training step cannot change the weights under them."""
        tr = model.transformer
        self.precision = precision
        self.T = _PRECISIONS[precision]
        self.ff3 = precision == "fp16ff"          # FF forward on hi/lo planes: lo planes of W1p / W2p / taps / gamma next to the usual images
        self.mx = self.ff3 and ff_mx_enabled()    # ... with the FF GEMMs' correction products on fp8 planes (W1p8 / W2p8: ops.Fp8Planes)
        quants = ops.QuantRowsGroup()
        T = self.T
        dev = model.start_tokens[0].device
        D = tr.dim
        self.layers = []
        # padded operand images are persistent per (layer, operand type): their pad rows / columns are zeroed once, a step only rewrites the
        # real entries (was: an 11 MB zero fill + ~10 small torch launches per layer per step)
        wbuf = model.__dict__.setdefault("_omlm_wbuf", {})
        packs = ops.CastPadGroup()                 # every layer's re-packs leave as ONE launch at the end of this constructor
        for li, (attn, _, ff) in enumerate(tr.layers):
            F = ff.inner_dim
            Fp = ceil_to(F, 64)      # 128-byte aligned bf16 rows for h1 / dh1 (2*Fp pitch) and whole k-tiles for FF-out
            w1 = ff.w_in.weight            # [2F, D]
            w2 = ff.w_out.weight           # [D, F]
            ent = {}
            if T == torch.float32:
                ent["Wq"], ent["Wkv"], ent["Wo"] = attn.to_q.weight, attn.to_kv.weight, attn.to_out[0].weight
            else:
                for name, w in (("Wq", attn.to_q.weight), ("Wkv", attn.to_kv.weight), ("Wo", attn.to_out[0].weight)):
                    ent[name] = h16_operand(w, T)
            bkey = (li, T, str(dev), F, D)
            if not persistent:
                W1p, W2p, convp, gammap = (torch.zeros(2 * Fp, D, dtype=T, device=dev), torch.zeros(D, Fp, dtype=T, device=dev),
                                           torch.zeros(3, 2 * Fp, dtype=T, device=dev), torch.zeros(Fp, dtype=T, device=dev))
            else:
                if bkey not in wbuf:
                    wbuf[bkey] = (torch.zeros(2 * Fp, D, dtype=T, device=dev), torch.zeros(D, Fp, dtype=T, device=dev),
                                  torch.zeros(3, 2 * Fp, dtype=T, device=dev), torch.zeros(Fp, dtype=T, device=dev))
                W1p, W2p, convp, gammap = wbuf[bkey]
            packs.add(w1.detach(), W1p, F, D, D, D)
            packs.add(w1.detach()[F:], W1p[Fp:], F, D, D, D)
            packs.add(w2.detach(), W2p, D, F, F, Fp)
            ent["W1p"], ent["W2p"], ent["F"], ent["Fp"] = W1p, W2p, F, Fp
            if self.ff3:
                lkey = bkey + ("lo",)
                if not persistent or lkey not in wbuf:
                    lo = (torch.zeros(2 * Fp, D, dtype=T, device=dev), torch.zeros(D, Fp, dtype=T, device=dev),
                          torch.zeros(3, 2 * Fp, dtype=T, device=dev), torch.zeros(Fp, dtype=T, device=dev))
                    if persistent:
                        wbuf[lkey] = lo
                else:
                    lo = wbuf[lkey]
                W1l, W2l, convl, gammal = lo
                if not (self.mx and persistent):       # (the training step of the fp8-corrected route never reads them; the cached decoder does)
                    packs.add(w1.detach(), W1l, F, D, D, D, lo=True)
                    packs.add(w1.detach()[F:], W1l[Fp:], F, D, D, D, lo=True)
                    packs.add(w2.detach(), W2l, D, F, F, Fp, lo=True)
                ent["W1p_lo"], ent["W2p_lo"], ent["convw_lo"], ent["gamma_mid_lo"] = W1l, W2l, convl, gammal
            if self.mx:
                mkey = bkey + ("mx",)
                if not persistent or mkey not in wbuf:
                    m8 = (ops.Fp8Planes(2 * Fp, D, dev), ops.Fp8Planes(D, Fp, dev))      # zero-filled once: pad rows / row tails stay zero
                    if persistent:
                        wbuf[mkey] = m8
                else:
                    m8 = wbuf[mkey]
                quants.add(w1.detach(), m8[0], 0, F, D, D)
                quants.add(w1.detach()[F:], m8[0], Fp, F, D, D)
                quants.add(w2.detach(), m8[1], 0, D, F, F)
                ent["W1p8"], ent["W2p8"] = m8
            if T in _H16 and with_transposes:
                # k-contiguous W^T copies: every input-gradient GEMM (dX = dY W) then runs in the fast NT form
                def wt(w, R, C, rows_pad=None, cols_pad=None):
                    t = torch.zeros(rows_pad or C, cols_pad or R, dtype=T, device=dev)
                    ops.transpose_cast(w.detach(), t, R, C, w.shape[-1], t.shape[-1])
                    return t
                HD = attn.to_q.weight.shape[0]
                ent["WqT"] = wt(attn.to_q.weight, HD, D)                      # [D, H*dh]
                ent["WkvT"] = wt(attn.to_kv.weight, attn.to_kv.weight.shape[0], D)
                ent["WoT"] = wt(attn.to_out[0].weight, D, HD)                 # [H*dh, D]
                W1pT = torch.zeros(D, 2 * Fp, dtype=T, device=dev)            # [D, 2Fp]: value half cols [0,F), gate half [Fp, Fp+F)
                ops.transpose_cast(w1.detach(), W1pT, F, D, D, 2 * Fp)
                ops.transpose_cast(w1.detach()[F:], W1pT[:, Fp:], F, D, D, 2 * Fp)
                ent["W1pT"] = W1pT
                ent["W2pT"] = wt(w2, D, F, rows_pad=Fp, cols_pad=D)           # [Fp, D], rows >= F zero
            # taps [3, 2Fp] (identity taps for plain FeedForward) and the padded LN gamma travel in the operand dtype: they are
            # re-read for every row, and as fp32 they were 70 % of the L2->L1 bytes of the conv-GEGLU-LN kernels
            cw = ff.conv_weight().detach().reshape(2 * F, 3)                 # reference ds_conv.weight [2F, 1, 3] -> tap-major [3, 2Fp]
            packs.add(cw, convp, F, 3, 3, 2 * Fp, transpose=True)
            packs.add(cw[F:], convp[:, Fp:], F, 3, 3, 2 * Fp, transpose=True)
            packs.add(ff.norm_mid.gamma.detach(), gammap, 1, F, F, Fp)
            ent["convw"], ent["gamma_mid"] = convp, gammap
            if self.ff3:
                packs.add(cw, ent["convw_lo"], F, 3, 3, 2 * Fp, transpose=True, lo=True)
                packs.add(cw[F:], ent["convw_lo"][:, Fp:], F, 3, 3, 2 * Fp, transpose=True, lo=True)
                packs.add(ff.norm_mid.gamma.detach(), ent["gamma_mid_lo"], 1, F, F, Fp, lo=True)
            cache = ff.__dict__.setdefault("_omlm_cmap", {})
            if (F, Fp, str(dev)) not in cache:            # static scatter map: uploaded once (no H2D inside graph capture)
                cm = torch.full((2 * Fp,), -1, dtype=torch.int32)
                cm[:F] = torch.arange(F, dtype=torch.int32)
                cm[Fp:Fp + F] = torch.arange(F, 2 * F, dtype=torch.int32)
                cache[(F, Fp, str(dev))] = cm.to(dev)
            ent["dW1_cmap"] = cache[(F, Fp, str(dev))]
            self.layers.append(ent)
        self.heads_lo = []
        if self.ff3:                               # the logit heads of "fp16ff" run on planes too (1.7 % of the FLOPs, 5-6 % of the error variance)
            for w in model.logit_weights:
                lo = torch.empty(w.shape, dtype=T, device=dev)
                packs.add(w.detach().reshape(-1, D), lo.view(-1, D), w.numel() // D, D, D, D, lo=True)
                self.heads_lo.append(lo)
        packs.flush()
        quants.flush()
        self.heads = []
        self.headsT = []
        for w in model.logit_weights:
            if T == torch.float32:
                self.heads.append(w)
            else:
                self.heads.append(h16_operand(w, T))
                if with_transposes:                                          # [Q, D, ldV], pad columns zero
                    Q, V1 = w.shape[0], w.shape[1]
                    ldV = ceil_to(V1, 8)
                    t = torch.zeros(Q, D, ldV, dtype=T, device=dev)
                    for qq in range(Q):
                        ops.transpose_cast(w.detach()[qq], t[qq], V1, D, D, ldV)
                    self.headsT.append(t)


def prepared_weights(model, precision: str) -> PreparedWeights:
    """Rebuilt on every grad-enabled forward (weights change every step; ~0.4 GB of traffic for musiclm_small);
    cached under no_grad (AR decoding) keyed on the parameters' autograd version counters."""
    key = (precision, sum(p._version for p in model.parameters()), tuple(p.data_ptr() for p in model.parameters()))
    cache = getattr(model, "_omlm_prepared", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    pw = PreparedWeights(model, precision, persistent=False)
    object.__setattr__(model, "_omlm_prepared", (key, pw))
    return pw


# ------------------------------------------------------------------------------------------------------
# relative position bias  ->  table [N, ldb] (row = i - j >= 0, column = head)
# ------------------------------------------------------------------------------------------------------
class RelposStepCache:
    """The rel-pos MLP once per OPTIMIZER step instead of once per micro-batch (round 6).  The reference recomputes
    the bias in every forward (transformer.py:402-405), but between two optimizer steps its weights do not change: the table of micro-batch
    2 .. k is the table of micro-batch 1, and the MLP's backward is linear in d(table), so the k micro-batches may add their d(table) into
    ONE buffer and the MLP runs backward once, on the sum.  Opt-in (SingleStageTrainer enables it when grad_accum_every > 1): the owner
    calls refresh() before the micro-batches of a step and flush() after the last one, before the gradient exchange.  The first training
    forward computes the table in line and is adopted (it fixes N); from then on the captured micro-step contains neither the MLP's forward
    nor its backward -- it reads `table` and accumulates into `dtable`, two persistent buffers.  Gradients equal the per-micro-batch form up
    to the order of fp32 additions."""

    def __init__(self, tr):
        self.tr, self.enabled, self.valid = tr, False, False
        self.n, self.table, self.dtable, self.saved = None, None, None, None

    def __getstate__(self):                          # (a pickled / deep-copied model starts without the buffers)
        return dict(tr=self.tr, enabled=False, valid=False, n=None, table=None, dtable=None, saved=None)

    def usable(self, n: int) -> bool:
        return self.enabled and self.valid and self.n == n

    def adopt(self, n: int, table: torch.Tensor, saved):
        self.n, self.table, self.saved = n, table, saved
        self.dtable = torch.zeros_like(table)
        self.valid = True

    def reset_accum(self):
        if self.dtable is not None:
            self.dtable.zero_()

    def refresh(self):
        """Start of an optimizer step: the table of the CURRENT weights into the persistent buffer, d(table) cleared."""
        if not self.enabled or self.n is None:
            return
        table, saved = relpos_forward(self.tr, self.n, True)
        self.table.copy_(table)
        self.saved = saved
        self.dtable.zero_()
        self.valid = True

    def flush(self):
        """After the last micro-batch: the MLP's backward on the accumulated d(table); the cache is stale from here on (the optimizer moves
        the weights), so forwards in between -- validation -- compute the table in line."""
        if self.enabled and self.valid and self.n is not None:
            relpos_backward(self.tr, self.n, self.saved, self.dtable)
        self.valid = False


def relpos_step_cache(tr) -> "RelposStepCache":
    c = tr.__dict__.get("_omlm_relpos_cache")
    if c is None:
        c = tr.__dict__["_omlm_relpos_cache"] = RelposStepCache(tr)
    return c


def relpos_forward(tr, n: int, save: bool):
    rp = tr.rel_pos_bias
    if rp is None:
        return None, None
    H = tr.heads
    ldb = ceil_to(H, 8)
    dev = tr.norm.gamma.device
    if tr.relative_position_bias_type == "t5":
        # 32-bucket embedding lookup (transformer.py:85-117); tiny, index plumbing done with torch
        from .transformer import t5_bucket_of_distance
        bucket = t5_bucket_of_distance(torch.arange(n, device=dev), rp.num_buckets, rp.max_distance)
        table = torch.zeros(n, ldb, device=dev, dtype=torch.float32)
        table[:, :H] = rp.relative_attention_bias.weight.detach()[bucket]
        return table, ("t5", bucket)
    lin = [rp.net[0][0], rp.net[1][0], rp.net[2][0], rp.net[3]]
    Hd = lin[0].weight.shape[0]
    if _RELPOS_FUSED and Hd in (256, 512) and H <= 16 and len(rp.net) == 4:
        # one launch for the whole MLP (csrc/optim_misc.hip relpos_mlp_fwd_kernel; round 5): a workgroup carries 8 rows through all layers
        saves = [torch.empty(n, Hd, device=dev) for _ in range(6)] if save else None
        table = torch.empty(n, ldb, device=dev)
        ops.relpos_mlp_fwd(lin[0].weight.detach().reshape(-1), lin[0].bias.detach(), lin[1].weight.detach(), lin[1].bias.detach(),
                           lin[2].weight.detach(), lin[2].bias.detach(), lin[3].weight.detach(), lin[3].bias.detach(), saves, table, n, Hd, H, ldb)
        return table, (("mlp_fused", saves) if save else None)
    pres, zs = [], []
    pre = torch.empty(n, Hd, device=dev)
    z = torch.empty(n, Hd, device=dev)
    ops.relpos_first_fwd(lin[0].weight.detach().reshape(-1), lin[0].bias.detach(), pre, z, n, Hd)
    pres.append(pre); zs.append(z)
    for k in (1, 2):
        # (the FORWARD GEMMs stay unsplit: a split-K sum is order-of-arrival, and 1e-7 of noise in the table is amplified by the 16-bit
        # roundings downstream -- run-to-run d(table) went from 2e-3 to 2e-2 in bf16 when they were split; the backward's GEMMs below do split)
        a = torch.empty(n, Hd, device=dev)
        ops.gemm(zs[-1], lin[k].weight.detach(), a, M=n, N=Hd, K=Hd, planes=_RELPOS_PLANES)
        pre = torch.empty(n, Hd, device=dev)
        z = torch.empty(n, Hd, device=dev)
        ops.bias_silu_fwd(a, lin[k].bias.detach(), pre, z, n, Hd)
        pres.append(pre); zs.append(z)
    a = torch.zeros(n, ldb, device=dev)
    ops.gemm(zs[-1], lin[3].weight.detach(), a, M=n, N=H, K=Hd, ldc=ldb)
    table = torch.empty(n, ldb, device=dev)
    ops.bias_add(a, lin[3].bias.detach(), table, n, H, ldb)
    return table, (("mlp", pres, zs) if save else None)


def relpos_backward(tr, n: int, saved, dtable: torch.Tensor):
    """dtable: [n, ldb] fp32 gradient of the bias table."""
    rp = tr.rel_pos_bias
    H = tr.heads
    ldb = dtable.shape[-1]
    dev = dtable.device
    if saved[0] == "t5":
        g = grad_of(rp.relative_attention_bias.weight)
        g.index_add_(0, saved[1], dtable[:, :H])
        return
    lin = [rp.net[0][0], rp.net[1][0], rp.net[2][0], rp.net[3]]
    Hd = lin[0].weight.shape[0]
    if saved[0] == "mlp_fused":
        scratch = torch.empty(3 * n * Hd, device=dev)
        grads = [grad_of(lin[0].weight).view(-1), grad_of(lin[0].bias), grad_of(lin[1].weight), grad_of(lin[1].bias),
                 grad_of(lin[2].weight), grad_of(lin[2].bias), grad_of(lin[3].weight), grad_of(lin[3].bias)]
        ops.relpos_mlp_bwd(dtable, lin[1].weight.detach(), lin[2].weight.detach(), lin[3].weight.detach(), saved[1], scratch, grads, n, Hd, H, ldb)
        return
    _, pres, zs = saved
    # last layer: table = z2 @ W3^T + b3
    ops.colsum_accumulate(dtable, grad_of(lin[3].bias), n, H, ldb)
    gw = grad_of(lin[3].weight)                                           # [H, Hd]
    ops.gemm(dtable, zs[2], gw, M=H, N=Hd, K=n, a_kmajor=True, b_kmajor=True, Cin=gw, lda=ldb)
    dz = torch.empty(n, Hd, device=dev)
    ops.gemm(dtable, lin[3].weight.detach(), dz, M=n, N=Hd, K=ldb, b_kmajor=True, lda=ldb, b_rows=H)
    for k in (2, 1):
        ds = torch.empty(n, Hd, device=dev)
        ops.silu_bwd(dz, pres[k], ds, n * Hd)
        ops.colsum_accumulate(ds, grad_of(lin[k].bias), n, Hd, Hd)
        gw = grad_of(lin[k].weight)
        ops.gemm(ds, zs[k - 1], gw, M=Hd, N=Hd, K=n, a_kmajor=True, b_kmajor=True, Cin=gw, planes=_RELPOS_PLANES)
        dz = torch.zeros(n, Hd, device=dev)        # zero-filled + Cin == C: lets the fp32 GEMM split its K over more workgroups (gemm.hip: `fine`)
        ops.gemm(ds, lin[k].weight.detach(), dz, M=n, N=Hd, K=Hd, b_kmajor=True, Cin=dz, planes=_RELPOS_PLANES)
    ds = torch.empty(n, Hd, device=dev)
    ops.silu_bwd(dz, pres[0], ds, n * Hd)
    ops.colsum_accumulate(ds, grad_of(lin[0].bias), n, Hd, Hd)
    ops.relpos_first_bwd(ds, grad_of(lin[0].weight).view(-1), n, Hd)


# ------------------------------------------------------------------------------------------------------
# trunk
# ------------------------------------------------------------------------------------------------------
class LayerSaved:
    __slots__ = ("x", "m1", "r1", "xn", "xc", "q_raw", "kv_raw", "q", "k", "v", "o", "lse", "abias",
                 "x1", "m2", "r2", "xn2", "h1", "h2", "m3", "r3", "seed", "p", "drop_bits", "gh", "h1_lo_tail")


def dropout_salt(tr, dev) -> torch.Tensor:
    """Per-forward dropout salt living in DEVICE memory: a counter bumped by a (graph-capturable) torch op on every
    training forward, snapshotted so that the backward of THIS forward regenerates the same masks.  Per-layer base
    seeds are fixed host constants; kernels combine both (omlm_ffmid_*: seed + *seed_dev * phi)."""
    st = tr.__dict__.get("_omlm_dropout")
    if st is None or st["counter"].device != dev:
        # the rank is mixed in: data-parallel replicas see different samples and must not share dropout masks
        rank = int(os.environ.get("RANK", "0"))
        g = torch.Generator().manual_seed((int(torch.initial_seed()) + 0x9E3779B1 * rank) & 0x7FFFFFFF)
        st = dict(counter=torch.zeros(1, dtype=torch.int64, device=dev),
                  seeds=[int(v) for v in torch.randint(1, 2 ** 62, (len(tr.layers),), generator=g)])
        tr.__dict__["_omlm_dropout"] = st
    st["counter"].add_(1)
    return st["counter"].clone(), st["seeds"]


def trunk_forward(tr, pw: PreparedWeights, x: torch.Tensor, keymask: Optional[torch.Tensor], B: int, N: int,
                  save: bool, training: bool, keep_h1_lo_tail: bool = False):
    """x: [B*N, D] fp32 (consumed as the layer-0 residual).  Returns (final LN output in operand dtype, saved).
    keep_h1_lo_tail ("fp16ff" prefill of the cached decoder): the lo plane of the last two h1 rows of every sample survives as sv.h1_lo_tail
    [B, 2, 2 Fp] fp32 (rows N-2, N-1; zeros where N < 2) -- the causal conv's state is the un-rounded h1."""
    T = pw.T
    dev = x.device
    M, D = x.shape
    H = tr.heads
    side = None
    if _RELPOS_ASYNC and tr.rel_pos_bias is not None and tr.relative_position_bias_type != "t5":
        main = torch.cuda.current_stream(dev)
        side = side_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            table, rp_saved = relpos_forward(tr, N, save)       # joined in front of the first attention kernel
    else:
        rc = tr.__dict__.get("_omlm_relpos_cache")
        if rc is not None and save and training and rc.usable(N):
            table, rp_saved = rc.table, ("cached", rc)            # (RelposStepCache: computed once for this optimizer step)
        else:
            table, rp_saved = relpos_forward(tr, N, save)
            # the FIRST training forward is adopted, once: a captured micro-step holds the addresses of the cache's buffers, so they are
            # never replaced (a stale cache -- validation between two optimizer steps -- computes in line and leaves the cache alone)
            if (rc is not None and rc.enabled and rc.n is None and save and training and rp_saved is not None and rp_saved[0] != "t5"
                    and not torch.cuda.is_current_stream_capturing()):
                rc.adopt(N, table, rp_saved)
                rp_saved = ("cached", rc)
    saved_layers: List[LayerSaved] = []
    abiases = None
    salt, seeds = (None, None)
    if training and any(float(ff.dropout_p) > 0 for _, _, ff in tr.layers):
        salt, seeds = dropout_salt(tr, dev)
    for li, ((attn, _, ff), w) in enumerate(zip(tr.layers, pw.layers)):
        sv = LayerSaved()
        F, Fp = w["F"], w["Fp"]
        m1 = torch.empty(M, device=dev); r1 = torch.empty(M, device=dev)
        xn = torch.empty(M, D, dtype=T, device=dev)
        xc = x if T == torch.float32 else torch.empty(M, D, dtype=T, device=dev)
        ops.layernorm_fwd(x, attn.norm.gamma.detach(), xn, None if T == torch.float32 else xc, m1, r1)
        q = torch.empty(M, H * DIM_HEAD, dtype=T, device=dev)
        k = torch.empty(M, DIM_HEAD, dtype=T, device=dev)
        v = torch.empty(M, DIM_HEAD, dtype=T, device=dev)
        fused_qk = T in _H16 and _QKNORM_FUSED
        if fused_qk:
            qn = torch.empty(M, H, device=dev)
            kn = torch.empty(M, device=dev)
            q_raw, kv_raw = qn, kn                                     # what the backward needs in place of the fp32 projections
            ops.gemm_qknorm(xn, w["Wq"], q, attn.q_scale.detach(), qn, H, M=M, N=H * DIM_HEAD, K=D)
            ops.gemm_qknorm(xc, w["Wkv"], k, attn.k_scale.detach(), kn, 1, M=M, N=2 * DIM_HEAD, K=D, C2=v, c2_col0=DIM_HEAD)   # K/V from the un-normalised residual (:228)
        else:
            q_raw = torch.empty(M, H * DIM_HEAD, device=dev)
            kv_raw = torch.empty(M, 2 * DIM_HEAD, device=dev)
            ops.gemm(xn, w["Wq"], q_raw, M=M, N=H * DIM_HEAD, K=D)
            ops.gemm(xc, w["Wkv"], kv_raw, M=M, N=2 * DIM_HEAD, K=D)      # K/V from the un-normalised residual (:228)
            ops.qk_norm_fwd(q_raw, kv_raw, attn.q_scale.detach(), attn.k_scale.detach(), q, k, v, H)
        o = torch.empty(M, H * DIM_HEAD, dtype=T, device=dev)
        lse = torch.empty(B, H, N, device=dev)
        if side is not None:
            torch.cuda.current_stream(dev).wait_stream(side)
            table.record_stream(torch.cuda.current_stream(dev))      # allocated under the side stream, read by the trunk's kernels
            side = None
        # the layer's bias table in the kernels' layout, with the fixed softmax reference point its scales allow (fp16: 15 octaves below the
        # bound, so that the probability numerators use half's normal range -- at the bound itself typical ones sat around 2^-12)
        # (round 6: the tables of all layers leave in ONE launch in front of the first attention kernel -- they differ only through the layers'
        # learned scales, and a launch per layer was six 14-us latency chains per step)
        if abiases is None:
            abiases = ops.AttnBias.group(table, N, H, dev, [a_.q_scale.detach() for a_, _, _ in tr.layers],
                                         [a_.k_scale.detach() for a_, _, _ in tr.layers], scale=ATTN_SCALE, half=T == torch.float16)
        abias = abiases[li]
        ops.attn_fwd(q, k, v, abias, keymask, o, lse, B, N, H, ATTN_SCALE)
        x1 = torch.empty(M, D, device=dev)
        ops.gemm(o, w["Wo"], x1, M=M, N=D, K=H * DIM_HEAD, Cin=x)
        # feed-forward
        m2 = torch.empty(M, device=dev); r2 = torch.empty(M, device=dev)
        xn2 = torch.empty(M, D, dtype=T, device=dev)
        h1 = torch.empty(M, 2 * Fp, dtype=T, device=dev)
        if pw.mx:
            # "fp16ff", round 6: the FF GEMMs as one half product + two fp8 corrections.  The LayerNorm (and, below, the conv-GEGLU-LN kernel)
            # leaves the hi plane -- what the fp16 backward keeps -- plus fp8 planes and row scales that die with the layer
            P1 = ops.Fp8Planes(M, D, dev, zero=False)
            ops.layernorm_fwd_mx(x1, ff.norm_in.gamma.detach(), xn2, P1, m2, r2)
            h1_lo = torch.empty(M, 2 * Fp, dtype=torch.uint8, device=dev)      # the lo plane as bf8 (e5m2) bytes: h1 = hi + lo to 2^-14
            ops.gemm_mx16(xn2, P1, w["W1p"], w["W1p8"], h1, h1_lo, M=M, N=2 * Fp, K=D)
        elif pw.ff3:
            # "fp16ff": LN output, h1 and h2 exist as hi/lo planes during this layer's forward; the lo planes die with the layer (their
            # consumers are the next launches of this stream), the hi planes are what the fp16 backward keeps
            xn2_lo = torch.empty(M, D, dtype=T, device=dev)
            ops.layernorm_fwd_planes(x1, ff.norm_in.gamma.detach(), xn2, xn2_lo, m2, r2)
            h1_lo = torch.empty(M, 2 * Fp, dtype=T, device=dev)
            ops.gemm_planes16(xn2, xn2_lo, w["W1p"], w["W1p_lo"], h1, h1_lo, M=M, N=2 * Fp, K=D)
        else:
            ops.layernorm_fwd(x1, ff.norm_in.gamma.detach(), xn2, None, m2, r2)
            ops.gemm(xn2, w["W1p"], h1, M=M, N=2 * Fp, K=D)
        h2 = torch.empty(M, Fp, dtype=T, device=dev)
        m3 = torch.empty(M, device=dev); r3 = torch.empty(M, device=dev)
        p = float(ff.dropout_p) if training else 0.0
        seed = seeds[li] if (p > 0 and seeds is not None) else 0
        # (the plane forward of "fp16ff" exists on the strip kernels only, which hand the keep mask over through drop_bits: present whenever p > 0)
        drop_bits = torch.empty(M, Fp // 8, dtype=torch.uint8, device=dev) if (p > 0 and (save or pw.ff3)) else None
        # the normalised GEGLU output is saved for the backward: its row-sum prepass reads it instead of recomputing conv + GELU, and
        # the fused second-generation backward (csrc/ffmid2.hip) requires it
        gh = torch.empty(M, Fp, dtype=T, device=dev) if (save and _FF_SAVE_GH) else None
        x2 = torch.empty(M, D, device=dev)
        if pw.ff3:
            if pw.mx:
                P2 = ops.Fp8Planes(M, Fp, dev, zero=False)
                ops.ffmid_fwd_mx(h1, h1_lo, w["convw"], w["convw_lo"], w["gamma_mid"], w["gamma_mid_lo"], h2, P2, m3, r3, N, F, Fp, p, seed,
                                 seed_dev=salt if p > 0 else None, drop_bits=drop_bits, gh=gh)
                ops.gemm_mx16(h2, P2, w["W2p"], w["W2p8"], x2, M=M, N=D, K=Fp, Cin=x1)
            else:
                h2_lo = torch.empty(M, Fp, dtype=T, device=dev)
                ops.ffmid_fwd_planes(h1, h1_lo, w["convw"], w["convw_lo"], w["gamma_mid"], w["gamma_mid_lo"], h2, h2_lo, m3, r3, N, F, Fp, p, seed,
                                     seed_dev=salt if p > 0 else None, drop_bits=drop_bits, gh=gh)
                ops.gemm_planes16(h2, h2_lo, w["W2p"], w["W2p_lo"], x2, M=M, N=D, K=Fp, Cin=x1)
                del xn2_lo, h2_lo
            if keep_h1_lo_tail and save:
                take = min(2, N)
                sv.h1_lo_tail = torch.zeros(B, 2, 2 * Fp, device=dev)
                tail = h1_lo.view(B, N, -1)[:, N - take:]
                if tail.dtype == torch.uint8:                              # bf8 bytes are the upper bytes of halves
                    tail = (tail.contiguous().to(torch.int16) << 8).view(torch.float16)
                sv.h1_lo_tail[:, 2 - take:].copy_(tail)
            del h1_lo
        else:
            ops.ffmid_fwd(h1, w["convw"], w["gamma_mid"], h2, m3, r3, N, F, Fp, p, seed, seed_dev=salt if p > 0 else None,
                          drop_bits=drop_bits, gh=gh)
            ops.gemm(h2, w["W2p"], x2, M=M, N=D, K=Fp, Cin=x1)
        if save:
            sv.x, sv.m1, sv.r1, sv.xn, sv.xc = x, m1, r1, xn, xc
            sv.q_raw, sv.kv_raw, sv.q, sv.k, sv.v, sv.o, sv.lse, sv.abias = q_raw, kv_raw, q, k, v, o, lse, abias
            sv.x1, sv.m2, sv.r2, sv.xn2, sv.h1, sv.h2, sv.m3, sv.r3, sv.seed, sv.p = x1, m2, r2, xn2, h1, h2, m3, r3, seed, p
            sv.drop_bits = drop_bits
            sv.gh = gh
            saved_layers.append(sv)
        x = x2
    if side is not None:                                         # depth 0: nothing consumed the table
        torch.cuda.current_stream(dev).wait_stream(side)
        table.record_stream(torch.cuda.current_stream(dev))
    mf = torch.empty(M, device=dev); rf = torch.empty(M, device=dev)
    y = torch.empty(M, D, dtype=T, device=dev)
    if pw.ff3:                                                   # the lo plane rides on the tensor until heads_forward has read it
        y_lo = torch.empty(M, D, dtype=T, device=dev)
        ops.layernorm_fwd_planes(x, tr.norm.gamma.detach(), y, y_lo, mf, rf)
        y._omlm_lo = y_lo
    else:
        ops.layernorm_fwd(x, tr.norm.gamma.detach(), y, None, mf, rf)
    saved = dict(layers=saved_layers, xL=x, mf=mf, rf=rf, table=table, rp=rp_saved, keymask=keymask, salt=salt) if save else None
    return y, saved


def trunk_backward(tr, pw: PreparedWeights, saved, dy: torch.Tensor, B: int, N: int, out_scale: float, head_wgrads=None):
    """dy: [M, D] fp32 gradient of the final LayerNorm output.  Accumulates parameter grads; returns
    d(trunk input) * out_scale (fp32).  out_scale carries the grad_shrink factor (utils.py:60-61)."""
    T = pw.T
    dev = dy.device
    M, D = dy.shape
    H = tr.heads
    table, keymask = saved["table"], saved["keymask"]
    rp_cached = saved["rp"] is not None and saved["rp"][0] == "cached"
    dtable = (saved["rp"][1].dtable if rp_cached else torch.zeros_like(table)) if table is not None else None
    nl = len(saved["layers"])
    rp_async = (_RELPOS_ASYNC and nl > 0 and dtable is not None and saved["rp"] is not None and saved["rp"][0] == "mlp")
    rp_side = None
    dres = torch.empty(M, D, device=dev)
    dres_c = dres if T == torch.float32 else torch.empty(M, D, dtype=T, device=dev)
    # the d(gamma) partial rows of every LayerNorm backward of this pass are summed by ONE launch at its end (13 launches of ~8 us before)
    cg = ops.ColsumGroup() if _LN_COLSUM_GROUP else None
    ops.layernorm_bwd(dy, saved["xL"], tr.norm.gamma.detach(), saved["mf"], saved["rf"], None, dres,
                      None if T == torch.float32 else dres_c, grad_of(tr.norm.gamma),
                      dx_scale=out_scale if nl == 0 else 1.0, defer=cg)
    ws = None
    # weight gradients have no consumer before the optimizer: in bf16 mode they are collected and issued as grouped launches of
    # full-K tiles (ops.WgradGroup) instead of 5 split-K GEMMs per layer; their operands stay alive until the flush
    wg = ops.WgradGroup() if (T in _H16 and _WGRAD_GROUP) else None
    for dYh, Xh, dWh, Mo, No in (head_wgrads or []):             # the logit heads' weight gradients (heads_backward) ride in the same launch
        if wg is not None:
            wg.add(dYh, Xh, dWh, M=Mo, N=No, K=dYh.shape[0])
        else:
            ops.gemm(dYh, Xh, dWh, M=Mo, N=No, K=dYh.shape[0], a_kmajor=True, b_kmajor=True, Cin=dWh)

    def wgrad(dY, X, dW, Mo, No, c_map=None):
        if wg is not None:
            wg.add(dY, X, dW, M=Mo, N=No, K=M, c_map=c_map)
            if len(wg.items) >= 40:
                wg.flush()
        else:
            ops.gemm(dY, X, dW, M=Mo, N=No, K=M, a_kmajor=True, b_kmajor=True, Cin=dW, c_map=c_map)

    for li in range(nl - 1, -1, -1):
        attn, _, ff = tr.layers[li]
        w, sv = pw.layers[li], saved["layers"][li]
        F, Fp = w["F"], w["Fp"]
        # ---- feed-forward block: x2 = x1 + h2 W2^T ----
        dh2 = torch.empty(M, Fp, dtype=T, device=dev)
        if "W2pT" in w: ops.gemm(dres_c, w["W2pT"], dh2, M=M, N=Fp, K=D)
        else: ops.gemm(dres_c, w["W2p"], dh2, M=M, N=Fp, K=D, b_kmajor=True)
        gW2 = grad_of(ff.w_out.weight)                                              # [D, F]
        wgrad(dres_c, sv.h2, gW2, D, F)
        if ws is None or ws.numel() < ops.ffmid_bwd_workspace_floats(F, Fp):
            ws = torch.empty(ops.ffmid_bwd_workspace_floats(F, Fp), device=dev)
        du = torch.empty(M, 2 * Fp, dtype=T, device=dev)
        dh1 = torch.empty(M, 2 * Fp, dtype=T, device=dev)
        gconv = grad_of(ff.conv_param()).view(-1) if ff.conv_param() is not None else None
        ops.ffmid_bwd(dh2, sv.h1, w["convw"], w["gamma_mid"], sv.m3, sv.r3, du, dh1,
                      grad_of(ff.norm_mid.gamma), gconv, ws, N, F, Fp, sv.p, sv.seed,
                      seed_dev=saved["salt"] if sv.p > 0 else None, drop_bits=sv.drop_bits, gh=sv.gh)
        del du, dh2
        dxn2 = torch.empty(M, D, dtype=T if _BF16_LN_GRAD else torch.float32, device=dev)     # consumed only by the LayerNorm backward
        if "W1pT" in w: ops.gemm(dh1, w["W1pT"], dxn2, M=M, N=D, K=2 * Fp)
        else: ops.gemm(dh1, w["W1p"], dxn2, M=M, N=D, K=2 * Fp, b_kmajor=True)
        gW1 = grad_of(ff.w_in.weight)                                               # [2F, D]
        wgrad(dh1, sv.xn2, gW1, 2 * Fp, D, c_map=w["dW1_cmap"])
        del dh1
        dx1 = torch.empty(M, D, device=dev)
        dx1_c = dx1 if T == torch.float32 else torch.empty(M, D, dtype=T, device=dev)
        ops.layernorm_bwd(dxn2, sv.x1, ff.norm_in.gamma.detach(), sv.m2, sv.r2, dres, dx1,
                          None if T == torch.float32 else dx1_c, grad_of(ff.norm_in.gamma), defer=cg)
        # ---- attention block: x1 = x + o Wo^T ----
        do = torch.empty(M, H * DIM_HEAD, dtype=T, device=dev)
        if "WoT" in w: ops.gemm(dx1_c, w["WoT"], do, M=M, N=H * DIM_HEAD, K=D)
        else: ops.gemm(dx1_c, w["Wo"], do, M=M, N=H * DIM_HEAD, K=D, b_kmajor=True)
        gWo = grad_of(attn.to_out[0].weight)
        wgrad(dx1_c, sv.o, gWo, D, H * DIM_HEAD)
        dq = torch.empty(M, H * DIM_HEAD, device=dev)
        dkv = torch.empty(2, M, DIM_HEAD, device=dev)          # one allocation: the dK/dV kernel zero-fills both with one fill
        dk, dv = dkv[0], dkv[1]
        delta = torch.empty(B, H, N, device=dev)
        ops.attn_bwd(sv.q, sv.k, sv.v, sv.abias, keymask, sv.o, do, sv.lse, delta, dq, dk, dv, dtable, B, N, H, ATTN_SCALE)
        if li == 0 and rp_async:
            # d(table) is complete: the MLP's backward leaves for the second stream while the trunk finishes this layer and the
            # grouped weight gradients (dtable / saved["rp"] stay referenced until the join at the end of this function)
            rp_side = side_stream(dev)
            for p_ in tr.rel_pos_bias.parameters():                # gradient buffers belong to the trunk's stream (allocated + zeroed
                grad_of(p_)                                         # here, before the fork), whoever accumulates into them
            rp_side.wait_stream(torch.cuda.current_stream(dev))
            dtable.record_stream(rp_side)
            for p_ in tr.rel_pos_bias.parameters():
                p_.grad.record_stream(rp_side)
            with torch.cuda.stream(rp_side):
                relpos_backward(tr, N, saved["rp"], dtable)
        dq_raw = torch.empty(M, H * DIM_HEAD, dtype=T, device=dev)
        dkv_raw = torch.empty(M, 2 * DIM_HEAD, dtype=T, device=dev)
        if T in _H16 and _QKNORM_FUSED:                                  # sv.q_raw / sv.kv_raw hold the norms [M, H] / [M]
            ops.qk_norm_bwd2(dq, dk, dv, sv.q, sv.k, sv.q_raw, sv.kv_raw, attn.q_scale.detach(), attn.k_scale.detach(),
                             dq_raw, dkv_raw, grad_of(attn.q_scale), grad_of(attn.k_scale), H)
        else:
            ops.qk_norm_bwd(dq, dk, dv, sv.q_raw, sv.kv_raw, attn.q_scale.detach(), attn.k_scale.detach(),
                            dq_raw, dkv_raw, grad_of(attn.q_scale), grad_of(attn.k_scale), H)
        dxn = torch.empty(M, D, dtype=T if _BF16_LN_GRAD else torch.float32, device=dev)
        kv16 = T in _H16 and _KV_DGRAD_H16
        tmp = torch.empty(M, D, dtype=T if kv16 else torch.float32, device=dev)      # kv16: the K/V term alone; else dx1 + the K/V term
        if "WqT" in w:
            ops.gemm(dq_raw, w["WqT"], dxn, M=M, N=D, K=H * DIM_HEAD)
            ops.gemm(dkv_raw, w["WkvT"], tmp, M=M, N=D, K=2 * DIM_HEAD, Cin=None if kv16 else dx1)
        else:
            ops.gemm(dq_raw, w["Wq"], dxn, M=M, N=D, K=H * DIM_HEAD, b_kmajor=True)
            ops.gemm(dkv_raw, w["Wkv"], tmp, M=M, N=D, K=2 * DIM_HEAD, b_kmajor=True, Cin=None if kv16 else dx1)
        gWq = grad_of(attn.to_q.weight)
        wgrad(dq_raw, sv.xn, gWq, H * DIM_HEAD, D)
        gWkv = grad_of(attn.to_kv.weight)
        wgrad(dkv_raw, sv.xc, gWkv, 2 * DIM_HEAD, D)
        dres = torch.empty(M, D, device=dev)
        dres_c = dres if T == torch.float32 else torch.empty(M, D, dtype=T, device=dev)
        last = li == 0
        ops.layernorm_bwd(dxn, sv.x, attn.norm.gamma.detach(), sv.m1, sv.r1, dx1 if kv16 else tmp, dres,
                          None if (T == torch.float32 or last) else dres_c, grad_of(attn.norm.gamma),
                          dx_scale=out_scale if last else 1.0, dres2=tmp if kv16 else None, defer=cg)
    if cg is not None:
        cg.flush()
    if wg is not None:
        wg.flush()
    if rp_side is not None:
        torch.cuda.current_stream(dev).wait_stream(rp_side)
    elif dtable is not None and saved["rp"] is not None and not rp_cached:      # (cached: RelposStepCache.flush runs it once per optimizer step)
        relpos_backward(tr, N, saved["rp"], dtable)
    return dres


# ------------------------------------------------------------------------------------------------------
# embedding gather, logit heads, loss
# ------------------------------------------------------------------------------------------------------
def build_ids(model, all_token_ids: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, List[int]]:
    """open_musiclm.py:116-130 + get_embeds (utils.py:133-134): flatten, add per-quantizer offsets, mark pads
    (id == -1 AFTER the offset add, exactly like the reference) and put the start-token marker (-2) in front."""
    parts, lens = [], []
    for seq, ids in zip(model.token_sequences, all_token_ids):
        ids = ids.reshape(ids.shape[0], -1)
        if seq.num_quantizers > 1:
            off = seq.codebook_size * (torch.arange(ids.shape[-1], device=ids.device) % seq.num_quantizers)
            ids = ids + off
        lens.append(ids.shape[-1])
        parts += [torch.full((ids.shape[0], 1), -2, device=ids.device, dtype=torch.int32), ids.to(torch.int32)]
    return torch.cat(parts, dim=1).contiguous(), lens


class PreparedIds:
    """ids32 [B, N] (start markers, offsets, conditioning masking already applied: ops.prepare_train_batch) + the per-sequence token counts,
    handed to run_forward in place of the list of id tensors."""
    __slots__ = ("ids32", "lens")

    def __init__(self, ids32: torch.Tensor, lens: Sequence[int]):
        self.ids32, self.lens = ids32, list(lens)


def get_layout(model, B: int, lens: Sequence[int], device, final_rows_only: bool) -> SeqLayout:
    cache = model.__dict__.setdefault("_omlm_layouts", {})
    key = (B, tuple(lens), str(device), final_rows_only)
    if key not in cache:
        if len(cache) > 64:
            cache.clear()
        cache[key] = build_layout(B, lens, [s.num_quantizers for s in model.token_sequences], device, final_rows_only)
    return cache[key]


def embed_forward(model, ids32: torch.Tensor, lay: SeqLayout) -> torch.Tensor:
    B, N = ids32.shape
    D = model.dim
    x = torch.empty(B * N, D, device=ids32.device)
    pos = [e.weight.detach() for e in model.absolute_position_embeddings] if model.use_absolute_position_embeddings else None
    ops.embed_fwd(ids32, lay.seg, lay.posidx, [e.weight.detach() for e in model.embeddings],
                  [s.detach() for s in model.start_tokens], pos, x.view(B, N, D))
    return x


def embed_backward(model, ids32: torch.Tensor, lay: SeqLayout, dx: torch.Tensor, alpha: float):
    B, N = ids32.shape
    dpos = [grad_of(e.weight) for e in model.absolute_position_embeddings] if model.use_absolute_position_embeddings else None
    ops.embed_bwd(ids32, lay.seg, lay.posidx, [grad_of(e.weight) for e in model.embeddings],
                  [grad_of(s) for s in model.start_tokens], dpos, dx.view(B, N, -1), alpha)


def heads_forward(model, pw: PreparedWeights, y: torch.Tensor, lay: SeqLayout, want: Sequence[bool]):
    """y: [B*N, D] operand dtype.  Returns per sequence a padded fp32 logits buffer [B*n_s, ldV] or None.
    (einsum 'q c d, b n q d -> b n q c' incl. the remainder positions, open_musiclm.py:163-186: the hidden
    position j of a sequence is scored by quantizer head j mod Q.)"""
    D = model.dim
    out = []
    y_lo = y.__dict__.pop("_omlm_lo", None) if pw.ff3 else None          # "fp16ff": trunk_forward left the final LayerNorm's lo plane on y
    for s, seq in enumerate(model.token_sequences):
        if not want[s]:
            out.append(None)
            continue
        V1 = seq.codebook_size + 1
        ldV = ceil_to(V1, 8)
        n_s = lay.n_out[s]
        # every row belongs to exactly one quantizer head and only the first V1 columns are ever read (CE, sampler, the views
        # handed back): no 147 MB zero fill per step
        buf = torch.empty(lay.B if lay.final_only else lay.B * n_s, ldV, device=y.device)
        for qq in range(seq.num_quantizers):
            ent = lay.head_maps.get((s, qq))
            if ent is None:
                continue
            a_map, c_map, rows = ent
            if y_lo is not None:
                ops.gemm_planes16(y, y_lo, pw.heads[s][qq], pw.heads_lo[s][qq], buf, M=rows, N=V1, K=D, a_map=a_map, c_map=c_map, ldc=ldV,
                                  a_rows=y.shape[0], b_rows=V1)
                continue
            ops.gemm(y, pw.heads[s][qq], buf, M=rows, N=V1, K=D, a_map=a_map, c_map=c_map, ldc=ldV,
                     a_rows=y.shape[0], b_rows=V1)
        out.append(buf)
    return out


def heads_backward(model, pw: PreparedWeights, y: torch.Tensor, lay: SeqLayout, dlogits: Sequence[Optional[torch.Tensor]],
                   deferred: Optional[list] = None):
    """dlogits[s]: [B*n_s, ldV] in the operand dtype with zeroed pad columns, or None.  Returns dy fp32 [B*N, D].
    deferred (16-bit operands): the heads' weight gradients are not launched here but appended as (dY, X, dW, M, N) for the trunk's
    grouped launch -- as their own row-mapped split-K GEMMs they ran at 158 TFLOP/s (3 x 128 us per step); gathered into contiguous
    rows they are 60 more full-K tiles of the ~900-tile group."""
    D = model.dim
    dy = torch.zeros(y.shape[0], D, device=y.device)
    for s, seq in enumerate(model.token_sequences):
        dl = dlogits[s]
        if dl is None:
            continue
        V1 = seq.codebook_size + 1
        ldV = dl.shape[-1]
        gW = grad_of(model.logit_weights[s])                                   # [Q, V1, D]
        for qq in range(seq.num_quantizers):
            ent = lay.head_maps.get((s, qq))
            if ent is None:
                continue
            a_map, c_map, rows = ent
            # dy[rows] = dlogits[rows] @ W_q          (each hidden row belongs to exactly one head)
            if pw.headsT:
                ops.gemm(dl, pw.headsT[s][qq], dy, M=rows, N=D, K=ldV, a_map=c_map, c_map=a_map, a_rows=dl.shape[0])
            else:
                ops.gemm(dl, pw.heads[s][qq], dy, M=rows, N=D, K=ldV, b_kmajor=True, a_map=c_map, c_map=a_map,
                         a_rows=dl.shape[0], b_rows=V1)
            # dW_q += dlogits[rows]^T @ y[rows]
            if deferred is not None and rows >= 1024:
                # gathered into contiguous rows, zero-padded to whole 64-row k-tiles (the grouped launch then keeps its SGPR-offset DMA form)
                rp = ceil_to(rows, 64)
                gd = torch.empty(rp, dl.shape[-1], dtype=dl.dtype, device=dl.device)
                gy = torch.empty(rp, y.shape[-1], dtype=y.dtype, device=y.device)
                torch.index_select(dl, 0, c_map, out=gd[:rows])
                torch.index_select(y, 0, a_map, out=gy[:rows])
                if rp > rows:
                    gd[rows:].zero_(); gy[rows:].zero_()
                deferred.append((gd, gy, gW[qq], V1, D))
                continue
            ops.gemm(dl, y, gW[qq], M=V1, N=D, K=rows, a_kmajor=True, b_kmajor=True, a_map=c_map, b_map=a_map,
                     Cin=gW[qq], lda=ldV, a_rows=dl.shape[0], b_rows=y.shape[0])
    return dy


class ForwardState:
    """Everything one forward leaves behind for its backward."""
    __slots__ = ("model", "pw", "ids32", "lay", "trunk", "y", "B", "N", "logits", "labels", "lse_rows", "coefs")


def run_forward(model, all_token_ids, self_attn_mask, only_final: bool, save: bool, precision: str,
                want: Optional[Sequence[bool]] = None, final_rows_only: bool = False):
    tr = model.transformer
    if tr.non_causal_prefix_size != 0:
        raise NotImplementedError("non_causal_prefix_size > 0 is not supported by the MI355X attention kernel "
                                  "(every shipped config uses 0)")
    require_gpu(model.start_tokens[0], "model parameters")
    ops.planes_begin()                       # bf16x3 operand planes live for this forward (+ its backward) only: ops.operand_planes
    if isinstance(all_token_ids, PreparedIds):
        ids32, lens = all_token_ids.ids32, all_token_ids.lens
    else:
        ids32, lens = build_ids(model, all_token_ids)
    require_gpu(ids32, "token ids")
    B, N = ids32.shape
    lay = get_layout(model, B, lens, ids32.device, final_rows_only)
    # k-contiguous W^T copies for the input-gradient GEMMs were worth 795 vs 644 TFLOP/s while the k-major GEMM path stalled on
    # its own DMA (gemm.hip: dma_issue); with that fixed, dX = dY W reads W k-major at the same rate (795 vs 778 TFLOP/s on the FF
    # shape) and the per-step transposes (183 MB written, 52 launches) are skipped.  OMLM_WT=1 brings them back (A/B).
    pw = prepared_weights(model, precision) if not save else PreparedWeights(model, precision, with_transposes=_WT)
    keymask = None
    if self_attn_mask is not None:
        assert self_attn_mask.shape == (B, N), f"self_attn_mask must be [{B}, {N}]"
        keymask = self_attn_mask.to(torch.uint8).contiguous()
    x = embed_forward(model, ids32, lay)
    y, tsaved = trunk_forward(tr, pw, x, keymask, B, N, save, model.training)
    nseq = len(model.token_sequences)
    if want is None:
        want = [(not only_final) or s == nseq - 1 for s in range(nseq)]
    logits = heads_forward(model, pw, y, lay, want)
    st = None
    if save:
        st = ForwardState()
        st.model, st.pw, st.ids32, st.lay, st.trunk, st.y, st.B, st.N, st.logits = model, pw, ids32, lay, tsaved, y, B, N, logits
    else:
        ops.planes_end()
    return logits, lay, st


def run_backward(st: ForwardState, dlogits: Sequence[Optional[torch.Tensor]]):
    model = st.model
    deferred = [] if (st.pw.T in _H16 and _WGRAD_GROUP) else None
    dy = heads_backward(model, st.pw, st.y, st.lay, dlogits, deferred)
    alpha = float(model.transformer.grad_shrink_alpha)
    dx = trunk_backward(model.transformer, st.pw, st.trunk, dy, st.B, st.N, out_scale=alpha, head_wgrads=deferred)
    embed_backward(model, st.ids32, st.lay, dx, 1.0)
    ops.planes_end()


def logits_views(model, lay: SeqLayout, bufs):
    out = []
    for s, (seq, buf) in enumerate(zip(model.token_sequences, bufs)):
        if buf is None:
            out.append(None)
        else:
            out.append(buf.view(lay.B, -1, buf.shape[-1])[:, :, : seq.codebook_size + 1])
    return out


class LogitsFunction(torch.autograd.Function):
    """TokenConditionedTransformer.forward as one autograd node (open_musiclm.py:100-190)."""

    @staticmethod
    def forward(ctx, model, all_token_ids, self_attn_mask, only_final, precision, *params):
        bufs, lay, st = run_forward(model, all_token_ids, self_attn_mask, only_final, True, precision)
        ctx.set_materialize_grads(False)          # unused logits -> None grads -> their head GEMMs are skipped
        ctx.st = st
        ctx.nparams = len(params)
        ctx.ls = loss_scale_state(model) if is_half(precision) else None
        views = logits_views(model, lay, bufs)
        ctx.present = [v is not None for v in views]
        return tuple(v for v in views if v is not None)

    @staticmethod
    def backward(ctx, *grads):
        st = ctx.st
        T = st.pw.T
        gi = iter(grads)
        dl = []
        for s, (seq, present) in enumerate(zip(st.model.token_sequences, ctx.present)):
            g = next(gi) if present else None
            if g is None:
                dl.append(None)
                continue
            V1 = seq.codebook_size + 1
            ldV = ceil_to(V1, 8)
            g2 = g.reshape(-1, V1).to(torch.float32).contiguous()
            if ctx.ls is not None:
                g2 = g2 * ctx.ls[0:1]                                # fp16: the device-side loss scale, removed by the optimizer
            d = torch.empty(g2.shape[0], ldV, dtype=T, device=g2.device)
            ops.cast_pad(g2, d, g2.shape[0], V1, V1, ldV)
            dl.append(d)
        run_backward(st, dl)
        ctx.st = None
        return (None,) * (5 + ctx.nparams)


class LossFunction(torch.autograd.Function):
    """Wrapper.forward(return_loss=True) (open_musiclm.py:378-410) fused: logits + cross entropy.

    Returns (loss, *logits) with the logits marked non-differentiable; the backward regenerates
    softmax - onehot from the saved logits / row lse directly in the GEMM operand dtype."""

    @staticmethod
    def forward(ctx, model, all_token_ids, labels, self_attn_mask, loss_weights, ignore_negative, precision, all_logits, *params):
        nseq = len(model.token_sequences)
        bufs, lay, st = run_forward(model, all_token_ids, self_attn_mask, False, True, precision,
                                    want=[bool(all_logits) or w > 0 for w in loss_weights])
        dev = next(b for b in bufs if b is not None).device
        total = 0
        total_dev = None                       # device-side part of the normaliser: non-ignored labels of padded sequences
        nll = torch.zeros(nseq, device=dev)
        st.labels, st.lse_rows, st.coefs = [], [], []
        for s, (seq, buf, lb, w) in enumerate(zip(model.token_sequences, bufs, labels, loss_weights)):
            if w > 0:
                lb32 = lb.reshape(-1).to(torch.int32).contiguous()
                assert lb32.numel() == buf.shape[0], (lb32.numel(), buf.shape)
                lse_rows = torch.empty(buf.shape[0], device=dev)
                ops.ce_fwd(buf, lb32, lse_rows, nll[s:s + 1], seq.codebook_size + 1)
                if ignore_negative[s]:
                    cnt = (lb32 >= 0).sum().to(torch.float32)
                    total_dev = cnt if total_dev is None else total_dev + cnt
                else:
                    total += lb32.numel()
                st.labels.append(lb32); st.lse_rows.append(lse_rows)
            else:
                st.labels.append(None); st.lse_rows.append(None)
        # loss = sum_s w_s * nll_sum_s / total   (== sum_s mean_s * n_s * w_s / sum n_s, :407-410); python-scalar
        # multiplies only: nothing is uploaded from the host here (graph-capture safe)
        loss = None
        if total_dev is None:
            for s, w in enumerate(loss_weights):
                if w > 0:
                    term = nll[s] * (float(w) / float(total))
                    loss = term if loss is None else loss + term
            st.coefs = [float(w) / float(total) if w > 0 else 0.0 for w in loss_weights]
            ctx.inv_total = None
        else:
            inv_total = 1.0 / (total_dev + float(total))             # device scalar: the count depends on the batch
            for s, w in enumerate(loss_weights):
                if w > 0:
                    term = nll[s] * float(w)
                    loss = term if loss is None else loss + term
            loss = loss * inv_total
            st.coefs = [float(w) if w > 0 else 0.0 for w in loss_weights]
            ctx.inv_total = inv_total
        ctx.st = st
        ctx.nparams = len(params)
        ctx.ls = loss_scale_state(model) if is_half(precision) else None
        views = logits_views(model, lay, bufs)
        ctx.mark_non_differentiable(*[v for v in views if v is not None])
        return (loss, *views)

    @staticmethod
    def backward(ctx, gloss, *unused):
        st = ctx.st
        T = st.pw.T
        g = gloss.reshape(1).to(torch.float32).contiguous()
        if ctx.inv_total is not None:
            g = (g * ctx.inv_total).reshape(1).contiguous()
        if ctx.ls is not None:
            g = (g * ctx.ls[0:1]).contiguous()                       # fp16: the device-side loss scale (loss_scale_state), removed by the optimizer
        dl = []
        for s, seq in enumerate(st.model.token_sequences):
            if st.labels[s] is None:
                dl.append(None)
                continue
            buf = st.logits[s]
            d = torch.empty(buf.shape[0], buf.shape[1], dtype=T, device=buf.device)
            ops.ce_bwd(buf, st.labels[s], st.lse_rows[s], g, st.coefs[s], d, seq.codebook_size + 1)
            dl.append(d)
        run_backward(st, dl)
        ctx.st = None
        return (None,) * (8 + ctx.nparams)
