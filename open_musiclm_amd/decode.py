"""KV-cached autoregressive decoding for TokenConditionedTransformer (host side of csrc/decode.hip).

The reference's ``generate`` (open_musiclm.py:253-326) re-runs the whole causal forward for every sampled id.  Because
every op of the trunk is causal, the same logits come out of computing one new row per step against a key/value cache
and the two-row state of the causal depthwise convolution -- that is what :class:`CachedDecoder` does:

    dec = CachedDecoder(model, batch, max_rows)
    logits = dec.prefill(cond_ids + [sampled_so_far])      # batched forward over the prompt rows, fills the caches
    logits = dec.step(new_ids, k)                           # one row: ids sampled from `logits`, k = index of that id

``step`` is one call into libomlm_hip.so (32 kernel launches: embedding gather, 5 per layer, logit head) plus the
counter advance; the row index lives on the device so that a step can be captured into a HIP graph.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import engine, hip, ops
from .hip import call, ptr, stream_ptr

MAX_DECODE_BATCH = 8

_PTRS = ["Wq", "Wkv", "Wo", "W1p", "W2p", "attn_gamma", "q_scale", "k_scale", "ffin_gamma", "convw", "mid_gamma",
         "Kc", "Vc", "hist"]


class DecodeArgs(C.Structure):
    """Mirror of ``omlm_decode_args`` (include/omlm.h)."""
    _fields_ = ([(n, C.c_int) for n in ("B", "D", "H", "L", "F", "Fp", "Nmax", "w_dtype", "round_bf16", "nsplit")] +
                [("eps", C.c_float), ("scale", C.c_float), ("pos_dev", C.c_void_p)] +
                [(n, C.POINTER(C.c_void_p)) for n in _PTRS] +
                [("bias_table", C.c_void_p), ("bias_ld", C.c_int),
                 ("final_gamma", C.c_void_p), ("head_W", C.c_void_p), ("V1", C.c_int), ("ldV", C.c_int),
                 ("emb_table", C.c_void_p), ("emb_row_offset", C.c_longlong), ("emb_rows", C.c_longlong)] +
                [(n, C.c_void_p) for n in ("x", "x1", "q", "parts", "u", "logits", "advance_pos", "advance_step", "ln_parts")] +
                [("W1p_lo", C.POINTER(C.c_void_p)), ("W2p_lo", C.POINTER(C.c_void_p)), ("head_W_lo", C.c_void_p)] +
                [("splitk_ws", C.c_void_p), ("splitk_cnt", C.c_void_p)])


def max_batch(model, precision: str) -> int:
    """Samples one decode call holds: 8 on the VALU step kernels; 16 where the matrix-core kernels serve the model (16-bit weights, dim 1024,
    at most 16 heads, feed-forward width <= 3072: omlm_decode_step's own conditions)."""
    tr = model.transformer
    inner = getattr(tr.layers[0][2], "inner_dim", 0) if len(tr.layers) else 0
    wide = (precision in ("bf16", "fp16", "fp16ff") and tr.dim == 1024 and tr.heads * engine.DIM_HEAD <= 1024 and 0 < engine.ceil_to(inner, 64) <= 3072
            and os.environ.get("OMLM_DECODE_MFMA", "1") != "0" and os.environ.get("OMLM_DECODE_V1", "0") != "1")
    return 16 if wide else MAX_DECODE_BATCH


def supports(model, batch: int, precision: Optional[str] = None) -> bool:
    tr = model.transformer
    limit = max_batch(model, precision) if precision is not None else MAX_DECODE_BATCH
    return batch <= limit and tr.non_causal_prefix_size == 0


class CachedDecoder:
    def __init__(self, model, batch: int, max_rows: int, precision: str):
        if batch > max_batch(model, precision):
            raise ValueError(f"cached decode handles up to {max_batch(model, precision)} samples per call here; got {batch}")
        self.model, self.B, self.Nmax, self.precision = model, batch, int(max_rows), precision
        tr = model.transformer
        dev = model.start_tokens[0].device
        hip.require_gpu(model.start_tokens[0], "model parameters")
        self.pw = engine.prepared_weights(model, precision)
        self.T = self.pw.T
        # "fp16ff": the steps read the FF-in / FF-out / head weights as hi + lo planes and keep LayerNorm outputs and h1 un-rounded, like the
        # three-product forward of the batched path (omlm_decode_args::W1p_lo)
        self.planes = bool(self.pw.ff3)
        L, D, H = len(tr.layers), tr.dim, tr.heads
        F, Fp = self.pw.layers[0]["F"], self.pw.layers[0]["Fp"]
        self.L, self.D, self.H, self.F, self.Fp = L, D, H, F, Fp
        B, Nmax = self.B, self.Nmax
        f32 = dict(device=dev, dtype=torch.float32)
        self.Kc = [torch.zeros(B, Nmax, engine.DIM_HEAD, **f32) for _ in range(L)]
        self.Vc = [torch.zeros(B, Nmax, engine.DIM_HEAD, **f32) for _ in range(L)]
        self.hist = [torch.zeros(B, 2, 2 * Fp, **f32) for _ in range(L)]
        self.x, self.x1 = torch.empty(B, D, **f32), torch.empty(B, D, **f32)
        self.q = torch.empty(B, H * engine.DIM_HEAD, **f32)
        self.nsplit = (Nmax + 63) // 64
        self.parts = torch.zeros(B, self.nsplit, H, 66, **f32)                 # attention partials (max, sum, o[64])
        self.pos_dev = torch.zeros(1, device=dev, dtype=torch.int32)           # row index, kept on the device
        self.u = torch.empty(B, Fp, **f32)
        seq = model.token_sequences[-1]
        self.Q, self.V1 = seq.num_quantizers, seq.codebook_size + 1
        self.ldV = engine.ceil_to(self.V1, 8)
        self.logits = torch.zeros(B, self.ldV, **f32)
        self.codebook = seq.codebook_size
        self.emb = model.embeddings[-1].weight.detach()
        # learned absolute position embeddings (open_musiclm.py:134-136: row p of the LAST sequence's table is added to the embedding of
        # its p-th id): the single-row steps add the row of the id they embed
        self.pos_emb = (model.absolute_position_embeddings[-1].weight.detach()
                        if getattr(model, "use_absolute_position_embeddings", False) else None)
        # rel-pos table for every distance the cache can hold: [Nmax, ld] fp32 (row = i - j, column = head)
        self.table, _ = engine.relpos_forward(tr, Nmax, False)
        self.rows = 0                       # rows already in the caches == index of the next row
        self._keep = []                     # python references that keep the pointer arrays' targets alive
        a = DecodeArgs()
        a.B, a.D, a.H, a.L, a.F, a.Fp, a.Nmax, a.nsplit = B, D, H, L, F, Fp, Nmax, self.nsplit
        a.pos_dev = self.pos_dev.data_ptr()
        a.w_dtype = ops.dcode(self.T)                              # 0 fp32, 1 bf16, 2 fp16 (served by the library's fp16 copy)
        a.round_bf16 = 0 if self.T == torch.float32 else 1         # round activations to the 16-bit operand type like the batched path
        a.eps, a.scale = 1e-5, float(engine.ATTN_SCALE)

        def arr(tensors: Sequence[torch.Tensor]):
            ts = [t.detach() for t in tensors]
            for t in ts:
                hip.require_gpu(t, "decode operand")
                assert t.is_contiguous()
            self._keep.append(ts)
            out = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
            self._keep.append(out)
            return C.cast(out, C.POINTER(C.c_void_p))
        lay = self.pw.layers
        a.Wq, a.Wkv, a.Wo = arr([w["Wq"] for w in lay]), arr([w["Wkv"] for w in lay]), arr([w["Wo"] for w in lay])
        a.W1p, a.W2p = arr([w["W1p"] for w in lay]), arr([w["W2p"] for w in lay])
        # fp32 views of the (operand-dtype) taps / gamma: same values as the batched path uses (fp16ff: hi + lo planes)
        if self.planes:
            a.convw = arr([w["convw"].float() + w["convw_lo"].float() for w in lay])
            a.mid_gamma = arr([w["gamma_mid"].float() + w["gamma_mid_lo"].float() for w in lay])
            a.W1p_lo, a.W2p_lo = arr([w["W1p_lo"] for w in lay]), arr([w["W2p_lo"] for w in lay])
        else:
            a.convw, a.mid_gamma = arr([w["convw"].float() for w in lay]), arr([w["gamma_mid"].float() for w in lay])
        a.attn_gamma = arr([attn.norm.gamma for attn, _, _ in tr.layers])
        a.q_scale = arr([attn.q_scale for attn, _, _ in tr.layers])
        a.k_scale = arr([attn.k_scale for attn, _, _ in tr.layers])
        a.ffin_gamma = arr([ff.norm_in.gamma for _, _, ff in tr.layers])
        a.Kc, a.Vc, a.hist = arr(self.Kc), arr(self.Vc), arr(self.hist)
        a.bias_table = self.table.data_ptr() if self.table is not None else None
        a.bias_ld = self.table.shape[-1] if self.table is not None else 0
        a.final_gamma = tr.norm.gamma.detach().data_ptr()
        a.V1, a.ldV = self.V1, self.ldV
        a.emb_table, a.emb_rows = self.emb.data_ptr(), self.emb.shape[0]
        for n in ("x", "x1", "q", "parts", "u", "logits"):
            setattr(a, n, getattr(self, n).data_ptr())
        # per-workgroup LayerNorm partial sums of the batched step kernels (OMLM_DECODE_LN_PARTS(D, Fp) floats x 3 producers)
        self.ln_parts = torch.zeros(3 * max((a.D + 15) // 16, (a.Fp + 7) // 8) * 32, device=self.x.device)      # [partial][16 samples][2]
        a.ln_parts = self.ln_parts.data_ptr()
        # split-K scratch of the batched FF-out launch (OMLM_DECODE_SPLITK_FLOATS): slabs + one zeroed arrival counter per 16 output rows
        self.splitk_ws = torch.empty(4 * ((a.D + 15) // 16) * 256, device=self.x.device)
        self.splitk_cnt = torch.zeros((a.D + 15) // 16, dtype=torch.int32, device=self.x.device)
        a.splitk_ws, a.splitk_cnt = self.splitk_ws.data_ptr(), self.splitk_cnt.data_ptr()
        self.args = a

    # ---- prompt: the batched forward over all known rows, keeping what the single-row steps need ---------------------
    def prefill(self, all_token_ids: List[torch.Tensor]) -> torch.Tensor:
        """all_token_ids: the conditioning sequences followed by the ids sampled so far (may be empty).  Returns the
        [B, ldV] logits of the last row (they predict the next id) and leaves the caches filled for rows < N."""
        model, tr = self.model, self.model.transformer
        ids32, lens = engine.build_ids(model, all_token_ids)
        B, N = ids32.shape
        assert B == self.B and N <= self.Nmax, (B, N, self.B, self.Nmax)
        lay = engine.get_layout(model, B, lens, ids32.device, True)
        x = engine.embed_forward(model, ids32, lay)
        y, saved = engine.trunk_forward(tr, self.pw, x, None, B, N, True, False, keep_h1_lo_tail=self.planes)
        nseq = len(model.token_sequences)
        logits = engine.heads_forward(model, self.pw, y, lay, [s == nseq - 1 for s in range(nseq)])[-1]
        for l, sv in enumerate(saved["layers"]):
            self.Kc[l][:, :N].copy_(sv.k.view(B, N, -1))
            self.Vc[l][:, :N].copy_(sv.v.view(B, N, -1))
            h1 = sv.h1.view(B, N, -1)
            self.hist[l].zero_()
            take = min(2, N)
            self.hist[l][:, 2 - take:].copy_(h1[:, N - take:])
            if self.planes:                                  # the conv state of "fp16ff" is the un-rounded h1 = hi + lo
                self.hist[l][:, 2 - take:].add_(sv.h1_lo_tail[:, 2 - take:])
        self.rows = N
        self.pos_dev.fill_(N)
        return logits

    def step(self, new_ids: torch.Tensor, k: int) -> torch.Tensor:
        """Append the row of ``new_ids`` ([B] int64: the k-th sampled id of every sample, k counted from 0) and return the
        [B, ldV] logits that predict id k + 1 (quantizer head (k + 1) mod Q)."""
        if self.rows >= self.Nmax:
            raise RuntimeError(f"decode cache full ({self.Nmax} rows)")
        a = self.args
        a.emb_row_offset = self.codebook * (k % self.Q) if self.Q > 1 else 0
        head = self.pw.heads[-1][(k + 1) % self.Q]
        a.head_W = head.data_ptr()
        a.head_W_lo = self.pw.heads_lo[-1][(k + 1) % self.Q].data_ptr() if self.planes else None
        ids = new_ids.contiguous()
        assert ids.dtype == torch.int64 and ids.numel() == self.B
        a.emb_table = self.emb.data_ptr()
        if self.pos_emb is not None:
            # the new row = token embedding (with the reference's offset / clamp rule of dec_embed_kernel) + position row k, formed here
            rows = (ids + a.emb_row_offset).clamp_(0, self.emb.shape[0] - 1)
            self.x.copy_(self.emb[rows] + self.pos_emb[k])
            a.emb_table = None
        a.advance_pos, a.advance_step = self.pos_dev.data_ptr(), None        # the head kernel moves the row index on
        call("omlm_decode_step", C.addressof(a), ptr(ids), stream_ptr())
        self.rows += 1
        return self.logits


class SamplingLoop:
    """sample -> embed -> 6 layers -> head -> advance, per id, for one CachedDecoder.

    All per-step state (row index, step counter, uniforms, id history) is device resident, so the cycle of each quantizer
    phase can be captured into a HIP graph (``use_graph=True``: after one eager cycle per phase, the remaining ids are graph
    replays, ~1 host launch per id instead of ~32).  Measured on MI355X the step is GPU-bound (32 dependent kernels per id) and
    graph replay is no faster than back-to-back eager launches, so eager is the default; the graph path
    is kept (and tested) for hosts that cannot keep up."""

    def __init__(self, dec: CachedDecoder, first_logits: torch.Tensor, uniforms: torch.Tensor, n0: int, n_new: int, topk: int,
                 temperature: float, forbid_by_phase: Sequence[bool], use_graph: bool = True):
        self.dec, self.n0, self.n_new, self.topk, self.temperature = dec, n0, n_new, topk, float(temperature)
        self.forbid = [bool(f) for f in forbid_by_phase]
        dev = dec.logits.device
        assert uniforms.shape == (n_new, dec.B, dec.V1) and uniforms.dtype == torch.float32 and uniforms.is_contiguous()
        self.U = uniforms
        self.hist = torch.zeros(n_new, dec.B, device=dev, dtype=torch.long)
        self.cur = torch.zeros(dec.B, device=dev, dtype=torch.long)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32)
        dec.logits.copy_(first_logits)
        self.use_graph = use_graph
        self.graphs = {}

    def _cycle(self, k: int, with_decode: bool):
        """Sample id number k (global index in the predicted sequence) from dec.logits, then compute its row."""
        dec, a = self.dec, self.dec.args
        phase = k % dec.Q
        if not with_decode:
            call("omlm_sample_topk_gumbel_at", ptr(dec.logits), ptr(self.U), ptr(self.step_dev), ptr(self.cur), ptr(self.hist),
                 dec.B, dec.V1, dec.ldV, self.topk, self.temperature, int(self.forbid[phase]), stream_ptr())
            return
        # 32 launches per id: the sampler also gathers the embedding row of the id it picked (the step's first launch), and the
        # head kernel (the step's last) moves the row index and the sampler's step counter on
        call("omlm_sample_embed_at", ptr(dec.logits), ptr(self.U), ptr(self.step_dev), ptr(self.cur), ptr(self.hist),
             dec.B, dec.V1, dec.ldV, self.topk, self.temperature, int(self.forbid[phase]),
             dec.emb.data_ptr(), dec.codebook * phase if dec.Q > 1 else 0, dec.emb.shape[0], ptr(dec.x), dec.D, stream_ptr())
        if dec.pos_emb is not None:
            # + absolute position row of id k = n0 + (device step counter): device-side indexing, so a captured cycle stays valid
            dec.x.add_(dec.pos_emb.index_select(0, (self.step_dev + self.n0).long()))
        a.emb_table = None
        a.head_W = dec.pw.heads[-1][(k + 1) % dec.Q].data_ptr()
        a.head_W_lo = dec.pw.heads_lo[-1][(k + 1) % dec.Q].data_ptr() if dec.planes else None
        a.advance_pos, a.advance_step = dec.pos_dev.data_ptr(), self.step_dev.data_ptr()
        call("omlm_decode_step", C.addressof(a), ptr(self.cur), stream_ptr())

    def run(self) -> torch.Tensor:
        """Returns the [n_new, B] sampled ids."""
        dec, Q = self.dec, self.dec.Q
        for i in range(self.n_new):
            k, last = self.n0 + i, i == self.n_new - 1
            if last:
                self._cycle(k, False)
                break
            if dec.rows >= dec.Nmax:
                raise RuntimeError(f"decode cache full ({dec.Nmax} rows)")
            g = self.graphs.get(k % Q)
            if g is not None:
                g.replay()
            elif self.use_graph and i >= Q and (self.n_new - 1 - i) >= 2 * Q:
                # every phase has run eagerly once: capture this phase's cycle (capture records, it does not execute) ...
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._cycle(k, True)
                    self.graphs[k % Q] = g
                    g.replay()                                   # ... and run it for this id
                except Exception:                                # pragma: no cover - capture support depends on the runtime
                    self.use_graph = False
                    torch.cuda.synchronize()
                    self._cycle(k, True)
            else:
                self._cycle(k, True)
            dec.rows += 1
        return self.hist
