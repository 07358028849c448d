"""Optimizer factory with the reference's signature (reference open_musiclm/optimizer.py:10-40) returning a
fused MI355X implementation: parameters, gradients and both Adam moments live in FLAT fp32 buffers, so that

  * backward kernels accumulate straight into the flat gradient buffer (param.grad are views of it),
  * data parallelism needs exactly ONE all-reduce of that buffer per optimizer step,
  * global-norm clipping + Adam/AdamW + zero_grad are two kernel launches per weight-decay group
    (sum of squares, then the fused update) with no host synchronisation.

Numerics follow torch.optim.Adam / AdamW (non-amsgrad, eps outside the bias-corrected sqrt).
"""
from __future__ import annotations

import os

from typing import Iterable, List, Optional

import torch
from torch.optim import lr_scheduler

from . import ops


def separate_weight_decayable_params(params):
    """optimizer.py:3-8: ndim >= 2 tensors are decayed, vectors / scalars are not."""
    wd_params = [p for p in params if p.ndim >= 2]
    no_wd_params = [p for p in params if p.ndim < 2]
    return wd_params, no_wd_params


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, param_groups, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0, decoupled=True):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled,
                        amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        super().__init__(param_groups, defaults)
        self._flat = None
        self._t = 0
        self._grads_clean = False
        self._gnorm_sq = None
        self.last_grad_norm_sq = None
        self._ls_state = None           # precision "fp16": the device-side loss-scale block (engine.loss_scale_state), else None

    # ---- flat storage --------------------------------------------------------------------------------
    def _all_params(self) -> List[torch.Tensor]:
        return [p for g in self.param_groups for p in g['params']]

    def _flatten(self):
        params = self._all_params()
        dev = params[0].device
        if dev.type != 'cuda':
            raise RuntimeError("FusedAdam needs the parameters on the MI355X (cuda) device; there is no CPU path")
        sizes = [p.numel() for p in params]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 7) // 8 * 8                     # 8-element granules: 16-byte aligned in the fp32 AND the bf16 buffer
        old = self._flat
        P = torch.zeros(tot, device=dev)
        G = torch.zeros(tot, device=dev)
        M = torch.zeros(tot, device=dev)
        V = torch.zeros(tot, device=dev)
        for i, (p, o, n) in enumerate(zip(params, offs, sizes)):
            P[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                G[o:o + n].copy_(p.grad.reshape(-1))
            if old is not None:
                M[o:o + n].copy_(old['M'][old['offs'][i]: old['offs'][i] + n])
                V[o:o + n].copy_(old['V'][old['offs'][i]: old['offs'][i] + n])
            p.data = P[o:o + n].view(p.shape)
            p.grad = G[o:o + n].view(p.shape)
        # 16-bit shadow of every parameter in the operand type of the model's precision (bf16; fp16 for precision "fp16" -- the
        # model tags its parameters, engine.tag_parameters), refreshed by the SAME kernel that updates the fp32 master (adamw p16
        # output): the engine's 16-bit GEMM operands are views of it, so the per-step weight-cast launches disappear.
        from . import engine
        prec = next((getattr(p, "_omlm_precision", None) for p in params if getattr(p, "_omlm_precision", None)), None)
        prec = prec or engine.default_precision()
        self._ls_state = None
        if engine.is_half(prec):
            owner = next((m for m in (engine.model_of(p) for p in params) if m is not None), None)
            if owner is None:
                raise RuntimeError("FusedAdam: fp16 parameters without their model (engine.tag_parameters): the loss-scale block is the model's")
            self._ls_state = engine.loss_scale_state(owner)
        P16 = torch.empty(tot, device=dev, dtype=torch.float16 if engine.is_half(prec) else torch.bfloat16)
        P16.copy_(P)                                    # one-off cast (cast_pad walks rows: a single 91M-element row is one workgroup)
        for p, o, n in zip(params, offs, sizes):
            p._omlm_bf16 = P16[o:o + n].view(p.shape)
            p._omlm_bf16_version = p._version
        self._flat = dict(P=P, G=G, M=M, V=V, P16=P16, offs=offs, sizes=sizes, total=tot)
        self._gnorm_sq = torch.zeros(1, device=dev)
        self._gnorm_partials = torch.empty(2048, device=dev)      # fixed-order grad-norm reduction: replicas clip identically
        # group ranges (groups are contiguous in the flat order by construction)
        r, k = [], 0
        for g in self.param_groups:
            n = len(g['params'])
            if n == 0:
                r.append((0, 0))
            else:
                r.append((offs[k], offs[k + n - 1] + (sizes[k + n - 1] + 7) // 8 * 8))
            k += n
        self._flat['ranges'] = r

    def _ensure_flat(self):
        if self._flat is None:
            self._flatten()
            return
        params = self._all_params()
        base_p, base_g = self._flat['P'].data_ptr(), self._flat['G'].data_ptr()
        for p, o in zip(params, self._flat['offs']):
            if p.data_ptr() != base_p + 4 * o or p.grad is None or p.grad.data_ptr() != base_g + 4 * o:
                self._flatten()                         # a .to() / load re-bound the storage: re-adopt it
                return

    @property
    def flat_grad(self) -> torch.Tensor:
        self._ensure_flat()
        return self._flat['G']

    @property
    def flat_param(self) -> torch.Tensor:
        self._ensure_flat()
        return self._flat['P']

    # ---- torch.optim API -------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = False):
        self._ensure_flat()
        if not self._grads_clean:
            self._flat['G'].zero_()
            self._grads_clean = True

    @torch.no_grad()
    def step(self, closure=None, max_grad_norm: Optional[float] = None, grad_scale: float = 1.0):
        """One optimizer step.  grad_scale multiplies every gradient first (1/world_size after a SUM all-reduce);
        max_grad_norm clips by the global norm of the scaled gradients (torch clip_grad_norm_ semantics).  In precision "fp16" the
        backward leaves loss_scale x the gradients in the flat buffer (engine.loss_scale); it is divided out here, and a step whose
        gradient norm is not finite (an fp16 overflow) is skipped on the device (adamw kernel)."""
        self._ensure_flat()
        f = self._flat
        self._t += 1                    # fp16: counts attempted steps; the Adam clock of that mode is the device's applied-step counter
        ls = self._ls_state
        gn = None
        if (max_grad_norm is not None and max_grad_norm > 0) or ls is not None:
            # fp16: the norm is ALWAYS formed -- it is the overflow detector (a non-finite norm skips the step on the device), with
            # max_norm = 0 meaning "guard only, no clipping"
            self._gnorm_sq.zero_()
            ops.sumsq_accumulate(f['G'], self._gnorm_sq, self._gnorm_partials)
            gn = self._gnorm_sq
        self._last_gnorm_sq_raw = gn
        for g, (a, b) in zip(self.param_groups, f['ranges']):
            if b <= a:
                continue
            beta1, beta2 = g['betas']
            ops.adamw_clip_step(f['P'][a:b], f['G'][a:b], f['M'][a:b], f['V'][a:b], f['P16'][a:b], lr=g['lr'], beta1=beta1,
                                beta2=beta2, eps=g['eps'], wd=g['weight_decay'], step=self._t, gscale=grad_scale,
                                gnorm_sq=gn, max_norm=max_grad_norm or 0.0, decoupled=g['decoupled'], zero_grad=True, ls_state=ls)
        if ls is not None:
            dyn = os.environ.get("OMLM_FP16_DYNAMIC", "1") != "0"
            ops.loss_scale_update(ls, gn, growth=2.0 if dyn else 1.0, backoff=0.5 if dyn else 1.0,
                                  interval=int(os.environ.get("OMLM_FP16_GROWTH_INTERVAL", "2000")), scale_min=1.0, scale_max=65536.0)
        self._grads_clean = True
        params = self._all_params()                     # the kernels wrote through raw pointers: tell autograd.
        # NB: _increment_version takes an ITERABLE of tensors; handing it one tensor iterates its rows (unbind), which
        # cost 49 ms of host time per step here before this was a single call on the list.
        if hasattr(torch._C, "_increment_version"):
            torch._C._increment_version(params)
        else:                                           # older torch: an in-place no-op bumps the counters
            torch._foreach_add_(params, 0.0)
        for p in params:
            p._omlm_bf16_version = p._version           # the shadow written by this very kernel is current
        return None

    def mark_grads_dirty(self):
        self._grads_clean = False

    @property
    def last_grad_norm_sq(self):
        """Squared global gradient norm of the last step in TRUE gradient units (the loss scale of precision "fp16" divided out; still
        multiplied by any grad_scale the caller passed), as a device tensor; None if the step formed no norm."""
        gn = getattr(self, "_last_gnorm_sq_raw", None)
        if gn is None or self._ls_state is None:
            return gn
        return gn / (self._ls_state[4] * self._ls_state[4])      # slot 4: the scale THAT step's gradients carried (snapshot taken by omlm_loss_scale_update before it moves the scale)

    @last_grad_norm_sq.setter
    def last_grad_norm_sq(self, v):
        self._last_gnorm_sq_raw = v

    @property
    def loss_scale(self) -> float:
        """Current loss scale (1 unless precision "fp16"; reading it synchronises)."""
        return float(self._ls_state[0].item()) if self._ls_state is not None else 1.0

    def loss_scale_report(self) -> dict:
        """fp16 mode: {'scale', 'skipped_steps', 'applied_steps'} read from the device (synchronises: call it where the host already waits,
        e.g. next to the trainer's loss read-back); {} otherwise."""
        if self._ls_state is None:
            return {}
        sc, _, sk, ap = [float(v) for v in self._ls_state.tolist()[:4]]
        return dict(scale=sc, skipped_steps=int(sk), applied_steps=int(ap))

    @torch.no_grad()
    def sync_replicas(self, dp, src: int = 0):
        """Make every data-parallel replica start from rank `src`'s state (what DDP's constructor broadcast does in the
        reference, trainer.py:286-304 via accelerator.prepare): ONE broadcast each of the flat parameter buffer and the two
        Adam moment buffers, then the bf16 operand shadow is refreshed from the received masters."""
        self._ensure_flat()
        if not dp.is_distributed:
            return
        f = self._flat
        for name in ('P', 'M', 'V'):
            dp.broadcast_(f[name], src=src)
        t = torch.tensor([self._t], device=f['P'].device, dtype=torch.int64)
        dp.broadcast_(t, src=src)
        self._t = int(t.item())
        if self._ls_state is not None:
            dp.broadcast_(self._ls_state, src=src)
        f['P16'].copy_(f['P'])
        params = self._all_params()
        if hasattr(torch._C, "_increment_version"):
            torch._C._increment_version(params)
        else:
            torch._foreach_add_(params, 0.0)
        for p in params:
            p._omlm_bf16_version = p._version

    def state_dict(self):
        """torch.optim.Adam(W)-shaped state_dict so checkpoints interoperate with the reference's trainer."""
        self._ensure_flat()
        f = self._flat
        state, k = {}, 0
        groups = []
        # the Adam clock: attempted steps, except in precision "fp16" where skipped (overflowed) steps do not count (device counter)
        t_now = self._t if self._ls_state is None else int(self._ls_state[3].item())
        for g in self.param_groups:
            idx = []
            for _ in g['params']:
                o, n = f['offs'][k], f['sizes'][k]
                shape = self._all_params()[k].shape
                state[k] = dict(step=torch.tensor(float(t_now)), exp_avg=f['M'][o:o + n].view(shape).clone(),
                                exp_avg_sq=f['V'][o:o + n].view(shape).clone())
                idx.append(k)
                k += 1
            groups.append({**{kk: vv for kk, vv in g.items() if kk != 'params'}, 'params': idx})
        out = dict(state=state if t_now > 0 else {}, param_groups=groups)
        if self._ls_state is not None:                  # extra top-level key (torch's Optimizer.load_state_dict ignores it)
            out["omlm_loss_scale"] = self.loss_scale_report()
        return out

    def load_state_dict(self, sd):
        self._ensure_flat()
        f = self._flat
        for g, sg in zip(self.param_groups, sd['param_groups']):
            for kk in ('lr', 'betas', 'eps', 'weight_decay'):
                if kk in sg:
                    g[kk] = sg[kk]
            if 'initial_lr' in sg:
                g['initial_lr'] = sg['initial_lr']
        t = 0
        for k, st in sd.get('state', {}).items():
            k = int(k)
            o, n = f['offs'][k], f['sizes'][k]
            f['M'][o:o + n].copy_(st['exp_avg'].reshape(-1).to(f['M'].device))
            f['V'][o:o + n].copy_(st['exp_avg_sq'].reshape(-1).to(f['V'].device))
            t = max(t, int(float(st['step'])))
        self._t = t
        if self._ls_state is not None:
            self._ls_state[3] = float(t)
            rep = sd.get("omlm_loss_scale")
            if rep:
                self._ls_state[0] = float(rep.get("scale", self._ls_state[0].item()))
                self._ls_state[2] = float(rep.get("skipped_steps", 0))


def get_optimizer(params: Iterable[torch.Tensor], lr=1e-4, wd=1e-2, betas=(0.9, 0.99), eps=1e-8,
                  filter_by_requires_grad=False, group_wd_params=True, **kwargs):
    """optimizer.py:10-34: Adam when wd == 0, else AdamW with ndim<2 parameters exempt from decay."""
    params = list(params)
    if filter_by_requires_grad:
        params = [p for p in params if p.requires_grad]
    if wd == 0:
        return FusedAdam([{'params': params}], lr=lr, betas=betas, eps=eps, weight_decay=0.0, decoupled=False)
    if group_wd_params:
        wd_params, no_wd_params = separate_weight_decayable_params(params)
        groups = [{'params': wd_params}, {'params': no_wd_params, 'weight_decay': 0}]
    else:
        groups = [{'params': params}]
    return FusedAdam(groups, lr=lr, betas=betas, eps=eps, weight_decay=wd, decoupled=True)


def get_linear_scheduler(optimizer, total_iters=10000, start_factor=1e-7):
    """optimizer.py:36-41"""
    return lr_scheduler.LinearLR(optimizer=optimizer, start_factor=start_factor, end_factor=1., total_iters=total_iters)
