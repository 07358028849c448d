"""Parameter containers of the trunk with the reference's module tree and state_dict keys
(reference open_musiclm/transformer.py).  The arithmetic lives in ``engine.py`` / libomlm_hip.so; the
sub-modules here only own parameters, so that checkpoints written by the reference load strictly and
``torch.manual_seed(s)`` + construction yields the very same initial weights as the reference.

state_dict schema:
  layers.{l}.0.{q_scale,k_scale,norm.gamma,norm.beta,to_q.weight,to_kv.weight,to_out.0.weight}
  layers.{l}.2.{0.gamma,0.beta,1.weight,2.ds_conv.weight,4.gamma,4.beta,6.weight}      (ConvFeedForward)
  layers.{l}.2.{0.gamma,0.beta,1.weight,3.gamma,3.beta,5.weight}                        (FeedForward)
  rel_pos_bias.net.{0,1,2}.0.{weight,bias}, rel_pos_bias.net.3.{weight,bias} | rel_pos_bias.relative_attention_bias.weight
  norm.{gamma,beta}
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import engine
from .utils import default, exists


def _fused(name):
    def forward(self, *a, **k):
        raise RuntimeError(f"{name}.forward is fused into the MI355X trunk kernels; call Transformer / "
                           "TokenConditionedTransformer instead")
    return forward


class LayerNorm(nn.Module):
    """transformer.py:24-31 (gamma learnable, beta a zero buffer that IS part of the state_dict)."""

    def __init__(self, dim):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(dim))
        self.register_buffer("beta", torch.zeros(dim))

    forward = _fused("LayerNorm")


class RelativePositionBias(nn.Module):
    """transformer.py:36-67.  The engine evaluates the MLP on the n distances 0..n-1 only and hands the
    [n, heads] table to the attention kernel instead of gathering an [h, n, n] matrix."""

    def __init__(self, *, dim, heads, layers=3):
        super().__init__()
        self.net = nn.ModuleList([])
        self.net.append(nn.Sequential(nn.Linear(1, dim), nn.SiLU()))
        for _ in range(layers - 1):
            self.net.append(nn.Sequential(nn.Linear(dim, dim), nn.SiLU()))
        self.net.append(nn.Linear(dim, heads))
        assert layers == 3, "the MI355X rel-pos path is written for the reference's 3 hidden layers"

    forward = _fused("RelativePositionBias")


def t5_bucket_of_distance(r: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """Bucket of relative position i - j = r >= 0 under transformer.py:85-105 with causal=True.

    The reference negates the relative position (n = -(i - j)) and clamps at 0, so every past key lands in
    bucket 0; the general formula is kept so the behaviour follows the reference for any r."""
    n = (-r).clamp(min=0)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float().clamp(min=1) / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).long()
    large = large.clamp(max=num_buckets - 1)
    return torch.where(is_small, n, large)


class T5RelativePositionBias(nn.Module):
    """transformer.py:69-117."""

    def __init__(self, *, heads, num_buckets=32, max_distance=128, causal=True):
        super().__init__()
        self.num_buckets, self.max_distance, self.causal = num_buckets, max_distance, causal
        self.relative_attention_bias = nn.Embedding(num_buckets, heads)

    forward = _fused("T5RelativePositionBias")


class CausalDSConv(nn.Module):
    """transformer.py:122-131."""

    def __init__(self, dim):
        super().__init__()
        self.ds_conv = nn.Conv1d(dim, dim, 3, bias=False, groups=dim)

    forward = _fused("CausalDSConv")


class GEGLU(nn.Module):
    """transformer.py:134-137."""
    forward = _fused("GEGLU")


class _FeedForwardBase(nn.Sequential):
    _idx = {}

    @property
    def norm_in(self): return self[self._idx["norm_in"]]
    @property
    def w_in(self): return self[self._idx["w_in"]]
    @property
    def norm_mid(self): return self[self._idx["norm_mid"]]
    @property
    def w_out(self): return self[self._idx["w_out"]]
    @property
    def dropout_p(self): return self[self._idx["dropout"]].p

    def forward(self, x):
        raise RuntimeError("feed-forward blocks are fused into the MI355X trunk kernels; call Transformer")


class ConvFeedForwardBlock(_FeedForwardBase):
    """transformer.py:140-150."""
    _idx = dict(norm_in=0, w_in=1, conv=2, norm_mid=4, dropout=5, w_out=6)

    def __init__(self, dim, mult=4, dropout=0.1):
        inner = int(dim * 2 * mult / 3)
        super().__init__(LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), CausalDSConv(inner * 2), GEGLU(),
                         LayerNorm(inner), nn.Dropout(dropout), nn.Linear(inner, dim, bias=False))
        self.inner_dim = inner

    def conv_param(self):
        return self[2].ds_conv.weight

    def conv_weight(self):
        return self[2].ds_conv.weight.detach().reshape(2 * self.inner_dim, 3)


class FeedForwardBlock(_FeedForwardBase):
    """transformer.py:152-161 (use_conv_ff=False): the same fused kernel with identity taps (0, 0, 1)."""
    _idx = dict(norm_in=0, w_in=1, norm_mid=3, dropout=4, w_out=5)

    def __init__(self, dim, mult=4, dropout=0.1):
        inner = int(dim * mult)
        super().__init__(LayerNorm(dim), nn.Linear(dim, inner * 2, bias=False), GEGLU(), LayerNorm(inner),
                         nn.Dropout(dropout), nn.Linear(inner, dim, bias=False))
        self.inner_dim = inner
        self._taps = None

    def conv_param(self):
        return None

    def conv_weight(self):
        dev = self[1].weight.device
        if self._taps is None or self._taps.device != dev:
            t = torch.zeros(2 * self.inner_dim, 3, device=dev)
            t[:, 2] = 1.0
            self._taps = t
        return self._taps


def ConvFeedForward(dim, mult=4, dropout=0.1):
    return ConvFeedForwardBlock(dim, mult, dropout)


def FeedForward(dim, mult=4, dropout=0.1):
    return FeedForwardBlock(dim, mult, dropout)


class Attention(nn.Module):
    """transformer.py:166-333: multi-query attention (one shared K/V head), l2-normalised q/k with learned
    per-dim scales, fixed logit scale 8.  Only the self-attention configuration used by
    TokenConditionedTransformer (no context, no null kv, no prefix) exists on the MI355X path."""

    def __init__(self, dim, causal=False, non_causal_prefix=0, dim_head=64, dim_context=None, heads=8,
                 norm_context=False, num_null_kv=0, dropout=0.1, scale=8, use_memory_efficient_attention=False):
        super().__init__()
        if dim_head != engine.DIM_HEAD or scale != engine.ATTN_SCALE:
            raise ValueError("the MI355X attention kernel is specialised for dim_head=64, scale=8 (reference defaults)")
        if num_null_kv != 0 or norm_context or (dim_context is not None and dim_context != dim):
            raise NotImplementedError("cross-attention / null-kv are not on the TokenConditionedTransformer path")
        if dropout != 0.0:
            raise NotImplementedError("attn_dropout > 0 is not supported (every shipped config uses 0.0)")
        self.heads, self.scale, self.causal = heads, scale, causal
        self.non_causal_prefix = non_causal_prefix
        self.dropout = dropout
        # accepted for config compatibility: the flash-style HIP kernel already is the memory-efficient path
        self.use_memory_efficient_attention = use_memory_efficient_attention
        inner_dim = dim_head * heads
        self.norm = LayerNorm(dim)
        self.context_norm = nn.Identity()
        self.attn_dropout = nn.Dropout(dropout)
        self.num_null_kv = 0
        self.null_kv = None
        self.to_q = nn.Linear(dim, inner_dim, bias=False)
        self.to_kv = nn.Linear(dim, dim_head * 2, bias=False)
        self.q_scale = nn.Parameter(torch.ones(dim_head))
        self.k_scale = nn.Parameter(torch.ones(dim_head))
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim, bias=False), nn.Dropout(dropout))

    forward = _fused("Attention")


class _TrunkFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tr, x, keymask, precision, *params):
        B, N, D = x.shape
        owner = tr.__dict__.get("_omlm_owner")
        pw = engine.PreparedWeights(owner, precision) if owner is not None else None
        if pw is None:
            raise RuntimeError("standalone Transformer.forward needs the module to be owned by a "
                               "TokenConditionedTransformer (weight preparation walks the owner)")
        xf = x.detach().reshape(B * N, D).contiguous().float()
        y, saved = engine.trunk_forward(tr, pw, xf, keymask, B, N, True, tr.training)
        ctx.tr, ctx.pw, ctx.saved, ctx.B, ctx.N, ctx.np = tr, pw, saved, B, N, len(params)
        lo = y.__dict__.pop("_omlm_lo", None)            # precision "fp16ff": the final LayerNorm's output arrives as hi/lo planes
        out = y.float() if lo is None else y.float() + lo.float()
        return out.view(B, N, D)

    @staticmethod
    def backward(ctx, dy):
        dx = engine.trunk_backward(ctx.tr, ctx.pw, ctx.saved, dy.reshape(-1, dy.shape[-1]).contiguous().float(),
                                   ctx.B, ctx.N, out_scale=float(ctx.tr.grad_shrink_alpha))
        return (None, dx.view(ctx.B, ctx.N, -1), None, None) + (None,) * ctx.np


class Transformer(nn.Module):
    """transformer.py:338-424."""

    def __init__(self, *, dim, depth, heads, dim_context=None, cross_attend=False, attn_dropout=0., ff_dropout=0.,
                 use_conv_ff=True, grad_shrink_alpha=0.1, cond_as_self_attn_prefix=False, non_causal_prefix_size=0,
                 relative_position_bias_type='continuous', **kwargs):
        super().__init__()
        assert not (cross_attend and cond_as_self_attn_prefix)
        if cross_attend or cond_as_self_attn_prefix:
            raise NotImplementedError("text-conditioning (cross attention / prefix) is not on the "
                                      "TokenConditionedTransformer path (has_condition=False in every factory)")
        if dim % 8 != 0:
            raise ValueError("dim must be a multiple of 8")
        self.dim, self.heads, self.depth = dim, heads, depth
        self.dim_context = default(dim_context, dim)
        self.cond_as_self_attn_prefix = cond_as_self_attn_prefix
        self.grad_shrink_alpha = grad_shrink_alpha
        self.non_causal_prefix_size = non_causal_prefix_size
        self.relative_position_bias_type = relative_position_bias_type
        self.layers = nn.ModuleList([])
        if relative_position_bias_type == 'continuous':
            self.rel_pos_bias = RelativePositionBias(dim=dim // 2, heads=heads)
        elif relative_position_bias_type == 't5':
            self.rel_pos_bias = T5RelativePositionBias(heads=heads, num_buckets=32, max_distance=128)
        elif relative_position_bias_type == 'none':
            self.rel_pos_bias = None
        else:
            raise ValueError(f'invalid relative position bias type: {relative_position_bias_type}')
        for _ in range(depth):
            self.layers.append(nn.ModuleList([
                Attention(dim=dim, heads=heads, dropout=attn_dropout, causal=True,
                          non_causal_prefix=non_causal_prefix_size, **kwargs),
                None,
                ConvFeedForward(dim=dim, dropout=ff_dropout) if use_conv_ff else FeedForward(dim=dim, dropout=ff_dropout),
            ]))
        self.norm = LayerNorm(dim)

    def forward(self, x, self_attn_mask=None, context=None, context_mask=None, attn_bias=None, precision=None):
        assert not exists(context), "conditioning context is not supported on the MI355X path"
        if exists(attn_bias):
            raise NotImplementedError("external attn_bias tensors are not supported; the rel-pos table is computed in-engine")
        keymask = self_attn_mask.to(torch.uint8).contiguous() if exists(self_attn_mask) else None
        precision = precision or engine.default_precision()
        return _TrunkFunction.apply(self, x, keymask, precision, *self.parameters())
