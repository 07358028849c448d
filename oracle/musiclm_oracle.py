"""CPU oracle for the open-musiclm TokenConditionedTransformer hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``open_musiclm_amd`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

It restates, as pure functions over a *reference-schema* ``state_dict`` (the
key list in SURVEY.md §8b), the arithmetic of

* ``open_musiclm/transformer.py``  (LayerNorm :24-31, RelativePositionBias
  :36-67, T5RelativePositionBias :69-117, CausalDSConv :122-131, GEGLU
  :134-137, ConvFeedForward :140-150, FeedForward :152-161, Attention
  :166-333 non-xformers branch, Transformer :338-424)
* ``open_musiclm/open_musiclm.py`` (TokenConditionedTransformer.forward
  :100-190, TokenConditionedTransformerWrapper.forward :328-410 and
  .generate :253-326)
* ``open_musiclm/utils.py``        (generate_mask_with_prob :49-56,
  grad_shrink :60-61, l2norm :68-69, gumbel_sample :71-76, top_k :78-84,
  mask_out_after_eos_id :86-93, append_eos_id :112-117, get_embeds :126-143)
* the residual-VQ / k-means nearest-codeword step that
  ``clap_quantized.py:75-87`` and ``hf_hubert_kmeans.py:78-87`` delegate to
  third-party code (vector-quantize-pytorch ``ResidualVQ`` eval path, sklearn
  ``MiniBatchKMeans.predict``).

Parity pinning: the transformer functions are pinned against the reference's
own modules by ``oracle/make_golden.py`` / ``make_golden_r2.py`` / ``make_golden_r5.py``
(which import /root/reference in the build container and write ``tests/golden/*.npz``).
The RVQ nearest-code step lives in the un-vendored, un-installed vector-quantize-pytorch:
no reference-held vector can exist offline, so it restates the library's published
``argmax(-cdist(x, embed))`` form (``cdist_form`` below) and is pinned bit for bit against
``torch.cdist`` itself (tests/golden/rvq_cdist_pin.npz) -- **pinned to the distance form,
not to the library**.  RVQ *fitting* (``rvq_fit_step``): **parity unpinned**.
The k-means assign is pinned against sklearn's ``predict``.

All functions run in whatever dtype the state dict / inputs carry (fp32 for
parity with the reference, fp64 for a tighter yardstick).  They are written
with differentiable torch ops so ``torch.autograd.grad`` yields oracle
gradients.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------
# configuration mirror (TokenSequenceInfo open_musiclm.py:23-30 + ctor kwargs)
# ----------------------------------------------------------------------------

@dataclass
class SeqInfo:
    codebook_size: int
    num_quantizers: int


@dataclass
class ModelSpec:
    token_sequences: List[SeqInfo]
    dim: int
    depth: int
    heads: int = 8
    dim_head: int = 64
    attn_scale: float = 8.0
    use_conv_ff: bool = True
    grad_shrink_alpha: float = 0.1
    relative_position_bias_type: str = "continuous"
    non_causal_prefix_size: int = 0
    use_absolute_position_embeddings: bool = False
    ff_dropout: float = 0.0          # oracle parity runs use 0 / eval mode
    eos_ids: List[int] = field(default_factory=list)

    def __post_init__(self):
        if not self.eos_ids:
            self.eos_ids = [s.codebook_size for s in self.token_sequences]


def semantic_spec(dim=1024, depth=6, heads=8, clap_q=12, **kw) -> ModelSpec:
    """open_musiclm.py:414-428"""
    return ModelSpec([SeqInfo(1024, clap_q), SeqInfo(1024, 1)], dim, depth, heads, **kw)


def coarse_spec(dim=1024, depth=6, heads=8, clap_q=12, coarse_q=3, **kw) -> ModelSpec:
    """open_musiclm.py:432-450"""
    return ModelSpec([SeqInfo(1024, clap_q), SeqInfo(1024, 1), SeqInfo(1024, coarse_q)],
                     dim, depth, heads, **kw)


def fine_spec(dim=1024, depth=6, heads=8, clap_q=12, coarse_q=3, fine_q=5, **kw) -> ModelSpec:
    """open_musiclm.py:454-472"""
    return ModelSpec([SeqInfo(1024, clap_q), SeqInfo(1024, coarse_q), SeqInfo(1024, fine_q)],
                     dim, depth, heads, **kw)


# ----------------------------------------------------------------------------
# transformer.py restatement
# ----------------------------------------------------------------------------

def layer_norm(x: Tensor, gamma: Tensor, eps: float = 1e-5) -> Tensor:
    """transformer.py:24-31 — mean-subtracting LN, learnable gamma, beta == 0."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)      # biased variance
    return (x - mu) * torch.rsqrt(var + eps) * gamma


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def gelu_erf(x: Tensor) -> Tensor:
    """F.gelu default (exact erf form) used by GEGLU transformer.py:134-137."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def rel_pos_table_continuous(sd: Dict[str, Tensor], prefix: str, n: int) -> Tensor:
    """transformer.py:55-67 restated as a 1-D table.

    Returns ``tab[h, r]`` for r = i - j in [-(n-1), n-1] stored at index
    r + n - 1, i.e. exactly the MLP output rows before the reference gathers
    them into an [h, n, n] matrix (``x[rel_pos]`` :66).
    """
    w0 = sd[prefix + "net.0.0.weight"]
    x = torch.arange(-n + 1, n, dtype=w0.dtype).unsqueeze(-1)
    li = 0
    while (prefix + f"net.{li}.0.weight") in sd:
        x = silu(F.linear(x, sd[prefix + f"net.{li}.0.weight"], sd[prefix + f"net.{li}.0.bias"]))
        li += 1
    x = F.linear(x, sd[prefix + f"net.{li}.weight"], sd[prefix + f"net.{li}.bias"])
    return x.t().contiguous()                              # [h, 2n-1]


def t5_bucket(rel: Tensor, num_buckets: int = 32, max_distance: int = 128) -> Tensor:
    """transformer.py:85-105, causal=True branch (the only one constructed, :369)."""
    n = (-rel).clamp(min=0)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.to(torch.float32) / max_exact)
                         / math.log(max_distance / max_exact) * (num_buckets - max_exact)).long()
    large = large.clamp(max=num_buckets - 1)
    return torch.where(is_small, n, large)


def rel_pos_table_t5(sd: Dict[str, Tensor], prefix: str, n: int) -> Tensor:
    """transformer.py:107-117 as a 1-D table over i-j (bucket(j-i... ) see :113)."""
    w = sd[prefix + "relative_attention_bias.weight"]     # [buckets, h]
    r = torch.arange(-n + 1, n)                            # r = i - j  (rel_pos[i, j] = i - j)
    b = t5_bucket(r, num_buckets=w.shape[0])
    return w[b].t().contiguous()                           # [h, 2n-1]


def rel_pos_bias_matrix(table: Tensor, n: int) -> Tensor:
    """Gather the [h, 2n-1] table into the reference's [h, n, n] layout (:57-58,:66-67)."""
    pos = torch.arange(n)
    rel = pos[:, None] - pos[None, :] + (n - 1)
    return table[:, rel]


def attention(sd: Dict[str, Tensor], p: str, x: Tensor, bias: Optional[Tensor],
              key_mask: Optional[Tensor], spec: ModelSpec) -> Tensor:
    """transformer.py:214-333 (self-attention, causal, num_null_kv=0, no prefix)."""
    b, n, _ = x.shape
    h, dh = spec.heads, spec.dim_head
    # NB: kv_input is bound to the *un-normalised* x at :228, before the pre-norm at :250,
    # so only the queries see LayerNorm(x); keys/values are projected from the raw residual.
    xn = layer_norm(x, sd[p + "norm.gamma"])                               # :250
    q = F.linear(xn, sd[p + "to_q.weight"])                                # :254
    kv = F.linear(x, sd[p + "to_kv.weight"])                               # :228,:254
    k, v = kv[..., :dh], kv[..., dh:]
    q = q.view(b, n, h, dh).permute(0, 2, 1, 3)                            # :265
    q = q / q.norm(dim=-1, keepdim=True).clamp(min=1e-12)                  # :269 F.normalize
    k = k / k.norm(dim=-1, keepdim=True).clamp(min=1e-12)
    q = q * sd[p + "q_scale"]                                              # :270
    k = k * sd[p + "k_scale"]                                              # :271
    sim = torch.einsum("bhid,bjd->bhij", q, k) * spec.attn_scale           # :304
    if bias is not None:
        sim = sim + bias                                                   # :306-308
    neg = -torch.finfo(sim.dtype).max
    if key_mask is not None:
        sim = sim.masked_fill(~key_mask[:, None, None, :], neg)            # :310-313
    causal = torch.ones(n, n, dtype=torch.bool).triu(1)                    # :315-317
    if spec.non_causal_prefix_size > 0:
        causal[:spec.non_causal_prefix_size, :spec.non_causal_prefix_size] = False   # :319-320
    sim = sim.masked_fill(causal, neg)                                     # :322
    attn = sim.softmax(dim=-1)                                             # :324
    out = torch.einsum("bhij,bjd->bhid", attn, v)                          # :328
    out = out.permute(0, 2, 1, 3).reshape(b, n, h * dh)                    # :331
    return F.linear(out, sd[p + "to_out.0.weight"])                        # :333


def causal_dwconv3(x: Tensor, w: Tensor) -> Tensor:
    """transformer.py:122-131: y[t] = w0 x[t-2] + w1 x[t-1] + w2 x[t], per channel."""
    w = w.reshape(-1, 3)
    xp = F.pad(x, (0, 0, 2, 0))
    return xp[:, :-2] * w[:, 0] + xp[:, 1:-1] * w[:, 1] + xp[:, 2:] * w[:, 2]


def conv_feed_forward(sd: Dict[str, Tensor], p: str, x: Tensor, spec: ModelSpec,
                      drop_mask: Optional[Tensor] = None) -> Tensor:
    """transformer.py:140-150."""
    xn = layer_norm(x, sd[p + "0.gamma"])
    hdn = F.linear(xn, sd[p + "1.weight"])
    hdn = causal_dwconv3(hdn, sd[p + "2.ds_conv.weight"])
    a, gate = hdn.chunk(2, dim=-1)                         # GEGLU :134-137
    g = gelu_erf(gate) * a
    g = layer_norm(g, sd[p + "4.gamma"])
    if drop_mask is not None:                              # nn.Dropout :148 with an injected keep-mask
        g = g * drop_mask / (1.0 - spec.ff_dropout)
    return F.linear(g, sd[p + "6.weight"])


def feed_forward(sd: Dict[str, Tensor], p: str, x: Tensor, spec: ModelSpec,
                 drop_mask: Optional[Tensor] = None) -> Tensor:
    """transformer.py:152-161 (use_conv_ff=False)."""
    xn = layer_norm(x, sd[p + "0.gamma"])
    hdn = F.linear(xn, sd[p + "1.weight"])
    a, gate = hdn.chunk(2, dim=-1)
    g = gelu_erf(gate) * a
    g = layer_norm(g, sd[p + "3.gamma"])
    if drop_mask is not None:
        g = g * drop_mask / (1.0 - spec.ff_dropout)
    return F.linear(g, sd[p + "5.weight"])


def grad_shrink(t: Tensor, alpha: float) -> Tensor:
    """utils.py:60-61."""
    return t * alpha + t.detach() * (1 - alpha)


def trunk(sd: Dict[str, Tensor], x: Tensor, key_mask: Optional[Tensor], spec: ModelSpec,
          prefix: str = "transformer.", drop_masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """transformer.py:385-424."""
    n = x.shape[1]
    x = grad_shrink(x, spec.grad_shrink_alpha)                              # :400
    bias = None
    if spec.relative_position_bias_type == "continuous":                    # :366-373,:405
        bias = rel_pos_bias_matrix(rel_pos_table_continuous(sd, prefix + "rel_pos_bias.", n), n)
    elif spec.relative_position_bias_type == "t5":
        bias = rel_pos_bias_matrix(rel_pos_table_t5(sd, prefix + "rel_pos_bias.", n), n)
    for l in range(spec.depth):                                             # :414-422
        lp = f"{prefix}layers.{l}."
        x = attention(sd, lp + "0.", x, bias, key_mask, spec) + x
        dm = None if drop_masks is None else drop_masks[l]
        if spec.use_conv_ff:
            x = conv_feed_forward(sd, lp + "2.", x, spec, dm) + x
        else:
            x = feed_forward(sd, lp + "2.", x, spec, dm) + x
    return layer_norm(x, sd[prefix + "norm.gamma"])                         # :424


# ----------------------------------------------------------------------------
# KV-cached formulation of the trunk (checker for csrc/decode.hip)
# ----------------------------------------------------------------------------
# The reference has no cached decode (open_musiclm.py:301-321 re-runs the whole forward per id).  Because attention is
# causal (:315-322), the depthwise conv only looks back 2 rows (:122-131) and LayerNorm is per row, row p of every layer
# depends on rows <= p only.  trunk_cached_rows() computes new rows against
#   K/V  : the l2-normalised, scaled keys and the values of the earlier rows of each layer,
#   hist : the FF-in outputs of the two previous rows (state of the causal conv),
# which is what the HIP decode kernels keep.  tests/test_oracle_golden.py checks it row by row against trunk().

def new_trunk_cache(spec: ModelSpec, batch: int) -> Dict[str, list]:
    return dict(k=[None] * spec.depth, v=[None] * spec.depth, hist=[None] * spec.depth, rows=0)


def trunk_cached_rows(sd: Dict[str, Tensor], x_new: Tensor, cache: Dict[str, list], spec: ModelSpec,
                      n_total: int, prefix: str = "transformer.") -> Tensor:
    """x_new: [B, m, d] embeddings of rows cache['rows'] .. cache['rows'] + m - 1 (m = 1 in decode, m = prompt length in the
    prefill).  n_total only sizes the rel-pos table (its entries do not depend on it).  Returns the final-LN hidden rows."""
    b, m, _ = x_new.shape
    h, dh = spec.heads, spec.dim_head
    p0 = cache["rows"]
    table = None
    if spec.relative_position_bias_type == "continuous":
        table = rel_pos_table_continuous(sd, prefix + "rel_pos_bias.", n_total)
    elif spec.relative_position_bias_type == "t5":
        table = rel_pos_table_t5(sd, prefix + "rel_pos_bias.", n_total)
    x = x_new
    for l in range(spec.depth):
        ap, fp = f"{prefix}layers.{l}.0.", f"{prefix}layers.{l}.2."
        xn = layer_norm(x, sd[ap + "norm.gamma"])
        q = F.linear(xn, sd[ap + "to_q.weight"]).view(b, m, h, dh)
        kv = F.linear(x, sd[ap + "to_kv.weight"])                    # un-normalised input, as in attention()
        k_new, v_new = kv[..., :dh], kv[..., dh:]
        q = q / q.norm(dim=-1, keepdim=True).clamp(min=1e-12) * sd[ap + "q_scale"]
        k_new = k_new / k_new.norm(dim=-1, keepdim=True).clamp(min=1e-12) * sd[ap + "k_scale"]
        k = k_new if cache["k"][l] is None else torch.cat([cache["k"][l], k_new], 1)
        v = v_new if cache["v"][l] is None else torch.cat([cache["v"][l], v_new], 1)
        cache["k"][l], cache["v"][l] = k, v
        sim = torch.einsum("bihd,bjd->bhij", q, k) * spec.attn_scale  # [b, h, m, p0 + m]
        i_idx = torch.arange(p0, p0 + m)[:, None]
        j_idx = torch.arange(p0 + m)[None, :]
        if table is not None:
            sim = sim + table[:, (i_idx - j_idx).clamp(min=-(n_total - 1)) + n_total - 1]
        sim = sim.masked_fill(j_idx > i_idx, -torch.finfo(sim.dtype).max)
        out = torch.einsum("bhij,bjd->bihd", sim.softmax(-1), v).reshape(b, m, h * dh)
        x = F.linear(out, sd[ap + "to_out.0.weight"]) + x
        xn = layer_norm(x, sd[fp + "0.gamma"])
        hdn = F.linear(xn, sd[fp + "1.weight"])
        if spec.use_conv_ff:
            prev = cache["hist"][l] if cache["hist"][l] is not None else torch.zeros(b, 2, hdn.shape[-1], dtype=hdn.dtype)
            ext = torch.cat([prev, hdn], 1)                          # rows p0-2 .. p0+m-1
            cache["hist"][l] = ext[:, -2:]
            w = sd[fp + "2.ds_conv.weight"].reshape(-1, 3)
            hdn = ext[:, :-2] * w[:, 0] + ext[:, 1:-1] * w[:, 1] + ext[:, 2:] * w[:, 2]
            a, gate = hdn.chunk(2, dim=-1)
            g = layer_norm(gelu_erf(gate) * a, sd[fp + "4.gamma"])
            x = F.linear(g, sd[fp + "6.weight"]) + x
        else:
            a, gate = hdn.chunk(2, dim=-1)
            g = layer_norm(gelu_erf(gate) * a, sd[fp + "3.gamma"])
            x = F.linear(g, sd[fp + "5.weight"]) + x
    cache["rows"] = p0 + m
    return layer_norm(x, sd[prefix + "norm.gamma"])


# ----------------------------------------------------------------------------
# open_musiclm.py restatement
# ----------------------------------------------------------------------------

def embed_sequences(sd: Dict[str, Tensor], all_token_ids: Sequence[Tensor], spec: ModelSpec
                    ) -> Tuple[Tensor, List[int]]:
    """open_musiclm.py:116-145 + utils.get_embeds :126-143.

    Returns the concatenated [B, N, d] input and the split points (:141-142,:149).
    """
    b = all_token_ids[0].shape[0]
    parts, split_at = [], []
    for i, (seq, ids) in enumerate(zip(spec.token_sequences, all_token_ids)):
        ids = ids.reshape(b, -1)
        if seq.num_quantizers > 1:                                          # :126-130
            off = seq.codebook_size * (torch.arange(ids.shape[-1]) % seq.num_quantizers)
            ids = ids + off
        table = sd[f"embeddings.{i}.weight"]
        pad = ids == -1                                                     # get_embeds
        emb = table[ids.masked_fill(pad, 0)]
        emb = emb.masked_fill(pad[..., None], 0.0)
        if spec.use_absolute_position_embeddings:                           # :134-136
            emb = emb + sd[f"absolute_position_embeddings.{i}.weight"][: emb.shape[1]][None]
        start = sd[f"start_tokens.{i}"].expand(b, 1, -1)                    # :139
        parts += [start, emb]
        n_tok = emb.shape[1] + 1
        split_at.append(n_tok if not split_at else split_at[-1] + n_tok)
    return torch.cat(parts, dim=1), split_at[:-1]


def logit_heads(sd: Dict[str, Tensor], hidden: Tensor, split_at: List[int], spec: ModelSpec,
                only_final: bool = False) -> List[Optional[Tensor]]:
    """open_musiclm.py:149-190."""
    pieces = list(torch.tensor_split(hidden, split_at, dim=1))
    pieces = [p[:, :-1] for p in pieces[:-1]] + [pieces[-1]]                # :156
    out: List[Optional[Tensor]] = []
    for i, (seq, p) in enumerate(zip(spec.token_sequences, pieces)):
        if only_final and i != len(pieces) - 1:
            out.append(None)
            continue
        w = sd[f"logit_weights.{i}"]                                        # [q, V+1, d]
        q = seq.num_quantizers
        pos_q = torch.arange(p.shape[1]) % q       # position -> quantizer, incl. the remainder (:177-182)
        out.append(torch.einsum("bnd,ncd->bnc", p, w[pos_q]))
    return out


def token_conditioned_forward(sd: Dict[str, Tensor], spec: ModelSpec, all_token_ids: Sequence[Tensor],
                              self_attn_mask: Optional[Tensor] = None, only_final: bool = False,
                              drop_masks: Optional[Sequence[Tensor]] = None,
                              return_hidden: bool = False):
    """TokenConditionedTransformer.forward open_musiclm.py:100-190."""
    x, split_at = embed_sequences(sd, all_token_ids, spec)
    hidden = trunk(sd, x, self_attn_mask, spec, drop_masks=drop_masks)
    logits = logit_heads(sd, hidden, split_at, spec, only_final)
    return (logits, hidden) if return_hidden else logits


def append_eos(ids: Tensor, eos_id: int) -> Tensor:
    """utils.py:112-117."""
    return torch.cat([ids, torch.full((ids.shape[0], 1), eos_id, dtype=ids.dtype)], dim=-1)


def build_training_inputs(all_token_ids: Sequence[Tensor], spec: ModelSpec, pad_id: int = -1
                          ) -> Tuple[List[Tensor], List[Tensor], Tensor]:
    """open_musiclm.py:340-371: flatten, append eos, labels, drop last, key mask."""
    b = all_token_ids[0].shape[0]
    ids = [append_eos(t.reshape(b, -1).long(), e) for t, e in zip(all_token_ids, spec.eos_ids)]
    labels = [t.clone() for t in ids]
    ids[-1] = ids[-1][:, :-1]
    mask_parts = []
    for k in range(len(ids) - 1):
        m = (ids[k] != pad_id) & (ids[k] != spec.eos_ids[k])
        ids[k] = ids[k].masked_fill(~m, 0)                                   # :363
        mask_parts.append(F.pad(m, (1, 0), value=True))                      # :366
    mask = torch.cat(mask_parts, dim=-1) if mask_parts else torch.empty(b, 0, dtype=torch.bool)
    mask = F.pad(mask, (0, ids[-1].shape[-1] + 1), value=True)               # :370-371
    return ids, labels, mask


def forgetful_mask_from_noise(noise: Tensor, mask_prob: float) -> Tensor:
    """utils.py:49-56 with the randn tensor injected (so CPU and GPU can share it)."""
    noise = noise.clone()
    seq = noise.shape[-1]
    noise[:, 0] = -torch.finfo(noise.dtype).max
    num_mask = min(int(seq * mask_prob), seq - 1)
    idx = noise.topk(num_mask, dim=-1).indices
    return ~torch.zeros_like(noise).scatter(1, idx, 1.0).bool()


def wrapper_forward_loss(sd: Dict[str, Tensor], spec: ModelSpec, all_token_ids: Sequence[Tensor],
                         loss_weights: Sequence[float], forget_noise: Optional[Tensor] = None,
                         mask_prob: float = 0.15, drop_masks: Optional[Sequence[Tensor]] = None):
    """TokenConditionedTransformerWrapper.forward(return_loss=True) open_musiclm.py:328-410.

    Returns (loss, all_logits [b c n], all_labels).
    """
    ids, labels, mask = build_training_inputs(all_token_ids, spec)
    if forget_noise is not None:                                             # :374-376
        mask = mask & forgetful_mask_from_noise(forget_noise, mask_prob)
    logits = token_conditioned_forward(sd, spec, ids, mask, drop_masks=drop_masks)
    logits = [l.transpose(1, 2) for l in logits]                             # :389
    total, running = 0, 0.0
    for lg, lb, w in zip(logits, labels, loss_weights):                      # :391-410
        if w > 0:
            n = lb.numel()
            running = running + F.cross_entropy(lg, lb) * n * w
            total += n
    return running / total, logits, labels


def top_k_filter(logits: Tensor, thres: float = 0.9) -> Tensor:
    """utils.py:78-84."""
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    val, ind = torch.topk(logits, k)
    out = torch.full_like(logits, float("-inf"))
    return out.scatter(1, ind, val)


def gumbel_argmax(logits: Tensor, uniform: Tensor, temperature: float) -> Tensor:
    """utils.py:65-76 with the uniform_(0,1) draw injected."""
    g = -torch.log(-torch.log(uniform + 1e-20) + 1e-20)
    return (logits / temperature + g).argmax(dim=-1)


def mask_out_after_eos(t: Tensor, eos_id: int, keep_eos: bool) -> Tensor:
    """utils.py:86-93."""
    m = (t == eos_id).float()
    if keep_eos:
        m = F.pad(m, (1, -1))
    return t.masked_fill(m.cumsum(dim=-1) > 0, -1)


def generate(sd: Dict[str, Tensor], spec: ModelSpec, conditioning_ids: Sequence[Tensor],
             max_time_steps: int, uniforms: Tensor, pred_ids: Optional[Tensor] = None,
             temperature: float = 1.0, filter_thres: float = 0.9,
             allow_eos_in_output: bool = False, include_eos_in_output: bool = False) -> Tensor:
    """TokenConditionedTransformerWrapper.generate open_musiclm.py:253-326.

    Full re-forward per sampled id (no KV cache), exactly like the reference.
    ``uniforms[step]`` is the [B, V+1] uniform draw of sampling step ``step``.
    allow_eos_in_output: the eos logit survives on the LAST quantizer of a time step only (:309-313); sampling goes on after an
    eos (the loop never stops early: a sampled eos id is embedded like any id, with the offset aliasing of :126-130), and
    everything behind the first eos -- the eos itself too unless include_eos_in_output -- leaves as -1 (:321-322, utils.py:86-93).
    """
    b = conditioning_ids[0].shape[0]
    cond = [append_eos(t.reshape(b, -1).long(), e) for t, e in zip(conditioning_ids, spec.eos_ids)]
    q = spec.token_sequences[-1].num_quantizers
    if pred_ids is None:
        init, sampled = 0, torch.empty(b, 0, dtype=torch.long)
    else:
        init, sampled = pred_ids.shape[1], pred_ids.reshape(b, -1).long()
    step = 0
    for _t in range(init, max_time_steps):
        for _ind in range(q):
            lg = token_conditioned_forward(sd, spec, cond + [sampled], None, only_final=True)[-1]
            last = lg[:, -1].clone()
            if not allow_eos_in_output or _ind != q - 1:
                last[:, -1] = float("-inf")                                   # :309-313
            nxt = gumbel_argmax(top_k_filter(last, filter_thres), uniforms[step], temperature)
            sampled = torch.cat([sampled, nxt[:, None]], dim=-1)
            step += 1
    sampled = mask_out_after_eos(sampled, spec.eos_ids[-1], keep_eos=include_eos_in_output)
    return sampled.reshape(b, -1, q)


# ----------------------------------------------------------------------------
# quantizers (rows 16-17 of SURVEY §8a)
# ----------------------------------------------------------------------------

def nearest_code(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """argmin_c sum_d (x_d - e_{c,d})^2 in fp32, d accumulated in index order
    with one fp32 rounding per add (no FMA contraction), ties -> lowest index.
    The k-means assign form (kmeans_assign below): pinned bit for bit against
    sklearn.MiniBatchKMeans.predict (tests/golden/kmeans_assign.npz and at 768 x 1024)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    n, d = x.shape
    dist = np.zeros((n, cb.shape[0]), dtype=np.float32)
    for j in range(d):
        diff = (x[:, j:j + 1] - cb[None, :, j]).astype(np.float32)
        dist = (dist + (diff * diff).astype(np.float32)).astype(np.float32)
    return dist.argmin(axis=1).astype(np.int64)


def cdist_form(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """[n, C] fp32 distances in the form of vector-quantize-pytorch's EuclideanCodebook (`dist = -cdist(flatten, embed)`
    -- torch.cdist in the 1.2-1.5 releases, the library's own expanded-form `cdist` later; setup.py:31 allows both):

        dist(c) = sqrt(max((x2 + e2_c) - 2 * xy_c, 0)),   x2 = sum_d x_d^2,  e2_c = sum_d e_{c,d}^2,  xy_c = sum_d x_d e_{c,d}

    every product and sum rounded to fp32 separately, d in index order (the summation order of a BLAS is its own; on
    inputs whose products and partial sums are exactly representable -- the pinning tests -- every order gives these bits)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    cb = np.ascontiguousarray(codebook, dtype=np.float32)
    n, d = x.shape
    x2 = np.zeros((n, 1), dtype=np.float32)
    e2 = np.zeros((1, cb.shape[0]), dtype=np.float32)
    xy = np.zeros((n, cb.shape[0]), dtype=np.float32)
    for j in range(d):
        xj, ej = x[:, j:j + 1], cb[None, :, j]
        x2 = (x2 + (xj * xj).astype(np.float32)).astype(np.float32)
        e2 = (e2 + (ej * ej).astype(np.float32)).astype(np.float32)
        xy = (xy + (xj * ej).astype(np.float32)).astype(np.float32)
    d2 = ((x2 + e2).astype(np.float32) - (np.float32(2.0) * xy).astype(np.float32)).astype(np.float32)
    return np.sqrt(np.maximum(d2, np.float32(0.0))).astype(np.float32)


def nearest_code_cdist(x: np.ndarray, codebook: np.ndarray) -> np.ndarray:
    """The library's pick: argmax_c of -cdist_form (torch.argmax: first maximum) == first minimum of the distance.
    The square root is part of the definition: distances that round to the same root are a tie for the lowest index."""
    return cdist_form(x, codebook).argmin(axis=1).astype(np.int64)


def rvq_encode(x: np.ndarray, codebooks: np.ndarray) -> np.ndarray:
    """Residual VQ eval path (clap_quantized.py:75-87 -> ResidualVQ.forward, third-party,
    un-vendored: vector-quantize-pytorch>=1.2.2, setup.py:31): per stage the code nearest to the running residual in
    the library's distance form (nearest_code_cdist), then r <- r - e_idx.  Pinned against torch.cdist (an installed
    third-party implementation of that form) on exactly-representable inputs incl. engineered ties and root-merged
    near-ties (tests/test_oracle_golden.py); on general fp32 inputs the ids of any two implementations -- the library on
    two BLAS back ends included -- differ wherever the summation order decides a near-tie.

    codebooks: [n_q, codebook_size, dim].  Returns indices [n, n_q] (the reference
    then rearranges 'n 1 c -> n c 1', :86)."""
    r = np.ascontiguousarray(x, dtype=np.float32).copy()
    out = np.zeros((r.shape[0], codebooks.shape[0]), dtype=np.int64)
    for s in range(codebooks.shape[0]):
        idx = nearest_code_cdist(r, codebooks[s])
        out[:, s] = idx
        r = (r - codebooks[s][idx].astype(np.float32)).astype(np.float32)
    return out


def rvq_fit_step(state: Dict[str, Tensor], x: Tensor, *, decay: float = 0.95, eps: float = 1e-5, kmeans_iters: int = 10,
                 threshold_dead: float = 0.0, init_picks: Optional[Sequence[Tensor]] = None,
                 expire_picks: Optional[Sequence[Tensor]] = None) -> Tuple[Tensor, float]:
    """One training-mode pass of the CLAP residual VQ: what ClapRVQTrainer.train_step (trainer.py:689-736) triggers through
    ClapQuantized.quantize(embeds, return_rvq_loss=True) with rq.train(True) (clap_quantized.py:75-84).  The arithmetic lives
    in the un-vendored, un-pinned vector-quantize-pytorch (setup.py:31 ">=1.2.2"; ctor at clap_quantized.py:38-46: euclidean
    codebook, kmeans_init=True, kmeans_iters=10 (library default), decay=0.95, commitment_weight=0, threshold_ema_dead_code):
    PARITY UNPINNED -- this restates the library's published algorithm (1.6-era EuclideanCodebook):

      per layer, on the running residual r:
        first batch: k-means (initial means = `init_picks[s]` rows of r -- the library draws randperm(n)[:K]; Lloyd iterations
                     keep a mean whose bucket is empty), embed = means, cluster_size = bucket counts, embed_avg = embed * counts
        idx = nearest code (the library's -cdist form, first maximum: nearest_code_cdist above);  q = embed[idx]  (codes BEFORE the update)
        EMA:  cluster_size <- d cs + (1-d) counts;  embed_avg <- d avg + (1-d) sums;
              embed <- embed_avg / ((cs + eps) / (sum cs + K eps) * sum cs)
        dead codes (threshold_dead > 0): codes with cs < threshold are re-seeded from rows `expire_picks[s]` of r, cs <- threshold,
              embed_avg <- sample * threshold
        r <- r - q
      loss = mse(sum of q, x)

    state: embed [S, K, D], embed_avg [S, K, D], cluster_size [S, K] (fp32) and initted [S] (bool), updated in place.
    Returns (indices [n, S] int64, loss)."""
    emb, avg, cs, initted = state["embed"], state["embed_avg"], state["cluster_size"], state["initted"]
    S, K, D = emb.shape
    r = x.detach().to(torch.float32).clone()
    n = r.shape[0]
    out = torch.zeros(n, S, dtype=torch.int64)
    qsum = torch.zeros_like(r)

    def bucket(rr, codes):
        idx = torch.from_numpy(nearest_code_cdist(rr.numpy(), codes.numpy()))
        counts = torch.bincount(idx, minlength=K).to(torch.float32)
        sums = torch.zeros(K, D, dtype=torch.float32).index_add_(0, idx, rr)
        return idx, counts, sums

    for s in range(S):
        if not bool(initted[s]):
            means = r[init_picks[s]].clone()
            counts = torch.zeros(K)
            for _ in range(kmeans_iters):
                _, counts, sums = bucket(r, means)
                means = torch.where((counts > 0)[:, None], sums / counts.clamp(min=1)[:, None], means)
            emb[s] = means
            cs[s] = counts
            avg[s] = means * counts[:, None]
            initted[s] = True
        idx, counts, sums = bucket(r, emb[s])
        q = emb[s][idx].clone()
        cs[s] = cs[s] * decay + counts * (1 - decay)
        avg[s] = avg[s] * decay + sums * (1 - decay)
        tot = cs[s].sum()
        smoothed = (cs[s] + eps) / (tot + K * eps) * tot
        emb[s] = avg[s] / smoothed[:, None]
        if threshold_dead > 0:
            dead = cs[s] < threshold_dead
            if bool(dead.any()):
                samp = r[expire_picks[s]]
                emb[s][dead] = samp[dead]
                cs[s][dead] = threshold_dead
                avg[s][dead] = samp[dead] * threshold_dead
        out[:, s] = idx
        qsum += q
        r = r - q
    loss = float(((qsum - x.to(torch.float32)) ** 2).mean())
    return out, loss


def kmeans_assign(x: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """hf_hubert_kmeans.py:78-87 assign step (MiniBatchKMeans.predict): nearest centroid."""
    return nearest_code(x, centroids)


# ----------------------------------------------------------------------------
# helpers for tests / bench
# ----------------------------------------------------------------------------

def init_state_dict(spec: ModelSpec, seed: int = 0, dtype=torch.float32) -> Dict[str, Tensor]:
    """Random parameters with the reference's key schema and init *distributions*
    (randn start tokens / logit weights, N(0,1) embeddings, kaiming-uniform linears).
    Not bit-identical to the reference's init stream — goldens carry real reference weights."""
    g = torch.Generator().manual_seed(seed)
    d, h, dh = spec.dim, spec.heads, spec.dim_head
    sd: Dict[str, Tensor] = {}

    def lin(out_f, in_f):
        bound = 1.0 / math.sqrt(in_f)
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound

    for i, s in enumerate(spec.token_sequences):
        sd[f"start_tokens.{i}"] = torch.randn(d, generator=g)
        sd[f"logit_weights.{i}"] = torch.randn(s.num_quantizers, s.codebook_size + 1, d, generator=g)
        sd[f"embeddings.{i}.weight"] = torch.randn((s.codebook_size + 1) * s.num_quantizers, d, generator=g)
    p = "transformer."
    if spec.relative_position_bias_type == "continuous":
        hd = d // 2
        sd[p + "rel_pos_bias.net.0.0.weight"] = lin(hd, 1)
        sd[p + "rel_pos_bias.net.0.0.bias"] = (torch.rand(hd, generator=g) * 2 - 1)
        for k in (1, 2):
            sd[p + f"rel_pos_bias.net.{k}.0.weight"] = lin(hd, hd)
            sd[p + f"rel_pos_bias.net.{k}.0.bias"] = (torch.rand(hd, generator=g) * 2 - 1) / math.sqrt(hd)
        sd[p + "rel_pos_bias.net.3.weight"] = lin(h, hd)
        sd[p + "rel_pos_bias.net.3.bias"] = (torch.rand(h, generator=g) * 2 - 1) / math.sqrt(hd)
    elif spec.relative_position_bias_type == "t5":
        sd[p + "rel_pos_bias.relative_attention_bias.weight"] = torch.randn(32, h, generator=g)
    for l in range(spec.depth):
        a = f"{p}layers.{l}.0."
        sd[a + "q_scale"] = 1 + 0.1 * torch.randn(dh, generator=g)
        sd[a + "k_scale"] = 1 + 0.1 * torch.randn(dh, generator=g)
        sd[a + "norm.gamma"] = 1 + 0.1 * torch.randn(d, generator=g)
        sd[a + "norm.beta"] = torch.zeros(d)
        sd[a + "to_q.weight"] = lin(h * dh, d)
        sd[a + "to_kv.weight"] = lin(2 * dh, d)
        sd[a + "to_out.0.weight"] = lin(d, h * dh)
        f_ = f"{p}layers.{l}.2."
        if spec.use_conv_ff:
            inner = int(d * 2 * 4 / 3)
            sd[f_ + "0.gamma"] = 1 + 0.1 * torch.randn(d, generator=g)
            sd[f_ + "0.beta"] = torch.zeros(d)
            sd[f_ + "1.weight"] = lin(2 * inner, d)
            sd[f_ + "2.ds_conv.weight"] = (torch.rand(2 * inner, 1, 3, generator=g) * 2 - 1) / math.sqrt(3)
            sd[f_ + "4.gamma"] = 1 + 0.1 * torch.randn(inner, generator=g)
            sd[f_ + "4.beta"] = torch.zeros(inner)
            sd[f_ + "6.weight"] = lin(d, inner)
        else:
            inner = d * 4
            sd[f_ + "0.gamma"] = 1 + 0.1 * torch.randn(d, generator=g)
            sd[f_ + "0.beta"] = torch.zeros(d)
            sd[f_ + "1.weight"] = lin(2 * inner, d)
            sd[f_ + "3.gamma"] = 1 + 0.1 * torch.randn(inner, generator=g)
            sd[f_ + "3.beta"] = torch.zeros(inner)
            sd[f_ + "5.weight"] = lin(d, inner)
    sd[p + "norm.gamma"] = 1 + 0.1 * torch.randn(d, generator=g)
    sd[p + "norm.beta"] = torch.zeros(d)
    return {k: v.to(dtype) for k, v in sd.items()}


def synthetic_ids(spec: ModelSpec, batch: int, lengths: Sequence[int], seed: int = 1234) -> List[Tensor]:
    """SURVEY §8d synthetic inputs: ids ~ U{0..V-1}; ``lengths[i]`` = time steps of sequence i."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for s, t in zip(spec.token_sequences, lengths):
        shape = (batch, t, s.num_quantizers) if s.num_quantizers > 1 else (batch, t)
        out.append(torch.randint(0, s.codebook_size, shape, generator=g))
    return out
