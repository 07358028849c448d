"""Host-side helpers with the reference's names and semantics (reference open_musiclm/utils.py).

These are the integer / bookkeeping pieces around the hot path (mask construction, eos handling, sampling
filters).  They operate on tiny id tensors and run as ordinary torch ops on whatever device the ids live
on; the heavy arithmetic (embedding gather, trunk, heads, loss, sampler) is in libomlm_hip.so.
"""
from __future__ import annotations

import os
import shutil
from pathlib import Path

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.utils.rnn import pad_sequence


def beartype_jit(func):
    """utils.py:13-15.  beartype is optional here: enabled only if importable and USE_BEARTYPE=1."""
    if os.environ.get('USE_BEARTYPE', '0') == '1':
        try:
            from beartype import beartype
            return beartype(func)
        except ImportError:
            pass
    return func


def exists(val):
    return val is not None


def default(val, d):
    return val if exists(val) else d


def ceil_div(numer, denom):
    return (numer + denom - 1) // denom


def remainder_needed_until_multiple(n, mult):
    return (ceil_div(n, mult) * mult) - n


def round_down_nearest_multiple(val, mult):
    return (val // mult) * mult


def curtail_to_multiple(t, mult):
    return t[..., :round_down_nearest_multiple(t.shape[-1], mult)]


def eval_decorator(fn):
    def inner(model, *args, **kwargs):
        was_training = model.training
        model.eval()
        out = fn(model, *args, **kwargs)
        model.train(was_training)
        return out
    return inner


def generate_mask_with_prob(shape, mask_prob, device):
    """utils.py:49-56: forgetful causal mask -- int(seq*p) key positions per row chosen by top-k of randn, position 0 kept."""
    n_keys = shape[-1]
    n_drop = min(int(n_keys * mask_prob), n_keys - 1)
    scores = torch.randn(shape, device=device)
    scores[:, 0] = torch.finfo(scores.dtype).min          # position 0 can never be among the top scores
    keep = torch.ones(shape, device=device, dtype=torch.bool)
    if n_drop > 0:
        keep.scatter_(1, scores.topk(n_drop, dim=-1).indices, False)
    return keep


def grad_shrink(t, alpha=0.1):
    """utils.py:60-61 (the engine applies the alpha factor in the trunk backward instead)."""
    return t * alpha + t.detach() * (1 - alpha)


def log(t, eps=1e-20):
    return torch.log(t + eps)


def l2norm(t):
    return F.normalize(t, dim=-1)


def gumbel_noise(t):
    u = torch.zeros_like(t).uniform_(0, 1)
    return -log(-log(u))


def gumbel_sample(t, temperature=1., dim=-1):
    return ((t / temperature) + gumbel_noise(t)).argmax(dim=dim)


def top_k(logits, thres=0.5):
    k = max(int((1 - thres) * logits.shape[-1]), 1)
    kept_val, kept_idx = logits.topk(k, dim=-1)
    filtered = logits.new_full(logits.shape, float('-inf'))
    return filtered.scatter(1, kept_idx, kept_val)


def mask_out_after_eos_id(t, eos_id, mask_value=-1, keep_eos=True):
    hit = (t == eos_id)
    if keep_eos:                                            # masking starts one position after the eos itself
        hit = F.pad(hit, (1, -1))
    seen = hit.to(torch.int32).cumsum(dim=-1) > 0
    return torch.where(seen, torch.full_like(t, mask_value), t)


def all_rows_have_eos_id(t, eos_id):
    return torch.any(t == eos_id, dim=-1).all()


def prob_mask_like(shape, prob, device):
    if prob == 1:
        return torch.ones(shape, device=device, dtype=torch.bool)
    elif prob == 0:
        return torch.zeros(shape, device=device, dtype=torch.bool)
    return torch.zeros(shape, device=device).float().uniform_(0, 1) < prob


def append_eos_id(ids, eos_id):
    b, device = ids.shape[0], ids.device
    eos_ids = torch.full((b, 1), eos_id, device=device, dtype=torch.long)
    return torch.cat((ids, eos_ids), dim=-1)


def batch_unique_consecutive(t, pad_value=0.):
    unique_arr = [torch.unique_consecutive(el) for el in t.unbind(dim=0)]
    return pad_sequence(unique_arr, batch_first=True, padding_value=pad_value)


def get_embeds(embeddings: nn.Embedding, codes: torch.Tensor, pad_id=-1, return_mask=False, mask_pad_pos_to=0):
    """utils.py:126-143.  Kept for API parity; TokenConditionedTransformer uses the fused gather kernel."""
    pad_mask = codes == pad_id
    codes_without_pad = codes.masked_fill(pad_mask, 0)
    embeds = embeddings(codes_without_pad)
    if exists(mask_pad_pos_to):
        embeds = embeds.masked_fill(pad_mask.unsqueeze(-1), mask_pad_pos_to)
    if return_mask:
        return embeds, ~pad_mask
    return embeds


def int16_to_float32(x):
    return (x / 32767.0).type(torch.float32)


def float32_to_int16(x):
    x = torch.clamp(x, min=-1., max=1.)
    return (x * 32767.).type(torch.int16)


def zero_mean_unit_var_norm(x):
    return (x - x.mean(dim=-1, keepdim=True)) / torch.sqrt(x.var(dim=-1, keepdim=True) + 1e-7)


def prepare_audio(data, sample_hz, target_sample_hz, normalize=True, target_length_seconds=None):
    """utils.py:157-166.  Resampling needs torchaudio (audio front-end, outside the hot path); equal rates need nothing
    (torchaudio.functional.resample returns its input unchanged when orig_freq == new_freq)."""
    if int(sample_hz) == int(target_sample_hz):
        def resample(x, *_a, **_k):
            return x
    else:
        try:
            from torchaudio.functional import resample
        except ImportError as e:  # pragma: no cover
            raise ImportError("prepare_audio needs torchaudio to resample (audio front-end is outside the MI355X hot path)") from e
    if data.shape[0] > 1:
        data = torch.mean(data, dim=0).unsqueeze(0)
    if normalize:
        data = zero_mean_unit_var_norm(data)
    if exists(target_length_seconds) and data.shape[1] > target_length_seconds * sample_hz:
        data = data[:, :int(target_length_seconds * sample_hz)]
    audio_for_wav2vec = resample(data, sample_hz, target_sample_hz)
    return int16_to_float32(float32_to_int16(audio_for_wav2vec))


def copy_file_to_folder(file_path: str, folder_path: str):
    config_file = Path(file_path)
    folder = Path(folder_path)
    shutil.copy(str(config_file), str(folder / config_file.name))
