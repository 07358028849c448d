"""TokenConditionedTransformer, its training / sampling wrapper, the three stages and MusicLM, with the
reference's public API (reference open_musiclm/open_musiclm.py) on top of the MI355X engine.

Host responsibilities kept here (all on tiny integer tensors): eos append, label construction, key-mask
construction incl. the forgetful mask, loss weighting bookkeeping, the AR sampling loop and the
sliding-window hierarchical decode.  Everything floating point goes through ``engine`` -> libomlm_hip.so.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import nn
from tqdm import tqdm

from . import decode, engine, ops
from .transformer import Transformer
from .utils import (append_eos_id, batch_unique_consecutive, beartype_jit, default, eval_decorator, exists,
                    float32_to_int16, generate_mask_with_prob, int16_to_float32, mask_out_after_eos_id)


@dataclass
class TokenSequenceInfo():
    """open_musiclm.py:23-30"""
    codebook_size: int
    num_quantizers: int
    unique_consecutive: bool


# Test hook: callable (n_draws, batch, V + 1) -> [n_draws, batch, V + 1] uniforms consumed by `generate` in place of the device
# RNG when no `uniforms` argument is given (the golden test of MusicLM.forward replays the reference's draws through it).
UNIFORM_SOURCE = None


def _flat(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(t.shape[0], -1)


class TokenConditionedTransformer(nn.Module):
    """open_musiclm.py:33-215.  Parameters and state_dict keys are identical to the reference:
    start_tokens.{i}, logit_weights.{i}, embeddings.{i}.weight, [absolute_position_embeddings.{i}.weight],
    transformer.*"""

    def __init__(self, *, token_sequences: List[TokenSequenceInfo], dim, depth, heads=8, attn_dropout=0.,
                 ff_dropout=0.1, has_condition=False, cond_as_self_attn_prefix=False, cond_drop_prob=0.5,
                 grad_shrink_alpha=0.1, use_absolute_position_embeddings=False,
                 max_absolute_position_embeddings=262, precision: Optional[str] = None, **kwargs):
        super().__init__()
        if len(token_sequences) > 4:
            raise ValueError("at most 4 token sequences are supported by the fused gather kernel")
        self.token_sequences = token_sequences
        self.dim = dim
        self.has_condition = has_condition
        self.cond_drop_prob = cond_drop_prob
        self.use_absolute_position_embeddings = use_absolute_position_embeddings
        self.precision = precision            # None -> engine.default_precision() ($OMLM_PRECISION, default bf16)

        self.start_tokens = torch.nn.ParameterList()
        self.logit_weights = torch.nn.ParameterList()
        self.embeddings = torch.nn.ModuleList()
        self.absolute_position_embeddings = torch.nn.ModuleList() if use_absolute_position_embeddings else None
        self.eos_ids = []
        for sequence in token_sequences:
            self.start_tokens.append(nn.Parameter(torch.randn(dim)))
            self.eos_ids.append(sequence.codebook_size)
            rows = sequence.codebook_size + 1
            self.embeddings.append(nn.Embedding(rows * sequence.num_quantizers, dim))
            self.logit_weights.append(nn.Parameter(torch.randn(sequence.num_quantizers, rows, dim)))
            if use_absolute_position_embeddings:
                self.absolute_position_embeddings.append(nn.Embedding(max_absolute_position_embeddings, dim))

        self.transformer = Transformer(dim=dim, depth=depth, heads=heads, attn_dropout=attn_dropout,
                                       ff_dropout=ff_dropout,
                                       cross_attend=has_condition and not cond_as_self_attn_prefix,
                                       cond_as_self_attn_prefix=cond_as_self_attn_prefix,
                                       grad_shrink_alpha=grad_shrink_alpha, **kwargs)
        self.transformer.__dict__["_omlm_owner"] = self
        engine.tag_parameters(self, precision)      # the fused optimizer reads the precision mode off the parameters (16-bit shadow type, loss scale)

    def __setstate__(self, state):
        """pickle / torch.load(model) / copy.deepcopy (EMA copies) / spawn: the parameter -> model registry of engine.tag_parameters is keyed by
        object identity, so a reconstructed model registers its own parameters again (the precision tag itself travels in the Parameter's
        __dict__); without this FusedAdam found 'fp16 parameters without their model' on such a copy."""
        super().__setstate__(state)
        self.transformer.__dict__["_omlm_owner"] = self
        self.__dict__.pop("_omlm_prepared", None)             # operand images of the ORIGINAL's parameters
        engine.tag_parameters(self, self.precision)

    @property
    def device(self):
        return next(self.parameters()).device

    def _precision(self) -> str:
        return self.precision or engine.default_precision()

    def forward(self, *, all_token_ids: List[torch.Tensor], self_attn_mask=None, cond_drop_prob=None,
                return_only_final_seq_logits=False):
        """Returns one [B, n_i, codebook_size+1] fp32 logits tensor per sequence (None for skipped ones).
        cond_drop_prob is accepted and ignored exactly like the reference (no text conditioning)."""
        assert len(all_token_ids) == len(self.token_sequences) == len(self.embeddings)
        ids = [_flat(t).to(self.device) for t in all_token_ids]
        prec = self._precision()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            outs = engine.LogitsFunction.apply(self, ids, self_attn_mask, bool(return_only_final_seq_logits), prec,
                                               *self.parameters())
            it = iter(outs)
            n = len(self.token_sequences)
            return [next(it) if (not return_only_final_seq_logits or s == n - 1) else None for s in range(n)]
        bufs, lay, _ = engine.run_forward(self, ids, self_attn_mask, bool(return_only_final_seq_logits), False, prec)
        return engine.logits_views(self, lay, bufs)

    def forward_with_cond_scale(self, *args, cond_scale=3, **kwargs):
        """open_musiclm.py:192-215: a no-op around forward() without text conditioning."""
        logits = self.forward(*args, cond_drop_prob=0., **kwargs)
        if cond_scale == 1 or not self.has_condition:
            return logits
        null_logits = self.forward(*args, cond_drop_prob=1., **kwargs)
        return [None if a is None else b + (a - b) * cond_scale for a, b in zip(logits, null_logits)]

    # ---- fast paths used by the wrapper ---------------------------------------------------------------
    def last_logits(self, ids: List[torch.Tensor]) -> torch.Tensor:
        """[B, ldV] logits of the final position of the final sequence only (AR decode step, no autograd)."""
        bufs, _, _ = engine.run_forward(self, ids, None, True, False, self._precision(), final_rows_only=True)
        return bufs[-1]

    def loss_and_logits(self, ids, labels, self_attn_mask, loss_weights, ignore_negative=None, all_logits=True):
        """ignore_negative[s]: labels < 0 of sequence s are ignore_index rows AND leave the loss normaliser (the padded labels of
        a unique_consecutive sequence, open_musiclm.py:396-404).  all_logits=False: the heads of zero-weight sequences are not
        evaluated (their logits come back as None)."""
        ign = tuple(bool(v) for v in ignore_negative) if ignore_negative is not None else (False,) * len(labels)
        return engine.LossFunction.apply(self, ids, labels, self_attn_mask, tuple(loss_weights), ign, self._precision(),
                                         bool(all_logits), *self.parameters())


_GENERATE_MASK_ORIG = generate_mask_with_prob


@beartype_jit
class TokenConditionedTransformerWrapper(nn.Module):
    """open_musiclm.py:219-410."""

    def __init__(self, *, transformer: TokenConditionedTransformer, pad_id=-1, unique_consecutive=True,
                 cross_entropy_loss_weights: Optional[List[float]] = None, mask_prob=0.15):
        super().__init__()
        self.transformer = transformer
        self.token_sequences = transformer.token_sequences
        self.unique_consecutive = unique_consecutive
        self.pad_id = pad_id
        self.cross_entropy_loss_weights = default(cross_entropy_loss_weights, [1 for _ in self.token_sequences])
        self.eos_ids = transformer.eos_ids
        self.mask_prob = mask_prob
        assert len(self.token_sequences) == len(self.eos_ids) == len(self.cross_entropy_loss_weights)

    @property
    def device(self):
        return next(self.parameters()).device

    @eval_decorator
    @torch.no_grad()
    def generate(self, *, conditioning_token_ids: List[torch.Tensor], pred_token_ids: Optional[torch.Tensor] = None,
                 max_time_steps=512, filter_thres=0.9, temperature=1., include_eos_in_output=False,
                 append_eos_to_conditioning_tokens=True, allow_eos_in_output=False, uniforms=None, **kwargs):
        """AR sampling (open_musiclm.py:253-326).  Every step re-runs the full causal forward over the grown
        sequence like the reference (results are identical to a KV-cached decode because the stack is strictly
        causal); only the last position's logits are formed, and eos suppression + top-k + Gumbel-argmax run in
        one sampler kernel.  ``uniforms`` ([steps, B, V+1]) injects the uniform draws (tests)."""
        assert len(conditioning_token_ids) == len(self.token_sequences) - 1
        batch, device = conditioning_token_ids[0].shape[0], self.device
        cond = [t.to(device) for t in conditioning_token_ids]
        if exists(pred_token_ids):
            assert pred_token_ids.shape[0] == batch
            first_step = pred_token_ids.shape[1]
            sampled = _flat(pred_token_ids).to(device).long()
        else:
            first_step = 0
            sampled = torch.empty((batch, 0), device=device, dtype=torch.long)
        pred_info, pred_eos_id = self.token_sequences[-1], self.eos_ids[-1]

        for i, info in enumerate(self.token_sequences[:-1]):
            if info.unique_consecutive:
                cond[i] = batch_unique_consecutive(cond[i], pad_value=self.pad_id)
        if pred_info.unique_consecutive:
            sampled = batch_unique_consecutive(sampled, pad_value=self.pad_id)
        if append_eos_to_conditioning_tokens:
            cond = [append_eos_id(_flat(t).long(), e) for t, e in zip(cond, self.eos_ids)]

        V1 = pred_info.codebook_size + 1
        k = max(int((1 - filter_thres) * V1), 1)
        Q = pred_info.num_quantizers
        step = 0
        nxt = torch.empty(batch, device=device, dtype=torch.long)
        n_new = max(max_time_steps - first_step, 0) * Q
        use_cache = kwargs.pop('use_cache', True) and decode.supports(self.transformer, 1) and n_new > 0
        if use_cache:
            # KV-cached decode (decode.py): one new row per sampled id instead of the reference's full re-forward.
            # Ids are sampled straight into a [steps, B] buffer that the next decode step reads: no per-step cat / copies.
            rows = sum(t.shape[-1] + 1 for t in cond) + 1 + sampled.shape[-1] + n_new
            n0 = sampled.shape[-1]
            if exists(uniforms):
                U = uniforms[:n_new].to(device).float().contiguous()
            elif UNIFORM_SOURCE is not None:
                U = UNIFORM_SOURCE(n_new, batch, V1).to(device).float().contiguous()
            else:
                U = torch.rand(n_new, batch, V1, device=device)
            forbid = [(not allow_eos_in_output) or (ind != Q - 1) for ind in range(Q)]
            use_graph = kwargs.pop('use_graph', False)
            # the step kernels hold up to MAX_DECODE_BATCH samples per call: larger batches run as consecutive groups (samples are
            # independent; every group streams the weights once per id)
            pieces = []
            group = decode.max_batch(self.transformer, self.transformer._precision())
            for b0 in range(0, batch, group):
                b1 = min(batch, b0 + group)
                dec = decode.CachedDecoder(self.transformer, b1 - b0, rows, self.transformer._precision())
                last = dec.prefill([t[b0:b1] for t in cond] + [sampled[b0:b1]])
                loop = decode.SamplingLoop(dec, last, U[:, b0:b1].contiguous(), n0, n_new, k, temperature, forbid, use_graph=use_graph)
                pieces.append(loop.run().t())                      # [b, n_new]
            sampled = torch.cat((sampled, torch.cat(pieces, dim=0)), dim=-1)
        else:
            if not exists(uniforms) and UNIFORM_SOURCE is not None and n_new > 0:
                uniforms = UNIFORM_SOURCE(n_new, batch, V1)
            for _t in tqdm(range(first_step, max_time_steps), desc='generating predicted tokens'):
                for ind in range(Q):
                    last = self.transformer.last_logits(cond + [sampled])
                    forbid = (not allow_eos_in_output) or (ind != Q - 1)
                    if exists(uniforms):
                        u = uniforms[step].to(device).float().contiguous()
                    else:
                        u = torch.empty(batch, V1, device=device).uniform_(0, 1)
                    ops.sample_topk_gumbel(last, u, nxt, V1, k, temperature, forbid)
                    sampled = torch.cat((sampled, nxt[:, None]), dim=-1)
                    step += 1
        sampled = mask_out_after_eos_id(sampled, pred_eos_id, keep_eos=include_eos_in_output)
        return sampled.reshape(batch, -1, Q)

    # ---- the trainers' optimizer-step path: id / label / mask construction as ONE kernel launch -------------------------------------
    def _fused_prepare_ok(self, all_token_ids, input_has_eos) -> bool:
        """return_loss=True, return_logits=False on GPU tensors without unique_consecutive sequences (shipped configs: False) and with the
        module's own generate_mask_with_prob (tests inject masks by replacing it: they take the torch path).  OMLM_FUSED_PREP=0: off."""
        if input_has_eos or os.environ.get("OMLM_FUSED_PREP", "1") == "0" or generate_mask_with_prob is not _GENERATE_MASK_ORIG:
            return False
        if self.unique_consecutive and any(info.unique_consecutive for info in self.token_sequences):
            return False
        n_tot = sum(int(t.numel() // t.shape[0]) + 1 for t in all_token_ids) + len(all_token_ids) - 1
        return all(t.is_cuda for t in all_token_ids) and n_tot <= 4096

    def _forward_loss_fused(self, all_token_ids):
        """Same arithmetic as _prepare + TokenConditionedTransformer.forward's id flattening + generate_mask_with_prob, in one launch
        (ops.prepare_train_batch); consumes the RNG exactly like the torch path (one randn of the mask's shape).  Labels come back as
        int32 tensors [B, len + 1]."""
        device = self.device
        ids = [_flat(t).to(device).long().contiguous() for t in all_token_ids]
        B = ids[0].shape[0]
        N = sum(t.shape[1] + 1 for t in ids) + len(ids) - 1
        scores, n_drop = None, 0
        if self.mask_prob > 0 and self.training:
            n_drop = min(int(N * self.mask_prob), N - 1)
            scores = torch.randn((B, N), device=device)
        weights = [float(w) for w in self.cross_entropy_loss_weights]
        seqs = self.token_sequences
        ids32, keymask, labels, lens = ops.prepare_train_batch(ids, [int(e) for e in self.eos_ids], [s.num_quantizers for s in seqs],
                                                               [s.codebook_size for s in seqs], self.pad_id, scores, n_drop,
                                                               [True] * len(ids))
        prepared = engine.PreparedIds(ids32, lens)
        ignore = [False] * len(ids)
        if torch.is_grad_enabled():
            loss, *logits = self.transformer.loss_and_logits(prepared, labels, keymask, weights, ignore, False)
        else:
            with torch.enable_grad():
                loss, *logits = self.transformer.loss_and_logits(prepared, labels, keymask, weights, ignore, False)
            loss = loss.detach()
        return loss, [l.transpose(1, 2) if l is not None else None for l in logits], labels

    def _prepare(self, all_token_ids, return_loss, input_has_eos):
        """eos append, labels, last-token drop, key mask (open_musiclm.py:340-376)."""
        batch, device = all_token_ids[0].shape[0], self.device
        ids = [_flat(t).to(device).long() for t in all_token_ids]
        if self.training:
            assert not input_has_eos, "train sequences (from clap, wav2vec, etc.) shouldn't come with an eos token"
        if not input_has_eos:
            ids = [append_eos_id(t, e) for t, e in zip(ids, self.eos_ids)]
        if self.unique_consecutive:
            for i, info in enumerate(self.token_sequences):
                if info.unique_consecutive:
                    ids[i] = batch_unique_consecutive(ids[i], pad_value=self.pad_id)
        labels = None
        if return_loss:
            labels = [t.clone() for t in ids]
            ids[-1] = ids[-1][:, :-1]
        pieces = []
        for i in range(len(ids) - 1):
            live = (ids[i] != self.pad_id) & (ids[i] != self.eos_ids[i])
            ids[i] = ids[i].masked_fill(~live, 0)
            pieces.append(F.pad(live, (1, 0), value=True))            # the sequence's start token is always attended
        mask = torch.cat(pieces, dim=-1) if pieces else torch.empty((batch, 0), device=device, dtype=torch.bool)
        mask = F.pad(mask, (0, ids[-1].shape[-1] + 1), value=True)   # predicted tokens + their start token
        if self.mask_prob > 0 and self.training:
            mask = mask & generate_mask_with_prob(mask.shape, self.mask_prob, device=mask.device)
        return ids, labels, mask

    def forward(self, *, all_token_ids: List[torch.Tensor], return_loss: bool = False, input_has_eos: bool = False,
                return_logits: bool = True, **kwargs):
        """return_logits=False (extension, return_loss=True only): the logits of sequences whose loss weight is 0 are not computed
        and come back as None -- the trainers' optimizer steps read only the loss (trainer.py:428-447)."""
        assert len(all_token_ids) == len(self.token_sequences)
        if return_loss and not return_logits and self._fused_prepare_ok(all_token_ids, input_has_eos):
            return self._forward_loss_fused(all_token_ids)
        ids, labels, mask = self._prepare(all_token_ids, return_loss, input_has_eos)
        if not return_loss:
            return self.transformer(all_token_ids=ids, self_attn_mask=mask, **kwargs)
        weights = [float(w) for w in self.cross_entropy_loss_weights]
        # unique_consecutive sequences carry pad_id labels: F.cross_entropy(ignore_index=pad_id) and num_logits = (labels != pad_id)
        # .sum() in the reference (:396-404); the fused loss ignores negative labels and counts the rest on the device
        ignore = [bool(info.unique_consecutive and self.unique_consecutive) for info in self.token_sequences]
        loss_labels = [lb.masked_fill(lb == self.pad_id, -1) if ig else lb for lb, ig in zip(labels, ignore)]
        if torch.is_grad_enabled():
            loss, *logits = self.transformer.loss_and_logits(ids, loss_labels, mask, weights, ignore, return_logits)
        else:
            with torch.enable_grad():
                loss, *logits = self.transformer.loss_and_logits(ids, loss_labels, mask, weights, ignore, return_logits)
            loss = loss.detach()
        all_logits = [l.transpose(1, 2) if l is not None else None for l in logits]               # 'b n c -> b c n' (:389)
        return loss, all_logits, labels


def create_semantic_transformer(dim=1024, depth=6, clap_codebook_size=1024, semantic_codebook_size=1024,
                                num_clap_quantizers=12, **kwargs):
    """open_musiclm.py:414-428"""
    seqs = [TokenSequenceInfo(clap_codebook_size, num_clap_quantizers, False),
            TokenSequenceInfo(semantic_codebook_size, 1, False)]
    return TokenConditionedTransformer(token_sequences=seqs, dim=dim, depth=depth, **kwargs)


def create_coarse_transformer(dim=512, depth=6, clap_codebook_size=1024, semantic_codebook_size=1024,
                              acoustic_codebook_size=1024, num_clap_quantizers=12, num_coarse_quantizers=4, **kwargs):
    """open_musiclm.py:432-450"""
    seqs = [TokenSequenceInfo(clap_codebook_size, num_clap_quantizers, False),
            TokenSequenceInfo(semantic_codebook_size, 1, False),
            TokenSequenceInfo(acoustic_codebook_size, num_coarse_quantizers, False)]
    return TokenConditionedTransformer(token_sequences=seqs, dim=dim, depth=depth, **kwargs)


def create_fine_transformer(dim=512, depth=6, clap_codebook_size=1024, acoustic_codebook_size=1024,
                            num_clap_quantizers=12, num_coarse_quantizers=4, num_fine_quantizers=8, **kwargs):
    """open_musiclm.py:454-472"""
    seqs = [TokenSequenceInfo(clap_codebook_size, num_clap_quantizers, False),
            TokenSequenceInfo(acoustic_codebook_size, num_coarse_quantizers, False),
            TokenSequenceInfo(acoustic_codebook_size, num_fine_quantizers, False)]
    return TokenConditionedTransformer(token_sequences=seqs, dim=dim, depth=depth, **kwargs)


def get_or_compute_clap_token_ids(clap_token_ids, clap, conditioning_audio, conditioning_text):
    """open_musiclm.py:476-485"""
    if not exists(clap_token_ids):
        assert exists(conditioning_audio) ^ exists(conditioning_text), "either condition on text or audio"
        assert exists(clap)
        clap_token_ids = clap(text_input=conditioning_text) if exists(conditioning_text) else clap(audio_input=conditioning_audio)
    return clap_token_ids


def get_or_compute_semantic_token_ids(semantic_token_ids, raw_audio, wav2vec):
    """open_musiclm.py:489-495"""
    if not exists(semantic_token_ids):
        assert exists(raw_audio)
        assert exists(wav2vec)
        semantic_token_ids = wav2vec(raw_audio, flatten=False)
    return semantic_token_ids


def get_or_compute_acoustic_token_ids(coarse_token_ids, fine_token_ids, raw_audio, neural_codec, num_coarse_quantizers: int):
    """open_musiclm.py:499-510"""
    if exists(raw_audio):
        assert not exists(coarse_token_ids) and not exists(fine_token_ids), "either provide coarse + fine ids or raw audio"
        assert exists(neural_codec), 'A neural audio codec must be provided if given raw wave for training'
        with torch.no_grad():
            neural_codec.eval()
            _, indices, _ = neural_codec(raw_audio, return_encoded=True)
            coarse_token_ids, fine_token_ids = indices[..., :num_coarse_quantizers], indices[..., num_coarse_quantizers:]
    return coarse_token_ids, fine_token_ids


class _Stage(nn.Module):
    """Shared plumbing of SemanticStage / CoarseStage / FineStage (open_musiclm.py:514-814)."""

    def _make_wrapper(self, transformer, pad_id, unique_consecutive, cross_entropy_loss_weights, mask_prob):
        self.transformer_wrapper = TokenConditionedTransformerWrapper(
            transformer=transformer, pad_id=pad_id, unique_consecutive=unique_consecutive,
            cross_entropy_loss_weights=cross_entropy_loss_weights, mask_prob=mask_prob)

    @property
    def device(self):
        return next(self.parameters()).device

    def _sample(self, conditioning, pred, max_time_steps, filter_thres, temperature, include_eos_in_output,
                append_eos_to_conditioning_tokens, **kwargs):
        return self.transformer_wrapper.generate(
            conditioning_token_ids=conditioning, pred_token_ids=pred, max_time_steps=max_time_steps,
            filter_thres=filter_thres, temperature=temperature, include_eos_in_output=include_eos_in_output,
            append_eos_to_conditioning_tokens=append_eos_to_conditioning_tokens, **kwargs)


class SemanticStage(_Stage):
    def __init__(self, *, semantic_transformer: TokenConditionedTransformer, wav2vec=None, clap=None, pad_id=-1,
                 unique_consecutive=False, cross_entropy_loss_weights: List[float] = None, mask_prob=0.15):
        super().__init__()
        self.wav2vec, self.clap = wav2vec, clap
        num_semantic_tokens = semantic_transformer.token_sequences[1].codebook_size
        if exists(wav2vec):
            assert wav2vec.codebook_size == num_semantic_tokens, \
                f'num_semantic_tokens on SemanticTransformer must be set to {wav2vec.codebook_size}'
        self._make_wrapper(semantic_transformer, pad_id, unique_consecutive, cross_entropy_loss_weights, mask_prob)

    @eval_decorator
    @torch.no_grad()
    def generate(self, *, conditioning_text=None, conditioning_audio=None, input_audio=None, clap_token_ids=None,
                 semantic_token_ids=None, filter_thres=0.9, temperature=1., max_time_steps=30 * 25,
                 include_eos_in_output=False, append_eos_to_conditioning_tokens=True, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, conditioning_audio, conditioning_text)
        if exists(semantic_token_ids) or exists(input_audio):
            semantic_token_ids = get_or_compute_semantic_token_ids(semantic_token_ids, input_audio, self.wav2vec)
        else:
            semantic_token_ids = None
        return self._sample([clap_token_ids], semantic_token_ids, max_time_steps, filter_thres, temperature,
                            include_eos_in_output, append_eos_to_conditioning_tokens, **kwargs)

    def forward(self, *, raw_wave_for_clap=None, raw_wave_for_semantic=None, clap_token_ids=None,
                semantic_token_ids=None, return_loss=False, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, raw_wave_for_clap, conditioning_text=None)
        semantic_token_ids = get_or_compute_semantic_token_ids(semantic_token_ids, raw_wave_for_semantic, self.wav2vec)
        return self.transformer_wrapper.forward(all_token_ids=[clap_token_ids, semantic_token_ids],
                                                return_loss=return_loss, **kwargs)


class CoarseStage(_Stage):
    def __init__(self, *, coarse_transformer: TokenConditionedTransformer, wav2vec=None, clap=None, neural_codec=None,
                 pad_id=-1, unique_consecutive=False, cross_entropy_loss_weights: List[float] = None, mask_prob=0.15):
        super().__init__()
        self.wav2vec, self.clap, self.neural_codec = wav2vec, clap, neural_codec
        num_semantic_tokens = coarse_transformer.token_sequences[1].codebook_size
        if exists(wav2vec):
            assert wav2vec.codebook_size == num_semantic_tokens, \
                f'num_semantic_tokens on CoarseTransformer must be set to {wav2vec.codebook_size}'
        self.num_coarse_quantizers = coarse_transformer.token_sequences[-1].num_quantizers
        self._make_wrapper(coarse_transformer, pad_id, unique_consecutive, cross_entropy_loss_weights, mask_prob)

    @eval_decorator
    @torch.no_grad()
    def generate(self, *, semantic_token_ids, coarse_token_ids=None, conditioning_text=None, conditioning_audio=None,
                 clap_token_ids=None, filter_thres=0.9, temperature=1., max_time_steps=10 * 600,
                 include_eos_in_output=False, append_eos_to_conditioning_tokens=True, reconstruct_wave=False, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, conditioning_audio, conditioning_text)
        sampled = self._sample([clap_token_ids, semantic_token_ids], coarse_token_ids, max_time_steps, filter_thres,
                               temperature, include_eos_in_output, append_eos_to_conditioning_tokens, **kwargs)
        if reconstruct_wave:
            assert exists(self.neural_codec)
            wave = self.neural_codec.decode_from_codebook_indices(sampled)
            return wave.squeeze(1)
        return sampled

    def forward(self, *, raw_wave_for_clap=None, raw_wave_for_semantic=None, raw_wave_for_acoustic=None,
                clap_token_ids=None, semantic_token_ids=None, coarse_token_ids=None, return_loss=False, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, raw_wave_for_clap, conditioning_text=None)
        semantic_token_ids = get_or_compute_semantic_token_ids(semantic_token_ids, raw_wave_for_semantic, self.wav2vec)
        coarse_token_ids, _ = get_or_compute_acoustic_token_ids(coarse_token_ids, None, raw_wave_for_acoustic,
                                                                self.neural_codec, self.num_coarse_quantizers)
        return self.transformer_wrapper.forward(all_token_ids=[clap_token_ids, semantic_token_ids, coarse_token_ids],
                                                return_loss=return_loss, **kwargs)


class FineStage(_Stage):
    def __init__(self, *, fine_transformer: TokenConditionedTransformer, clap=None, neural_codec=None, pad_id=-1,
                 unique_consecutive=False, cross_entropy_loss_weights: List[float] = None, mask_prob=0.15):
        super().__init__()
        self.clap, self.neural_codec = clap, neural_codec
        self.num_coarse_quantizers = fine_transformer.token_sequences[1].num_quantizers
        self._make_wrapper(fine_transformer, pad_id, unique_consecutive, cross_entropy_loss_weights, mask_prob)

    @eval_decorator
    @torch.no_grad()
    def generate(self, *, coarse_token_ids, fine_token_ids=None, conditioning_text=None, conditioning_audio=None,
                 clap_token_ids=None, filter_thres=0.9, temperature=1., max_time_steps=3 * 600,
                 include_eos_in_output=False, append_eos_to_conditioning_tokens=True, reconstruct_wave=False, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, conditioning_audio, conditioning_text)
        sampled = self._sample([clap_token_ids, coarse_token_ids], fine_token_ids, max_time_steps, filter_thres,
                               temperature, include_eos_in_output, append_eos_to_conditioning_tokens, **kwargs)
        if reconstruct_wave:
            assert exists(self.neural_codec)
            wave = self.neural_codec.decode_from_codebook_indices(torch.cat((coarse_token_ids, sampled), dim=-1))
            return wave.squeeze(1)
        return sampled

    def forward(self, *, raw_wave_for_clap=None, raw_wave_for_acoustic=None, clap_token_ids=None,
                coarse_token_ids=None, fine_token_ids=None, return_loss=False, **kwargs):
        clap_token_ids = get_or_compute_clap_token_ids(clap_token_ids, self.clap, raw_wave_for_clap, conditioning_text=None)
        coarse_token_ids, fine_token_ids = get_or_compute_acoustic_token_ids(
            coarse_token_ids, fine_token_ids, raw_wave_for_acoustic, self.neural_codec, self.num_coarse_quantizers)
        assert exists(coarse_token_ids) and exists(fine_token_ids)
        return self.transformer_wrapper.forward(all_token_ids=[clap_token_ids, coarse_token_ids, fine_token_ids],
                                                return_loss=return_loss, **kwargs)


def _windows(t: torch.Tensor, size: int, step: int):
    """[B, T, ...] -> list of [B, size, ...] windows at stride `step` (tensor.unfold semantics, :958-959)."""
    n = (t.shape[1] - size) // step + 1
    return [t[:, i * step: i * step + size] for i in range(max(n, 0))]


class MusicLM(nn.Module):
    """open_musiclm.py:818-1071: hierarchical semantic -> coarse -> fine sliding-window decode.

    Extensions over the reference signature (all optional, defaults reproduce the reference):
      clap_token_ids  -- bypass the CLAP text tower with pre-quantised conditioning ids (synthetic benchmarks);
      return_tokens   -- return (semantic, coarse, fine) id tensors instead of decoding a waveform.
    ``generate`` is an alias of ``forward``."""

    def __init__(self, *, wav2vec=None, clap=None, neural_codec=None, semantic_transformer: TokenConditionedTransformer,
                 coarse_transformer: TokenConditionedTransformer, fine_transformer: TokenConditionedTransformer):
        super().__init__()
        assert semantic_transformer.token_sequences[1].codebook_size == coarse_transformer.token_sequences[1].codebook_size
        assert coarse_transformer.token_sequences[2].codebook_size == fine_transformer.token_sequences[2].codebook_size
        assert coarse_transformer.token_sequences[2].num_quantizers == fine_transformer.token_sequences[1].num_quantizers
        self.semantic = SemanticStage(semantic_transformer=semantic_transformer, wav2vec=wav2vec, clap=clap)
        self.coarse = CoarseStage(coarse_transformer=coarse_transformer, wav2vec=wav2vec, clap=clap, neural_codec=neural_codec)
        self.fine = FineStage(fine_transformer=fine_transformer, clap=clap, neural_codec=neural_codec)
        self.wav2vec, self.clap, self.neural_codec = wav2vec, clap, neural_codec

    @property
    def device(self):
        return next(self.parameters()).device

    @eval_decorator
    @torch.no_grad()
    def forward(self, *, text: Optional[List[str]] = None, prime_wave=None, prime_wave_sample_hz=None, output_seconds=8,
                semantic_window_seconds=10, coarse_window_seconds=4, fine_window_seconds=2,
                semantic_steps_per_second=50, acoustic_steps_per_second=75, return_coarse_generated_wave=False,
                mask_out_generated_fine_tokens=False, semantic_sliding_window_step_percent=0.5,
                coarse_sliding_window_step_percent=0.5, fine_sliding_window_step_percent=1,
                clap_token_ids=None, return_tokens=False):
        if not exists(clap_token_ids):
            assert exists(text), 'text needs to be passed in if one of the transformer requires conditioning'
            clap_token_ids = get_or_compute_clap_token_ids(None, self.clap, conditioning_audio=None, conditioning_text=text)

        sem_hz, ac_hz = semantic_steps_per_second, acoustic_steps_per_second
        # audio continuation (open_musiclm.py:896-926)
        prime_coarse_all = prime_fine_all = None
        prime_sem = prime_coarse = prime_fine = None
        sem_adjust = coarse_adjust = fine_adjust = 0
        if exists(prime_wave):
            from .utils import prepare_audio
            assert exists(prime_wave_sample_hz)
            wave_w2v = prepare_audio(prime_wave, prime_wave_sample_hz, self.wav2vec.target_sample_hz, normalize=True,
                                     target_length_seconds=semantic_window_seconds)
            wave_codec = prepare_audio(prime_wave, prime_wave_sample_hz, self.neural_codec.sample_rate, normalize=False,
                                       target_length_seconds=semantic_window_seconds)
            c_sem = get_or_compute_semantic_token_ids(None, wave_w2v, self.wav2vec)
            c_coarse, c_fine = get_or_compute_acoustic_token_ids(
                None, None, wave_codec, self.neural_codec, self.coarse.transformer_wrapper.token_sequences[2].num_quantizers)
            n_sem = int(sem_hz * semantic_window_seconds * (1 - semantic_sliding_window_step_percent))
            n_coarse = int(ac_hz * coarse_window_seconds * (1 - coarse_sliding_window_step_percent))
            n_fine = int(ac_hz * fine_window_seconds * (1 - fine_sliding_window_step_percent))
            prime_coarse_all, prime_fine_all = c_coarse, c_fine
            prime_sem = c_sem[:, -n_sem:] if c_sem.shape[1] >= n_sem else c_sem
            prime_coarse = c_coarse[:, -n_coarse:]
            prime_fine = c_fine[:, -n_fine:] if n_fine > 0 else None
            sem_adjust = n_sem - int(sem_hz * coarse_window_seconds * (1 - coarse_sliding_window_step_percent))
            coarse_adjust = n_coarse - int(ac_hz * fine_window_seconds * (1 - fine_sliding_window_step_percent))
            fine_adjust = n_fine

        # ---- semantic stage (:930-952): one window, then 50%-overlap continuation windows ----
        sem = self.semantic.generate(clap_token_ids=clap_token_ids, semantic_token_ids=prime_sem,
                                     max_time_steps=int(min(output_seconds, semantic_window_seconds) * sem_hz),
                                     include_eos_in_output=False, append_eos_to_conditioning_tokens=True)
        while sem.shape[1] < int(output_seconds * sem_hz):
            keep = int(semantic_window_seconds * sem_hz * (1 - semantic_sliding_window_step_percent))
            nxt = self.semantic.generate(clap_token_ids=clap_token_ids, semantic_token_ids=sem[:, -keep:],
                                         max_time_steps=int(semantic_window_seconds * sem_hz),
                                         include_eos_in_output=False, append_eos_to_conditioning_tokens=True)
            sem = torch.cat([sem, nxt[:, keep:]], dim=1)
        sem_all = sem
        sem = sem[:, sem_adjust:]

        # ---- coarse stage (:956-984): semantic windows of coarse_window*50-1 ids at 50% stride ----
        win = int(coarse_window_seconds * sem_hz - 1)
        coarse = None
        for sem_win in _windows(sem, win, int(win * coarse_sliding_window_step_percent)):
            if exists(coarse):
                keep = int(coarse_window_seconds * ac_hz * (1 - coarse_sliding_window_step_percent))
                cond_coarse = coarse[:, -keep:]
            else:
                keep, cond_coarse = 0, prime_coarse
            pred = self.coarse.generate(clap_token_ids=clap_token_ids, semantic_token_ids=sem_win,
                                        coarse_token_ids=cond_coarse,
                                        max_time_steps=int(coarse_window_seconds * ac_hz), reconstruct_wave=False,
                                        include_eos_in_output=False, append_eos_to_conditioning_tokens=True,
                                        temperature=0.95)
            coarse = pred if not exists(coarse) else torch.cat([coarse, pred[:, keep:]], dim=1)
        if return_coarse_generated_wave:
            return self.neural_codec.decode_from_codebook_indices(coarse).squeeze(1)
        coarse = coarse[:, coarse_adjust:]

        # ---- fine stage (:996-1026): non-overlapping (100% stride) coarse windows ----
        fwin = int(fine_window_seconds * ac_hz)
        fstep = int(fwin * fine_sliding_window_step_percent)
        fine = None
        for coarse_win in _windows(coarse, fwin, fstep):
            if exists(fine):
                keep = int(fwin * (1 - fine_sliding_window_step_percent))
                cond_fine = fine[:, -keep:] if keep > 0 else None
            else:
                keep, cond_fine = 0, prime_fine
            pred = self.fine.generate(clap_token_ids=clap_token_ids, coarse_token_ids=coarse_win, fine_token_ids=cond_fine,
                                      max_time_steps=fwin, reconstruct_wave=False, include_eos_in_output=False,
                                      append_eos_to_conditioning_tokens=True, temperature=0.4)
            fine = pred if not exists(fine) else torch.cat([fine, pred[:, keep:]], dim=1)
        fine = fine[:, fine_adjust:]
        if exists(prime_coarse_all) and exists(prime_fine_all):
            fine = torch.cat([prime_fine_all, fine], dim=1)
            coarse = torch.cat([prime_coarse_all, coarse], dim=1)
        if return_tokens:
            return sem_all, coarse, fine
        acoustic = torch.cat([coarse[:, : fine.shape[1]], fine], dim=-1) if coarse.shape[1] != fine.shape[1] \
            else torch.cat([coarse, fine], dim=-1)
        assert exists(self.neural_codec), "a neural codec is required to decode tokens to a waveform (or pass return_tokens=True)"
        return self.neural_codec.decode_from_codebook_indices(acoustic).squeeze(1)

    generate = forward

    @eval_decorator
    @torch.no_grad()
    def generate_top_match(self, *, text: List[str], num_samples=4, num_top_matches: int = 1, **kwargs):
        """open_musiclm.py:1039-1071: sample num_samples waves per prompt, rank by CLAP text/audio cosine similarity."""
        try:
            from torchaudio.functional import resample
        except ImportError as e:  # pragma: no cover
            raise ImportError("generate_top_match needs torchaudio to resample for CLAP") from e
        all_samples, all_similarities = [], []
        for prompt in text:
            batch_text = [prompt] * num_samples
            samples = self.forward(text=batch_text, **kwargs)
            text_latents = self.clap(text_input=[prompt], return_embedding=True).repeat(num_samples, 1)
            clap_input = int16_to_float32(float32_to_int16(resample(samples, self.neural_codec.sample_rate, self.clap.sample_rate)))
            audio_latents = self.clap(audio_input=clap_input, return_embedding=True)
            sim = F.cosine_similarity(text_latents, audio_latents, dim=-1)
            top = sim.topk(num_top_matches, dim=0, sorted=True).indices
            all_similarities.append(sim[top].detach().cpu())
            all_samples.append(samples[top])
        return all_samples, all_similarities
