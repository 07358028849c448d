"""SingleStageTrainer with the reference's constructor / method surface (reference open_musiclm/trainer.py:111-560),
re-hosted on the MI355X engine:

  * no HF accelerate / DDP: one process per GPU (parallel.DataParallel over RCCL); gradients of all
    grad_accum_every micro-batches accumulate in the optimizer's flat buffer and are exchanged with ONE
    SUM all-reduce per optimizer step (the reference all-reduces on every micro-batch backward, :439);
  * clip_grad_norm_ + Adam(W) + zero_grad + LinearLR warm-up are the fused optimizer step (:444-449);
  * the per-micro-batch `loss.item()` host sync of the reference (:441) is replaced by one device-side
    accumulation that is read back once per optimizer step.

Checkpoint files keep the reference's names and formats ({stage}.transformer|optimizer|scheduler.{step}.pt, :536-549)
so scripts/train_utils.py resumes from them unchanged.
"""
from __future__ import annotations

import itertools
import json
import os
import sys
import time
from dataclasses import asdict, is_dataclass
from pathlib import Path
from shutil import rmtree
from typing import List, Optional

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader, Dataset, random_split

from .data import PreprocessedDataset, SoundDataset, get_dataloader, get_preprocessed_dataloader
from .open_musiclm import (CoarseStage, FineStage, SemanticStage, TokenConditionedTransformer)
from .optimizer import get_linear_scheduler, get_optimizer
from .parallel import DataParallel
from .utils import copy_file_to_folder, default, exists


def cycle(dl, sampler=None):
    epoch = 0
    while True:
        if sampler is not None:                    # DistributedSampler: a new shuffle every pass (else every epoch repeats)
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


def _rank_sampler(dp, ds):
    """DistributedSampler over `ds` for this rank, or None in a single process."""
    if not dp.is_distributed:
        return None
    from torch.utils.data.distributed import DistributedSampler
    return DistributedSampler(ds, num_replicas=dp.world_size, rank=dp.rank, shuffle=True)


def _gather_rows(dp, t: torch.Tensor) -> torch.Tensor:
    """all-gather of [rows, d] along dim 0; a gloo group (CPU collectives) gets the tensor on the host and returns it to its device."""
    if not dp.is_distributed:
        return t
    import torch.distributed as dist
    dev = t.device
    if dist.get_backend() == "gloo" and dev.type != "cpu":
        return dp.all_gather_cat(t.cpu().contiguous()).to(dev)
    return dp.all_gather_cat(t.contiguous())


def yes_or_no(question):
    if not sys.stdin or not sys.stdin.isatty():       # non-interactive launch (torchrun, CI): keep existing results
        return False
    answer = input(f'{question} (y/n) ')
    return answer.lower() in ('yes', 'y')


def accum_log(log, new_logs):
    for key, new_value in new_logs.items():
        log[key] = log.get(key, 0.) + new_value
    return log


def noop(*args, **kwargs):
    pass


class _JsonlTracker:
    """Stand-in for accelerate's trackers (tensorboard / wandb are not hot-path): metrics go to a JSONL file."""

    def __init__(self, logging_dir, run_name, config, enabled):
        self.f = None
        if enabled and logging_dir:
            os.makedirs(logging_dir, exist_ok=True)
            self.f = open(os.path.join(logging_dir, f"{run_name}.jsonl"), "a")
            self.f.write(json.dumps({"config": config}, default=str) + "\n")

    def log(self, values, step):
        if self.f:
            self.f.write(json.dumps({"step": step, **values}) + "\n")
            self.f.flush()


class SingleStageTrainer(nn.Module):
    """trainer.py:111-560.  semantic: needs audio_conditioner + wav2vec (or preprocessed / token datasets);
    coarse: + neural_codec; fine: audio_conditioner + neural_codec."""

    def __init__(self, transformer: TokenConditionedTransformer, stage, *, num_train_steps, batch_size,
                 model_config=None, training_config=None, dataset: Optional[Dataset] = None, wav2vec=None,
                 neural_codec=None, audio_conditioner=None, data_max_length_seconds=1,
                 ignore_files: Optional[List[str]] = None, cross_entropy_loss_weights: Optional[List[float]] = None,
                 ignore_load_errors=True, folder=None, use_preprocessed_data=False, lr=3e-4, lr_warmup=0,
                 grad_accum_every=1, wd=0., max_grad_norm=0.5, valid_frac=0.05, random_split_seed=42,
                 save_results_every=100, save_predicted_tokens=True, save_reconstructed_wave=True,
                 save_model_every=1000, results_folder='./results', accelerate_kwargs: dict = {},
                 config_paths: Optional[List[str]] = None, dataset_yields_tokens: Optional[bool] = None,
                 use_hip_graph: Optional[bool] = None):
        super().__init__()
        # accelerate_kwargs is accepted for script compatibility: log_with / logging_dir select the JSONL tracker
        self.dp = DataParallel(device=transformer.device)
        if self.dp.is_distributed and transformer.device.type == 'cuda':
            # One process drives ONE GPU: cuda:LOCAL_RANK.  The reference's scripts build the model with device='cuda' (= cuda:0
            # on every rank) and rely on accelerator.prepare to place it (trainer.py:286-304); here the trainer moves the
            # transformer and the frozen front-ends before the optimizer adopts the parameters.
            target = torch.device('cuda', self.dp.local_rank % max(torch.cuda.device_count(), 1))
            torch.cuda.set_device(target)
            if transformer.device != target:
                transformer.to(target)
                for m in (wav2vec, neural_codec, audio_conditioner):
                    if isinstance(m, nn.Module):
                        m.to(target)
        self.log_with = accelerate_kwargs.get('log_with')
        self.logging_dir = accelerate_kwargs.get('logging_dir') or accelerate_kwargs.get('project_dir')

        self.use_preprocessed_data = use_preprocessed_data
        tokens_in = use_preprocessed_data or bool(default(dataset_yields_tokens, exists(dataset) and not exists(audio_conditioner)))
        self.model_config, self.training_config = model_config, training_config
        self.transformer = transformer
        self.wav2vec, self.audio_conditioner, self.neural_codec = wav2vec, audio_conditioner, neural_codec
        self.stage = stage

        if stage == 'semantic':
            assert tokens_in or (exists(audio_conditioner) and exists(wav2vec))
            self.train_wrapper = SemanticStage(semantic_transformer=transformer, wav2vec=wav2vec, clap=audio_conditioner,
                                               cross_entropy_loss_weights=default(cross_entropy_loss_weights, [0., 1.]))
            token_fields, wave_fields = ('clap_token_ids', 'semantic_token_ids'), ('raw_wave_for_clap', 'raw_wave_for_semantic')
        elif stage == 'coarse':
            assert tokens_in or (exists(wav2vec) and exists(audio_conditioner) and exists(neural_codec))
            self.train_wrapper = CoarseStage(coarse_transformer=transformer, neural_codec=neural_codec, wav2vec=wav2vec,
                                             clap=audio_conditioner,
                                             cross_entropy_loss_weights=default(cross_entropy_loss_weights, [0., 0., 1.]))
            token_fields = ('clap_token_ids', 'semantic_token_ids', 'coarse_token_ids')
            wave_fields = ('raw_wave_for_clap', 'raw_wave_for_semantic', 'raw_wave_for_acoustic')
        elif stage == 'fine':
            assert tokens_in or (exists(audio_conditioner) and exists(neural_codec))
            self.train_wrapper = FineStage(fine_transformer=transformer, clap=audio_conditioner, neural_codec=neural_codec,
                                           cross_entropy_loss_weights=default(cross_entropy_loss_weights, [0., 0., 1.]))
            token_fields = ('clap_token_ids', 'coarse_token_ids', 'fine_token_ids')
            wave_fields = ('raw_wave_for_clap', 'raw_wave_for_acoustic')
        else:
            raise ValueError(f'invalid stage: {stage}')
        self.ds_fields = token_fields if tokens_in else wave_fields

        self.register_buffer('steps', torch.Tensor([0]))
        self.num_train_steps, self.batch_size, self.grad_accum_every = num_train_steps, batch_size, grad_accum_every

        self.optim = get_optimizer(transformer.parameters(), lr=lr, wd=wd)
        self.scheduler = get_linear_scheduler(self.optim, total_iters=lr_warmup) if lr_warmup > 0 else None
        self.max_grad_norm = max_grad_norm

        # ---- data ----
        if self.use_preprocessed_data:
            g = self.model_config.global_cfg
            self.ds = PreprocessedDataset(folder, stage=self.stage,
                                          semantic_window_seconds=int(g.semantic_audio_length_seconds),
                                          coarse_window_seconds=int(g.coarse_audio_length_seconds),
                                          fine_window_seconds=int(g.fine_audio_length_seconds),
                                          semantic_steps_per_second=self.model_config.hubert_kmeans_cfg.output_hz,
                                          acoustic_steps_per_second=self.model_config.encodec_cfg.output_hz)
        else:
            self.ds = dataset
            if not exists(self.ds):
                assert exists(folder), 'folder must be passed in, if not passing in a custom dataset for text conditioned audio synthesis training'
                self.ds = SoundDataset(folder, max_length_seconds=data_max_length_seconds, ignore_files=default(ignore_files, []),
                                       ignore_load_errors=ignore_load_errors)
        if valid_frac > 0:
            train_size = int((1 - valid_frac) * len(self.ds))
            valid_size = len(self.ds) - train_size
            self.ds, self.valid_ds = random_split(self.ds, [train_size, valid_size],
                                                  generator=torch.Generator().manual_seed(random_split_seed))
            self.print(f'training with dataset of {len(self.ds)} samples and validating with randomly splitted {len(self.valid_ds)} samples')
        else:
            self.valid_ds = self.ds
            self.print(f'training with shared training and valid dataset of {len(self.ds)} samples')

        make_dl = get_preprocessed_dataloader if (self.use_preprocessed_data or tokens_in) else get_dataloader
        sampler = vsampler = None
        if self.dp.is_distributed:        # per-rank shard of the data (accelerate.prepare(dl) equivalent; batch_size is per process)
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(self.ds, num_replicas=self.dp.world_size, rank=self.dp.rank, shuffle=True)
            vsampler = DistributedSampler(self.valid_ds, num_replicas=self.dp.world_size, rank=self.dp.rank, shuffle=True)
        self.dl = make_dl(self.ds, batch_size=batch_size, shuffle=sampler is None, sampler=sampler)
        self.valid_dl = make_dl(self.valid_ds, batch_size=batch_size, shuffle=vsampler is None, sampler=vsampler)
        self.dl_iter, self.valid_dl_iter = cycle(self.dl, sampler), cycle(self.valid_dl, vsampler)
        if self.dp.is_distributed and transformer.device.type == 'cuda':
            # replicas start identical (rank 0's weights / moments), but draw different forgetful masks and dropout masks
            self.optim.sync_replicas(self.dp)
            torch.cuda.manual_seed(int(torch.initial_seed() + 7919 * (self.dp.rank + 1)) & 0x7FFFFFFFFFFF)

        self.save_model_every, self.save_results_every = save_model_every, save_results_every
        self.save_predicted_tokens, self.save_reconstructed_wave = save_predicted_tokens, save_reconstructed_wave
        self.results_folder = Path(results_folder)
        if self.is_main and len([*self.results_folder.glob('**/*')]) > 0 and yes_or_no('do you want to clear previous experiment checkpoints and results?'):
            rmtree(str(self.results_folder))
        self.results_folder.mkdir(parents=True, exist_ok=True)
        self.dp.barrier()
        if exists(save_reconstructed_wave):
            self.waves_folder = self.results_folder / 'reconstructed_waves'
            self.waves_folder.mkdir(parents=True, exist_ok=True)
        if exists(save_predicted_tokens):
            self.tokens_folder = self.results_folder / 'tokens'
            self.tokens_folder.mkdir(parents=True, exist_ok=True)

        hps = {}
        if exists(model_config) and is_dataclass(model_config):
            hps.update(asdict(model_config.global_cfg))
            hps.update(asdict(getattr(model_config, f'{stage}_cfg')))
        if exists(training_config) and is_dataclass(training_config):
            hps.update(asdict(getattr(training_config, f'{stage}_trainer_cfg')))
        self.tracker = _JsonlTracker(self.logging_dir, f"{stage}_stage_{int(time.time() * 1000)}", hps,
                                     enabled=self.is_main and exists(self.log_with))
        if self.is_main and exists(config_paths):
            configs_folder = self.results_folder / "configs"
            configs_folder.mkdir(parents=True, exist_ok=True)
            for config_path in config_paths:
                copy_file_to_folder(config_path, configs_folder)
        # static-shape token training -> capture the micro-step into a HIP graph (graph.py); raw-audio front-ends stay eager
        self.use_hip_graph = tokens_in if use_hip_graph is None else use_hip_graph
        self._graphed = None

    # ---- checkpointing (trainer.py:359-391) ----------------------------------------------------------
    def save(self, model_path, optim_path, scheduler_path=None):
        torch.save({k: v.detach().cpu() for k, v in self.transformer.state_dict().items()}, model_path)
        torch.save(self.optim.state_dict(), optim_path)
        if exists(self.scheduler):
            assert exists(scheduler_path)
            torch.save(self.scheduler.state_dict(), scheduler_path)

    def load(self, model_path, optim_path, scheduler_path=None, steps=0):
        model_path, optim_path = Path(model_path), Path(optim_path)
        assert model_path.exists() and optim_path.exists()
        self.transformer.load_state_dict(torch.load(model_path, map_location=self.device))
        self.optim.load_state_dict(torch.load(optim_path, map_location=self.device, weights_only=False))
        if exists(self.scheduler):
            assert exists(scheduler_path), 'the config specifies lr warmup is used, but no scheduler checkpoint is given. try setting lr_warmup to 0.'
            scheduler_path = Path(scheduler_path)
            assert scheduler_path.exists()
            self.scheduler.load_state_dict(torch.load(scheduler_path, map_location=self.device, weights_only=False))
        self._graphed = None        # weights changed under the captured graph: re-capture on the next micro-step
        # the restored loss-scale block carries the CUMULATIVE skipped-step count, and the restored scheduler was already rewound for those
        # skips: without this the first train_step after a resume reported them again and rewound the warm-up a second time
        rep = self.optim.loss_scale_report() if hasattr(self.optim, 'loss_scale_report') else {}
        self._skipped_seen = rep.get('skipped_steps', 0)
        if self.dp.is_distributed and self.device.type == 'cuda':
            self.optim.sync_replicas(self.dp)
        if steps > 0:
            assert int(self.steps.item()) == 0, 'steps should be 0 when loading a checkpoint for the first time'
            self.steps += steps

    def print(self, msg):
        self.dp.print(msg)

    def generate(self, *args, **kwargs):
        return self.train_wrapper.generate(*args, **kwargs)

    @property
    def device(self):
        return self.transformer.device

    @property
    def is_distributed(self):
        return self.dp.is_distributed

    @property
    def is_main(self):
        return self.dp.is_main

    @property
    def is_local_main(self):
        return self.dp.is_local_main

    def _next_batch(self, it):
        batch = next(it)
        if isinstance(batch, torch.Tensor):
            batch = (batch,)
        return {k: v.to(self.device, non_blocking=True) for k, v in zip(self.ds_fields, batch)}

    # ---- one optimizer step (trainer.py:415-552) -------------------------------------------------------
    def micro_step(self, data_kwargs):
        """forward + backward of ONE micro-batch; gradients accumulate in the optimizer's flat buffer."""
        if self.use_hip_graph and self.device.type == 'cuda':
            if self._graphed is None:
                from .graph import GraphedForwardBackward
                self._graphed = GraphedForwardBackward(lambda **kw: self.train_wrapper(**kw, return_loss=True, return_logits=False)[0],
                                                       loss_scale=1.0 / self.grad_accum_every)

                def discard():                          # warm-up steps really ran: throw their gradients away
                    self.optim.mark_grads_dirty()
                    self.optim.zero_grad()
                    rc = self._relpos_cache()
                    if rc is not None:
                        rc.reset_accum()
                self._graphed.prepare(data_kwargs, after_warmup=discard)
                if self._graphed.capture_error:
                    self.print(f'HIP graph capture unavailable ({self._graphed.capture_error}); launching eagerly')
            loss = self._graphed(**data_kwargs)
            self.optim.mark_grads_dirty()
            return loss.clone()
        loss, _, _ = self.train_wrapper(**data_kwargs, return_loss=True, return_logits=False)
        (loss / self.grad_accum_every).backward()
        self.optim.mark_grads_dirty()
        return loss.detach()

    def _relpos_cache(self):
        """engine.RelposStepCache of the transformer's trunk when gradient accumulation makes it pay (grad_accum_every > 1, learned MLP bias,
        HIP path); OMLM_RELPOS_CACHE=0 keeps the per-micro-batch form."""
        if self.grad_accum_every <= 1 or self.device.type != 'cuda' or os.environ.get("OMLM_RELPOS_CACHE", "1") == "0":
            return None
        tr = getattr(self.transformer, "transformer", None)
        if tr is None or getattr(tr, "rel_pos_bias", None) is None or getattr(tr, "relative_position_bias_type", "") == "t5":
            return None
        from . import engine
        rc = engine.relpos_step_cache(tr)
        rc.enabled = True
        return rc

    def optimizer_step(self):
        """ONE gradient exchange + fused clip/Adam(W)/zero_grad + scheduler tick.  Precision "fp16": a step whose gradients overflowed is
        skipped ON THE DEVICE (no host read here), so the tick happens regardless; train_step, which reads the skip counter next to its loss
        read-back anyway, takes the tick back (_rewind_scheduler) -- warm-up and decay then advance with the Adam clock, on applied steps only."""
        self.dp.allreduce_sum_(self.optim.flat_grad)
        self.optim.step(max_grad_norm=self.max_grad_norm, grad_scale=self.dp.grad_scale())
        if exists(self.scheduler):
            self.scheduler.step()

    def _rewind_scheduler(self, n: int):
        """Take back the scheduler ticks of n optimizer steps the device skipped (fp16 overflow): LinearLR is a closed form of last_epoch."""
        sch = self.scheduler
        if not exists(sch) or n <= 0 or not hasattr(sch, "_get_closed_form_lr"):
            return
        sch.last_epoch = max(sch.last_epoch - n, 0)
        lrs = sch._get_closed_form_lr()
        for g, lr in zip(sch.optimizer.param_groups, lrs):
            g['lr'] = lr
        sch._last_lr = list(lrs)

    def train_step(self):
        steps = int(self.steps.item())
        self.transformer.train()
        self.train_wrapper.train()
        self.optim.zero_grad()
        loss_acc = torch.zeros((), device=self.device)
        rc = self._relpos_cache()
        if rc is not None:
            rc.refresh()                                # the rel-pos table of this step's weights, once for all its micro-batches
        for _ in range(self.grad_accum_every):
            loss_acc += self.micro_step(self._next_batch(self.dl_iter))
        if rc is not None:
            rc.flush()                                  # the MLP's backward on the micro-batches' summed d(table), before the exchange
        self.optimizer_step()
        logs = {'loss': float(loss_acc.item()) / self.grad_accum_every}       # single host sync per optimizer step
        if self.device.type == 'cuda':
            # the host has just waited for the device: a token id / label past its table (wrong codebook size, damaged token store)
            # was skipped by the kernels and flagged -- surface it like torch's device assert would
            from . import ops
            ops.raise_on_index_error(self.device)
            rep = self.optim.loss_scale_report() if hasattr(self.optim, "loss_scale_report") else {}
            if rep:                                  # precision "fp16": overflowed steps are skipped on the device -- say so
                logs.update(loss_scale=rep["scale"], skipped_steps=rep["skipped_steps"])
                if rep["skipped_steps"] > getattr(self, "_skipped_seen", 0):
                    self.print(f"{steps}: fp16 gradient overflow -- optimizer step skipped ({rep['skipped_steps']} so far), "
                               f"loss scale now {rep['scale']:g}")
                    self._rewind_scheduler(rep["skipped_steps"] - getattr(self, "_skipped_seen", 0))
                    self._skipped_seen = rep["skipped_steps"]
        self.print(f"{steps}: loss: {logs['loss']}")

        valid_loss = valid_accuracy = None
        if not (steps % self.save_results_every):
            valid_loss, valid_accuracy = self.validate(steps)
        self.tracker.log({"train_loss": logs['loss'], "valid_loss": valid_loss, "valid_accuracy": valid_accuracy}, step=steps)

        if self.is_main and not (steps % self.save_model_every):
            self.print(f'{steps}: saving model to {str(self.results_folder)}')
            self.save(str(self.results_folder / f'{self.stage}.transformer.{steps}.pt'),
                      str(self.results_folder / f'{self.stage}.optimizer.{steps}.pt'),
                      str(self.results_folder / f'{self.stage}.scheduler.{steps}.pt'))
            if exists(self.audio_conditioner) and getattr(self.audio_conditioner, 'learn_rvq', False):
                torch.save(self.audio_conditioner.rq.state_dict(), str(self.results_folder / f'{self.stage}.conditioner_rvq.{steps}.pt'))
        self.steps += 1
        return logs

    @torch.no_grad()
    def validate(self, steps):
        """trainer.py:457-526: teacher-forced validation loss / token accuracy, token dumps, optional wave reconstruction."""
        self.train_wrapper.eval()
        data_kwargs = self._next_batch(self.valid_dl_iter)
        valid_loss, all_logits, all_labels = self.train_wrapper(**data_kwargs, return_loss=True)
        valid_loss = float(self.dp.reduce_mean(valid_loss.detach().reshape(1)).item())
        pred = self.dp.all_gather_cat(all_logits[-1].argmax(1).contiguous()).cpu().long()
        gt = self.dp.all_gather_cat(all_labels[-1].contiguous()).cpu().long()
        valid_accuracy = (pred == gt).float().mean().item()
        self.print(f'{steps}: valid loss {valid_loss}, valid acc {valid_accuracy}')
        if self.is_main and self.save_predicted_tokens:
            inter = torch.empty((pred.shape[0] + gt.shape[0], pred.shape[1]), dtype=pred.dtype)
            inter[0::2], inter[1::2] = pred, gt
            np.savetxt(str(self.tokens_folder / f'{self.stage}.tokens.{steps}.txt'), inter, fmt='%-6s',
                       header='predicted and ground truth tokens from the validation set. row 0%2 is predicted, 1%2 is ground truth\n ')
        if self.is_main and self.save_reconstructed_wave and self.stage in ('coarse', 'fine') and exists(self.neural_codec):
            toks = all_logits[-1].argmax(1)[:, :-1]
            toks[toks == self.transformer.eos_ids[-1]] = 0
            q = self.transformer.token_sequences[-1].num_quantizers
            toks = toks.reshape(toks.shape[0], -1, q)
            if self.stage == 'fine':
                cq = self.transformer.token_sequences[-2].num_quantizers
                coarse = all_labels[-2][:, :-1].reshape(toks.shape[0], -1, cq)
                toks = torch.cat((coarse, toks), dim=-1)
            waves = self.neural_codec.decode_from_codebook_indices(toks).cpu()
            try:
                import torchaudio
                for i, wave in enumerate(waves[:4]):
                    torchaudio.save(str(self.waves_folder / f'{self.stage}.reconstructed_wave_{i}.{steps}.wav'), wave, self.neural_codec.sample_rate)
            except ImportError:
                pass
        return valid_loss, valid_accuracy

    def train(self, log_fn=noop):
        while self.steps < self.num_train_steps:
            logs = self.train_step()
            log_fn(logs)
        self.print('training complete')


class ClapRVQTrainer(nn.Module):
    """Learn the residual vector quantizer that turns CLAP embeddings into discrete tokens (trainer.py:564-741).

    Same constructor and step contract as the reference: every step accumulates ``accumulate_batches`` batches of embeddings,
    concatenates them over ranks, and rank 0 runs ONE training-mode pass of the residual VQ over them
    (``audio_conditioner.quantize(embeds, return_rvq_loss=True)`` with ``learn_rvq``: k-means init on the first pass, then EMA
    codebook updates -- csrc/vq_fit.hip); ``clap.rvq.{steps}.pt`` checkpoints carry the library's key layout.
    The CLAP towers are third-party pretrained networks that are not part of this build: when ``audio_conditioner.clap`` is None
    the dataset must yield the embeddings themselves ([D] or [n, D] float tensors, e.g. precomputed CLAP embeddings)."""

    def __init__(self, *, num_train_steps, batch_size, accumulate_batches: Optional[int] = None, audio_conditioner=None,
                 dataset: Optional[Dataset] = None, ignore_files: Optional[List[str]] = None, ignore_load_errors: bool = True,
                 folder=None, wd=0., max_grad_norm=0.5, data_max_length_seconds=10, valid_frac=0.05, random_split_seed=42,
                 save_results_every=100, save_model_every=1000, results_folder='./results', accelerate_kwargs: dict = {},
                 config_paths: Optional[List[str]] = None):
        super().__init__()
        assert exists(audio_conditioner), 'audio_conditioner (ClapQuantized) must be passed in'
        dev = audio_conditioner.rq.codebooks.device
        self.dp = DataParallel(device=dev)
        self.log_with = accelerate_kwargs.get('log_with')
        self.logging_dir = accelerate_kwargs.get('logging_dir') or accelerate_kwargs.get('project_dir')
        self.audio_conditioner = audio_conditioner
        self.embeds_in = not exists(getattr(audio_conditioner, 'clap', None))
        self.ds = dataset
        self.num_train_steps = num_train_steps
        self.accumulate_batches = accumulate_batches
        self.register_buffer('steps', torch.Tensor([0]))
        if not exists(self.ds):
            assert exists(folder), 'folder must be passed in, if not passing in a custom dataset for text conditioned audio synthesis training'
            self.ds = SoundDataset(folder, max_length_seconds=data_max_length_seconds, target_sample_hz=audio_conditioner.sample_rate,
                                   seq_len_multiple_of=None, ignore_files=default(ignore_files, []), ignore_load_errors=ignore_load_errors)
        if valid_frac > 0:
            train_size = int((1 - valid_frac) * len(self.ds))
            valid_size = len(self.ds) - train_size
            self.ds, self.valid_ds = random_split(self.ds, [train_size, valid_size],
                                                  generator=torch.Generator().manual_seed(random_split_seed))
            self.print(f'training with dataset of {len(self.ds)} samples and validating with randomly splitted {len(self.valid_ds)} samples')
        else:
            self.valid_ds = self.ds
            self.print(f'training with shared training and valid dataset of {len(self.ds)} samples')
        # every rank draws ITS shard of the data (what accelerator.prepare(dl) gives the reference, trainer.py:664-669): with plain
        # shuffle=True loaders identically seeded ranks would all contribute the same batches to the gather below
        sampler = _rank_sampler(self.dp, self.ds)
        self.dl = get_dataloader(self.ds, batch_size=batch_size, shuffle=sampler is None, sampler=sampler)
        self.valid_dl = get_dataloader(self.valid_ds, batch_size=batch_size, shuffle=True)
        self.dl_iter, self.valid_dl_iter = cycle(self.dl, sampler), cycle(self.valid_dl)
        self.save_model_every, self.save_results_every = save_model_every, save_results_every
        self.results_folder = Path(results_folder)
        if self.is_main and len([*self.results_folder.glob('**/*')]) > 0 and \
                yes_or_no('do you want to clear previous experiment checkpoints and results?'):
            rmtree(str(self.results_folder))
        self.results_folder.mkdir(parents=True, exist_ok=True)
        hps = {"num_train_steps": num_train_steps, "batch_size": batch_size, "accumulate_batches": accumulate_batches}
        self.tracker = _JsonlTracker(self.logging_dir, "clap_rvq", hps, self.is_main and exists(self.log_with))
        if self.is_main and exists(config_paths):
            configs_folder = self.results_folder / "configs"
            configs_folder.mkdir(parents=True, exist_ok=True)
            for config_path in config_paths:
                copy_file_to_folder(config_path, configs_folder)

    def print(self, msg):
        if self.is_main:
            print(msg)

    @property
    def device(self):
        return self.audio_conditioner.rq.codebooks.device

    @property
    def is_distributed(self):
        return self.dp.is_distributed

    @property
    def is_main(self):
        return self.dp.rank == 0

    @property
    def is_local_main(self):
        return self.dp.local_rank == 0

    def _embed(self, batch) -> torch.Tensor:
        item = batch[0] if isinstance(batch, (list, tuple)) else batch
        if self.embeds_in:
            emb = item.to(self.device).float()
            return emb.reshape(-1, emb.shape[-1])
        return self.audio_conditioner.forward(audio_input=item.to(self.device), return_embedding=True)

    def train_step(self):
        steps = int(self.steps.item())
        self.audio_conditioner.learn_rvq = True
        iters = default(self.accumulate_batches, 1)
        iters = -(-iters // self.dp.world_size)
        embeds = torch.cat([self._embed(next(self.dl_iter)) for _ in range(iters)], dim=0)
        embeds = _gather_rows(self.dp, embeds)
        logs = {}
        if self.is_main:
            loss = self.audio_conditioner.quantize(embeds, return_rvq_loss=True)
            self.print(f'loss: {loss}')
            valid_loss = None
            if not (steps % self.save_results_every):
                with torch.no_grad():
                    self.audio_conditioner.learn_rvq = False
                    valid_loss = self.audio_conditioner.quantize(self._embed(next(self.valid_dl_iter)), return_rvq_loss=True)
                self.print(f'{steps}: valid loss {valid_loss}')
            logs = {"train_loss": loss, "valid_loss": valid_loss}
            self.tracker.log(logs, step=steps)
            if not (steps % self.save_model_every):
                torch.save(self.audio_conditioner.rq.state_dict(), str(self.results_folder / f'clap.rvq.{steps}.pt'))
                self.print(f'{steps}: saving model to {str(self.results_folder)}')
        self.steps += 1
        return logs

    def train(self, log_fn=noop):
        while self.steps < self.num_train_steps:
            logs = self.train_step()
            log_fn(logs)
        self.print('training complete')


class HfHubertKmeansTrainer(nn.Module):
    """Trainer for the k-means part of HfHubertWithKmeans (trainer.py:748-905): 1) collect `feature_extraction_num_steps` batches
    of features, 2) fit sklearn's MiniBatchKMeans on them (`learn_kmeans`, the reference's own call) and dump `kmeans.joblib`.
    The MERT / HuBERT extractor is a third-party pretrained network that is not part of this build: when `hubert_kmeans.hubert`
    is None the dataset must yield the features themselves ([t, f] or [b, t, f] float tensors, e.g. precomputed MERT layer-7
    embeddings); with an extractor present the reference's `forward(wav_input=..., return_embed=True)` path is used."""

    def __init__(self, *, feature_extraction_num_steps: int, feature_extraction_batch_size: int, hubert_kmeans,
                 dataset: Optional[Dataset] = None, ignore_files: Optional[List[str]] = None, ignore_load_errors: bool = True,
                 folder=None, data_max_length_seconds=1, results_folder='./results', accelerate_kwargs: dict = {},
                 config_paths: Optional[List[str]] = None):
        super().__init__()
        self.dp = DataParallel(device=torch.device('cpu'), backend='gloo')
        self.ds = dataset
        self.feature_extraction_num_steps = feature_extraction_num_steps
        self.feature_extraction_batch_size = feature_extraction_batch_size
        self.hubert_kmeans = hubert_kmeans
        self.features_in = not exists(getattr(hubert_kmeans, 'hubert', None))
        self.register_buffer('steps', torch.Tensor([0]))
        if not exists(self.ds):
            assert exists(folder), 'folder must be passed in, if not passing in a custom dataset for text conditioned audio synthesis training'
            self.ds = SoundDataset(folder, max_length_seconds=data_max_length_seconds, normalize=True,
                                   target_sample_hz=hubert_kmeans.target_sample_hz, seq_len_multiple_of=hubert_kmeans.seq_len_multiple_of,
                                   ignore_files=default(ignore_files, []), ignore_load_errors=ignore_load_errors)
        self.print(f'training on {feature_extraction_num_steps * feature_extraction_batch_size} out of {len(self.ds)} samples')
        sampler = _rank_sampler(self.dp, self.ds)                            # per-rank shard (trainer.py:812-815 in the reference)
        self.dl = get_dataloader(self.ds, batch_size=feature_extraction_batch_size, shuffle=sampler is None, sampler=sampler)
        self.dl_iter = cycle(self.dl, sampler)
        self.results_folder = Path(results_folder)
        if self.is_main and len([*self.results_folder.glob('**/*')]) > 0 and \
                yes_or_no('do you want to clear previous experiment checkpoints and results?'):
            rmtree(str(self.results_folder))
        self.results_folder.mkdir(parents=True, exist_ok=True)
        if self.is_main and exists(config_paths):
            configs_folder = self.results_folder / "configs"
            configs_folder.mkdir(parents=True, exist_ok=True)
            for config_path in config_paths:
                copy_file_to_folder(config_path, configs_folder)

    def print(self, msg):
        if self.is_main:
            print(msg)

    @property
    def device(self):
        return torch.device('cpu')

    @property
    def is_distributed(self):
        return self.dp.is_distributed

    @property
    def is_main(self):
        return self.dp.rank == 0

    @property
    def is_local_main(self):
        return self.dp.local_rank == 0

    def extract_hubert_features(self):
        batch = next(self.dl_iter)
        item = batch[0] if isinstance(batch, (list, tuple)) else batch
        if self.features_in:
            embed = item.float()
            embed = embed.reshape(-1, embed.shape[-1])                       # 'b t f -> (b t) f'
        else:
            dev = next(self.hubert_kmeans.parameters(), item).device       # the waveform goes to the extractor's device
            embed = self.hubert_kmeans.forward(wav_input=item.to(dev), return_embed=True)
            embed = embed.reshape(-1, embed.shape[-1])
        embed = _gather_rows(self.dp, embed.detach())
        return embed.cpu().numpy()

    def train(self, log_fn=noop, seed=0, **kmeans_kwargs):
        from .hf_hubert_kmeans import learn_kmeans
        self.print('step 1: extracting features. must wait for this to complete before training kmeans.')
        features = []
        num_steps = -(-self.feature_extraction_num_steps // self.dp.world_size)
        while self.steps < num_steps:
            self.print(f'{int(self.steps.item())} / {num_steps} steps')
            features.append(self.extract_hubert_features())
            self.steps += 1
        features = np.concatenate(features, axis=0)
        features = features[~np.any(np.isnan(features), axis=-1)]
        self.print('step 2: training kmeans')
        if self.is_main:
            learn_kmeans(features, seed, str(self.results_folder / 'kmeans.joblib'),
                         n_clusters=self.hubert_kmeans.codebook_size, **kmeans_kwargs)
        self.print('training complete')
