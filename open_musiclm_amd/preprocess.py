"""DataPreprocessor (reference open_musiclm/preprocess.py:82-284): the offline ETL that turns audio into the sqlite token store
`data.PreprocessedDataset` reads -- table tokens(idx integer primary key, path text, clap array, semantic array, coarse array,
fine array), arrays as uint16 np.save blobs: clap [windows, 12, 1] (sliding `clap_audio_length_seconds` windows, 1 s hop),
semantic [1, T_s], coarse [1, T_a, Qc], fine [1, T_a, Qf].

The three tokenizers (CLAP + RVQ, MERT + k-means, Encodec) are third-party pretrained networks that are not part of this build, and
the reference's audio loader needs torchaudio: here the preprocessor takes ANY objects with the reference's tokenizer call signatures
(`audio_conditioner(audio_input=...)`, `wav2vec(wave, flatten=False)`, `neural_codec(wave, return_encoded=True)` -> `(_, indices, _)`)
and ANY dataset / iterable of the reference's batches `{'data': (wave_for_clap, wave_for_semantic, wave_for_acoustic),
'file_path': [...]}` (what `SoundDatasetForPreprocessing` + its collate produce), so the store is written exactly as the
reference writes it: same windows, same routing functions, same dtypes, same sqlite adapters, same resume rule."""
from __future__ import annotations

import math
from itertools import islice
from pathlib import Path
from shutil import rmtree
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from .data import init_sqlite
from .open_musiclm import (get_or_compute_acoustic_token_ids, get_or_compute_clap_token_ids,
                           get_or_compute_semantic_token_ids)
from .parallel import DataParallel
from .utils import copy_file_to_folder, exists


def cycle(dl):
    while True:
        for data in dl:
            yield data


def noop(*args, **kwargs):
    pass


class DataPreprocessor(nn.Module):
    def __init__(self, *, num_coarse_quantizers=3, wav2vec=None, neural_codec=None, audio_conditioner=None,
                 max_audio_length_seconds=180, random_crop=True, clap_audio_length_seconds=10, semantic_audio_length_seconds=10,
                 clap_batch_size=32, num_crops=1, ignore_files: Optional[List[str]] = None, ignore_load_errors=True,
                 replace_existing=False, folder=None, dataset=None, shard_dataset=True, results_folder='./data/fma_preprocessed',
                 accelerate_kwargs: dict = {}, config_paths: Optional[List[str]] = None, **kwargs):
        super().__init__()
        self.dp = DataParallel(device=torch.device('cpu'), backend='gloo')
        self.wav2vec, self.audio_conditioner, self.neural_codec = wav2vec, audio_conditioner, neural_codec
        self.num_coarse_quantizers = num_coarse_quantizers
        self.max_audio_length_seconds = max_audio_length_seconds
        self.clap_audio_length_seconds = int(clap_audio_length_seconds)
        self.semantic_audio_length_seconds = int(semantic_audio_length_seconds)
        assert self.clap_audio_length_seconds == self.semantic_audio_length_seconds, 'clap window must be equal to semantic window for now'
        self.clap_batch_size, self.num_crops, self.replace_existing = clap_batch_size, num_crops, replace_existing
        self.register_buffer('steps', torch.Tensor([0]))
        assert exists(wav2vec) and exists(audio_conditioner) and exists(neural_codec)
        self.ds_fields = ('raw_wave_for_clap', 'raw_wave_for_semantic', 'raw_wave_for_acoustic')
        if not exists(dataset):
            assert exists(folder), 'audio folder must be passed in for preprocessing'
            raise ImportError("reading an audio folder needs torchaudio (SoundDatasetForPreprocessing), which is not part of this "
                              "build: pass dataset= an iterable of {'data': (clap wave, semantic wave, acoustic wave), 'file_path': [...]}")
        self.ds = dataset
        # the reference hands its dataloader to accelerate, which deals batch i to rank i % world (wrapping round at the end);
        # len(self.ds) stays the whole dataset's, as process() divides it by the world size
        self.dl_iter = cycle(self.ds)
        if shard_dataset and self.dp.world_size > 1:
            self.dl_iter = islice(self.dl_iter, self.dp.rank, None, self.dp.world_size)
        self.results_folder = Path(results_folder)
        if self.is_main:
            if len([*self.results_folder.glob('**/*')]) > 0 and _yes_or_no('do you want to clear previous experiment checkpoints and results?'):
                rmtree(str(self.results_folder))
            self.results_folder.mkdir(parents=True, exist_ok=True)
        if self.is_main and exists(config_paths):
            configs_folder = self.results_folder / "configs"
            configs_folder.mkdir(parents=True, exist_ok=True)
            for config_path in config_paths:
                copy_file_to_folder(config_path, configs_folder)
        if self.is_main:
            self.conn, self.cursor = init_sqlite(str(self.results_folder / 'preprocessed.db'))
            self.cursor.execute("CREATE TABLE IF NOT EXISTS tokens(idx integer primary key, path text, clap array, semantic array, coarse array, fine array)")
            self.conn.commit()
        self.dp.barrier()
        if not self.is_main:
            self.conn, self.cursor = init_sqlite(str(self.results_folder / 'preprocessed.db'))

    def print(self, msg):
        if self.is_main:
            print(msg)

    @property
    def is_distributed(self):
        return self.dp.is_distributed

    @property
    def is_main(self):
        return self.dp.rank == 0

    @property
    def is_local_main(self):
        return self.dp.local_rank == 0

    def generate_tokens_from_batch(self, raw_wave_for_clap, raw_wave_for_semantic, raw_wave_for_acoustic):
        """preprocess.py:229-250.  clap windows: `clap_audio_length_seconds` long, one second apart, at the conditioner's rate."""
        sr = self.audio_conditioner.sample_rate
        clap_split = raw_wave_for_clap.unfold(-1, sr * self.clap_audio_length_seconds, sr).squeeze(0)
        ids = []
        for i in range(0, clap_split.shape[0], self.clap_batch_size):
            ids.append(get_or_compute_clap_token_ids(None, self.audio_conditioner, clap_split[i:i + self.clap_batch_size, :], None))
        clap_token_ids = torch.cat(ids, dim=0)
        semantic_token_ids = get_or_compute_semantic_token_ids(None, raw_wave_for_semantic, self.wav2vec)
        coarse_token_ids, fine_token_ids = get_or_compute_acoustic_token_ids(None, None, raw_wave_for_acoustic, self.neural_codec,
                                                                             self.num_coarse_quantizers)
        return clap_token_ids, semantic_token_ids, (coarse_token_ids, fine_token_ids)

    def process(self, log_fn=noop):
        """preprocess.py:252-284: row idx = iteration * world + rank; rows already present are skipped unless replace_existing."""
        world, rank = self.dp.world_size, self.dp.rank
        iters = math.ceil(self.num_crops * len(self.ds) / world)
        for it in range(iters):
            inputs = next(self.dl_iter)
            if exists(inputs):
                idx = it * world + rank
                if not self.replace_existing:
                    self.cursor.execute("SELECT * FROM tokens WHERE idx=?", (idx,))
                    if len(self.cursor.fetchall()) > 0:
                        continue
                data_kwargs = dict(zip(self.ds_fields, inputs['data']))
                clap_ids, sem_ids, (coarse_ids, fine_ids) = self.generate_tokens_from_batch(**data_kwargs)
                arrays = [t.detach().cpu().numpy().astype(np.uint16) for t in (clap_ids, sem_ids, coarse_ids, fine_ids)]   # uint16: space
                # the reference INSERTs in both modes, which raises on the primary key as soon as replace_existing meets a row
                verb = "INSERT OR REPLACE" if self.replace_existing else "INSERT"
                self.cursor.execute(f"{verb} INTO tokens VALUES (?, ?, ?, ?, ?, ?)", (idx, inputs['file_path'][0], *arrays))
                self.conn.commit()
            self.steps += 1
        self.print('processing complete')


def _yes_or_no(question):
    import sys
    if not sys.stdin or not sys.stdin.isatty():
        return False
    return input(f'{question} (y/n) ').lower() in ('yes', 'y')
