"""Token datasets feeding the trainer.

* ``SyntheticTokenDataset`` -- seeded random token ids in the exact shapes the reference's PreprocessedDataset
  yields for a stage (reference data.py:350-429), used by bench.py and the tests.
* ``PreprocessedDataset``   -- reader of the reference's sqlite token store (data.py:304-439, writer
  preprocess.py:198-200,273-280): table tokens(idx, path, clap, semantic, coarse, fine) with np.save blobs.
* ``SoundDataset``          -- raw-audio dataset: needs torchaudio (audio front-end is outside the hot path).
"""
from __future__ import annotations

import io
import random
import sqlite3
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .utils import exists


class SyntheticTokenDataset(Dataset):
    """item -> tuple of int32 tensors shaped like one PreprocessedDataset item (leading dim 1)."""

    def __init__(self, stage, *, length=4096, seed=1234, semantic_window_seconds=10, coarse_window_seconds=4,
                 fine_window_seconds=2, semantic_steps_per_second=50, acoustic_steps_per_second=75,
                 num_clap_quantizers=12, num_coarse_quantizers=3, num_fine_quantizers=5, codebook_size=1024):
        self.stage, self.length, self.seed, self.V = stage, length, seed, codebook_size
        clap = (1, num_clap_quantizers, 1)
        if stage == 'semantic':
            self.shapes = [clap, (1, semantic_window_seconds * semantic_steps_per_second - 1)]
        elif stage == 'coarse':
            self.shapes = [clap, (1, coarse_window_seconds * semantic_steps_per_second - 1),
                           (1, coarse_window_seconds * acoustic_steps_per_second, num_coarse_quantizers)]
        elif stage == 'fine':
            self.shapes = [clap, (1, fine_window_seconds * acoustic_steps_per_second, num_coarse_quantizers),
                           (1, fine_window_seconds * acoustic_steps_per_second, num_fine_quantizers)]
        else:
            raise ValueError(f'invalid stage {stage}')

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        return tuple(torch.randint(0, self.V, s, generator=g, dtype=torch.int32) for s in self.shapes)


def _np_from_blob(blob):
    return np.load(io.BytesIO(blob))


def _blob_from_np(arr):
    """The reference's adapt_array (data.py:33-40): the array's np.save image as a blob."""
    out = io.BytesIO()
    np.save(out, arr)
    out.seek(0)
    return sqlite3.Binary(out.read())


def init_sqlite(path):
    """Same adapters as the reference (data.py:33-56: array <-> np.save bytes), for reading and for writing the token store."""
    sqlite3.register_adapter(np.ndarray, _blob_from_np)
    sqlite3.register_converter("array", _np_from_blob)
    conn = sqlite3.connect(path, detect_types=sqlite3.PARSE_DECLTYPES)
    return conn, conn.cursor()


class PreprocessedDataset(Dataset):
    """Random second-aligned crops from the preprocessed token store (reference data.py:304-429)."""

    def __init__(self, folder, stage, semantic_window_seconds=10, coarse_window_seconds=4, fine_window_seconds=2,
                 semantic_steps_per_second=50, acoustic_steps_per_second=75):
        path = Path(folder)
        assert path.exists(), 'folder does not exist'
        self.stage = stage
        self.sem_win, self.coarse_win, self.fine_win = semantic_window_seconds, coarse_window_seconds, fine_window_seconds
        self.sem_hz, self.ac_hz = semantic_steps_per_second, acoustic_steps_per_second
        self.conn, self.cursor = init_sqlite(str(path / 'preprocessed.db'))
        self.cursor.execute('SELECT idx from tokens')
        self.ids = [r[0] for r in self.cursor.fetchall()]

    def __len__(self):
        return len(self.ids)

    def _audio_seconds(self, clap=None, semantic=None, coarse=None, fine=None):
        cands = []
        if exists(clap): cands.append(clap.shape[0] + self.sem_win - 1)           # one clap row per sliding second
        if exists(semantic): cands.append((semantic.shape[1] + 1) // self.sem_hz)
        if exists(coarse): cands.append(coarse.shape[1] // self.ac_hz)
        if exists(fine): cands.append(fine.shape[1] // self.ac_hz)
        cands = [int(c) for c in cands]
        assert len(set(cands)) == 1, 'audio lengths are not equal'
        return cands[0]

    def _crop(self, seconds, outer, inner=None):
        o0 = random.randint(0, seconds - outer)
        if inner is None:
            return o0, o0 + outer, None, None
        i0 = random.randint(o0, o0 + outer - inner)
        return o0, o0 + outer, i0, i0 + inner

    def _sem(self, t, a, b):
        return t[:, a * self.sem_hz: b * self.sem_hz - 1]

    def _ac(self, t, a, b):
        return t[:, a * self.ac_hz: b * self.ac_hz]

    def __getitem__(self, idx):
        cols = {'semantic': 'clap, semantic', 'coarse': 'clap, semantic, coarse', 'fine': 'clap, coarse, fine'}[self.stage]
        row = self.cursor.execute(f'SELECT {cols} FROM tokens where idx = ?', (self.ids[idx],)).fetchone()
        row = [torch.from_numpy(r.astype(np.int32)) for r in row]
        if self.stage == 'semantic':
            clap, sem = row
            o0, o1, _, _ = self._crop(self._audio_seconds(clap=clap, semantic=sem), self.sem_win)
            return clap[o0].unsqueeze(0), self._sem(sem, o0, o1)
        if self.stage == 'coarse':
            clap, sem, coarse = row
            o0, o1, i0, i1 = self._crop(self._audio_seconds(clap=clap, semantic=sem, coarse=coarse), self.sem_win, self.coarse_win)
            return clap[o0].unsqueeze(0), self._sem(sem, i0, i1), self._ac(coarse, i0, i1)
        clap, coarse, fine = row
        o0, o1, i0, i1 = self._crop(self._audio_seconds(clap=clap, coarse=coarse, fine=fine), self.sem_win, self.fine_win)
        return clap[o0].unsqueeze(0), self._ac(coarse, i0, i1), self._ac(fine, i0, i1)


def concatenate_fn(batch):
    """data.py:433-435: items carry a leading dim of 1; a batch is their concatenation (per field)."""
    if isinstance(batch[0], (tuple, list)):
        return tuple(torch.cat([item[i] for item in batch], dim=0) for i in range(len(batch[0])))
    return torch.cat(batch, dim=0)


def get_preprocessed_dataloader(ds, **kwargs):
    return DataLoader(ds, collate_fn=concatenate_fn, **kwargs)


class SoundDataset(Dataset):
    def __init__(self, *a, **k):
        raise ImportError("SoundDataset needs torchaudio and the pretrained tokenizers (CLAP / MERT / Encodec), which are "
                          "outside the MI355X hot path; train from preprocessed tokens (use_preprocessed_data=True) or pass "
                          "a token dataset via dataset=")


def get_dataloader(ds, **kwargs):
    return DataLoader(ds, **kwargs)
