"""Round-2 golden vectors, produced by running the REAL reference (/root/reference) on CPU (build container only):

    python oracle/make_golden_r2.py

  tests/golden/musiclm_forward.npz   reference MusicLM.forward (open_musiclm.py:864-1035) on tiny stages: conditioning ids
                                     injected through a stand-in `clap`, every uniform draw of gumbel_noise (utils.py:73-75)
                                     recorded in call order, every stage.generate output and the final [coarse | fine] ids
                                     recorded (a stand-in neural codec captures them): pins the window stitcher.
  tests/golden/kmeans_fit.npz        the reference's learn_kmeans (hf_hubert_kmeans.py:121-149: sklearn MiniBatchKMeans under
                                     np.random.seed) on seeded features: pins the k-means FIT of HfHubertKmeansTrainer.
  tests/golden/preprocessed/         a small token store written with the reference's sqlite adapters (data.py:32-52,
                                     preprocess.py:198-200) + the crops the reference's PreprocessedDataset (data.py:304-429)
                                     returns under random.seed(s): pins our reader crop for crop.
"""
from __future__ import annotations

import importlib
import os
import random
import shutil
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.make_golden import OUT, import_reference, to_np  # noqa: E402


def make_musiclm_forward(ref):
    ref_utils = importlib.import_module("open_musiclm.utils")
    torch.manual_seed(21)
    tiny = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.0)
    cb = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    sem = ref.create_semantic_transformer(**tiny, clap_codebook_size=32, semantic_codebook_size=48)
    coarse = ref.create_coarse_transformer(**tiny, num_coarse_quantizers=3, **cb)
    fine = ref.create_fine_transformer(**tiny, num_coarse_quantizers=3, num_fine_quantizers=5,
                                       clap_codebook_size=32, acoustic_codebook_size=40)
    B = 2
    clap_ids = torch.randint(0, 32, (B, 12, 1), generator=torch.Generator().manual_seed(3))

    class Clap:                                    # stand-in for ClapQuantized: returns the injected conditioning ids
        def __call__(self, *, text_input=None, audio_input=None, **kw):
            return clap_ids.clone()

    captured = {}

    class Codec:                                   # stand-in for the Encodec wrapper: captures the final acoustic ids
        sample_rate = 24000
        def decode_from_codebook_indices(self, ids):
            captured["acoustic"] = ids.clone()
            return torch.zeros(ids.shape[0], 1, 8)

    mlm = ref.MusicLM(wav2vec=None, clap=Clap(), neural_codec=Codec(), semantic_transformer=sem, coarse_transformer=coarse,
                      fine_transformer=fine)
    uniforms, calls = [], []
    orig_noise = ref_utils.gumbel_noise

    def recording_noise(t):
        u = torch.zeros_like(t).uniform_(0, 1)
        uniforms.append(u.clone())
        return -ref_utils.log(-ref_utils.log(u))
    ref_utils.gumbel_noise = recording_noise
    for name in ("semantic", "coarse", "fine"):
        stage = getattr(mlm, name)
        def wrap(fn, name=name):
            def inner(*a, **k):
                first = len(uniforms)
                out = fn(*a, **k)
                calls.append((name, first, len(uniforms), out.clone()))
                return out
            return inner
        stage.generate = wrap(stage.generate)
    kw = dict(output_seconds=4, semantic_window_seconds=2, coarse_window_seconds=2, fine_window_seconds=1,
              semantic_steps_per_second=6, acoustic_steps_per_second=4)
    try:
        torch.manual_seed(99)
        mlm(text=["x"] * B, **kw)
    finally:
        ref_utils.gumbel_noise = orig_noise
    out = {}
    for pfx, m in (("sem", sem), ("coarse", coarse), ("fine", fine)):
        out.update({f"sd.{pfx}." + k: v for k, v in to_np(m.state_dict()).items()})
    out["clap_ids"] = clap_ids.numpy()
    out["acoustic"] = captured["acoustic"].numpy()
    out["n_calls"] = np.int64(len(calls))
    for i, (name, a, b, ids) in enumerate(calls):
        out[f"call.{i}.stage"] = np.array(name)
        out[f"call.{i}.ids"] = ids.numpy()
        # the draws of this generate call, in order; every draw is [B, V + 1]
        out[f"call.{i}.uniforms"] = torch.stack(uniforms[a:b]).numpy() if b > a else np.zeros((0, B, 1), np.float32)
    out["meta.kwargs"] = np.array(repr(kw))
    out["meta.tiny"] = np.array(repr(tiny))
    np.savez_compressed(os.path.join(OUT, "musiclm_forward.npz"), **out)
    print(f"[musiclm_forward] {len(calls)} stage.generate calls ({[c[0] for c in calls]}), {len(uniforms)} uniform draws, "
          f"acoustic ids {tuple(captured['acoustic'].shape)}")


def make_preprocessed(ref):
    data = importlib.import_module("open_musiclm.data")
    folder = os.path.join(OUT, "preprocessed")
    shutil.rmtree(folder, ignore_errors=True)
    os.makedirs(folder)
    conn, cur = data.init_sqlite(os.path.join(folder, "preprocessed.db"))
    # schema and blob format of the reference's writer (preprocess.py:198-200): uint16 arrays through adapt_array
    cur.execute("create table if not exists tokens(idx integer primary key, path text, clap array, semantic array, coarse array, fine array)")
    rng = np.random.RandomState(7)
    sem_hz, ac_hz, sem_win = 5, 3, 4               # small rates keep the store tiny; window arithmetic is rate-independent
    for idx, secs in enumerate([6, 9, 4, 12]):
        clap = rng.randint(0, 1024, (secs - sem_win + 1, 12, 1)).astype(np.uint16)
        semantic = rng.randint(0, 1024, (1, secs * sem_hz - 1)).astype(np.uint16)
        coarse = rng.randint(0, 1024, (1, secs * ac_hz, 3)).astype(np.uint16)
        fine = rng.randint(0, 1024, (1, secs * ac_hz, 5)).astype(np.uint16)
        cur.execute("insert into tokens(idx, path, clap, semantic, coarse, fine) values(?, ?, ?, ?, ?, ?)",
                    (idx * 3 + 1, f"file{idx}.wav", clap, semantic, coarse, fine))
    conn.commit()
    out = {}
    for stage in ("semantic", "coarse", "fine"):
        ds = data.PreprocessedDataset(folder, stage=stage, semantic_window_seconds=sem_win, coarse_window_seconds=2,
                                      fine_window_seconds=1, semantic_steps_per_second=sem_hz, acoustic_steps_per_second=ac_hz)
        for seed in (0, 1, 2):
            random.seed(1000 + seed)
            for i in range(len(ds)):
                for f, t in enumerate(ds[i]):
                    out[f"{stage}.{seed}.{i}.{f}"] = t.numpy()
        ds.conn.close()
    conn.close()
    out["meta"] = np.array(repr(dict(semantic_window_seconds=sem_win, coarse_window_seconds=2, fine_window_seconds=1,
                                     semantic_steps_per_second=sem_hz, acoustic_steps_per_second=ac_hz)))
    np.savez_compressed(os.path.join(OUT, "preprocessed_crops.npz"), **out)
    print(f"[preprocessed] store + {len(out) - 1} reference crops written")



def make_preprocess_store(ref):
    """The token store as the REFERENCE's DataPreprocessor.process writes it (preprocess.py:252-284), tokenizers and audio batches
    from oracle/preprocess_standins.py.  The reference's constructor needs torchaudio + an audio folder + accelerate, so the object
    is assembled field by field and only its own generate_tokens_from_batch / process run; world size 1 and a 2-rank store."""
    import types
    from oracle import preprocess_standins as S
    pp = importlib.import_module("open_musiclm.preprocess")
    data = importlib.import_module("open_musiclm.data")
    base = os.path.join(OUT, "preprocess_store")
    shutil.rmtree(base, ignore_errors=True)

    def run(folder, shards, clap_batch_size):
        os.makedirs(folder)
        for rank, items in enumerate(shards):
            obj = pp.DataPreprocessor.__new__(pp.DataPreprocessor)
            torch.nn.Module.__init__(obj)
            obj.accelerator = types.SimpleNamespace(num_processes=len(shards), process_index=rank, unwrap_model=lambda m: m,
                                                    is_main_process=rank == 0, print=print)
            obj.wav2vec, obj.audio_conditioner, obj.neural_codec = S.Wav2Vec(), S.Clap(), S.Codec()
            obj.num_coarse_quantizers, obj.clap_audio_length_seconds, obj.clap_batch_size = 3, S.WINDOW_S, clap_batch_size
            obj.num_crops, obj.replace_existing = 1, False
            obj.register_buffer("steps", torch.Tensor([0]))
            obj.ds_fields = ("raw_wave_for_clap", "raw_wave_for_semantic", "raw_wave_for_acoustic")
            obj.ds = list(range(len(shards) * len(items)))               # process() only takes its length (the UNsharded dataset's)
            obj.dl_iter = pp.cycle(items)
            obj.conn, obj.cursor = data.init_sqlite(os.path.join(folder, "preprocessed.db"))
            obj.cursor.execute("CREATE TABLE IF NOT EXISTS tokens(idx integer primary key, path text, clap array, semantic array, coarse array, fine array)")
            obj.process()
            obj.conn.close()

    items = S.batches()
    run(os.path.join(base, "world1"), [items], clap_batch_size=2)
    run(os.path.join(base, "world2"), [items[0:4:2], items[1:4:2]], clap_batch_size=32)
    print("[preprocess_store] reference-written stores: world1 (5 files), world2 (2 ranks x 2 files)")

def make_kmeans_fit(ref):
    import tempfile
    hk = importlib.import_module("open_musiclm.hf_hubert_kmeans")
    rng = np.random.RandomState(11)
    centers = rng.randn(16, 16).astype(np.float32) * 2
    feats = (centers[rng.randint(0, 16, 3000)] + 0.3 * rng.randn(3000, 16)).astype(np.float32)
    kw = dict(n_clusters=16, max_iter=30, batch_size=512, n_init=3, max_no_improvement=20)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "kmeans.joblib")
        hk.learn_kmeans(feats, 5, path, **kw)
        import joblib
        km = joblib.load(path)
    np.savez_compressed(os.path.join(OUT, "kmeans_fit.npz"), features=feats, centers=km.cluster_centers_.astype(np.float64),
                        seed=np.array(5), kwargs=np.array(repr(kw)))
    print(f"[kmeans_fit] reference learn_kmeans: {km.cluster_centers_.shape} centroids, inertia {km.inertia_:.4f}")


if __name__ == "__main__":
    ref = import_reference()
    only = set(sys.argv[1:])
    if not only or "musiclm_forward" in only:
        make_musiclm_forward(ref)
    if not only or "preprocessed" in only:
        make_preprocessed(ref)
    if not only or "kmeans_fit" in only:
        make_kmeans_fit(ref)
    if not only or "preprocess_store" in only:
        make_preprocess_store(ref)
