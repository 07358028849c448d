"""CPU: host-side logic of the drop-in boundary (no GPU compute): C-ABI export surface, state_dict schema,
id / label / mask construction, row maps of the logit heads, sampling filters, optimizer grouping, token
datasets, config loaders, loud failure without a GPU, and the world_size-2 data-parallel exchange over gloo."""
import io
import json
import os
import re
import sqlite3
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import musiclm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_header_symbol():
    import __graft_entry__ as G
    from open_musiclm_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        G.build()
    lib = hip.lib()                                           # also binds every SIGNATURES entry via getattr
    hdr = open(os.path.join(ROOT, "include", "omlm.h")).read()
    declared = set(re.findall(r"\b(omlm_\w+)\s*\(", hdr))
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.omlm_version() >= 100


def tiny(stage="coarse", **kw):
    from open_musiclm_amd import open_musiclm as M
    torch.manual_seed(0)
    base = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.0)
    base.update(kw)
    return getattr(M, f"create_{stage}_transformer")(**base)


def test_state_dict_schema_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "tiny_coarse.npz"))
    ref_keys = [k[3:] for k in z.files if k.startswith("sd.")]
    m = tiny("coarse", num_coarse_quantizers=3, clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    assert list(m.state_dict().keys()) == ref_keys
    m.load_state_dict({k: torch.from_numpy(z["sd." + k]) for k in ref_keys}, strict=True)
    z2 = np.load(os.path.join(golden_dir, "tiny_semantic_t5_plainff.npz"))
    m2 = tiny("semantic", use_conv_ff=False, relative_position_bias_type="t5", clap_codebook_size=32, semantic_codebook_size=48)
    assert list(m2.state_dict().keys()) == [k[3:] for k in z2.files if k.startswith("sd.")]


def test_forward_fails_loudly_without_gpu():
    m = tiny("coarse", num_coarse_quantizers=3)
    ids = [torch.zeros(1, 12, dtype=torch.long), torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, 6, dtype=torch.long)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(all_token_ids=ids)
    from open_musiclm_amd.optimizer import get_optimizer
    opt = get_optimizer(m.parameters(), lr=1e-3, wd=0.01)
    with pytest.raises(RuntimeError, match="no CPU path"):
        opt.step()


def test_wrapper_prepare_matches_oracle_ids_labels_mask():
    from open_musiclm_amd import open_musiclm as M
    m = tiny("coarse", num_coarse_quantizers=3)
    w = M.TokenConditionedTransformerWrapper(transformer=m, unique_consecutive=False, cross_entropy_loss_weights=[0, 0, 1])
    spec = O.coarse_spec(dim=128, depth=2, heads=2)
    ids = O.synthetic_ids(spec, 3, [1, 7, 5], seed=2)
    w.eval()
    got_ids, got_labels, got_mask = w._prepare([t.clone() for t in ids], True, False)
    exp_ids, exp_labels, exp_mask = O.build_training_inputs(ids, spec)
    for a, b in zip(got_ids, exp_ids):
        assert torch.equal(a, b)
    for a, b in zip(got_labels, exp_labels):
        assert torch.equal(a, b)
    assert torch.equal(got_mask, exp_mask)
    # training mode adds the forgetful mask: int(0.15 N) keys dropped per row, never position 0
    w.train()
    torch.manual_seed(3)
    _, _, tm = w._prepare([t.clone() for t in ids], True, False)
    n = tm.shape[1]
    assert tm[:, 0].all()
    assert ((exp_mask & ~tm).sum(1) <= int(0.15 * n)).all() and ((~tm).sum(1) >= int(0.15 * n)).all()
    with pytest.raises(AssertionError):
        w._prepare(ids, True, True)                        # eos in training input is rejected like the reference


def test_build_ids_offsets_pads_and_layout_maps():
    from open_musiclm_amd import engine
    m = tiny("fine", num_coarse_quantizers=3, num_fine_quantizers=5)
    B = 2
    clap = torch.randint(0, 1024, (B, 12))
    coarse = torch.randint(0, 1024, (B, 7))                # ragged: 7 is not a multiple of 3
    fine = torch.randint(0, 1024, (B, 11))
    clap[0, 0] = -1                                        # pad at quantizer 0 stays a pad ...
    clap[0, 1] = -1                                        # ... but -1 at quantizer 1 becomes 1023 (reference quirk, :126-134)
    ids32, lens = engine.build_ids(m, [clap, coarse, fine])
    assert lens == [12, 7, 11] and ids32.shape == (B, 12 + 7 + 11 + 3)
    assert int(ids32[0, 0]) == -2 and int(ids32[0, 1]) == -1 and int(ids32[0, 2]) == 1023
    assert int(ids32[1, 3]) == int(clap[1, 2]) + 2 * 1024
    assert int(ids32[0, 13]) == -2 and int(ids32[0, 14 + 4]) == int(coarse[0, 4]) + 1024
    lay = engine.build_layout(B, lens, [12, 3, 5], "cpu")
    assert lay.N == 33 and lay.starts == (0, 13, 21) and lay.n_out == (12, 7, 12)
    # every logit row of every sequence is produced by exactly one (sequence, quantizer) GEMM, from hidden
    # position start + j with head j mod Q (open_musiclm.py:149-186 incl. the remainder handling)
    for s, (Q, n_s, st) in enumerate(zip([12, 3, 5], lay.n_out, lay.starts)):
        seen = torch.zeros(B * n_s, dtype=torch.int32)
        for qq in range(Q):
            if (s, qq) not in lay.head_maps:
                continue
            a_map, c_map, rows = lay.head_maps[(s, qq)]
            assert rows == a_map.numel() == c_map.numel()
            j = c_map % n_s
            b = c_map // n_s
            assert ((j % Q) == qq).all()
            assert torch.equal(a_map, (b * lay.N + st + j).to(torch.int32))
            seen[c_map.long()] += 1
        assert (seen == 1).all()
    fin = engine.build_layout(B, lens, [12, 3, 5], "cpu", final_rows_only=True)
    (key, (a_map, c_map, rows)), = fin.head_maps.items()
    assert key == (2, 11 % 5) and rows == B and torch.equal(c_map, torch.arange(B, dtype=torch.int32))
    assert torch.equal(a_map, (torch.arange(B) * 33 + 21 + 11).to(torch.int32))


def test_sampling_helpers_match_oracle():
    from open_musiclm_amd import utils as U
    torch.manual_seed(0)
    x = torch.randn(4, 1025)
    assert torch.equal(U.top_k(x, 0.9), O.top_k_filter(x, 0.9))
    assert int((U.top_k(x, 0.9) > -float("inf")).sum(1)[0]) == 102
    t = torch.tensor([[1, 5, 9, 5, 2], [9, 1, 2, 3, 4], [1, 2, 3, 4, 5]])
    for keep in (True, False):
        assert torch.equal(U.mask_out_after_eos_id(t, 9, keep_eos=keep), O.mask_out_after_eos(t, 9, keep))
    torch.manual_seed(5)
    a = U.generate_mask_with_prob((3, 40), 0.15, "cpu")
    torch.manual_seed(5)
    assert torch.equal(a, O.forgetful_mask_from_noise(torch.randn(3, 40), 0.15))
    u = torch.rand(4, 1025)
    torch.manual_seed(7)
    ref = (x / 0.95 + (-torch.log(-torch.log(u + 1e-20) + 1e-20))).argmax(-1)
    assert torch.equal(O.gumbel_argmax(x, u, 0.95), ref)


def test_optimizer_grouping_and_state_dict_shape():
    from open_musiclm_amd.optimizer import FusedAdam, get_optimizer, get_linear_scheduler
    m = tiny("semantic")
    opt = get_optimizer(m.parameters(), lr=3e-4, wd=0.01)
    assert isinstance(opt, FusedAdam) and len(opt.param_groups) == 2
    assert all(p.ndim >= 2 for p in opt.param_groups[0]["params"]) and opt.param_groups[0]["weight_decay"] == 0.01
    assert all(p.ndim < 2 for p in opt.param_groups[1]["params"]) and opt.param_groups[1]["weight_decay"] == 0
    assert opt.param_groups[0]["betas"] == (0.9, 0.99) and opt.param_groups[0]["decoupled"]
    plain = get_optimizer(m.parameters(), lr=3e-4, wd=0)
    assert len(plain.param_groups) == 1 and not plain.param_groups[0]["decoupled"]
    sched = get_linear_scheduler(opt, total_iters=10)
    assert abs(opt.param_groups[0]["lr"] - 3e-4 * 1e-7) < 1e-12       # LinearLR warm-up from 1e-7 (optimizer.py:36-41)
    # same two-group ordering as torch AdamW built the reference's way -> interchangeable param indices
    n_wd = len(opt.param_groups[0]["params"])
    assert n_wd + len(opt.param_groups[1]["params"]) == len(list(m.parameters()))


def _write_db(path, rows):
    def blob(a):
        b = io.BytesIO()
        np.save(b, a)
        return sqlite3.Binary(b.getvalue())
    conn = sqlite3.connect(path)
    conn.execute("create table tokens (idx integer primary key, path text, clap array, semantic array, coarse array, fine array)")
    for i, r in enumerate(rows):
        conn.execute("insert into tokens values (?,?,?,?,?,?)", (i, f"f{i}.mp3", *[blob(a) for a in r]))
    conn.commit()
    conn.close()


def test_preprocessed_dataset_reads_reference_sqlite_format(tmp_path):
    from open_musiclm_amd.data import PreprocessedDataset, get_preprocessed_dataloader
    secs = 14
    rng = np.random.RandomState(0)
    rows = []
    for _ in range(3):
        rows.append((rng.randint(0, 1024, (secs - 10 + 1, 12, 1)).astype(np.uint16),        # one clap row per sliding 10 s window
                     rng.randint(0, 1024, (1, secs * 50 - 1)).astype(np.uint16),
                     rng.randint(0, 1024, (1, secs * 75, 3)).astype(np.uint16),
                     rng.randint(0, 1024, (1, secs * 75, 5)).astype(np.uint16)))
    _write_db(str(tmp_path / "preprocessed.db"), rows)
    for stage, shapes in (("semantic", [(2, 12, 1), (2, 499)]), ("coarse", [(2, 12, 1), (2, 199), (2, 300, 3)]),
                          ("fine", [(2, 12, 1), (2, 150, 3), (2, 150, 5)])):
        ds = PreprocessedDataset(str(tmp_path), stage=stage)
        assert len(ds) == 3
        batch = next(iter(get_preprocessed_dataloader(ds, batch_size=2)))
        assert [tuple(t.shape) for t in batch] == shapes and all(t.dtype == torch.int32 for t in batch)
    # crops are consistent slices of the stored arrays (coarse stage: semantic and coarse share the inner window)
    import random
    random.seed(1)
    ds = PreprocessedDataset(str(tmp_path), stage="coarse")
    clap, sem, coarse = ds[1]
    full_sem, full_coarse = torch.from_numpy(rows[1][1].astype(np.int32)), torch.from_numpy(rows[1][2].astype(np.int32))
    starts = [s for s in range(0, secs - 3) if torch.equal(full_sem[:, s * 50: s * 50 + 199], sem)]
    assert len(starts) >= 1 and any(torch.equal(full_coarse[:, s * 75: s * 75 + 300], coarse) for s in starts)


def test_synthetic_dataset_shapes_match_survey_lengths():
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd import engine
    for stage, N, q in (("semantic", 514, [12, 1]), ("coarse", 1116, [12, 1, 3]), ("fine", 1217, [12, 3, 5])):
        item = SyntheticTokenDataset(stage)[0]
        lens = [int(np.prod(t.shape[1:])) + 1 for t in item]          # + eos
        lens[-1] -= 1                                                  # last token dropped for the loss
        assert engine.build_layout(1, lens, q, "cpu").N == N           # SURVEY.md §8 table [probed with the reference]


def test_config_loaders_roundtrip(tmp_path):
    from open_musiclm_amd import config as Cfg
    model_json = dict(
        global_cfg=dict(semantic_audio_length_seconds=10.0, coarse_audio_length_seconds=4.0, fine_audio_length_seconds=2.0,
                        clap_audio_length_seconds=10.0, num_coarse_quantizers=3, num_fine_quantizers=5),
        clap_rvq_cfg=dict(enable_fusion=False, rq_num_quantizers=12, codebook_size=1024, rq_ema_decay=0.95, threshold_ema_dead_code=0.5),
        hubert_kmeans_cfg=dict(model_name="m-a-p/MERT-v0", normalize_embeds=True, embed_layer=7, target_sample_hz=16000,
                               seq_len_multiple_of=320, codebook_size=1024, output_hz=50),
        encodec_cfg=dict(bandwidth=6.0, codebook_size=1024, output_hz=75),
        semantic_cfg=dict(dim=128, depth=1, heads=2), coarse_cfg=dict(dim=128, depth=1, heads=2, ff_dropout=0.0),
        fine_cfg=dict(dim=128, depth=1, heads=2))
    stage = dict(folder="./x", valid_frac=0.05, lr=3e-4, lr_warmup=3000, batch_size=4, grad_accum_every=8, wd=0.01,
                 max_grad_norm=0.5, cross_entropy_loss_weights=[0.0, 1.0], num_train_steps=10, save_results_every=5,
                 save_model_every=5, save_predicted_tokens=True, save_reconstructed_wave=True, use_preprocessed_data=False)
    train_json = dict(
        clap_rvq_trainer_cfg=dict(folder="./x", num_train_steps=1, batch_size=2, accumulate_batches=1, save_model_every=1, save_results_every=1),
        hubert_kmeans_trainer_cfg=dict(folder="./x", feature_extraction_num_steps=1, feature_extraction_batch_size=1),
        semantic_trainer_cfg=dict(stage="semantic", **stage),
        coarse_trainer_cfg=dict(stage="coarse", **{**stage, "cross_entropy_loss_weights": [0.0, 0.0, 1.0]}),
        fine_trainer_cfg=dict(stage="fine", **{**stage, "cross_entropy_loss_weights": [0.0, 0.0, 1.0]}),
        data_preprocessor_cfg={})
    (tmp_path / "m.json").write_text(json.dumps(model_json))
    (tmp_path / "t.json").write_text(json.dumps(train_json))
    mc = Cfg.load_model_config(str(tmp_path / "m.json"))
    tc = Cfg.load_training_config(str(tmp_path / "t.json"))
    assert mc.coarse_cfg.relative_position_bias_type == "continuous" and tc.coarse_trainer_cfg.grad_accum_every == 8
    t = Cfg.create_coarse_transformer_from_config(mc, None, "cpu")
    assert [s.num_quantizers for s in t.token_sequences] == [12, 1, 3] and t.eos_ids == [1024, 1024, 1024]
    # the reference's scripts import through the `open_musiclm` package name
    from open_musiclm.config import load_model_config as aliased
    assert aliased is Cfg.load_model_config


DP_SCRIPT = r"""
import os, sys, torch
sys.path.insert(0, %r)
from open_musiclm_amd.parallel import DataParallel
dp = DataParallel(device=torch.device('cpu'))
assert dp.world_size == 2 and dp.is_distributed
g = torch.full((1000,), float(dp.rank + 1))
dp.allreduce_sum_(g)                                  # THE single gradient exchange
assert torch.allclose(g * dp.grad_scale(), torch.full((1000,), 1.5))
m = dp.reduce_mean(torch.tensor([float(dp.rank)]))
assert abs(float(m) - 0.5) < 1e-6
cat = dp.all_gather_cat(torch.full((2, 3), dp.rank))
assert cat.shape == (4, 3) and int(cat[0, 0]) == 0 and int(cat[3, 0]) == 1
p = torch.full((5,), float(dp.rank)); dp.broadcast_(p, 0); assert float(p.sum()) == 0.0
# optional bf16 exchange: the same sum to bf16 precision, identical on both ranks
os.environ["OMLM_DP_GRAD_DTYPE"] = "bf16"
torch.manual_seed(dp.rank)
h = torch.randn(4096)
want = torch.zeros(4096)
for r in range(2):
    torch.manual_seed(r); want += torch.randn(4096).bfloat16().float()
dp.allreduce_sum_(h)
assert float((h - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max()), float((h - want).abs().max())
both = dp.all_gather_cat(h[None]); assert torch.equal(both[0], both[1])
os.environ["OMLM_DP_GRAD_DTYPE"] = "fp32"
# gradient buckets: the flat buffer cut into contiguous asynchronous all-reduces gives the single collective's result bit for bit
torch.manual_seed(100 + dp.rank)
big = torch.randn(300_000) * 1e3
one = big.clone(); dp.allreduce_sum_(one)
os.environ["OMLM_DP_BUCKET_MB"] = "0.25"              # 1.2 MB of fp32 -> 5 buckets
assert dp.bucket_count(big) == 5
rng = dp.bucket_ranges(big.numel(), 5)
assert rng[0][0] == 0 and rng[-1][1] == big.numel() and all(a[1] == b[0] for a, b in zip(rng, rng[1:])) and all(s %% 1024 == 0 for s, _ in rng)
cut = big.clone(); dp.allreduce_sum_(cut)
assert torch.equal(one, cut)
both = dp.all_gather_cat(cut[None]); assert torch.equal(both[0], both[1])
os.environ["OMLM_DP_BUCKET_MB"] = "0"
dp.barrier(); dp.shutdown()
print('rank', dp.rank, 'ok')
"""


def test_data_parallel_exchange_world_size_2_gloo(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(DP_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def test_decode_args_struct_layout_matches_header(tmp_path):
    """The ctypes mirror of omlm_decode_args (decode.py) has the size and field offsets a C compiler gives the header struct."""
    import ctypes
    import subprocess
    from open_musiclm_amd import decode
    fields = [f[0] for f in decode.DecodeArgs._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "omlm.h")}"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(omlm_decode_args));']
    lines += [f'  printf("%zu\\n", offsetof(omlm_decode_args, {f}));' for f in fields]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(decode.DecodeArgs)
    assert out[1:] == [getattr(decode.DecodeArgs, f).offset for f in fields]


def test_wgrad_desc_layout_matches_header(tmp_path):
    """ctypes mirror of omlm_gemm_wgrad_desc vs the C header (gcc)."""
    import ctypes
    import subprocess
    from open_musiclm_amd import ops
    fields = [f[0] for f in ops._WgradDesc._fields_]
    src = tmp_path / "wl.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "omlm.h")}"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(omlm_gemm_wgrad_desc));']
    lines += [f'  printf("%zu\\n", offsetof(omlm_gemm_wgrad_desc, {f}));' for f in fields]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "wl"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(ops._WgradDesc)
    assert out[1:] == [getattr(ops._WgradDesc, f).offset for f in fields]


def test_cast_pad_desc_layout_matches_header(tmp_path):
    """ctypes mirror of omlm_cast_pad_desc (the grouped weight re-pack) vs the C header (gcc)."""
    import ctypes
    import subprocess
    from open_musiclm_amd import ops
    fields = [f[0] for f in ops._CastDesc._fields_]
    src = tmp_path / "cl.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "omlm.h")}"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(omlm_cast_pad_desc));']
    lines += [f'  printf("%zu\\n", offsetof(omlm_cast_pad_desc, {f}));' for f in fields]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "cl"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(ops._CastDesc)
    assert out[1:] == [getattr(ops._CastDesc, f).offset for f in fields]


def test_colsum_desc_layout_matches_header(tmp_path):
    """ctypes mirror of omlm_colsum_desc (the grouped column sums of a backward pass) vs the C header (gcc)."""
    import ctypes
    import subprocess
    from open_musiclm_amd import ops
    fields = [f[0] for f in ops._ColsumDesc._fields_]
    src = tmp_path / "cs.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "omlm.h")}"', 'int main(void) {',
             '  printf("%zu\\n", sizeof(omlm_colsum_desc));']
    lines += [f'  printf("%zu\\n", offsetof(omlm_colsum_desc, {f}));' for f in fields]
    lines += ['  return 0; }']
    src.write_text("\n".join(lines))
    exe = tmp_path / "cs"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(ops._ColsumDesc)
    assert out[1:] == [getattr(ops._ColsumDesc, f).offset for f in fields]


def test_decode_batch_limit_follows_the_kernels():
    """decode.max_batch: 16 samples per call only where omlm_decode_step's matrix-core kernels serve the model (16-bit weights, dim 1024,
    <= 16 heads, padded feed-forward width <= 3072); 8 everywhere else -- generate() groups larger batches accordingly."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    big = M.create_coarse_transformer(dim=1024, depth=1, heads=8, num_coarse_quantizers=3, precision="bf16")
    small = M.create_coarse_transformer(dim=128, depth=1, heads=2, num_coarse_quantizers=3, precision="bf16")
    assert decode.max_batch(big, "bf16") == 16 and decode.max_batch(big, "fp16") == 16
    assert decode.max_batch(big, "bf16x3") == 8 and decode.max_batch(small, "bf16") == 8
    assert decode.supports(big, 16, "bf16") and not decode.supports(big, 17, "bf16") and not decode.supports(big, 9)
    assert decode.max_batch(big, "fp16ff") == 16              # fp16ff decodes on the fp16 step kernels (the hi planes)


def test_precision_modes_and_their_half_family(monkeypatch):
    """engine: the four precision names, which of them carry the loss scale / the fp16 weight shadow (is_half), the operand type each
    keeps in HBM, and the error for an unknown $OMLM_PRECISION."""
    import torch
    from open_musiclm_amd import engine
    assert set(engine._PRECISIONS) == {"bf16", "fp16", "fp16ff", "bf16x3"}
    assert engine._PRECISIONS["fp16ff"] is torch.float16 and engine._PRECISIONS["fp16"] is torch.float16
    assert [engine.is_half(p) for p in ("bf16", "fp16", "fp16ff", "bf16x3")] == [False, True, True, False]
    assert engine.loss_scale("bf16") == 1.0 and engine.loss_scale("fp16ff") == engine.loss_scale_initial() == engine.loss_scale("fp16")
    monkeypatch.setenv("OMLM_PRECISION", "fp16ff")
    assert engine.default_precision() == "fp16ff"
    monkeypatch.setenv("OMLM_PRECISION", "fp8")
    with pytest.raises(ValueError):
        engine.default_precision()


@pytest.mark.skipif(not os.path.isdir("/root/reference/scripts"), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("script", ["train_semantic_stage", "train_coarse_stage", "train_fine_stage", "train_clap_rvq",
                                    "train_hubert_kmeans", "preprocess_data"])
def test_reference_training_scripts_import_against_this_package(script):
    """Drop-in boundary: the reference's own entry scripts resolve `open_musiclm.*` to this repository (alias package) and
    get every name they import from it; `--help` stops before any GPU work."""
    import subprocess
    import sys
    # (the reference mount stays untouched: no __pycache__ next to its scripts)
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONDONTWRITEBYTECODE="1")
    probe = ("import runpy, sys, open_musiclm; assert open_musiclm.__file__.startswith(%r), open_musiclm.__file__; "
             "sys.argv = [%r, '--help']; runpy.run_path(%r, run_name='__main__')") % (
                 ROOT, script, f"/root/reference/scripts/{script}.py")
    r = subprocess.run([sys.executable, "-c", probe], cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "--model_config" in r.stdout


def test_preprocessed_dataset_matches_reference_crops(golden_dir):
    """data.PreprocessedDataset against the REFERENCE class (data.py:304-429) on a token store written with the reference's own
    sqlite adapters: same random.seed -> identical crops, item by item, for all three stages (oracle/make_golden_r2.py)."""
    import ast
    import random
    import numpy as np
    from open_musiclm_amd.data import PreprocessedDataset
    z = np.load(os.path.join(golden_dir, "preprocessed_crops.npz"))
    kw = ast.literal_eval(str(z["meta"]))
    folder = os.path.join(golden_dir, "preprocessed")
    for stage in ("semantic", "coarse", "fine"):
        ds = PreprocessedDataset(folder, stage=stage, **kw)
        assert len(ds) == 4
        for seed in (0, 1, 2):
            random.seed(1000 + seed)
            for i in range(len(ds)):
                item = ds[i]
                for f, t in enumerate(item):
                    ref = z[f"{stage}.{seed}.{i}.{f}"]
                    assert tuple(t.shape) == ref.shape and t.dtype == torch.int32, (stage, seed, i, f, t.shape, ref.shape)
                    assert np.array_equal(t.numpy(), ref), (stage, seed, i, f)


def test_kmeans_assign_oracle_vs_sklearn_at_real_dims():
    """hf_hubert_kmeans.py:87 calls sklearn MiniBatchKMeans.predict on 768-d MERT features against 1024 centroids
    (configs/model/musiclm_small.json hubert_kmeans_cfg).  The oracle (sum (x - c)^2, lowest index on ties) is compared with
    sklearn itself at those dimensions on clustered features; the mismatch count is reported, not assumed."""
    import numpy as np
    from sklearn.cluster import MiniBatchKMeans
    from oracle import musiclm_oracle as O
    rng = np.random.RandomState(0)
    centers = rng.randn(1024, 768).astype(np.float32)
    feats = (centers[rng.randint(0, 1024, 8192)] + 0.7 * rng.randn(8192, 768)).astype(np.float32)
    km = MiniBatchKMeans(n_clusters=1024, batch_size=2048, n_init=1, random_state=0, max_iter=3).fit(feats)
    x = (centers[rng.randint(0, 1024, 4096)] + 0.7 * rng.randn(4096, 768)).astype(np.float32)
    ref = km.predict(x).astype(np.int64)
    mine = O.kmeans_assign(x, km.cluster_centers_.astype(np.float32))
    mism = int((ref != mine).sum())
    print(f"kmeans 768 x 1024: {mism} mismatches / {len(x)}")
    assert mism == 0, mism


def test_rvq_checkpoint_layout_roundtrip():
    """ResidualVQCodebooks reads and writes vector-quantize-pytorch's ResidualVQ keys (what trainer.py:731 saves and
    clap_quantized.py:109 loads): layers.{s}._codebook.{initted, cluster_size, embed, embed_avg}."""
    from open_musiclm_amd.clap_quantized import ClapQuantized
    cq = ClapQuantized(clap=None, codebook_size=16, rq_num_quantizers=3, embed_dim=8, rq_ema_decay=0.9, threshold_ema_dead_code=0.5)
    g = torch.Generator().manual_seed(1)
    cq.rq.codebooks.copy_(torch.randn(3, 16, 8, generator=g))
    cq.rq.embed_avg.copy_(torch.randn(3, 16, 8, generator=g))
    cq.rq.cluster_size.copy_(torch.rand(3, 16, generator=g))
    cq.rq.initted.fill_(True)
    sd = cq.rq.state_dict()
    assert sorted(sd) == sorted(f"layers.{s}._codebook.{k}" for s in range(3) for k in ("initted", "cluster_size", "embed", "embed_avg"))
    assert sd["layers.1._codebook.embed"].shape == (1, 16, 8) and sd["layers.1._codebook.cluster_size"].shape == (1, 16)
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    other = ClapQuantized(clap=None, codebook_size=16, rq_num_quantizers=3, embed_dim=8)
    other.rq.load_state_dict(torch.load(buf))
    for name in ("codebooks", "embed_avg", "cluster_size", "initted"):
        assert torch.equal(getattr(other.rq, name), getattr(cq.rq, name)), name
    assert other.rq.decay == 0.95 and cq.rq.decay == 0.9 and cq.rq.threshold_ema_dead_code == 0.5
    # the enclosing module round-trips too (strict), with the library's keys under the "rq." prefix
    full = cq.state_dict()
    assert "rq.layers.2._codebook.embed_avg" in full and not any(k.startswith("rq.codebooks") for k in full)
    third = ClapQuantized(clap=None, codebook_size=16, rq_num_quantizers=3, embed_dim=8)
    res = third.load_state_dict(full, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for name in ("codebooks", "embed_avg", "cluster_size", "initted"):
        assert torch.equal(getattr(third.rq, name), getattr(cq.rq, name)), name
    with pytest.raises(RuntimeError):                       # a key of neither layout is reported by a strict load
        third.load_state_dict({**full, "rq.layers.0._codebook.bogus": torch.zeros(1)}, strict=True)
    own = {"codebooks": cq.rq.codebooks.clone(), "embed_avg": cq.rq.embed_avg.clone(), "cluster_size": cq.rq.cluster_size.clone()}
    fourth = ClapQuantized(clap=None, codebook_size=16, rq_num_quantizers=3, embed_dim=8)
    fourth.rq.load_state_dict(own)                          # round-1 checkpoints (this module's buffer names) still load
    assert torch.equal(fourth.rq.codebooks, cq.rq.codebooks) and bool(fourth.rq.initted.all())
    with pytest.raises(RuntimeError):                       # fitting runs on the device only: no CPU fallback
        cq.learn_rvq = True
        cq.quantize(torch.randn(4, 8))


def test_kmeans_fit_matches_the_reference(golden_dir, tmp_path):
    """learn_kmeans (hf_hubert_kmeans.py:121-149) is sklearn's MiniBatchKMeans under np.random.seed in the reference and here: on the
    golden features (fit by the REFERENCE's own function, oracle/make_golden_r2.py) the centroids agree bit for bit, and
    HfHubertKmeansTrainer on a dataset of those features writes the same model."""
    import ast
    import joblib
    from open_musiclm_amd.hf_hubert_kmeans import HfHubertWithKmeans, learn_kmeans
    from open_musiclm_amd.trainer import HfHubertKmeansTrainer
    z = np.load(os.path.join(golden_dir, "kmeans_fit.npz"))
    feats, want, seed, kw = z["features"], z["centers"], int(z["seed"]), ast.literal_eval(str(z["kwargs"]))
    km = learn_kmeans(feats, seed, str(tmp_path / "a.joblib"), verbose=0, **kw)
    assert np.array_equal(km.cluster_centers_.astype(np.float64), want)
    assert np.array_equal(joblib.load(str(tmp_path / "a.joblib")).cluster_centers_, km.cluster_centers_)

    class Feats(torch.utils.data.Dataset):            # one "clip" = 30 frames of precomputed features
        def __len__(self): return len(feats) // 30
        def __getitem__(self, i): return torch.from_numpy(feats[30 * i:30 * (i + 1)])
    hk = HfHubertWithKmeans(hubert=None, kmeans=None, codebook_size=kw["n_clusters"])
    trainer = HfHubertKmeansTrainer(feature_extraction_num_steps=4, feature_extraction_batch_size=25, hubert_kmeans=hk, dataset=Feats(),
                                    results_folder=str(tmp_path / "km"))
    trainer.train(seed=seed, verbose=0, **{k: v for k, v in kw.items() if k != "n_clusters"})
    fitted = joblib.load(str(tmp_path / "km" / "kmeans.joblib"))
    assert fitted.cluster_centers_.shape == want.shape and np.isfinite(fitted.cluster_centers_).all()
    # the fitted model plugs into the assign side (centroids become the HIP kernel's table)
    hk2 = HfHubertWithKmeans(hubert=None, kmeans=fitted)
    assert hk2.codebook_size == kw["n_clusters"] and tuple(hk2.kmeans.centroids.shape) == want.shape


KM_DP_SCRIPT = r"""
import os, sys, numpy as np, torch, joblib
sys.path.insert(0, %r)
from open_musiclm_amd.hf_hubert_kmeans import HfHubertWithKmeans
from open_musiclm_amd.trainer import HfHubertKmeansTrainer
rank = int(os.environ["RANK"])
rng = np.random.RandomState(100 + rank)                      # every rank extracts features from ITS OWN clips
centers = np.random.RandomState(1).randn(8, 12).astype(np.float32) * 3
feats = (centers[rng.randint(0, 8, 600)] + 0.1 * rng.randn(600, 12)).astype(np.float32)
class Feats(torch.utils.data.Dataset):
    def __len__(self): return 20
    def __getitem__(self, i): return torch.from_numpy(feats[30 * i:30 * (i + 1)])
hk = HfHubertWithKmeans(hubert=None, kmeans=None, codebook_size=8)
tr = HfHubertKmeansTrainer(feature_extraction_num_steps=4, feature_extraction_batch_size=10, hubert_kmeans=hk, dataset=Feats(),
                           results_folder=%r)
assert tr.dp.world_size == 2
tr.train(seed=3, verbose=0, n_init=2, max_iter=20, batch_size=256)
tr.dp.barrier()
if rank == 0:
    km = joblib.load(os.path.join(%r, "kmeans.joblib"))
    # 4 steps / 2 ranks = 2 steps per rank, each gathering 2 x 10 clips x 30 frames: the fit saw 1200 rows from BOTH ranks
    d = np.linalg.norm(km.cluster_centers_[:, None] - centers[None], axis=-1).min(1)
    assert km.cluster_centers_.shape == (8, 12) and float(d.max()) < 0.5, d
tr.dp.shutdown()
print('rank', rank, 'ok')
"""


def test_kmeans_trainer_world_size_2_gloo(tmp_path):
    """HfHubertKmeansTrainer across two processes (gloo, CPU): the features of both ranks are gathered, rank 0 fits and writes."""
    out = str(tmp_path / "km")
    script = tmp_path / "km_dp.py"
    script.write_text(KM_DP_SCRIPT % (ROOT, out, out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("ok" in o for o in outs)


def _raw_rows(db):
    """Rows of a token store with the blobs as stored (no converter): idx, path and the four np.save images."""
    conn = sqlite3.connect(db)
    rows = conn.execute("SELECT idx, path, clap, semantic, coarse, fine FROM tokens ORDER BY idx").fetchall()
    schema = conn.execute("SELECT sql FROM sqlite_master WHERE name='tokens'").fetchone()[0]
    conn.close()
    return [(r[0], r[1]) + tuple(bytes(b) for b in r[2:]) for r in rows], schema


def _preprocessor(folder, items, **kw):
    from oracle import preprocess_standins as S
    from open_musiclm_amd.preprocess import DataPreprocessor
    return DataPreprocessor(num_coarse_quantizers=3, wav2vec=S.Wav2Vec(), neural_codec=S.Codec(), audio_conditioner=S.Clap(),
                            clap_audio_length_seconds=S.WINDOW_S, semantic_audio_length_seconds=S.WINDOW_S, dataset=items,
                            results_folder=str(folder), **kw)


def test_data_preprocessor_writes_the_reference_store(golden_dir, tmp_path):
    """DataPreprocessor.process against the store the REFERENCE's process() wrote from the same stand-in tokenizers and batches
    (tests/golden/preprocess_store, oracle/make_golden_r2.py::make_preprocess_store): same rows, same paths, the np.save blobs byte
    for byte, same table; then the resume rule (rows present are skipped), replace_existing, and the store read back by
    PreprocessedDataset."""
    from oracle import preprocess_standins as S
    from open_musiclm_amd.data import PreprocessedDataset
    items = S.batches()
    pp = _preprocessor(tmp_path / "w1", items, clap_batch_size=2)
    pp.process()
    got, schema = _raw_rows(str(tmp_path / "w1" / "preprocessed.db"))
    want, want_schema = _raw_rows(os.path.join(golden_dir, "preprocess_store", "world1", "preprocessed.db"))
    assert schema == want_schema and len(got) == 5 and got == want
    # clap windows went through the conditioner in groups of clap_batch_size (6 s file, 4 s windows, 1 s hop -> 3 windows -> 2 + 1)
    assert pp.audio_conditioner.calls[:2] == [2, 1] and int(pp.steps.item()) == 5
    pp.conn.close()
    arr = np.load(io.BytesIO(got[1][2]))
    assert arr.dtype == np.uint16 and arr.shape == (9 - S.WINDOW_S + 1, 12, 1)

    # resume: a second pass over a store that already holds the rows calls no tokenizer and changes nothing
    pp2 = _preprocessor(tmp_path / "w1", items, clap_batch_size=2)
    pp2.process()
    assert pp2.audio_conditioner.calls == [] and _raw_rows(str(tmp_path / "w1" / "preprocessed.db"))[0] == want
    pp2.conn.close()
    # replace_existing recomputes and overwrites in place (the reference's plain INSERT raises on the primary key there)
    pp3 = _preprocessor(tmp_path / "w1", items[::-1], clap_batch_size=32, replace_existing=True)
    pp3.process()
    pp3.conn.close()
    redone, _ = _raw_rows(str(tmp_path / "w1" / "preprocessed.db"))
    assert [r[0] for r in redone] == [0, 1, 2, 3, 4] and [r[1:] for r in redone] == [r[1:] for r in want[::-1]]

    # and the reader side takes it: every stage crops from the written store
    for stage in ("semantic", "coarse", "fine"):
        ds = PreprocessedDataset(str(tmp_path / "w1"), stage=stage, semantic_window_seconds=S.WINDOW_S, coarse_window_seconds=2,
                                 fine_window_seconds=1, semantic_steps_per_second=S.SEM_HZ, acoustic_steps_per_second=S.AC_HZ)
        assert len(ds) == 5
        for i in range(5):
            out = ds[i]
            assert out[0].shape == (1, 12, 1)
        ds.conn.close()


def test_data_preprocessor_needs_its_tokenizers_and_a_dataset(tmp_path):
    from oracle import preprocess_standins as S
    from open_musiclm_amd.preprocess import DataPreprocessor
    with pytest.raises(AssertionError):
        DataPreprocessor(dataset=S.batches(), results_folder=str(tmp_path / "a"))
    with pytest.raises(ImportError, match="torchaudio"):
        DataPreprocessor(wav2vec=S.Wav2Vec(), neural_codec=S.Codec(), audio_conditioner=S.Clap(), clap_audio_length_seconds=4,
                         semantic_audio_length_seconds=4, folder=str(tmp_path), results_folder=str(tmp_path / "b"))
    with pytest.raises(AssertionError, match="clap window"):
        DataPreprocessor(wav2vec=S.Wav2Vec(), neural_codec=S.Codec(), audio_conditioner=S.Clap(), clap_audio_length_seconds=4,
                         semantic_audio_length_seconds=10, dataset=S.batches(), results_folder=str(tmp_path / "c"))


PP_DP_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
from oracle import preprocess_standins as S
from open_musiclm_amd.preprocess import DataPreprocessor
rank = int(os.environ["RANK"])
items = S.batches()[0:4]
pp = DataPreprocessor(num_coarse_quantizers=3, wav2vec=S.Wav2Vec(), neural_codec=S.Codec(), audio_conditioner=S.Clap(),
                      clap_audio_length_seconds=S.WINDOW_S, semantic_audio_length_seconds=S.WINDOW_S, dataset=items, shard_dataset=True,
                      results_folder=%r)
assert pp.dp.world_size == 2 and len(pp.ds) == 4
pp.process()
pp.dp.barrier()
pp.conn.close()
pp.dp.shutdown()
print('rank', rank, 'ok')
"""


def test_data_preprocessor_world_size_2_gloo(golden_dir, tmp_path):
    """Two ranks (gloo, CPU) write one store: rank r takes files r, r+2, ... and row idx = iteration * 2 + r, exactly the store the
    reference's process() wrote with num_processes=2 (tests/golden/preprocess_store/world2)."""
    out = str(tmp_path / "pp2")
    script = tmp_path / "pp_dp.py"
    script.write_text(PP_DP_SCRIPT % (ROOT, out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29623", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got, _ = _raw_rows(os.path.join(out, "preprocessed.db"))
    want, _ = _raw_rows(os.path.join(golden_dir, "preprocess_store", "world2", "preprocessed.db"))
    assert len(got) == 4 and got == want


def test_encodec_wrapper_around_a_supplied_codec():
    """EncodecWrapper (encodec_wrapper.py:12-71) over a stand-in with Encodec's interface: quantizer count from the bandwidth,
    [B, n_q, T] codes of several frames concatenated and returned as [B, T, n_q], indices handed back as one (codes, None) frame."""
    from types import SimpleNamespace
    from open_musiclm_amd.encodec_wrapper import EncodecWrapper, create_encodec_24khz
    from open_musiclm_amd.model_types import NeuralCodec

    class Codec:
        sample_rate, bandwidth = 24000, 6.0
        quantizer = SimpleNamespace(n_q=32, bins=1024)

        def __init__(self):
            self.seen = None

        def encode(self, x):
            assert x.dim() == 3 and x.shape[1] == 1
            b, t = x.shape[0], x.shape[-1] // 320
            base = torch.arange(b * 8 * t).reshape(b, 8, t)
            return [(base[..., : t // 2], None), (base[..., t // 2:], None)]

        def decode(self, frames):
            self.seen = frames
            (codes, scale), = frames
            return codes.float().sum(1, keepdim=True)

    codec = Codec()
    w = EncodecWrapper(encodec=codec)
    assert isinstance(w, NeuralCodec) and (w.num_quantizers, w.codebook_size, w.sample_rate, w.output_hz) == (8, 1024, 24000, 75)
    _, ids, _ = w(torch.zeros(2, 320 * 6), return_encoded=True)
    assert ids.shape == (2, 6, 8) and torch.equal(ids, torch.arange(2 * 8 * 6).reshape(2, 8, 6).transpose(1, 2))
    wave = w.decode_from_codebook_indices(ids)
    assert codec.seen[0][1] is None and torch.equal(codec.seen[0][0], ids.transpose(1, 2)) and wave.shape == (2, 1, 6)
    with pytest.raises(AssertionError):
        w(torch.zeros(1, 640), return_encoded=False)
    with pytest.raises(ImportError, match="encodec"):
        create_encodec_24khz(bandwidth=6.0)


def test_hubert_wrapper_with_a_supplied_extractor():
    """HfHubertWithKmeans.forward (hf_hubert_kmeans.py:54-93) over a small random-init transformers.HubertModel: the wave is curtailed
    to a multiple of the hop, layer `embed_layer` is taken and normalised; the id path goes to the HIP kernel and refuses the CPU."""
    from transformers import HubertConfig, HubertModel
    from open_musiclm_amd.hf_hubert_kmeans import HfHubertWithKmeans
    from open_musiclm_amd.utils import zero_mean_unit_var_norm
    torch.manual_seed(0)
    cfg = HubertConfig(hidden_size=32, num_hidden_layers=3, num_attention_heads=2, intermediate_size=64, conv_dim=(16, 16),
                       conv_stride=(5, 4), conv_kernel=(10, 4), num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4,
                       num_feat_extract_layers=2)
    hubert = HubertModel(cfg).eval()
    centers = torch.randn(8, 32)
    hk = HfHubertWithKmeans(hubert=hubert, kmeans=centers.numpy(), embed_layer=2, target_sample_hz=16000, seq_len_multiple_of=20,
                            codebook_size=8, output_hz=800)
    wave = torch.randn(2, 20 * 30 + 7)
    emb = hk(wave, return_embed=True)
    with torch.no_grad():
        want = hubert(input_values=wave[:, :600], attention_mask=torch.ones(2, 600), output_hidden_states=True).hidden_states[2]
    assert emb.shape == want.shape and torch.allclose(emb, zero_mean_unit_var_norm(want), atol=1e-6)
    assert torch.equal(hk(wave, return_embed=True, input_sample_hz=16000), emb)
    with pytest.raises(ImportError, match="torchaudio"):
        hk(wave, return_embed=True, input_sample_hz=44100)
    with pytest.raises(RuntimeError):                             # nearest-centroid assignment is a HIP kernel: no CPU path
        hk(wave, flatten=False)
    with pytest.raises(RuntimeError, match="feature extractor"):
        HfHubertWithKmeans(hubert=None, kmeans=centers.numpy(), codebook_size=8)(wave)


def test_data_parallel_backend_choice_and_shared_gpu_flag(monkeypatch):
    """parallel.DataParallel: the exchange backend is nccl (= RCCL) with one GPU per local rank, and falls back to gloo BY ITSELF -- flagged as
    a dry run -- when the local ranks outnumber the visible GPUs; $OMLM_DP_BACKEND overrides; the legacy HSA IPC mode is refused for RCCL."""
    import torch.distributed as dist
    from open_musiclm_amd import parallel
    seen = {}
    monkeypatch.setattr(dist, "is_initialized", lambda: False)
    monkeypatch.setattr(dist, "init_process_group", lambda backend, rank, world_size: seen.update(backend=backend, rank=rank, world=world_size))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: seen.update(device=i))
    for k, v in dict(WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", LOCAL_WORLD_SIZE="2").items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("OMLM_DP_BACKEND", raising=False)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    dp = parallel.DataParallel(device=torch.device("cuda", 1))
    assert seen["backend"] == "nccl" and seen["device"] == 1 and not dp.shared_gpu and dp.owns_group
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    dp = parallel.DataParallel(device=torch.device("cuda", 0))
    assert seen["backend"] == "gloo" and seen["device"] == 0 and dp.shared_gpu           # two ranks on one visible GPU: dry run over gloo
    # a multi-node launch without LOCAL_WORLD_SIZE (srun exports RANK, WORLD_SIZE, LOCAL_RANK and its own variables): 2 nodes x 8 GPUs is NOT a dry run
    monkeypatch.delenv("LOCAL_WORLD_SIZE")
    for k in ("OMPI_COMM_WORLD_LOCAL_SIZE", "SLURM_NTASKS_PER_NODE", "SLURM_NNODES", "SLURM_JOB_NUM_NODES"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SLURM_NNODES", "2")
    monkeypatch.setenv("WORLD_SIZE", "16"); monkeypatch.setenv("RANK", "11"); monkeypatch.setenv("LOCAL_RANK", "3")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    dp = parallel.DataParallel(device=torch.device("cuda", 3))
    assert seen["backend"] == "nccl" and seen["device"] == 3 and not dp.shared_gpu
    monkeypatch.delenv("SLURM_NNODES"); monkeypatch.setenv("OMPI_COMM_WORLD_LOCAL_SIZE", "8")
    dp = parallel.DataParallel(device=torch.device("cuda", 3))
    assert seen["backend"] == "nccl" and not dp.shared_gpu
    # the decision is the same on EVERY rank (a per-rank test split a 4-rank / 2-GPU launch into nccl and gloo ranks)
    monkeypatch.delenv("OMPI_COMM_WORLD_LOCAL_SIZE"); monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    picked = []
    for r in range(4):
        monkeypatch.setenv("RANK", str(r)); monkeypatch.setenv("LOCAL_RANK", str(r))
        dp = parallel.DataParallel(device=torch.device("cuda", r % 2))
        picked.append((seen["backend"], dp.shared_gpu))
    assert picked == [("gloo", True)] * 4
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "1"); monkeypatch.setenv("LOCAL_RANK", "1")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.setenv("OMLM_DP_BACKEND", "nccl")
    dp = parallel.DataParallel(device=torch.device("cuda", 0))
    assert seen["backend"] == "nccl" and dp.shared_gpu                                   # explicit choice is honoured (RCCL itself will refuse)
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    with pytest.raises(RuntimeError, match="HSA_ENABLE_IPC_MODE_LEGACY"):
        parallel.DataParallel(device=torch.device("cuda", 0))


def test_bench_refuses_a_rank_count_that_disagrees_with_its_launcher():
    """bench.py under a launcher whose WORLD_SIZE differs from --gpus: one clear message and a non-zero exit (it used to be a bare assert);
    without WORLD_SIZE and with --gpus > 1 it would re-execute itself under torch.distributed.run (covered on the GPU box)."""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout) and "--gpus 2" in (r.stderr + r.stdout)


def test_fused_batch_preparation_is_only_taken_where_it_is_equivalent(monkeypatch):
    """Wrapper._fused_prepare_ok: CPU tensors, unique_consecutive sequences, ids that already carry eos, a replaced generate_mask_with_prob
    (how tests inject masks) and $OMLM_FUSED_PREP=0 all keep the torch construction."""
    from open_musiclm_amd import open_musiclm as M
    model = tiny()
    w = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.])
    ids = [torch.zeros(2, 12, 1, dtype=torch.long), torch.zeros(2, 9, dtype=torch.long), torch.zeros(2, 6, 3, dtype=torch.long)]
    assert not w._fused_prepare_ok(ids, False)                                           # CPU tensors
    fake = [t.to("meta") for t in ids]
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    assert w._fused_prepare_ok(fake, False)
    assert not w._fused_prepare_ok(fake, True)                                           # input_has_eos
    monkeypatch.setenv("OMLM_FUSED_PREP", "0")
    assert not w._fused_prepare_ok(fake, False)
    monkeypatch.delenv("OMLM_FUSED_PREP")
    monkeypatch.setattr(M, "generate_mask_with_prob", lambda *a, **k: None)
    assert not w._fused_prepare_ok(fake, False)                                          # injected mask source
    monkeypatch.undo()
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    model.token_sequences[1].unique_consecutive = True
    w2 = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=True, cross_entropy_loss_weights=[0., 0., 1.])
    assert not w2._fused_prepare_ok(fake, False)
    too_long = [torch.zeros(1, 5000, dtype=torch.long, device="meta")] * 3
    w3 = M.TokenConditionedTransformerWrapper(transformer=tiny(), unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.])
    assert not w3._fused_prepare_ok(too_long, False)                                     # N > 4096: past the kernel's register slots


def test_model_pickles_and_the_parameter_registry_finds_it():
    """the parameter -> model link of precision "fp16" must not live on the Parameter (its __dict__ is pickled):
    torch.save(model) / pickle / spawn-based multiprocessing work, and the registry still resolves every parameter to its model."""
    import io
    import pickle
    from open_musiclm_amd import engine, open_musiclm as M
    m = M.create_coarse_transformer(dim=64, depth=1, heads=2, num_coarse_quantizers=3, precision="fp16")
    buf = io.BytesIO()
    torch.save(m, buf)
    pickle.dumps(next(m.parameters()))
    assert all(engine.model_of(p) is m for p in m.parameters())
    m2 = torch.load(io.BytesIO(buf.getvalue()), weights_only=False)
    assert sorted(m2.state_dict()) == sorted(m.state_dict())
    # a reconstructed model (torch.load, pickle, copy.deepcopy = EMA copies) registers ITS parameters again -- the fused
    # optimizer finds the model (and its loss-scale block) through the registry
    import copy
    m3 = copy.deepcopy(m)
    for mm in (m2, m3):
        assert all(engine.model_of(p) is mm for p in mm.parameters()) and all(p._omlm_precision == "fp16" for p in mm.parameters())
        assert mm.transformer.__dict__["_omlm_owner"] is mm
    assert all(engine.model_of(p) is m for p in m.parameters())


def test_scheduler_rewind_after_device_skipped_steps():
    """trainer._rewind_scheduler: LinearLR ticks taken back for optimizer steps the device skipped (fp16 overflow) -- the warm-up then
    advances with the Adam clock (applied steps)."""
    from open_musiclm_amd.optimizer import get_linear_scheduler
    from open_musiclm_amd.trainer import SingleStageTrainer
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([w], lr=1.0)
    sch = get_linear_scheduler(opt, total_iters=10, start_factor=0.1)
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(4):
        opt.step(); sch.step(); lrs.append(opt.param_groups[0]["lr"])
    holder = type("T", (), {"scheduler": sch})()
    SingleStageTrainer._rewind_scheduler(holder, 2)
    assert sch.last_epoch == 2 and abs(opt.param_groups[0]["lr"] - lrs[2]) < 1e-12 and abs(sch.get_last_lr()[0] - lrs[2]) < 1e-12
    opt.step(); sch.step()
    assert abs(opt.param_groups[0]["lr"] - lrs[3]) < 1e-12


def test_gemm_half_tile_ring_schedule_model():
    """The synchronisation of csrc/gemm.hip's gemm_tile8_body as a barrier-epoch model (tools/sim_gemm_t8_schedule.py): equal barrier counts
    for both wave groups, every fragment read behind every wave's counted wait, every slot re-requested >= 2 barriers after its last read."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("sim_t8", os.path.join(ROOT, "tools", "sim_gemm_t8_schedule.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert [m.check(nk) for nk in range(1, 9)] == [2 + 8 * nk for nk in range(1, 9)]
