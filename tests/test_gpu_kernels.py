"""GPU: every HIP kernel of libomlm_hip.so, called through the C ABI, against a plain PyTorch reference of the
same op (fp32 / fp64, torch ops on the same device).  Tolerances are written next to each check:
 * fp32 operands ("bf16x3" split) must reproduce fp32 math to ~1e-5 relative;
 * bf16 operands are compared against the reference evaluated on bf16-rounded inputs (so only accumulation
   order differs) to ~1e-5, or to the stated bf16 tolerance where outputs are rounded to bf16;
 * integer outputs (quantizer ids, sampled ids) must be bit-exact.
Every metric is also appended to gpurun_out/kernel_report.json for post-mortem reading."""
import json
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "kernel_report.json")


def report(name, **metrics):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        try:
            data = json.load(open(REPORT))
        except Exception:
            data = {}
    data[name] = {k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in metrics.items()}
    json.dump(data, open(REPORT, "w"), indent=1)


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from open_musiclm_amd import ops as o
    from open_musiclm_amd import hip
    hip.lib()       # fail loudly if the extension is missing
    return o


def test_probe_transpose_read(ops, dev):
    out = torch.zeros(64 * 4, dtype=torch.int16, device=dev)
    ops.probe_tr16(out)
    got = out.cpu().numpy().reshape(64, 4)
    exp = np.zeros((64, 4), dtype=np.int64)
    for l in range(64):
        g0, c = 16 * (l // 16), l % 16
        for j in range(4):
            exp[l, j] = 4 * (g0 + 4 * j + c // 4) + c % 4
    report("probe_tr16", match=bool((got == exp).all()), got=got.tolist())
    assert (got == exp).all(), f"ds_read_b64_tr_b16 semantic differs from the assumed one:\n{got}"


GEMM_CASES = [
    # M, N, K, a_kmaj, b_kmaj
    (200, 136, 192, False, False),
    (257, 1032, 64, False, False),
    (130, 72, 200, False, True),
    (96, 264, 1000, True, True),
    (1025, 128, 333, True, True),
    (384, 512, 512, False, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,akm,bkm", GEMM_CASES)
def test_gemm_layouts(ops, dev, dtype, M, N, K, akm, bkm):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    c8 = lambda x: (x + 7) // 8 * 8
    Am = torch.randn(M, K, generator=g).to(dtype)          # logical operands (asymmetric random data)
    Bm = torch.randn(N, K, generator=g).to(dtype)
    Kp = c8(K)

    def store(mat, kmaj, n):
        if kmaj:                                            # [K, ld8(n)]: pad columns hold garbage that must not leak
            st = torch.randn(K, c8(n), generator=g).to(dtype)
            st[:, :n] = mat.t()
            return st
        st = torch.zeros(n, Kp, dtype=dtype)                # [n, Kp]: contraction zero-padded to 8
        st[:, :K] = mat
        return st

    A, B = store(Am, akm, M).to(dev), store(Bm, bkm, N).to(dev)
    Kcall = K if (akm and bkm) else Kp                      # k-major operands are bounded by their row count
    Cin = torch.randn(M, N, generator=g).to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(A, B, C, M=M, N=N, K=Kcall, a_kmajor=akm, b_kmajor=bkm, Cin=Cin, alpha=0.5,
             a_rows=K if akm else M, b_rows=K if bkm else N)
    ref = 0.5 * (Am.double() @ Bm.double().t()).to(dev) + Cin.double()
    e = relerr(C, ref)
    report(f"gemm[{dtype},{M},{N},{K},{akm},{bkm}]", relerr=e)
    assert not torch.isnan(C).any()
    assert e < 2e-5, e          # fp32-grade: bf16x3 split, or exact products of bf16 inputs, fp32 accumulation


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,bkm,out16,resid", [
    (8192, 4096, 1024, False, True, False),      # 512 whole tiles: two per workgroup, counted first-iteration waits
    (8200, 4104, 192, True, True, False),        # ragged on both sides: edge tiles (drained waits), odd k-tile count (the stage parity flips per tile)
    (7700, 4100, 128, False, True, False),       # row pitch not a multiple of 8: the element-wise store path behind the counted wait
    (13000, 2752, 1024, True, True, False),      # the d(h2) shape class: 11 tile columns, walk of 2..3 tiles per workgroup
    (9000, 3000, 320, False, False, True),       # fp32 output + residual (the CIN instantiation), 5 k-tiles
    (40000, 512, 256, False, True, False),       # 128x128 tiles, two workgroups per CU: 1252 tiles on 512 walkers
    (33000, 512, 192, True, False, True),        # the same route with fp32 output + residual, ragged last row tile, odd k-tile count
])
def test_gemm_persistent_walk_equals_the_one_tile_grid(ops, dev, dtype, M, N, K, bkm, out16, resid):
    """The persistent form of the 256x256 kernel (one workgroup per CU walks several tiles; the next tile's first k-tile is requested under the
    current tile's epilogue) must produce the same bits as the one-tile grid (OMLM_GEMM_PERSIST=0) -- same products, same order -- and
    both must match fp64 on the rounded operands.  Repeated so that a race between the epilogue's LDS patches and the landing DMA pieces
    (or a short counted wait) would show as run-to-run differences."""
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + K)
    A = torch.randn(M, K, generator=g).to(dtype).to(dev)
    Bm = torch.randn(N, K, generator=g).to(dtype)
    if bkm:                                                 # [K, N padded to 8]
        B = torch.zeros(K, (N + 7) // 8 * 8, dtype=dtype)
        B[:, :N] = Bm.t()
        B = B.to(dev)
    else:
        B = Bm.to(dev)
    Cin = torch.randn(M, N, generator=g).to(dev) if resid else None
    odt = dtype if out16 else torch.float32
    ref = A.double() @ Bm.to(dev).double().t()
    if resid:
        ref = ref + Cin.double()
    outs = {}
    old = os.environ.get("OMLM_GEMM_PERSIST")
    try:
        for mode in ("0", "1"):
            os.environ["OMLM_GEMM_PERSIST"] = mode
            runs = []
            for rep in range(6 if mode == "1" else 1):
                C = torch.full((M, N), float("nan"), device=dev, dtype=odt)
                ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=False, b_kmajor=bkm, Cin=Cin, a_rows=M, b_rows=K if bkm else N)
                runs.append(C)
            outs[mode] = runs
    finally:
        if old is None:
            os.environ.pop("OMLM_GEMM_PERSIST", None)
        else:
            os.environ["OMLM_GEMM_PERSIST"] = old
    torch.cuda.synchronize()
    e = relerr(outs["1"][0].float(), ref)
    same = all(torch.equal(outs["0"][0], c) for c in outs["1"])
    report(f"gemm_persist[{dtype},{M},{N},{K},{bkm},{out16},{resid}]", relerr=e, bit_equal_to_one_tile_grid=same)
    assert not torch.isnan(outs["1"][0].float()).any()
    assert same, "persistent walk differs from the one-tile grid (or from itself between runs)"
    assert e < (8e-3 if out16 else 2e-5), e      # 16-bit output: one rounding of the result (2^-8 / 2^-11 of the row's scale); fp32: accumulation order only


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,akm,bkm,out16,resid", [
    (4096, 2048, 1024, False, False, True, False),     # NT, 16-bit out: 16 k-tiles (even)
    (3000, 1100, 192, False, True, True, False),       # NN, ragged M and N (edge tiles: out-of-range rows / columns staged as zeros), 3 k-tiles (odd: the B0 register sets swap roles)
    (2048, 1300, 64, False, False, False, True),       # ONE k-tile: prologue -> four phases -> drain; fp32 out + residual
    (1500, 1024, 2752, False, False, False, True),     # FF-out's K = 43 k-tiles, fp32 out + residual, ragged M
    (1280, 1536, 4096, True, True, False, False),      # both operands k-major (the weight-gradient layout), full-K tiles, plain fp32 store
    (1032, 1288, 128, True, False, True, False),       # k-major A, k-contiguous B, ragged, 2 k-tiles
])
def test_gemm_half_tile_ring_schedule_equals_the_rotated_loop(ops, dev, dtype, M, N, K, akm, bkm, out16, resid):
    """gemm_tile8_body (round 5: half-tile ring, counted vmcnt(10) per phase, two wave groups one barrier apart) multiplies the same
    fragments in the same order per accumulator as the rotated loop it replaces on long contractions: OMLM_GEMM_T8=1 (every eligible
    256 x 256 launch) must equal OMLM_GEMM_T8=0 BIT FOR BIT in every layout / output form, and fp64 on the rounded operands.  Six
    repeats: a short counted wait or a slot re-requested too early would show as run-to-run differences."""
    g = torch.Generator(device="cpu").manual_seed(M + 5 * N + K)
    c8 = lambda x: (x + 7) // 8 * 8
    Am, Bm = torch.randn(M, K, generator=g).to(dtype), torch.randn(N, K, generator=g).to(dtype)

    def store(mat, kmaj, n):
        if kmaj:
            st = torch.randn(K, c8(n), generator=g).to(dtype)
            st[:, :n] = mat.t()
            return st
        return mat.contiguous()
    A, B = store(Am, akm, M).to(dev), store(Bm, bkm, N).to(dev)
    Cin = torch.randn(M, N, generator=g).to(dev) if resid else None
    odt = dtype if out16 else torch.float32
    ref = Am.to(dev).double() @ Bm.to(dev).double().t()
    if resid:
        ref = ref + Cin.double()
    outs = {}
    old = {k: os.environ.get(k) for k in ("OMLM_GEMM_T8", "OMLM_GEMM_TILE")}
    try:
        os.environ["OMLM_GEMM_TILE"] = "256x256"           # the wide tile whatever the host's shape rule says
        for mode in ("0", "1"):
            os.environ["OMLM_GEMM_T8"] = mode
            runs = []
            for rep in range(6 if mode == "1" else 1):
                C = torch.full((M, c8(N)), float("nan"), device=dev, dtype=odt)
                ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=akm, b_kmajor=bkm, Cin=Cin, ldcin=N if resid else None,
                         a_rows=K if akm else M, b_rows=K if bkm else N)
                runs.append(C[:, :N].clone())
            outs[mode] = runs
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    torch.cuda.synchronize()
    e = relerr(outs["1"][0].float(), ref)
    same = all(torch.equal(outs["0"][0], c) for c in outs["1"])
    report(f"gemm_t8[{dtype},{M},{N},{K},{akm},{bkm},{out16},{resid}]", relerr=e, bit_equal_to_rotated_loop=same)
    assert not torch.isnan(outs["1"][0].float()).any()
    assert same, "the half-tile-ring schedule differs from the rotated loop (or from itself between runs)"
    assert e < (8e-3 if out16 else 2e-5), e


def test_gemm_half_tile_ring_on_operand_planes_and_split_k(ops, dev):
    """The same schedule under the hi/lo-plane route (3x k-loop through three plane pairs: 'bf16x3') and with split-K partial sums
    (fp32 atomics into C: order is free, values agree to rounding)."""
    g = torch.Generator(device="cpu").manual_seed(99)
    M, N, K = 2048, 1024, 1024
    A, B = torch.randn(M, K, generator=g).to(dev), torch.randn(N, K, generator=g).to(dev)
    dY, X = torch.randn(16384, 512, generator=g).half().to(dev), torch.randn(16384, 768, generator=g).half().to(dev)
    outs = {}
    old = {k: os.environ.get(k) for k in ("OMLM_GEMM_T8", "OMLM_GEMM_TILE")}
    try:
        os.environ["OMLM_GEMM_TILE"] = "256x256"
        for mode in ("0", "1"):
            os.environ["OMLM_GEMM_T8"] = mode
            C = torch.full((M, N), float("nan"), device=dev)
            ops.gemm(A, B, C, M=M, N=N, K=K)                                  # fp32 operands above the plane threshold
            # split-K: accumulate-into-C with few tiles and a long contraction (a weight-gradient shape, k-major fp16 operands)
            dW = torch.zeros(512, 768, device=dev)
            ops.gemm(dY, X, dW, M=512, N=768, K=16384, a_kmajor=True, b_kmajor=True, Cin=dW)
            outs[mode] = (C, dW, dY, X)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t()
    e3 = relerr(outs["1"][0], ref)
    same3 = torch.equal(outs["0"][0], outs["1"][0])
    refw = outs["1"][2].double().t() @ outs["1"][3].double()
    ew = relerr(outs["1"][1], refw)
    report("gemm_t8_planes_splitk", planes_relerr=e3, planes_bit_equal=same3, splitk_relerr=ew, splitk_vs_rotated=relerr(outs["1"][1], outs["0"][1].double()))
    assert same3 and e3 < 2e-5, (same3, e3)
    assert ew < 2e-5 and relerr(outs["1"][1], outs["0"][1].double()) < 1e-5


@pytest.mark.parametrize("M,N,K,akm,bkm,accumulate", [(1100, 520, 264, False, True, False), (600, 264, 5000, True, True, True),
                                                       (777, 1032, 520, False, False, False), (300, 512, 2048, True, False, True)])
def test_gemm_fp32_on_operand_planes(ops, dev, M, N, K, akm, bkm, accumulate):
    """fp32 operands above the size threshold take the hi/lo-plane route (omlm_split_planes + omlm_gemm_planes: one bf16 tile-kernel
    launch with a 3x k-loop) -- same fp32-grade bar as the register-staged kernel, all four layouts, in-place accumulation
    (split-K) included, and both routes must agree."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    c8 = lambda x: (x + 7) // 8 * 8
    Am, Bm = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Kp = c8(K)

    def store(mat, kmaj, n):
        if kmaj:
            st = torch.randn(K, c8(n), generator=g)
            st[:, :n] = mat.t()
            return st
        st = torch.zeros(n, Kp)
        st[:, :K] = mat
        return st
    A, B = store(Am, akm, M).to(dev), store(Bm, bkm, N).to(dev)
    Kcall = K if (akm and bkm) else Kp
    C0 = torch.randn(M, N, generator=g).to(dev)
    ref = (Am.double() @ Bm.double().t()).to(dev) + C0.double()
    outs = []
    for planes in (True, False):
        ops._X3_PLANES = planes
        try:
            if accumulate:
                C = C0.clone()
                ops.gemm(A, B, C, M=M, N=N, K=Kcall, a_kmajor=akm, b_kmajor=bkm, Cin=C, a_rows=K if akm else M, b_rows=K if bkm else N)
            else:
                C = torch.full((M, N), float("nan"), device=dev)
                ops.gemm(A, B, C, M=M, N=N, K=Kcall, a_kmajor=akm, b_kmajor=bkm, Cin=C0, a_rows=K if akm else M, b_rows=K if bkm else N)
        finally:
            ops._X3_PLANES = True
        outs.append(C)
    e_planes, e_reg, e_ab = relerr(outs[0], ref), relerr(outs[1], ref), relerr(outs[0], outs[1])
    report(f"gemm_planes[{M},{N},{K},{akm},{bkm}]", planes=e_planes, register_staged=e_reg, ab=e_ab)
    assert not torch.isnan(outs[0]).any() and e_planes < 2e-5 and e_reg < 2e-5 and e_ab < 2e-5
    # the plane cache lives only inside a scope (engine: one forward + its backward).  Outside: nothing is cached, so a tensor that a
    # kernel rewrote through its raw pointer (no version bump) is split again; inside: one split per tensor, a torch in-place op
    # (version bump) re-splits, and closing the scope drops everything.
    kw = dict(M=M, N=N, K=Kcall, a_kmajor=akm, b_kmajor=bkm, a_rows=K if akm else M, b_rows=K if bkm else N)
    assert len(ops._PLANES) == 0
    C2 = torch.empty(M, N, device=dev)
    ops.gemm(A, B, C2, **kw)
    assert len(ops._PLANES) == 0
    ops.cast_pad((2.0 * A).contiguous(), A, A.shape[0], A.shape[1], A.shape[1], A.shape[1])      # raw-pointer rewrite of A (fp32 -> fp32 copy)
    C3 = torch.empty(M, N, device=dev)
    ops.gemm(A, B, C3, **kw)
    assert relerr(C3, 2.0 * C2) < 2e-5
    ops.planes_begin()
    try:
        ops.gemm(A, B, C3, **kw)
        n_in = len(ops._PLANES)
        assert n_in == 2
        ops.gemm(A, B, C3, **kw)
        assert len(ops._PLANES) == n_in
        A.mul_(0.5)
        C4 = torch.empty(M, N, device=dev)
        ops.gemm(A, B, C4, **kw)
        assert relerr(C4, C2) < 2e-5
    finally:
        ops.planes_end()
    assert len(ops._PLANES) == 0


def test_gemm_row_maps_and_bf16_out(ops, dev):
    g = torch.Generator().manual_seed(5)
    rows, D, V = 300, 128, 41
    y = torch.randn(rows, D, generator=g).to(dev).bfloat16()
    W = torch.randn(V, D, generator=g).to(dev).bfloat16()
    a_map = torch.randperm(rows, generator=g)[:77].to(torch.int32).to(dev)
    c_map = torch.randperm(150, generator=g)[:77].to(torch.int32).to(dev)
    c_map[5] = -1                                       # dropped row
    out = torch.zeros(150, 48, device=dev, dtype=torch.bfloat16)
    ops.gemm(y, W, out, M=77, N=V, K=D, a_map=a_map, c_map=c_map, ldc=48, a_rows=rows, b_rows=V)
    ref = torch.zeros(150, 48, dtype=torch.float64, device=dev)
    prod = y.double()[a_map.long()] @ W.double().t()
    for r in range(77):
        if int(c_map[r]) >= 0:
            ref[int(c_map[r]), :V] = prod[r]
    e = relerr(out, ref)
    report("gemm_row_maps_bf16_out", relerr=e)
    assert e < 5e-3             # output rounded to bf16: 2^-9 relative


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_gemm_wgrad_group(ops, dev, splits, dtype):
    """Grouped weight-gradient launch (dW_i += dY_i^T X_i, all problems in one grid) vs fp64, with ragged extents (M = 130,
    N = 2730: partial tiles, ldc not a multiple of 4), a scattering c_map with dropped rows, accumulation into existing
    gradients, and forced K-splits (atomics) as well as the full-K form."""
    g = torch.Generator().manual_seed(11 + splits)
    K = 1500
    shapes = [(512, 1024, None), (130, 1024, None), (1024, 2730, None), (704, 256, "map"), (1024, 512, None)]
    grp = ops.WgradGroup()
    keep = []
    for M, N, cm in shapes:
        ldM, ldN = (M + 7) // 8 * 8, (N + 7) // 8 * 8
        dY = torch.randn(K, ldM, generator=g).to(dev).to(dtype)
        X = torch.randn(K, ldN, generator=g).to(dev).to(dtype)
        rowsC = M
        c_map = None
        if cm:
            rowsC = M - 40
            perm = torch.randperm(M, generator=g)
            c_map = torch.full((M,), -1, dtype=torch.int32)
            c_map[perm[:rowsC]] = torch.arange(rowsC, dtype=torch.int32)          # 40 logical rows are dropped
            c_map = c_map.to(dev)
        dW0 = torch.randn(rowsC, N, generator=g).to(dev)
        dW = dW0.clone()
        grp.add(dY, X, dW, M=M, N=N, K=K, c_map=c_map)
        keep.append((dY, X, dW, dW0, c_map, M, N))
    grp.flush(splits=splits)
    worst = 0.0
    for dY, X, dW, dW0, c_map, M, N in keep:
        prod = dY.double()[:, :M].t() @ X.double()[:, :N]
        ref = dW0.double().clone()
        if c_map is None:
            ref += prod
        else:
            sel = c_map.long() >= 0
            ref[c_map.long()[sel]] += prod[sel]
        worst = max(worst, relerr(dW, ref))
    report(f"gemm_wgrad_group[splits={splits},{dtype}]", relerr=worst)
    assert worst < 2e-5, worst


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_wgrad_group_at_bench_shape(ops, dev, dtype):
    """The grouped launch at the shape bench.py times: K = 32 x 1116 tokens, the five weight gradients of two coarse-small layers
    (dW1 5504 x 1024, dW2 1024 x 2752, dWq, dWkv, dWo) accumulated into existing gradients, against fp64 products."""
    g = torch.Generator().manual_seed(5)
    K = 32 * 1116
    shapes = [(5504, 1024), (1024, 2752), (512, 1024), (128, 1024), (1024, 512)] * 2
    grp = ops.WgradGroup()
    keep = []
    for M, N in shapes:
        dY = torch.randn(K, M, generator=g).to(dev).to(dtype)
        X = torch.randn(K, N, generator=g).to(dev).to(dtype)
        dW0 = torch.randn(M, N, generator=g).to(dev)
        dW = dW0.clone()
        grp.add(dY, X, dW, M=M, N=N, K=K)
        keep.append((dY, X, dW, dW0))
    grp.flush()
    worst = 0.0
    for dY, X, dW, dW0 in keep:
        ref = dW0.double() + dY.double().t() @ X.double()
        worst = max(worst, relerr(dW, ref))
    report(f"gemm_wgrad_group_bench_shape[{dtype}]", relerr=worst, K=K)
    assert worst < 2e-5, worst


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_cast_pad_group(ops, dev, dtype):
    """The grouped weight re-pack (omlm_cast_pad_group) against torch: padded row copies with zero pad columns, sub-views with a storage
    offset, transposed tap tables, a single-row gamma -- 70 problems, i.e. more than one launch of 64 descriptors."""
    g = torch.Generator().manual_seed(4)
    grp = ops.CastPadGroup()
    checks = []
    for i in range(14):
        F, D = 37 + 3 * i, 24 + 8 * (i % 3)
        Fp = (F + 7) // 8 * 8 + 8 * (i % 2)
        w1 = torch.randn(2 * F, D, generator=g).to(dev)
        W1p = torch.full((2 * Fp, D), 7.0, device=dev, dtype=dtype)
        grp.add(w1, W1p, F, D, D, D)
        grp.add(w1[F:], W1p[Fp:], F, D, D, D)
        w2 = torch.randn(D, F, generator=g).to(dev)
        W2p = torch.full((D, Fp), 7.0, device=dev, dtype=dtype)
        grp.add(w2, W2p, D, F, F, Fp)
        cw = torch.randn(2 * F, 3, generator=g).to(dev)
        taps = torch.zeros(3, 2 * Fp, device=dev, dtype=dtype)
        grp.add(cw, taps, F, 3, 3, 2 * Fp, transpose=True)
        grp.add(cw[F:], taps[:, Fp:], F, 3, 3, 2 * Fp, transpose=True)
        checks.append((w1, W1p, w2, W2p, cw, taps, F, Fp))
    grp.flush()
    for w1, W1p, w2, W2p, cw, taps, F, Fp in checks:
        assert torch.equal(W1p[:F], w1[:F].to(dtype)) and torch.equal(W1p[Fp:Fp + F], w1[F:].to(dtype))
        assert float((W1p[F:Fp] - 7.0).abs().max()) == 0.0                          # rows between the halves are not touched
        assert torch.equal(W2p[:, :F], w2.to(dtype)) and float(W2p[:, F:].abs().max()) == 0.0     # pad columns zeroed
        assert torch.equal(taps[:, :F], cw[:F].t().to(dtype)) and torch.equal(taps[:, Fp:Fp + F], cw[F:].t().to(dtype))
        assert float(taps[:, F:Fp].abs().max()) == 0.0


@pytest.mark.parametrize("M,D", [(333, 1024), (6700, 1024), (333, 768)])       # one row per workgroup; 3-4 rows per workgroup (the row-pair walk of the
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])   # d = 1024 backward: both register sets, both LDS parities); the general kernel
def test_layernorm_fwd_bwd(ops, dev, dtype, M, D):
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    y = torch.empty(M, D, device=dev, dtype=dtype)
    xc = torch.empty(M, D, device=dev, dtype=dtype)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, y, xc, mean, rstd)
    xr = x.double().requires_grad_(True)
    gr = gamma.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr, (D,), gr, None, 1e-5)
    tol = 1e-5 if dtype == torch.float32 else 5e-3
    e_y, e_c = relerr(y, yr.detach()), relerr(xc, x)
    dy = torch.randn(M, D, generator=g).to(dev)
    dres = torch.randn(M, D, generator=g).to(dev)
    yr.backward(dy.double())
    dx = torch.empty(M, D, device=dev)
    dxc = torch.empty(M, D, device=dev, dtype=dtype)
    dgamma = torch.zeros(D, device=dev)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx, dxc, dgamma, dx_scale=0.1)
    e_dx = relerr(dx, 0.1 * (xr.grad + dres.double()))
    e_dg = relerr(dgamma, gr.grad)
    report(f"layernorm[{dtype},{M},{D}]", y=e_y, xcast=e_c, dx=e_dx, dgamma=e_dg, dxcast=relerr(dxc, dx))
    assert e_y < tol and e_c < tol and e_dx < 1e-5 and e_dg < 1e-4 and relerr(dxc, dx) < tol
    if dtype in (torch.bfloat16, torch.float16):
        # dy handed over in the 16-bit operand type (what the input-gradient GEMM's epilogue writes in bf16 / fp16 mode): exact for the rounded values
        dyb = dy.to(dtype)
        xr2, gr2 = x.double().requires_grad_(True), gamma.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xr2, (D,), gr2, None, 1e-5).backward(dyb.double())
        dx2, dg2 = torch.empty(M, D, device=dev), torch.zeros(D, device=dev)
        ops.layernorm_bwd(dyb, x, gamma, mean, rstd, dres, dx2, None, dg2, dx_scale=0.1)
        e2, eg2 = relerr(dx2, 0.1 * (xr2.grad + dres.double())), relerr(dg2, gr2.grad)
        report(f"layernorm[{dtype},{M},{D} dy]", dx=e2, dgamma=eg2)
        assert e2 < 1e-5 and eg2 < 1e-4
    dx0, dg0 = torch.empty(M, D, device=dev), torch.zeros(D, device=dev)        # no incoming residual gradient (the final LayerNorm of the trunk)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, None, dx0, None, dg0)
    assert relerr(dx0, xr.grad) < 1e-5 and relerr(dg0, gr.grad) < 1e-4
    # a second residual-gradient term in the cast type (omlm_layernorm_bwd2: the K/V projection's input gradient), with and without a
    # cast output: exact for the values handed over
    d2 = torch.randn(M, D, generator=g).to(dev).to(dtype)
    for with_cast in (True, False):
        dx3, dg3 = torch.empty(M, D, device=dev), torch.zeros(D, device=dev)
        dxc3 = torch.empty(M, D, device=dev, dtype=dtype) if with_cast else None
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx3, dxc3, dg3, dx_scale=0.1, dres2=d2)
        e3 = relerr(dx3, 0.1 * (xr.grad + dres.double() + d2.double()))
        report(f"layernorm[{dtype},{M},{D} dres2 cast={with_cast}]", dx=e3)
        assert e3 < 1e-5 and relerr(dg3, gr.grad) < 1e-4
        if with_cast:
            assert relerr(dxc3, dx3) < tol


def test_layernorm_bwd_grouped_column_sums(ops, dev):
    """Round 5: LayerNorm backwards whose d(gamma) partial rows are summed by ONE omlm_colsum_group launch (engine.trunk_backward's
    ColsumGroup) give the same dx bit for bit and the same d(gamma) as the call that sums its own rows -- three problems of different
    row counts and widths in one group, accumulated onto non-zero d(gamma) buffers."""
    g = torch.Generator().manual_seed(9)
    grp = ops.ColsumGroup()
    cases = []
    for M, D in ((333, 1024), (6700, 1024), (150, 768)):
        x = (torch.randn(M, D, generator=g) * 2 + 0.5).to(dev)
        gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
        y = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        ops.layernorm_fwd(x, gamma, y, None, mean, rstd)
        dy, dres = torch.randn(M, D, generator=g).to(dev), torch.randn(M, D, generator=g).to(dev)
        start = torch.randn(D, generator=g).to(dev)
        dx_a, dg_a = torch.empty(M, D, device=dev), start.clone()
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx_a, None, dg_a)
        dx_b, dg_b = torch.empty(M, D, device=dev), start.clone()
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres, dx_b, None, dg_b, defer=grp)
        assert torch.equal(dg_b, start)                                   # nothing summed before the flush
        xr, gr = x.double().requires_grad_(True), gamma.double().requires_grad_(True)
        torch.nn.functional.layer_norm(xr, (D,), gr, None, 1e-5).backward(dy.double())
        cases.append((M, D, dx_a, dg_a, dx_b, dg_b, start, gr.grad))
    assert len(grp.items) == 3
    grp.flush()
    assert not grp.items
    for M, D, dx_a, dg_a, dx_b, dg_b, start, ref in cases:
        assert torch.equal(dx_a, dx_b)
        e_own, e_grp = relerr(dg_a - start, ref), relerr(dg_b - start, ref)
        report(f"layernorm grouped colsum[{M},{D}]", own=e_own, grouped=e_grp)
        assert e_grp < 1e-4 and e_own < 1e-4
    grp.flush()                                                           # empty group: no launch, no error


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_qk_norm_fwd_bwd(ops, dev, dtype):
    M, H = 150, 3
    g = torch.Generator().manual_seed(2)
    q_raw = torch.randn(M, H * 64, generator=g).to(dev)
    kv_raw = torch.randn(M, 128, generator=g).to(dev)
    qs = (1 + 0.2 * torch.randn(64, generator=g)).to(dev)
    ks = (1 + 0.2 * torch.randn(64, generator=g)).to(dev)
    q = torch.empty(M, H * 64, device=dev, dtype=dtype)
    k = torch.empty(M, 64, device=dev, dtype=dtype)
    v = torch.empty(M, 64, device=dev, dtype=dtype)
    ops.qk_norm_fwd(q_raw, kv_raw, qs, ks, q, k, v, H)
    qr, kvr = q_raw.double().requires_grad_(True), kv_raw.double().requires_grad_(True)
    qsr, ksr = qs.double().requires_grad_(True), ks.double().requires_grad_(True)
    qn = torch.nn.functional.normalize(qr.view(M, H, 64), dim=-1) * qsr
    kn = torch.nn.functional.normalize(kvr[:, :64], dim=-1) * ksr
    vn = kvr[:, 64:]
    tol = 1e-5 if dtype == torch.float32 else 5e-3
    e = max(relerr(q, qn.reshape(M, -1).detach()), relerr(k, kn.detach()), relerr(v, vn.detach()))
    dq, dk, dv = (torch.randn(M, H * 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev),
                  torch.randn(M, 64, generator=g).to(dev))
    (qn.reshape(M, -1) * dq.double()).sum().add((kn * dk.double()).sum()).add((vn * dv.double()).sum()).backward()
    dq_raw = torch.empty(M, H * 64, device=dev, dtype=dtype)
    dkv_raw = torch.empty(M, 128, device=dev, dtype=dtype)
    dqs, dks = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ops.qk_norm_bwd(dq, dk, dv, q_raw, kv_raw, qs, ks, dq_raw, dkv_raw, dqs, dks, H)
    eb = max(relerr(dq_raw, qr.grad), relerr(dkv_raw, kvr.grad))
    es = max(relerr(dqs, qsr.grad), relerr(dks, ksr.grad))
    report(f"qk_norm[{dtype}]", fwd=e, bwd=eb, dscale=es)
    assert e < tol and eb < tol and es < 1e-4


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", [(300, 1024), (129, 200), (35, 64)])
def test_gemm_qknorm_fused_projection_and_its_backward(ops, dev, dtype, M, K):
    """The q / k projections with l2-norm + learned scale as GEMM epilogue (omlm_gemm_qknorm: q with H = 3 heads; k | v with v passed
    through into its own buffer) against fp64 on the same 16-bit operands, and omlm_qk_norm_bwd2 (the backward from the normalised
    16-bit outputs + saved norms) against autograd of the fp64 formula.  Ragged M (128-row tiles), K with and without whole 64-deep k-tiles."""
    H = 3
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).to(dev).to(dtype)
    A2 = torch.randn(M, K, generator=g).to(dev).to(dtype)
    Wq = (torch.randn(H * 64, K, generator=g) / math.sqrt(K)).to(dev).to(dtype)
    Wkv = (torch.randn(128, K, generator=g) / math.sqrt(K)).to(dev).to(dtype)
    qs = (1 + 0.2 * torch.randn(64, generator=g)).to(dev)
    ks = (1 + 0.2 * torch.randn(64, generator=g)).to(dev)
    q = torch.full((M, H * 64), 7.0, device=dev, dtype=dtype)
    k = torch.full((M, 64), 7.0, device=dev, dtype=dtype)
    v = torch.full((M, 64), 7.0, device=dev, dtype=dtype)
    qn, kn = torch.zeros(M, H, device=dev), torch.zeros(M, device=dev)
    ops.gemm_qknorm(A, Wq, q, qs, qn, H, M=M, N=H * 64, K=K)
    ops.gemm_qknorm(A2, Wkv, k, ks, kn, 1, M=M, N=128, K=K, C2=v, c2_col0=64)
    qr = (A.double() @ Wq.double().t()).requires_grad_(True)
    kvr = (A2.double() @ Wkv.double().t()).requires_grad_(True)
    qsr, ksr = qs.double().requires_grad_(True), ks.double().requires_grad_(True)
    q_ref = torch.nn.functional.normalize(qr.view(M, H, 64), dim=-1) * qsr
    k_ref = torch.nn.functional.normalize(kvr[:, :64], dim=-1) * ksr
    v_ref = kvr[:, 64:]
    tol = 5e-3 if dtype == torch.bfloat16 else 7e-4                  # one rounding of the outputs to the operand type
    e = max(relerr(q, q_ref.reshape(M, -1).detach()), relerr(k, k_ref.detach()), relerr(v, v_ref.detach()))
    en = max(relerr(qn, qr.detach().view(M, H, 64).norm(dim=-1)), relerr(kn, kvr.detach()[:, :64].norm(dim=-1)))
    dq, dk, dv = (torch.randn(M, H * 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev))
    (q_ref.reshape(M, -1) * dq.double()).sum().add((k_ref * dk.double()).sum()).add((v_ref * dv.double()).sum()).backward()
    dq_raw = torch.empty(M, H * 64, device=dev, dtype=dtype)
    dkv_raw = torch.empty(M, 128, device=dev, dtype=dtype)
    dqs, dks = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
    ops.qk_norm_bwd2(dq, dk, dv, q, k, qn, kn, qs, ks, dq_raw, dkv_raw, dqs, dks, H)
    eb = max(relerr(dq_raw, qr.grad), relerr(dkv_raw, kvr.grad))
    es = max(relerr(dqs, qsr.grad), relerr(dks, ksr.grad))
    report(f"gemm_qknorm[{dtype},{M},{K}]", fwd=e, norms=en, bwd=eb, dscale=es)
    # backward: xh comes from the ROUNDED outputs (2^-9 / 2^-11 per element) -- the projection term carries that rounding
    assert e < tol and en < 2e-5 and eb < 3 * tol and es < 3 * tol, (e, en, eb, es)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_qknorm_persistent_walk_equals_the_one_tile_grid(ops, dev, dtype):
    """omlm_gemm_qknorm on more tiles than two per CU takes the persistent walk (l2-norm epilogue instantiation): same bits as the
    one-tile grid for the normalised outputs, the pass-through columns and the saved norms, run to run."""
    M, K, H = 20000, 256, 8
    g = torch.Generator().manual_seed(9)
    A = torch.randn(M, K, generator=g).to(dev).to(dtype)
    W = (torch.randn(H * 64 + 64, K, generator=g) / math.sqrt(K)).to(dev).to(dtype)      # 8 normalised heads + 64 pass-through columns
    qs = (1 + 0.2 * torch.randn(64, generator=g)).to(dev)
    res = {}
    old = os.environ.get("OMLM_GEMM_PERSIST")
    try:
        for mode in ("0", "1"):
            os.environ["OMLM_GEMM_PERSIST"] = mode
            runs = []
            for rep in range(4 if mode == "1" else 1):
                q = torch.full((M, H * 64), 7.0, device=dev, dtype=dtype)
                v = torch.full((M, 64), 7.0, device=dev, dtype=dtype)
                qn = torch.zeros(M, H, device=dev)
                ops.gemm_qknorm(A, W, q, qs, qn, H, M=M, N=H * 64 + 64, K=K, C2=v, c2_col0=H * 64)
                runs.append((q, v, qn))
            res[mode] = runs
    finally:
        if old is None:
            os.environ.pop("OMLM_GEMM_PERSIST", None)
        else:
            os.environ["OMLM_GEMM_PERSIST"] = old
    torch.cuda.synchronize()
    ref = res["0"][0]
    same = all(torch.equal(a, b) for r in res["1"] for a, b in zip(r, ref))
    raw = A.double() @ W.double().t()
    q_ref = torch.nn.functional.normalize(raw[:, :H * 64].view(M, H, 64), dim=-1) * qs.double()
    e = max(relerr(ref[0], q_ref.reshape(M, -1)), relerr(ref[1], raw[:, H * 64:]))
    report(f"gemm_qknorm_persist[{dtype}]", fwd=e, bit_equal_to_one_tile_grid=same)
    assert same and e < (5e-3 if dtype == torch.bfloat16 else 7e-4), (same, e)


def naive_attention(q, k, v, bias, keymask, H, scale=8.0):
    """q [B,N,H*64], k,v [B,N,64] double; bias [N, >=H]; keymask [B,N] bool."""
    B, N, _ = q.shape
    qh = q.view(B, N, H, 64).permute(0, 2, 1, 3)
    sim = torch.einsum("bhid,bjd->bhij", qh, k) * scale
    idx = (torch.arange(N, device=q.device)[:, None] - torch.arange(N, device=q.device)[None, :]).clamp(min=0)
    if bias is not None:
        sim = sim + bias[:, :H].t()[:, idx]
    neg = -torch.finfo(torch.float32).max
    if keymask is not None:
        sim = sim.masked_fill(~keymask[:, None, None, :], neg)
    sim = sim.masked_fill(torch.ones(N, N, dtype=torch.bool, device=q.device).triu(1), neg)
    out = torch.einsum("bhij,bjd->bhid", sim.softmax(-1), v)
    return out.permute(0, 2, 1, 3).reshape(B, N, H * 64)


# the last two cases are musiclm_large's fine stage (BASELINE config 4): 16 heads, N = 1817 positions
@pytest.mark.parametrize("dtype,B,N,H", [(torch.float32, 2, 77, 2), (torch.bfloat16, 2, 77, 2),
                                         (torch.bfloat16, 1, 200, 5), (torch.float32, 1, 130, 8),
                                         (torch.bfloat16, 1, 1817, 16), (torch.float32, 1, 1817, 16),
                                         (torch.float16, 2, 77, 2), (torch.float16, 1, 200, 5), (torch.float16, 1, 1817, 16),
                                         # the bench shapes themselves (BASELINE configs 2 and 4 at their per-GPU batch), fp64 reference on the GPU
                                         (torch.bfloat16, 32, 1116, 8), (torch.float16, 32, 1116, 8), (torch.bfloat16, 8, 1817, 16)])
def test_attention_fwd_bwd(ops, dev, dtype, B, N, H):
    g = torch.Generator().manual_seed(N + H)
    M = B * N
    unit = lambda t: torch.nn.functional.normalize(t, dim=-1)
    q = unit(torch.randn(B, N, H, 64, generator=g)).reshape(M, H * 64).to(dev)
    k = unit(torch.randn(M, 64, generator=g)).to(dev)
    v = torch.randn(M, 64, generator=g).to(dev)
    ldb = (H + 7) // 8 * 8
    bias = torch.zeros(N, ldb)
    bias[:, :H] = torch.randn(N, H, generator=g) * 2
    bias = bias.to(dev)
    keymask = (torch.rand(B, N, generator=g) > 0.2)
    keymask[:, 0] = True
    keymask = keymask.to(dev)
    qd, kd, vd = q.to(dtype), k.to(dtype), v.to(dtype)
    out = torch.empty(M, H * 64, device=dev, dtype=dtype)
    lse = torch.empty(B, H, N, device=dev)
    ops.attn_fwd(qd, kd, vd, bias, keymask.to(torch.uint8), out, lse, B, N, H, 8.0)
    qr = qd.double().view(B, N, H * 64).requires_grad_(True)
    kr = kd.double().view(B, N, 64).requires_grad_(True)
    vr = vd.double().view(B, N, 64).requires_grad_(True)
    br = bias.double().requires_grad_(True)
    ref = naive_attention(qr, kr, vr, br, keymask, H)
    e_f = relerr(out.view(B, N, -1), ref.detach())
    # fp32 operands: bf16x3 forward -> fp32-grade; bf16 operands: P and the output are rounded to bf16
    tol_f = 2e-5 if dtype == torch.float32 else (1e-2 if dtype == torch.bfloat16 else 2e-3)       # P and the output rounded to 2^-9 / 2^-12
    if dtype == torch.bfloat16:
        # the same forward with the fixed softmax reference point (q, k are unit vectors here: |q.k| <= 1): same softmax
        ab = ops.AttnBias(bias, N, H, dev, qk_bound=1.0, scale=8.0)
        assert float(ab.tableT.view(-1, ab.tableT.numel() // ((H + 7) // 8 * 8))[0, -2]) == 1.0      # the fixed path is taken
        out2 = torch.empty_like(out); lse2 = torch.empty_like(lse)
        ops.attn_fwd(qd, kd, vd, ab, keymask.to(torch.uint8), out2, lse2, B, N, H, 8.0)
        e_f2 = relerr(out2.view(B, N, -1), ref.detach())
        e_lse = float((lse2 - lse).abs().max())
        report(f"attention_fixed_ref[{B},{N},{H}]", fwd=e_f2, lse_diff=e_lse)
        # lse comes from the MFMA denominator (bf16-rounded P, like the numerator): log2-domain agreement to ~2^-7
        assert e_f2 < tol_f and e_lse < 1e-2, (e_f2, e_lse)
    if dtype == torch.float16:
        # half operands: the fixed reference point sits 15 octaves under the bound and is taken while 2 c |q.k| + the table's range < 28 --
        # not with the +-2 table above (the flag stays 0: online softmax, checked above), so a narrow table here
        ab0 = ops.AttnBias(bias, N, H, dev, qk_bound=1.0, scale=8.0, half=True)
        assert float(ab0.tableT.view(-1, ab0.tableT.numel() // ((H + 7) // 8 * 8))[0, -2]) == 0.0
        out0 = torch.empty_like(out); lse0 = torch.empty_like(lse)
        ops.attn_fwd(qd, kd, vd, ab0, keymask.to(torch.uint8), out0, lse0, B, N, H, 8.0)
        assert torch.equal(out0, out) and torch.equal(lse0, lse)                                       # same online kernel, same bits
        bias_s = bias * 0.05
        ab = ops.AttnBias(bias_s, N, H, dev, qk_bound=1.0, scale=8.0, half=True)
        assert float(ab.tableT.view(-1, ab.tableT.numel() // ((H + 7) // 8 * 8))[0, -2]) == 1.0      # the fixed path is taken
        out2 = torch.empty_like(out); lse2 = torch.empty_like(lse); out3 = torch.empty_like(out); lse3 = torch.empty_like(lse)
        ops.attn_fwd(qd, kd, vd, ab, keymask.to(torch.uint8), out2, lse2, B, N, H, 8.0)
        ops.attn_fwd(qd, kd, vd, bias_s, keymask.to(torch.uint8), out3, lse3, B, N, H, 8.0)          # raw table: online softmax
        ref_s = naive_attention(qd.double().view(B, N, H * 64), kd.double().view(B, N, 64), vd.double().view(B, N, 64), bias_s.double(), keymask, H)
        e_f2, e_f3 = relerr(out2.view(B, N, -1), ref_s), relerr(out3.view(B, N, -1), ref_s)
        e_lse = float((lse2 - lse3).abs().max())
        report(f"attention_fixed_ref_half[{B},{N},{H}]", fwd=e_f2, fwd_online=e_f3, lse_diff=e_lse)
        assert e_f2 < tol_f and e_f3 < tol_f and e_lse < 2e-3, (e_f2, e_f3, e_lse)
    do = torch.randn(B, N, H * 64, generator=g).to(dev)
    ref.backward(do.double())
    dq = torch.empty(M, H * 64, device=dev)
    dk = torch.empty(M, 64, device=dev)
    dv = torch.empty(M, 64, device=dev)
    dbias = torch.zeros(N, ldb, device=dev)
    delta = torch.empty(B, H, N, device=dev)
    ops.attn_bwd(qd, kd, vd, bias, keymask.to(torch.uint8), out, do.reshape(M, -1).to(dtype).contiguous(), lse, delta,
                 dq, dk, dv, dbias, B, N, H, 8.0)
    e_q, e_k, e_v = relerr(dq.view(B, N, -1), qr.grad), relerr(dk.view(B, N, -1), kr.grad), relerr(dv.view(B, N, -1), vr.grad)
    e_b = relerr(dbias[:, :H], br.grad[:, :H])
    # the C ABI's null-workspace form (atomics into the table) accumulates the same d(bias) on top of what is there
    dbias2 = dbias.clone()
    ops.attn_bwd(qd, kd, vd, bias, keymask.to(torch.uint8), out, do.reshape(M, -1).to(dtype).contiguous(), lse, delta,
                 dq, dk, dv, dbias2, B, N, H, 8.0, workspace=False)
    e_b2 = relerr(dbias2[:, :H] * 0.5, br.grad[:, :H])
    assert float(dbias[:, H:].abs().max() if ldb > H else 0.0) == 0.0
    report(f"attention[{dtype},{B},{N},{H}]", fwd=e_f, dq=e_q, dk=e_k, dv=e_v, dbias=e_b, dbias_atomic=e_b2)
    assert e_b2 < (4e-3 if dtype == torch.float16 else 2e-2), e_b2
    assert e_f < tol_f, e_f
    # backward always runs single-pass bf16 MFMA (P, dS, dO rounded to bf16): 2^-8-level relative error
    assert max(e_q, e_k, e_v, e_b) < (4e-3 if dtype == torch.float16 else 2e-2), (e_q, e_k, e_v, e_b)


def ffmid_reference(h1, convw, gamma, F, Fp, nseq, drop=None):
    """h1 [M, 2Fp] double in the padded layout; returns h2 [M, F]."""
    M = h1.shape[0]
    x = torch.cat([h1[:, :F], h1[:, Fp:Fp + F]], dim=-1).view(M // nseq, nseq, 2 * F)
    xp = torch.nn.functional.pad(x, (0, 0, 2, 0))
    u = xp[:, :-2] * convw[:, 0] + xp[:, 1:-1] * convw[:, 1] + xp[:, 2:] * convw[:, 2]
    a, gate = u[..., :F], u[..., F:]
    gl = torch.nn.functional.gelu(gate) * a
    y = torch.nn.functional.layer_norm(gl, (F,), gamma, None, 1e-5)
    return y.reshape(M, F)


@pytest.mark.parametrize("dtype,F,save_gh", [(torch.float32, 341, False), (torch.bfloat16, 341, False), (torch.float32, 2730, False),
                                             (torch.bfloat16, 341, True), (torch.bfloat16, 2730, True), (torch.float32, 341, True),
                                             (torch.float16, 341, False), (torch.float16, 2730, True)])
def test_ffmid_fwd_bwd(ops, dev, dtype, F, save_gh):
    """save_gh: the forward also stores the normalised GEGLU output and the backward's first sweep runs from it."""
    nseq, Bn = 19, 3
    M = nseq * Bn
    Fp = (F + 7) // 8 * 8
    g = torch.Generator().manual_seed(F)
    h1 = torch.zeros(M, 2 * Fp)
    h1[:, :F] = torch.randn(M, F, generator=g)
    h1[:, Fp:Fp + F] = torch.randn(M, F, generator=g)
    # taps / gamma are operands of the kernel dtype: the reference uses the same (rounded) values
    convw = (torch.randn(2 * F, 3, generator=g) * 0.5).to(dev).to(dtype).float()
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(dev).to(dtype).float()
    h1d = h1.to(dev).to(dtype)
    h2 = torch.empty(M, Fp, device=dev, dtype=dtype)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    taps, gpad = ops.pack_conv_taps(convw, F, Fp).to(dtype), ops.pad_vector(gamma, Fp).to(dtype)
    gh = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype) if save_gh else None
    ops.ffmid_fwd(h1d, taps, gpad, h2, mean, rstd, nseq, F, Fp, 0.0, 0, gh=gh)
    h1r = h1d.double().requires_grad_(True)
    cr, gr = convw.double().requires_grad_(True), gamma.double().requires_grad_(True)
    ref = ffmid_reference(h1r, cr, gr, F, Fp, nseq)
    tol = 2e-5 if dtype == torch.float32 else 8e-3      # fp32: incl. the 1.5e-7 abs error of the A&S erf
    e_f = relerr(h2[:, :F], ref.detach())
    pad_zero = bool((h2[:, F:] == 0).all())
    dh2 = torch.zeros(M, Fp)
    dh2[:, :F] = torch.randn(M, F, generator=g)
    dh2d = dh2.to(dev).to(dtype)
    ref.backward(dh2d.double()[:, :F])
    du = torch.empty(M, 2 * Fp, device=dev, dtype=dtype)
    dh1 = torch.empty(M, 2 * Fp, device=dev, dtype=dtype)
    dgamma, dconv = torch.zeros(F, device=dev), torch.zeros(2 * F * 3, device=dev)
    ws = torch.empty(ops.ffmid_bwd_workspace_floats(F, Fp), device=dev)
    ops.ffmid_bwd(dh2d, h1d, taps, gpad, mean, rstd, du, dh1, dgamma, dconv, ws, nseq, F, Fp, 0.0, 0, gh=gh)
    if save_gh:
        assert bool((gh[:, F:] == 0).all()) and not torch.isnan(gh.float()).any()
    gref = h1r.grad
    e_x = max(relerr(dh1[:, :F], gref[:, :F]), relerr(dh1[:, Fp:Fp + F], gref[:, Fp:Fp + F]))
    e_g, e_c = relerr(dgamma, gr.grad), relerr(dconv.view(2 * F, 3), cr.grad)
    report(f"ffmid[{dtype},{F},gh={save_gh}]", fwd=e_f, dh1=e_x, dgamma=e_g, dconv=e_c, pad_zero=pad_zero)
    assert pad_zero and e_f < tol and e_x < (2e-5 if dtype == torch.float32 else 2e-2)
    assert e_g < (1e-4 if dtype == torch.float32 else 2e-2) and e_c < (1e-4 if dtype == torch.float32 else 2e-2)


def test_ffmid_dropout_statistics_and_replay(ops, dev):
    """First-generation (wave-per-row) kernels: Philox keep-mask, regenerated or read back from the stored bits."""
    ops.ffmid_set_impl(0)
    try:
        _ffmid_gen1_dropout_checks(ops, dev)
    finally:
        ops.ffmid_set_impl(1)


def _ffmid_gen1_dropout_checks(ops, dev):
    F, nseq = 341, 16
    Fp, M = 344, 64
    g = torch.Generator().manual_seed(9)
    h1 = torch.randn(M, 2 * Fp, generator=g).to(dev)
    convw = ops.pack_conv_taps(torch.randn(2 * F, 3, generator=g).to(dev), F, Fp)
    gamma = ops.pad_vector(torch.ones(F, device=dev), Fp)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    base = torch.empty(M, Fp, device=dev)
    ops.ffmid_fwd(h1, convw, gamma, base, mean, rstd, nseq, F, Fp, 0.0, 0)
    a, b, c = (torch.empty(M, Fp, device=dev) for _ in range(3))
    ops.ffmid_fwd(h1, convw, gamma, a, mean, rstd, nseq, F, Fp, 0.1, 1234)
    ops.ffmid_fwd(h1, convw, gamma, b, mean, rstd, nseq, F, Fp, 0.1, 1234)
    ops.ffmid_fwd(h1, convw, gamma, c, mean, rstd, nseq, F, Fp, 0.1, 99)
    kept = (a[:, :F] != 0)
    frac = 1 - kept.float().mean().item()
    scale_ok = relerr(a[:, :F][kept], (base[:, :F] / 0.9)[kept])
    report("ffmid_dropout", dropped_frac=frac, scale=scale_ok)
    assert torch.equal(a, b) and not torch.equal(a, c)          # mask is a pure function of (seed, element)
    salt = torch.tensor([5], dtype=torch.int64, device=dev)
    d, e = torch.empty(M, Fp, device=dev), torch.empty(M, Fp, device=dev)
    ops.ffmid_fwd(h1, convw, gamma, d, mean, rstd, nseq, F, Fp, 0.1, 1234, seed_dev=salt)
    salt += 1                                                   # what a graph replay does between steps
    ops.ffmid_fwd(h1, convw, gamma, e, mean, rstd, nseq, F, Fp, 0.1, 1234, seed_dev=salt)
    assert not torch.equal(d, a) and not torch.equal(d, e)
    assert abs(frac - 0.1) < 0.02 and scale_ok < 1e-5
    # the stored keep-mask (1 bit per element) reproduces the regenerated one in the backward, bit for bit
    bits = torch.empty(M, Fp // 8, dtype=torch.uint8, device=dev)
    a2 = torch.empty(M, Fp, device=dev)
    ops.ffmid_fwd(h1, convw, gamma, a2, mean, rstd, nseq, F, Fp, 0.1, 1234, drop_bits=bits)
    assert torch.equal(a2, a)
    unpacked = ((bits[:, :, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(M, Fp).bool()
    assert torch.equal(unpacked[:, :F], kept | (base[:, :F] == 0))
    dh2 = torch.randn(M, Fp, generator=g).to(dev)
    outs = []
    for db in (None, bits):
        du, dh1 = torch.empty(M, 2 * Fp, device=dev), torch.empty(M, 2 * Fp, device=dev)
        dgamma, dconv = torch.zeros(F, device=dev), torch.zeros(2 * F * 3, device=dev)
        ws = torch.empty(ops.ffmid_bwd_workspace_floats(F, Fp), device=dev)
        ops.ffmid_bwd(dh2, h1, convw, gamma, mean, rstd, du, dh1, dgamma, dconv, ws, nseq, F, Fp, 0.1, 1234, drop_bits=db)
        outs.append((dh1, dgamma, dconv))
    assert torch.equal(outs[0][0], outs[1][0])                  # dh1: deterministic, must match bit for bit
    assert relerr(outs[0][1], outs[1][1]) < 1e-5 and relerr(outs[0][2], outs[1][2]) < 1e-5   # atomically reduced partials


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("F,nseq,Bn,p", [(341, 80, 2, 0.0), (341, 80, 2, 0.1), (2730, 45, 2, 0.1), (1024, 37, 3, 0.1)])
def test_ffmid_strip_kernels(ops, dev, F, nseq, Bn, p, dtype):
    """Second-generation (column-strip) kernels: several strips per sample (conv / conv^T halos across strip boundaries), the chunk
    holding the F boundary, dropout through the stored keep bits -- against fp64 autograd with the SAME mask, and against the
    first-generation kernels on identical inputs."""
    M = nseq * Bn
    Fp = (F + 7) // 8 * 8
    lo = dtype in (torch.bfloat16, torch.float16)
    t_f, t_g, t_ab = (8e-3, 2e-2, 2e-2) if lo else (2e-5, 1e-4, 1e-4)
    g = torch.Generator().manual_seed(F + nseq)
    h1 = torch.zeros(M, 2 * Fp)
    h1[:, :F] = torch.randn(M, F, generator=g)
    h1[:, Fp:Fp + F] = torch.randn(M, F, generator=g)
    convw = (torch.randn(2 * F, 3, generator=g) * 0.5).to(dev).to(dtype).float()
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(dev).to(dtype).float()
    h1d = h1.to(dev).to(dtype)
    taps, gpad = ops.pack_conv_taps(convw, F, Fp).to(dtype), ops.pad_vector(gamma, Fp).to(dtype)
    dh2 = torch.zeros(M, Fp)
    dh2[:, :F] = torch.randn(M, F, generator=g)
    dh2d = dh2.to(dev).to(dtype)
    res = {}
    try:
        for impl in (1, 0):
            ops.ffmid_set_impl(impl)
            h2 = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
            gh = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            bits = torch.zeros(M, Fp // 8, dtype=torch.uint8, device=dev) if p > 0 else None
            ops.ffmid_fwd(h1d, taps, gpad, h2, mean, rstd, nseq, F, Fp, p, 4321, drop_bits=bits, gh=gh)
            du = torch.empty(M, 2 * Fp, device=dev, dtype=dtype)
            dh1 = torch.full((M, 2 * Fp), float("nan"), device=dev, dtype=dtype)
            dgamma, dconv = torch.zeros(F, device=dev), torch.zeros(2 * F * 3, device=dev)
            ws = torch.empty(ops.ffmid_bwd_workspace_floats(F, Fp), device=dev)
            # the backward of BOTH generations runs from the strip forward's mask / gh, so their outputs are comparable
            b_use, gh_use = (bits, gh) if impl == 1 else (res[1]["bits"], res[1]["gh"])
            ops.ffmid_bwd(dh2d, h1d, taps, gpad, res[1]["mean"] if impl == 0 else mean, res[1]["rstd"] if impl == 0 else rstd,
                          du, dh1, dgamma, dconv, ws, nseq, F, Fp, p, 4321, drop_bits=b_use, gh=gh_use)
            res[impl] = dict(h2=h2, gh=gh, mean=mean, rstd=rstd, bits=bits, dh1=dh1, dgamma=dgamma, dconv=dconv)
    finally:
        ops.ffmid_set_impl(1)
    new, old = res[1], res[0]
    # fp64 reference with the new forward's mask
    h1r = h1d.double().requires_grad_(True)
    cr, gr = convw.double().requires_grad_(True), gamma.double().requires_grad_(True)
    ref = ffmid_reference(h1r, cr, gr, F, Fp, nseq)
    if p > 0:
        keep = ((new["bits"][:, :, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(M, Fp)[:, :F].double()
        frac = 1.0 - float(keep.mean())
        assert abs(frac - p) < 0.01, frac
        refd = ref * keep / (1 - p)
    else:
        refd = ref
    refd.backward(dh2d.double()[:, :F])
    e_f = relerr(new["h2"][:, :F], refd.detach())
    e_gh = relerr(new["gh"][:, :F], (ref / gr).detach())
    gref = h1r.grad
    e_x = max(relerr(new["dh1"][:, :F], gref[:, :F]), relerr(new["dh1"][:, Fp:Fp + F], gref[:, Fp:Fp + F]))
    e_g, e_c = relerr(new["dgamma"], gr.grad), relerr(new["dconv"].view(2 * F, 3), cr.grad)
    pads = bool((new["h2"][:, F:] == 0).all() and (new["gh"][:, F:] == 0).all() and (new["dh1"][:, F:Fp] == 0).all()
                and (new["dh1"][:, Fp + F:] == 0).all())
    e_stat = max(relerr(new["mean"], old["mean"]), relerr(new["rstd"], old["rstd"]))
    e_ab = max(relerr(new["dh1"], old["dh1"].float()), relerr(new["dgamma"], old["dgamma"]), relerr(new["dconv"], old["dconv"]))
    report(f"ffmid_strip[{dtype},{F},{nseq},p={p}]", fwd=e_f, gh=e_gh, dh1=e_x, dgamma=e_g, dconv=e_c, stats_vs_gen1=e_stat,
           bwd_vs_gen1=e_ab, pads_zero=pads)
    assert pads and e_f < t_f and e_gh < t_f and e_x < (t_g if lo else 2e-5) and e_g < t_g and e_c < t_g
    assert e_stat < 1e-4 and e_ab < t_ab
    if p == 0:
        assert relerr(new["h2"].float(), old["h2"].float()) < t_f


def test_ffmid_strip_dropout_replay_bf16(ops, dev):
    """The strip forward's keep-mask is a pure function of (seed, salt, element) and its stored bits are what it applied."""
    F, nseq, Fp, M = 341, 40, 344, 80
    g = torch.Generator().manual_seed(11)
    h1 = torch.randn(M, 2 * Fp, generator=g).to(dev).bfloat16()
    convw = ops.pack_conv_taps(torch.randn(2 * F, 3, generator=g).to(dev), F, Fp).bfloat16()
    gamma = ops.pad_vector(torch.ones(F, device=dev), Fp).bfloat16()
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)

    def run(p, seed, salt=None):
        out = torch.empty(M, Fp, device=dev, dtype=torch.bfloat16)
        bits = torch.zeros(M, Fp // 8, dtype=torch.uint8, device=dev)
        ops.ffmid_fwd(h1, convw, gamma, out, mean, rstd, nseq, F, Fp, p, seed, seed_dev=salt, drop_bits=bits)
        return out, bits
    base, _ = run(0.0, 0)
    a, ba = run(0.1, 1234)
    b, bb = run(0.1, 1234)
    c, bc = run(0.1, 99)
    assert torch.equal(a, b) and torch.equal(ba, bb) and not torch.equal(ba, bc)
    salt = torch.tensor([5], dtype=torch.int64, device=dev)
    d, bd = run(0.1, 1234, salt)
    salt += 1
    e, be = run(0.1, 1234, salt)
    assert not torch.equal(bd, ba) and not torch.equal(bd, be)
    keep = ((ba[:, :, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(M, Fp)[:, :F].bool()
    frac = 1 - keep.float().mean().item()
    assert abs(frac - 0.1) < 0.01, frac
    assert bool((a[:, :F][~keep] == 0).all())
    scale = relerr(a[:, :F][keep].float(), (base[:, :F].float() / 0.9)[keep])
    # neighbouring rows / chunks must not share masks (the counter really is (row, chunk))
    assert float((ba[0] == ba[1]).float().mean()) < 0.2 and float((ba[:, 0] == ba[:, 1]).float().mean()) < 0.2
    report("ffmid_strip_dropout", dropped_frac=frac, scale=scale)
    assert scale < 1e-2          # bf16 rounding of y * (1 / 0.9) against the rounded base


def test_embed_gather_fwd_bwd(ops, dev):
    B, D = 3, 64
    g = torch.Generator().manual_seed(3)
    tables = [torch.randn(30, D, generator=g).to(dev), torch.randn(11, D, generator=g).to(dev)]
    starts = [torch.randn(D, generator=g).to(dev), torch.randn(D, generator=g).to(dev)]
    lens = [5, 4]
    N = sum(lens) + 2
    ids = torch.full((B, N), -2, dtype=torch.int32)
    ids[:, 1:6] = torch.randint(0, 30, (B, 5), generator=g, dtype=torch.int32)
    ids[:, 7:] = torch.randint(0, 11, (B, 4), generator=g, dtype=torch.int32)
    ids[0, 2] = -1
    seg = torch.tensor([0] * 6 + [1] * 5, dtype=torch.int32)
    posidx = torch.zeros(N, dtype=torch.int32)
    ids, seg, posidx = ids.to(dev), seg.to(dev), posidx.to(dev)
    out = torch.empty(B, N, D, device=dev)
    ops.embed_fwd(ids, seg, posidx, tables, starts, None, out)
    ref = torch.zeros(B, N, D, device=dev)
    for b in range(B):
        for n in range(N):
            s, i = int(seg[n]), int(ids[b, n])
            ref[b, n] = starts[s] if i == -2 else (tables[s][i] if i >= 0 else 0)
    assert torch.equal(out, ref)                                 # a gather is a copy: bit-exact
    dx = torch.randn(B, N, D, generator=g).to(dev)
    dt = [torch.zeros_like(t) for t in tables]
    ds = [torch.zeros_like(s) for s in starts]
    ops.embed_bwd(ids, seg, posidx, dt, ds, None, dx, 0.1)
    rt = [torch.zeros_like(t, dtype=torch.float64) for t in tables]
    rs = [torch.zeros_like(s, dtype=torch.float64) for s in starts]
    for b in range(B):
        for n in range(N):
            s, i = int(seg[n]), int(ids[b, n])
            if i == -2:
                rs[s] += 0.1 * dx[b, n].double()
            elif i >= 0:
                rt[s][i] += 0.1 * dx[b, n].double()
    e = max(max(relerr(a, b_) for a, b_ in zip(dt, rt)), max(relerr(a, b_) for a, b_ in zip(ds, rs)))
    report("embed", bwd=e)
    assert e < 1e-5


def test_cross_entropy_fwd_bwd(ops, dev):
    R, V, ld = 97, 41, 48
    g = torch.Generator().manual_seed(4)
    logits = torch.zeros(R, ld)
    logits[:, :V] = torch.randn(R, V, generator=g) * 5
    labels = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    lg, lb = logits.to(dev), labels.to(dev)
    lse, nll = torch.empty(R, device=dev), torch.zeros(1, device=dev)
    ops.ce_fwd(lg, lb, lse, nll, V)
    lr = lg[:, :V].double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lr, lb.long(), reduction="sum")
    e_f = abs(float(nll) - float(loss)) / float(loss)
    gs = torch.tensor([0.25], device=dev)
    (loss * 0.25 * 3.0).backward()
    dl = torch.full((R, ld), float("nan"), device=dev)
    ops.ce_bwd(lg, lb, lse, gs, 3.0, dl, V)
    e_b = relerr(dl[:, :V], lr.grad)
    report("cross_entropy", fwd=e_f, bwd=e_b)
    assert e_f < 1e-5 and e_b < 1e-4 and bool((dl[:, V:] == 0).all())


def test_adamw_clip_matches_torch(ops, dev):
    n = 10007
    g = torch.Generator().manual_seed(6)
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref_p], lr=3e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    P, M, V = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    nsq = torch.zeros(1, device=dev)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.001)
        ref_p.grad = grad.double().clone() / 2.0                  # grad_scale 0.5 (2-rank mean)
        torch.nn.utils.clip_grad_norm_([ref_p], 0.5)
        opt.step()
        G = grad.clone().to(dev)
        nsq.zero_()
        ops.sumsq_accumulate(G, nsq)
        ops.adamw_clip_step(P, G, M, V, None, lr=3e-4, beta1=0.9, beta2=0.99, eps=1e-8, wd=0.01, step=step, gscale=0.5,
                            gnorm_sq=nsq, max_norm=0.5, decoupled=True, zero_grad=True)
        assert float(G.abs().max()) == 0.0
    e = relerr(P.cpu(), ref_p.detach())
    report("adamw", relerr=e)
    assert e < 1e-6


def test_rvq_and_kmeans_bit_exact(ops, dev, golden_dir):
    from oracle import musiclm_oracle as O
    rng = np.random.RandomState(0)
    cb = rng.randn(12, 256, 512).astype(np.float32)
    x = rng.randn(33, 512).astype(np.float32)
    exp = O.rvq_encode(x, cb)
    cbT = torch.from_numpy(cb).to(dev).transpose(1, 2).contiguous()
    idx = torch.empty(33, 12, dtype=torch.int32, device=dev)
    res = torch.empty(33, 512, device=dev)
    ops.rvq_encode(torch.from_numpy(x).to(dev), cbT, idx, res, 33, 512, 256, 12)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), exp)           # bit-exact ids vs the oracle's fixed-order -cdist form
    z = np.load(os.path.join(golden_dir, "kmeans_assign.npz"))
    cT = torch.from_numpy(z["centroids"]).to(dev).t().contiguous()
    out = torch.empty(len(z["x"]), dtype=torch.int32, device=dev)
    ops.nearest_centroid(torch.from_numpy(z["x"]).to(dev), cT, out, len(z["x"]), z["x"].shape[1], cT.shape[1])
    assert np.array_equal(out.cpu().numpy().astype(np.int64), z["assign"])   # == sklearn predict fixture
    # ties -> lowest index; duplicated codeword
    cb2 = np.repeat(rng.randn(1, 4, 8).astype(np.float32), 1, axis=0)
    cb2[0, 3] = cb2[0, 1]
    x2 = cb2[0, 1][None].copy()
    i2 = torch.empty(1, 1, dtype=torch.int32, device=dev)
    ops.rvq_encode(torch.from_numpy(x2).to(dev), torch.from_numpy(cb2).to(dev).transpose(1, 2).contiguous(), i2, None, 1, 8, 4, 1)
    assert int(i2) == 1
    zz = np.load(os.path.join(golden_dir, "rvq_cdist_pin.npz"))              # ids written by torch.cdist (oracle/make_golden.py)
    cq = torch.from_numpy(zz["codebooks"]).to(dev).transpose(1, 2).contiguous()
    i3 = torch.empty(zz["indices"].shape, dtype=torch.int32, device=dev)
    ops.rvq_encode(torch.from_numpy(zz["x"]).to(dev), cq, i3, None, *zz["x"].shape, zz["codebooks"].shape[1], zz["codebooks"].shape[0])
    assert np.array_equal(i3.cpu().numpy().astype(np.int64), zz["indices"])
    report("rvq_kmeans", exact=True)


def test_rvq_ids_bit_exact_against_torch_cdist_at_real_dims(ops, dev):
    """The pin of SURVEY 8a row 16: ClapQuantized.quantize's ids (clap_quantized.py:75-87 -> vector-quantize-pytorch's
    `argmax(-cdist(x, embed))`) from the HIP kernel against torch.cdist ITSELF at the shipped dimensions -- 512-d, 1024 codes, 12
    residual stages, 10240 rows -- on exactly representable inputs (tests/rvq_cases.py: every summation order gives the same bits, so
    the ids are a property of the distance form), with engineered exact ties and the root-merged near-tie.  0 flips allowed."""
    import rvq_cases as RC
    from open_musiclm_amd.clap_quantized import ClapQuantized
    n, D, K, S = 10240, 512, 1024, 12
    x, cb, info = RC.exact_rvq_case(n, D, K, S, seed=21)
    want = RC.cdist_chain(x, cb)
    cq = ClapQuantized(clap=None, codebook_size=K, rq_num_quantizers=S, embed_dim=D).to(dev)
    cq.rq.codebooks.copy_(torch.from_numpy(cb))
    got = cq.quantize(torch.from_numpy(x).to(dev))[..., 0].cpu().numpy()
    rows = info["torch_rows"]
    flips = int((got[rows] != want[rows]).any(1).sum())
    report("rvq_vs_torch_cdist", rows=len(rows), stages=S, flipped_rows=flips)
    assert flips == 0
    RC.check_engineered(got, info)
    assert np.array_equal(RC.expanded_chain(x, cb)[rows], want[rows])


def test_kmeans_assign_real_dims_vs_sklearn(ops, dev):
    """HfHubertWithKmeans assign step at the shipped dimensions (768-d MERT features, 1024 centroids, hubert_kmeans_cfg of
    configs/model/musiclm_small.json) through the product class, against sklearn's MiniBatchKMeans.predict itself (the call the
    reference makes, hf_hubert_kmeans.py:87) on clustered features.  The mismatch count is reported; the bar is bit-exact ids."""
    from sklearn.cluster import MiniBatchKMeans
    from open_musiclm_amd.hf_hubert_kmeans import HfHubertWithKmeans
    rng = np.random.RandomState(0)
    centers = rng.randn(1024, 768).astype(np.float32)
    feats = (centers[rng.randint(0, 1024, 8192)] + 0.7 * rng.randn(8192, 768)).astype(np.float32)
    km = MiniBatchKMeans(n_clusters=1024, batch_size=2048, n_init=1, random_state=0, max_iter=3).fit(feats)
    x = (centers[rng.randint(0, 1024, 4096)] + 0.7 * rng.randn(4096, 768)).astype(np.float32)
    ref = km.predict(x).astype(np.int64)
    w2v = HfHubertWithKmeans(hubert=None, kmeans=km, normalize_embeds=False).to(dev)
    got = w2v.kmeans.predict(torch.from_numpy(x).to(dev)).cpu().numpy()
    mism = int((got != ref).sum())
    report("kmeans_768x1024", mismatches=mism, n=len(x))
    assert mism == 0, mism


def test_sampler_matches_oracle(ops, dev):
    from oracle import musiclm_oracle as O
    B, V = 5, 1025
    g = torch.Generator().manual_seed(8)
    logits = torch.randn(B, 1032, generator=g) * 4
    u = torch.rand(B, V, generator=g)
    last = logits[:, :V].clone()
    last[:, -1] = float("-inf")
    exp = O.gumbel_argmax(O.top_k_filter(last, 0.9), u, 0.95)
    out = torch.empty(B, dtype=torch.long, device=dev)
    ops.sample_topk_gumbel(logits.to(dev), u.to(dev), out, V, max(int(0.1 * V), 1), 0.95, True)
    report("sampler", equal=bool(torch.equal(out.cpu(), exp)))
    assert torch.equal(out.cpu(), exp)


def test_rvq_fit_step_matches_the_restated_update_rules_parity_unpinned(ops, dev):
    """PARITY UNPINNED: the update rules come from vector-quantize-pytorch (>= 1.2.2, setup.py:31 of the reference), which is not vendored
    in /root/reference and not installed here -- oracle.rvq_fit_step restates its published algorithm and is checked against nothing
    but itself.  Training-mode residual VQ on the device (csrc/vq_fit.hip + the nearest-codeword kernel) against oracle.rvq_fit_step with
    the same initial picks: k-means init, EMA updates and dead-code re-seeding.  Sums are fp32 atomics on the device, so codes
    agree to rounding and the assignments almost everywhere."""
    from open_musiclm_amd.clap_quantized import ClapQuantized
    from oracle import musiclm_oracle as O
    S, K, D, n = 3, 64, 32, 1500
    centers = torch.randn(K, D, generator=torch.Generator().manual_seed(9))

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        return centers[torch.randint(0, K, (n,), generator=g)] + 0.1 * torch.randn(n, D, generator=g)
    cq = ClapQuantized(clap=None, codebook_size=K, rq_num_quantizers=S, embed_dim=D, rq_ema_decay=0.9, learn_rvq=True,
                       threshold_ema_dead_code=0.3).to(dev)
    state = dict(embed=torch.zeros(S, K, D), embed_avg=torch.zeros(S, K, D), cluster_size=torch.zeros(S, K),
                 initted=torch.zeros(S, dtype=torch.bool))
    pick_log = []

    def picks(nn_, k_):
        g = torch.Generator().manual_seed(1000 + len(pick_log))
        p = torch.randperm(nn_, generator=g)[:k_]
        pick_log.append(p)
        return p
    worst_code, worst_loss, agree = 0.0, 0.0, 1.0
    for step in range(4):
        x = batch(50 + step)
        pick_log.clear()
        cq.rq.init_pick_source = picks
        cq.rq.expire_pick_source = picks
        idx, q = cq.rq.fit_step(x.to(dev))
        loss = float(((q.cpu() - x) ** 2).mean())
        # replay with the same picks in the same order of consumption (per layer: init pick if needed, then expiry pick if needed)
        log = list(pick_log)

        class Feed:
            def __init__(self): self.i = 0
            def __getitem__(self, s):
                p = log[self.i]; self.i += 1
                return p
        feed = Feed()
        # the oracle indexes init_picks[s] / expire_picks[s] exactly when the device path called its pick source
        o_idx, o_loss = O.rvq_fit_step(state, x, decay=0.9, threshold_dead=0.3, init_picks=feed, expire_picks=feed)
        assert feed.i == len(log), (feed.i, len(log))
        worst_code = max(worst_code, relerr(cq.rq.codebooks.cpu(), state["embed"]))
        worst_loss = max(worst_loss, abs(loss - o_loss) / o_loss)
        agree = min(agree, float((idx.cpu() == o_idx).float().mean()))
        assert relerr(cq.rq.cluster_size.cpu(), state["cluster_size"]) < 1e-3
        # keep the two trajectories on the same codes: rounding-level differences must not compound through near-ties
        state["embed"].copy_(cq.rq.codebooks.cpu()); state["embed_avg"].copy_(cq.rq.embed_avg.cpu())
        state["cluster_size"].copy_(cq.rq.cluster_size.cpu())
    report("rvq_fit", codes=worst_code, loss=worst_loss, index_agreement=agree)
    # the device sums are fp32 atomics (order-dependent at the 1e-7 level): a near-tie may flip a single assignment, which moves
    # that bucket's mean by ~1/count -- only then is the looser code bar in force
    assert bool(cq.rq.initted.all()) and worst_loss < 1e-3 and agree > 0.995
    assert worst_code < (1e-3 if agree == 1.0 else 5e-2), (worst_code, agree)
    # eval-mode encode with the fitted codebooks stays bit-exact against the stated chain
    cq.learn_rvq = False
    x = batch(99)[:64]
    got = cq.quantize(x.to(dev)).squeeze(-1).cpu()
    assert torch.equal(got, torch.from_numpy(O.rvq_encode(x.numpy(), cq.rq.codebooks.cpu().numpy())))


def test_clap_rvq_trainer_on_embeddings(dev, tmp_path):
    """ClapRVQTrainer (trainer.py:564-741) with a dataset of precomputed embeddings: steps run, the loss falls, and the saved
    checkpoint carries the library's key layout and reloads into a fresh quantizer that encodes identically."""
    from open_musiclm_amd.clap_quantized import ClapQuantized
    from open_musiclm_amd.trainer import ClapRVQTrainer
    torch.manual_seed(0)                             # k-means init picks and the loader shuffle come from the global generators
    K, D, S = 32, 16, 4
    centers = torch.randn(K, D, generator=torch.Generator().manual_seed(3))
    g = torch.Generator().manual_seed(4)
    data = centers[torch.randint(0, K, (2048,), generator=g)] + 0.05 * torch.randn(2048, D, generator=g)
    cq = ClapQuantized(clap=None, codebook_size=K, rq_num_quantizers=S, embed_dim=D, learn_rvq=True).to(dev)
    trainer = ClapRVQTrainer(num_train_steps=5, batch_size=256, accumulate_batches=2, audio_conditioner=cq,
                             dataset=torch.utils.data.TensorDataset(data), valid_frac=0.05, save_model_every=2,
                             save_results_every=2, results_folder=str(tmp_path / "rvq"))
    logs = []
    trainer.train(log_fn=logs.append)
    assert len(logs) == 5 and logs[-1]["train_loss"] < 0.05 ** 2 * 6 and logs[0]["valid_loss"] is not None
    ck = torch.load(str(tmp_path / "rvq" / "clap.rvq.4.pt"), map_location="cpu")
    assert "layers.0._codebook.embed" in ck and ck["layers.3._codebook.embed"].shape == (1, K, D)
    fresh = ClapQuantized(clap=None, codebook_size=K, rq_num_quantizers=S, embed_dim=D).to(dev)
    fresh.rq.load_state_dict(ck)
    cq.learn_rvq = False
    x = data[:100].to(dev)
    assert torch.equal(fresh.quantize(x), cq.quantize(x))


def test_index_guards_skip_and_flag(ops, dev):
    """A token id / position / label past its table is skipped (no out-of-bounds access) and flagged; torch's embedding and
    cross entropy raise a device assert in the same situation (embed_ce.hip)."""
    B, D = 2, 32
    tables = [torch.randn(10, D, device=dev)]
    starts = [torch.randn(D, device=dev)]
    ids = torch.tensor([[-2, 3, 10, 9], [-2, 11, 0, -1]], dtype=torch.int32, device=dev)      # 10 and 11 are past the 10-row table
    seg = torch.zeros(4, dtype=torch.int32, device=dev)
    posidx = torch.zeros(4, dtype=torch.int32, device=dev)
    out = torch.full((B, 4, D), float("nan"), device=dev)
    ops.raise_on_index_error(dev)                                   # clear
    ops.embed_fwd(ids, seg, posidx, tables, starts, None, out)
    assert torch.equal(out[0, 1], tables[0][3]) and torch.equal(out[0, 3], tables[0][9])
    assert bool((out[0, 2] == 0).all()) and bool((out[1, 1] == 0).all()) and torch.equal(out[1, 0], starts[0])
    with pytest.raises(IndexError, match="token id"):
        ops.raise_on_index_error(dev)
    ops.raise_on_index_error(dev)                                   # flag was cleared by the raise
    dt, ds = [torch.zeros(10, D, device=dev)], [torch.zeros(D, device=dev)]
    dx = torch.ones(B, 4, D, device=dev)
    ops.embed_bwd(ids, seg, posidx, dt, ds, None, dx, 1.0)
    assert float(dt[0].sum()) == 3 * D and float(ds[0].sum()) == 2 * D      # rows 3, 9, 0 once each; the two bad ids nowhere
    with pytest.raises(IndexError):
        ops.raise_on_index_error(dev)
    V, ld = 5, 8
    logits = torch.randn(3, ld, device=dev)
    labels = torch.tensor([1, 7, -1], dtype=torch.int32, device=dev)           # 7 >= V
    lse, nll = torch.empty(3, device=dev), torch.zeros(1, device=dev)
    ops.ce_fwd(logits, labels, lse, nll, V)
    ref = torch.logsumexp(logits[0, :V], 0) - logits[0, 1]
    assert abs(float(nll) - float(ref)) < 1e-5
    with pytest.raises(IndexError, match="label"):
        ops.raise_on_index_error(dev)
    d = torch.full((3, ld), float("nan"), device=dev)
    ops.ce_bwd(logits, labels, lse, None, 1.0, d, V)
    assert bool((d[1] == 0).all()) and bool((d[2] == 0).all()) and torch.isfinite(d).all()


@pytest.mark.parametrize("n,Hd,H", [(1116, 512, 8), (1817, 512, 16), (37, 256, 3)])
def test_relpos_mlp_fused_kernels_vs_fp64(ops, dev, n, Hd, H):
    """The rel-pos MLP (transformer.py:36-67) as one forward launch and two backward launches (round 5, csrc/optim_misc.hip) against torch
    autograd in fp64 with nn.Linear-scale weights: the bias table, the saved activations' consistency (backward from the forward's own
    saves) and all eight parameter gradients, ACCUMULATED onto non-zero buffers; the launches are deterministic (bit-equal repeats)."""
    g = torch.Generator(device="cpu").manual_seed(n + Hd + H)
    ldb = (H + 7) // 8 * 8
    u = lambda *shape, fan: ((torch.rand(*shape, generator=g) * 2 - 1) / fan ** 0.5)
    w0, b0 = u(Hd, fan=1), u(Hd, fan=1)
    W1, b1, W2, b2 = u(Hd, Hd, fan=Hd), u(Hd, fan=Hd), u(Hd, Hd, fan=Hd), u(Hd, fan=Hd)
    W3, b3 = u(H, Hd, fan=Hd), u(H, fan=Hd)
    dtab = torch.zeros(n, ldb)
    dtab[:, :H] = torch.randn(n, H, generator=g)
    P = [t.double().requires_grad_(True) for t in (w0, b0, W1, b1, W2, b2, W3, b3)]
    x = torch.arange(n, dtype=torch.float64)[:, None]
    silu = torch.nn.functional.silu
    h = silu(x * P[0][None, :] + P[1])
    h = silu(h @ P[2].t() + P[3])
    h = silu(h @ P[4].t() + P[5])
    tab = h @ P[6].t() + P[7]
    tab.backward(dtab[:, :H].double())
    D = [t.to(dev) for t in (w0, b0, W1, b1, W2, b2, W3, b3)]
    saves = [torch.empty(n, Hd, device=dev) for _ in range(6)]
    table = torch.full((n, ldb), float("nan"), device=dev)
    ops.relpos_mlp_fwd(*D, saves, table, n, Hd, H, ldb)
    table2 = torch.full((n, ldb), float("nan"), device=dev)
    ops.relpos_mlp_fwd(*D, None, table2, n, Hd, H, ldb)                      # the no-save form (decode / eval)
    e_tab = relerr(table[:, :H].cpu(), tab.detach())
    init = [torch.randn(t.shape, generator=g) for t in (w0, b0, W1, b1, W2, b2, W3, b3)]
    runs = []
    for rep in range(2):
        grads = [t.clone().to(dev) for t in init]
        scratch = torch.empty(3 * n * Hd, device=dev)
        ops.relpos_mlp_bwd(dtab.to(dev), D[2], D[4], D[6], saves, scratch, grads, n, Hd, H, ldb)
        runs.append(grads)
    torch.cuda.synchronize()
    errs = {nm: relerr(gk.cpu().double() - i0.double(), p.grad) for nm, gk, i0, p in zip("w0 b0 W1 b1 W2 b2 W3 b3".split(), runs[0], init, P)}
    same = all(torch.equal(a_, b_) for a_, b_ in zip(runs[0], runs[1]))
    times = {}
    for nm, fn in (("fwd_us", lambda: ops.relpos_mlp_fwd(*D, saves, table, n, Hd, H, ldb)),
                   ("bwd_us", lambda: ops.relpos_mlp_bwd(dtab.to(dev), D[2], D[4], D[6], saves, scratch, grads, n, Hd, H, ldb))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        times[nm] = e0.elapsed_time(e1) * 100.0
    report(f"relpos_mlp_fused[{n},{Hd},{H}]", table=e_tab, grads=errs, deterministic=same, **times)
    assert torch.equal(table, table2) and not torch.isnan(table).any()
    assert (table[:, H:] == 0).all()
    assert e_tab < 1e-5, e_tab
    assert same
    assert max(errs.values()) < 5e-5, errs            # fp32 accumulation over up to 1817 rows, added to O(1) initial values


# ---------------------------------------------------------------------------------------------------------------------
# precision "fp16ff" (round 5): the ConvFeedForward forward on hi/lo planes of the 16-bit operand type
# ---------------------------------------------------------------------------------------------------------------------
def hilo(x, dtype):
    """x fp32 -> (hi, lo) planes of `dtype`: hi = rne(x), lo = rne(x - hi)."""
    hi = x.to(dtype)
    return hi, (x - hi.float()).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,planes_out", [(1100, 1280, 1024, True),      # 256 x 256 tiles on the half-tile-ring kernel, ragged M
                                              (2300, 1024, 2752, False),     # FF-out form: fp32 result + residual, 256 x 256 with a peeled tail of m-tiles
                                              (200, 136, 72, True),          # 128 x 128 tile, ragged everything, K not a whole k-tile
                                              (2100, 600, 320, False),       # 256 x 128 tile
                                              (333, 5504, 1024, True)])      # FF-in width at a short M (N >= 2048: 256 x 256)
def test_gemm_planes16_vs_fp64(ops, dev, dtype, M, N, K, planes_out):
    """omlm_gemm_planes16: C = A B^T (+ Cin) with both operands as hi/lo planes -- three products, fp32 accumulation.  Against the fp64
    product of the values the planes hold (hi + lo).  The bar is the accumulation's: 16-bit operand rounding is gone (fp16 planes carry
    ~21 bits, bf16 planes 16: the lo * lo product that is left out is 2^-22 / 2^-16 relative to a term)."""
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.05).to(dev)
    Ah, Al = hilo(A, dtype)
    Bh, Bl = hilo(B, dtype)
    ref = (Ah.double() + Al.double()) @ (Bh.double() + Bl.double()).t()
    bar = 3e-6 if dtype == torch.float16 else 6e-5
    if planes_out:
        C, Cl = torch.full((M, N), float("nan"), device=dev, dtype=dtype), torch.full((M, N), float("nan"), device=dev, dtype=dtype)
        ops.gemm_planes16(Ah, Al, Bh, Bl, C, Cl, M=M, N=N, K=K)
        e_sum = relerr(C.double() + Cl.double(), ref)
        # the hi plane is the correctly rounded result (up to the accumulation error moving a value across a rounding boundary)
        e_hi = relerr(C, ref)
        flips = float((C != ref.to(dtype)).float().mean())
        report(f"gemm_planes16[{dtype},{M},{N},{K},planes]", hi_plus_lo=e_sum, hi=e_hi, hi_not_rne_frac=flips)
        # hi + lo: the result to ~2^-21 (fp16 planes; lo may be subnormal below 6e-5) / 2^-16 (bf16 planes)
        assert e_sum < (4e-6 if dtype == torch.float16 else 6e-5), e_sum
        assert e_hi < (6e-4 if dtype == torch.float16 else 5e-3) and flips < (2e-3 if dtype == torch.float16 else 1e-2)      # (bf16 planes: 2^-16 against an ulp of 2^-8)
    else:
        Cin = torch.randn(M, N, generator=g).to(dev)
        C = torch.full((M, N), float("nan"), device=dev)
        ops.gemm_planes16(Ah, Al, Bh, Bl, C, M=M, N=N, K=K, Cin=Cin)
        e = relerr(C, ref + Cin.double())
        # what single-plane operands give on the same data, for the report
        C1 = torch.empty(M, N, device=dev)
        ops.gemm(Ah, Bh, C1, M=M, N=N, K=K, Cin=Cin)
        report(f"gemm_planes16[{dtype},{M},{N},{K},fp32]", err=e, single_plane_err=relerr(C1, ref + Cin.double()))
        assert e < bar, e


def torch_fp8_planes(ops, hi, lo, dev, bound=None):
    """Reference quantiser of omlm_gemm_mx16's fp8 planes, in torch on the CPU: one power-of-two scale per row (bound / 2^e <= 256), hi8 =
    e4m3(hi 2^-e), lo8 = e4m3(lo 2^-(e - 11)).  Returns (ops.Fp8Planes on the device, dequantised hi8 [fp64], dequantised lo8 [fp64])."""
    hi, lo = hi.float().cpu(), lo.float().cpu()
    R, K = hi.shape
    b = hi.abs().amax(1) if bound is None else bound.float().cpu()
    e = torch.ceil(torch.log2(b.clamp(min=2.0 ** -100))) - 8.0
    q = lambda v, ee: (v * torch.exp2(-ee)[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn)
    h8, l8 = q(hi, e), q(lo, e - 11.0)
    P = ops.Fp8Planes(R, K, dev)
    P.planes[0, :R, :K] = h8.view(torch.uint8).to(dev)
    P.planes[1, :R, :K] = l8.view(torch.uint8).to(dev)
    P.scale[:R] = (e + 127).to(torch.uint8).to(dev)
    return P, h8.double() * torch.exp2(e)[:, None].double(), l8.double() * torch.exp2(e - 11.0)[:, None].double()


@pytest.mark.parametrize("M,N,K,planes_out", [(256, 256, 128, False),        # one tile, one half tile pair + 1 + 1 fp8 tiles
                                              (1100, 1280, 1024, True),      # ragged M, planes out (the FF-in form)
                                              (2300, 1024, 2752, False),     # FF-out form: K = 43 half tiles (pad tile) + 2 x 22 fp8 tiles (the last one half empty), fp32 + residual
                                              (333, 5504, 1024, True),       # FF-in width
                                              (8448, 2048, 1024, True),      # 264 tiles = one machine round + 8: the tail runs as a slice-storing split-K through the workspace
                                              (8448, 2048, 1024, False),
                                              (3200, 5504, 1024, True),      # 286 tiles: 242 full-round tiles (not a whole round) + 44 tail tiles x 4 slices in ONE grid (gemm_mx_fused_kernel)
                                              (3200, 5504, 1024, False)])
def test_gemm_mx16_vs_fp64(ops, dev, M, N, K, planes_out):
    """omlm_gemm_mx16: C = A_hi B_hi^T on half MFMAs + the two correction products on fp8 MFMAs with one scale per row.  (1) The kernel's own
    arithmetic: against the fp64 evaluation of exactly those three products on the values its planes hold -- the bar is fp32 accumulation.
    (2) What it is for: against the fp64 product of the un-split operands -- the fp8 rounding of the corrections leaves ~2^-15 per term."""
    g = torch.Generator().manual_seed(M + N + K)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    A[::7] *= 30.0                                             # rows of very different size: every row has its own scale
    B[3::5] *= 0.01
    Ah, Al = hilo(A.to(dev), torch.float16)
    Bh, Bl = hilo(B.to(dev), torch.float16)
    A8, a8h, a8l = torch_fp8_planes(ops, Ah, Al, dev)
    B8, b8h, b8l = torch_fp8_planes(ops, Bh, Bl, dev)
    ref_k = (Ah.double().cpu() @ Bh.double().cpu().t() + a8h @ b8l.t() + a8l @ b8h.t()).to(dev)          # what the kernel computes
    ref = (Ah.double() + Al.double()) @ (Bh.double() + Bl.double()).t()                                  # what it stands for
    scale = float(ref.abs().max())
    if planes_out:
        C, Cl = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16), torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        ops.gemm_mx16(Ah, A8, Bh, B8, C, Cl, M=M, N=N, K=K)
        got = C.double() + Cl.double()
        e_k, e = float((got - ref_k).abs().max()) / scale, float((got - ref).abs().max()) / scale
        # (planes out: the sum is known to the lo plane's own rounding, 2^-22 of the value, and subnormal below 6e-5)
        assert e_k < 4e-6, e_k
        # the lo plane as bf8 (e5m2) bytes: same hi plane bit for bit; the bytes are the upper bytes of halves, within e5m2's rounding
        # (3 significant bits) of the half lo plane, so hi + lo8 knows the value to 2^-11 2^-3
        C8, Cl8 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16), torch.full((M, N), 0x7f, device=dev, dtype=torch.uint8)
        ops.gemm_mx16(Ah, A8, Bh, B8, C8, Cl8, M=M, N=N, K=K)
        assert torch.equal(C8, C)
        lo8 = (Cl8.to(torch.int16) << 8).view(torch.float16).double()
        d8 = (lo8 - Cl.double()).abs()
        assert bool((d8 <= 0.126 * Cl.double().abs() + 2.0 ** -17).all()), float((d8 / (Cl.double().abs() + 1e-30)).max())
        e8 = float((C8.double() + lo8 - ref_k).abs().max()) / scale
        report(f"gemm_mx16_bf8_lo[{M},{N},{K}]", vs_own_products=e8)
        assert e8 < 2.0 ** -13, e8
    else:
        Cin = torch.randn(M, N, generator=g).to(dev)
        C = torch.full((M, N), float("nan"), device=dev)
        ops.gemm_mx16(Ah, A8, Bh, B8, C, M=M, N=N, K=K, Cin=Cin)
        e_k, e = float((C.double() - Cin.double() - ref_k).abs().max()) / scale, float((C.double() - Cin.double() - ref).abs().max()) / scale
        assert e_k < 2e-6, e_k
    # the tail's k-slices ride in the full-round launch (gemm_mx_fused_kernel); OMLM_MX_FUSE_TAIL=0 runs them as a launch of their own:
    # same tiles, same k order, same slice sum -- bit for bit
    old = os.environ.get("OMLM_MX_FUSE_TAIL")
    try:
        os.environ["OMLM_MX_FUSE_TAIL"] = "0"
        if planes_out:
            C2, Cl2 = torch.full_like(C, float("nan")), torch.full_like(Cl, float("nan"))
            ops.gemm_mx16(Ah, A8, Bh, B8, C2, Cl2, M=M, N=N, K=K)
            assert torch.equal(C2, C) and torch.equal(Cl2, Cl)
        else:
            C2 = torch.full_like(C, float("nan"))
            ops.gemm_mx16(Ah, A8, Bh, B8, C2, M=M, N=N, K=K, Cin=Cin)
            assert torch.equal(C2, C)
    finally:
        if old is None:
            os.environ.pop("OMLM_MX_FUSE_TAIL", None)
        else:
            os.environ["OMLM_MX_FUSE_TAIL"] = old
    # per-row view of the residual error (rows differ by 30x in size): the corrections' fp8 rounding, relative to the row's own largest output
    C1 = torch.empty(M, N, device=dev)
    ops.gemm(Ah, Bh, C1, M=M, N=N, K=K)
    e1 = float((C1.double() - ref).abs().max()) / scale
    report(f"gemm_mx16[{M},{N},{K},{'planes' if planes_out else 'fp32'}]", vs_own_products=e_k, vs_exact=e, single_half_product=e1)
    assert e < 6e-5 and e < e1 / 8, (e, e1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,D", [(333, 1024), (6700, 1024), (150, 768)])
def test_layernorm_fwd_planes(ops, dev, dtype, M, D):
    """omlm_layernorm_fwd_planes: the hi plane is bit for bit omlm_layernorm_fwd's output, hi + lo is the fp32 result to the lo plane's
    rounding; statistics identical."""
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    y0 = torch.empty(M, D, device=dev, dtype=dtype)
    m0, r0 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, y0, None, m0, r0)
    y, yl = torch.full((M, D), float("nan"), device=dev, dtype=dtype), torch.full((M, D), float("nan"), device=dev, dtype=dtype)
    m1, r1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd_planes(x, gamma, y, yl, m1, r1)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), None, 1e-5)
    e = relerr(y.double() + yl.double(), ref)
    report(f"layernorm_fwd_planes[{dtype},{M},{D}]", hi_plus_lo=e, hi=relerr(y, ref))
    assert torch.equal(y, y0) and torch.equal(m0, m1) and torch.equal(r0, r1)
    assert e < (3e-6 if dtype == torch.float16 else 3e-5), e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("F,nseq,Bn,p", [(341, 80, 2, 0.0), (2730, 45, 2, 0.1), (1024, 37, 3, 0.1)])
def test_ffmid_fwd_planes_vs_fp64(ops, dev, dtype, F, nseq, Bn, p):
    """omlm_ffmid_fwd_planes: conv3 + GEGLU + LayerNorm(F) + dropout on h1 / taps / gamma read as hi + lo, h2 left as planes.  Against fp64
    on the values the planes hold, with the kernel's own keep mask; statistics, keep bits and gh against the single-plane forward's."""
    M = nseq * Bn
    Fp = (F + 63) // 64 * 64
    g = torch.Generator().manual_seed(F + nseq)
    h1 = torch.zeros(M, 2 * Fp)
    h1[:, :F] = torch.randn(M, F, generator=g)
    h1[:, Fp:Fp + F] = torch.randn(M, F, generator=g)
    convw = (torch.randn(2 * F, 3, generator=g) * 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(dev)
    h1h, h1l = hilo(h1.to(dev), dtype)
    taps32, g32 = ops.pack_conv_taps(convw, F, Fp), ops.pad_vector(gamma, Fp)
    tph, tpl = hilo(taps32, dtype)
    gph, gpl = hilo(g32, dtype)
    h2, h2l = (torch.full((M, Fp), float("nan"), device=dev, dtype=dtype) for _ in range(2))
    gh = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    bits = torch.zeros(M, Fp // 8, dtype=torch.uint8, device=dev) if p > 0 else None
    ops.ffmid_fwd_planes(h1h, h1l, tph, tpl, gph, gpl, h2, h2l, mean, rstd, nseq, F, Fp, p, 4321, drop_bits=bits, gh=gh)
    # reference on the plane values
    cw = (tph.double() + tpl.double())                                   # [3, 2Fp] tap-major
    cwr = torch.cat([cw[:, :F], cw[:, Fp:Fp + F]], dim=1).t().contiguous()                 # [2F, 3]
    gmr = (gph.double() + gpl.double())[:F]
    ref_gh = ffmid_reference(h1h.double() + h1l.double(), cwr, torch.ones(F, dtype=torch.float64, device=dev), F, Fp, nseq)
    ref = ref_gh * gmr
    if p > 0:
        keep = ((bits[:, :, None] >> torch.arange(8, device=dev, dtype=torch.uint8)) & 1).reshape(M, Fp).bool()[:, :F]
        ref = torch.where(keep, ref / (1 - p), torch.zeros_like(ref))
        frac = 1 - keep.float().mean().item()
        assert abs(frac - p) < 0.02, frac
    e = relerr(h2[:, :F].double() + h2l[:, :F].double(), ref)
    e_gh = relerr(gh[:, :F], ref_gh)
    # the single-plane forward on the hi planes: same keep bits (a function of seed and position), statistics within its operand rounding
    h2s = torch.empty(M, Fp, device=dev, dtype=dtype)
    ms, rs = torch.empty(M, device=dev), torch.empty(M, device=dev)
    bits_s = torch.zeros(M, Fp // 8, dtype=torch.uint8, device=dev) if p > 0 else None
    ops.ffmid_fwd(h1h, tph, gph, h2s, ms, rs, nseq, F, Fp, p, 4321, drop_bits=bits_s, gh=None)
    e_single = relerr(h2s[:, :F], ref)
    report(f"ffmid_fwd_planes[{dtype},{F},{nseq},{p}]", hi_plus_lo=e, gh=e_gh, single_plane=e_single)
    assert bool((h2[:, F:] == 0).all()) and bool((h2l[:, F:] == 0).all())
    if p > 0:
        assert torch.equal(bits, bits_s)
    # erf by A&S 7.1.26 (1.5e-7 abs) bounds the kernel at ~2e-6 of the output range; the single-plane forward sits at its operand rounding
    assert e < (1e-5 if dtype == torch.float16 else 6e-5), e
    assert e_gh < (1e-3 if dtype == torch.float16 else 8e-3) and e < 0.05 * e_single


def _fp8_bytes(v, e):
    """e4m3 bytes of v 2^-e (torch's conversion on the CPU: RNE), e per row"""
    return (v.float().cpu() * torch.exp2(-e.float().cpu())[:, None]).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)


def _check_mx_planes(P, rows, K, hi, lo_ref, name, bound):
    """The fp8 planes of an operand against their definition: the scale is the one the analytic bound sqrt(width) max|gamma| gives (and bounds
    every row), hi8 is bit for bit e4m3(hi 2^-e), lo8 dequantises to the lo part within e4m3's rounding (relative to the scale's step where the
    entry is subnormal), the row tails up to a whole 128-byte k-tile are zero."""
    e = P.scale[:rows].cpu().to(torch.int32) - 127
    rowmax = hi.float().abs().amax(1).cpu()
    assert bool((torch.exp2((e + 8).float()) >= rowmax).all()), name + ": a row exceeds its scale"
    assert bool((e == e[0]).all()) and 2.0 ** (int(e[0]) + 8) >= bound and 2.0 ** (int(e[0]) + 7) < bound * 1.001, (name, int(e[0]), bound)
    assert torch.equal(P.planes[0, :rows, :K].cpu(), _fp8_bytes(hi, e)), name + ": hi8 bytes"
    dq_lo = P.planes[1, :rows, :K].cpu().view(torch.float8_e4m3fn).double() * torch.exp2((e - 11).double())[:, None]
    lo_ref = lo_ref.double().cpu()
    # e4m3: half an ulp of a 3-bit mantissa = 2^-4 of the entry, or half a subnormal step 2^(e - 11 - 10) below the normal range (+ fp32 noise of lo)
    tol = lo_ref.abs() * 0.0635 + torch.exp2((e - 21).double())[:, None] + 2e-7 * hi.double().abs().cpu()      # (last term: fp32 evaluation order of y itself)
    assert bool(((dq_lo - lo_ref).abs() <= tol).all()), name
    err = float((dq_lo - lo_ref).abs().max()) / max(float(lo_ref.abs().max()), 1e-30)
    K128 = (K + 127) // 128 * 128
    assert K128 <= P.planes.shape[2] and (K128 == K or float(P.planes[:, :rows, K:K128].max()) == 0), name + ": row tails"
    return err


@pytest.mark.parametrize("M,D", [(333, 1024), (6700, 1024), (150, 768), (40, 64)])
def test_layernorm_fwd_mx(ops, dev, M, D):
    """omlm_layernorm_fwd_mx: the half hi plane and the statistics are bit for bit omlm_layernorm_fwd's; the fp8 planes and row scales are
    what omlm_gemm_mx16 defines (written into an UNINITIALISED buffer: the kernel owns the row tails)."""
    g = torch.Generator().manual_seed(M)
    x = (torch.randn(M, D, generator=g) * 3 + 1).to(dev)
    x[5] *= 40.0; x[7] = x[7] * 1e-3 + 2.0                     # rows of very different spread
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    y0 = torch.empty(M, D, device=dev, dtype=torch.float16)
    m0, r0 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd(x, gamma, y0, None, m0, r0)
    P = ops.Fp8Planes(M, D, dev, zero=False)
    P.planes.fill_(0x7f); P.scale.fill_(0xff)                  # fp8 NaN bytes / NaN scales wherever the kernel does not write
    y = torch.full((M, D), float("nan"), device=dev, dtype=torch.float16)
    m1, r1 = torch.empty(M, device=dev), torch.empty(M, device=dev)
    ops.layernorm_fwd_mx(x, gamma, y, P, m1, r1)
    if D == 1024:             # the product's width: same reduction code as omlm_layernorm_fwd's row kernel -> the same bits
        assert torch.equal(y, y0) and torch.equal(m0, m1) and torch.equal(r0, r1)
    else:                     # other widths: the same sums in another instruction selection -- rstd within an ulp, y within a rounding flip
        assert torch.equal(m0, m1) and float(((r0 - r1) / r0).abs().max()) < 3e-7 and float((y != y0).float().mean()) < 1e-3
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), None, 1e-5)
    assert relerr(y, ref) < 6e-4
    # lo = y - hi with y as the KERNEL's fp32 arithmetic has it (its own statistics; on the near-constant row 7, rstd ~ 230, fp32 statistics are
    # 2e-5 off the fp64 ones -- 100 x a lo value)
    yk = ((x - m1[:, None]) * r1[:, None] * gamma).double()
    err = _check_mx_planes(P, M, D, y, yk - y.double(), f"layernorm_fwd_mx[{M},{D}]", math.sqrt(D) * float(gamma.abs().max()))
    report(f"layernorm_fwd_mx[{M},{D}]", lo8_rel_err=err)


@pytest.mark.parametrize("F,nseq,Bn,p", [(341, 80, 2, 0.0), (2730, 45, 2, 0.1), (1024, 37, 3, 0.1)])
def test_ffmid_fwd_mx(ops, dev, F, nseq, Bn, p):
    """omlm_ffmid_fwd_mx against omlm_ffmid_fwd_planes on the same planes: h2's hi plane, the statistics, gh and the keep bits are bit for bit
    the same; the fp8 planes hold hi / lo as omlm_gemm_mx16 defines them."""
    M = nseq * Bn
    Fp = (F + 63) // 64 * 64
    dtype = torch.float16
    g = torch.Generator().manual_seed(F + nseq)
    h1 = torch.zeros(M, 2 * Fp)
    h1[:, :F] = torch.randn(M, F, generator=g)
    h1[:, Fp:Fp + F] = torch.randn(M, F, generator=g)
    h1[3] *= 25.0
    convw = (torch.randn(2 * F, 3, generator=g) * 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(F, generator=g)).to(dev)
    h1h, h1l = hilo(h1.to(dev), dtype)
    # the MX form reads h1's lo plane as bf8 (e5m2) bytes -- the upper bytes of halves: the planes run gets the same values as a half plane
    h1l8 = ((h1l.view(torch.int16).to(torch.int32) + 0x80) >> 8).clamp(-128, 127).to(torch.int8).view(torch.uint8)      # (round half up on the magnitude bits: any e5m2 values do)
    h1l = (h1l8.to(torch.int16) << 8).view(torch.float16)
    taps32, g32 = ops.pack_conv_taps(convw, F, Fp), ops.pad_vector(gamma, Fp)
    tph, tpl = hilo(taps32, dtype)
    gph, gpl = hilo(g32, dtype)
    def run(mx):
        h2 = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
        gh = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
        m, r = torch.empty(M, device=dev), torch.empty(M, device=dev)
        bits = torch.zeros(M, Fp // 8, dtype=torch.uint8, device=dev) if p > 0 else None
        if mx:
            P = ops.Fp8Planes(M, Fp, dev, zero=False)
            P.planes.fill_(0x7f); P.scale.fill_(0xff)
            ops.ffmid_fwd_mx(h1h, h1l8, tph, tpl, gph, gpl, h2, P, m, r, nseq, F, Fp, p, 4321, drop_bits=bits, gh=gh)
            return h2, P, gh, m, r, bits
        h2l = torch.full((M, Fp), float("nan"), device=dev, dtype=dtype)
        ops.ffmid_fwd_planes(h1h, h1l, tph, tpl, gph, gpl, h2, h2l, m, r, nseq, F, Fp, p, 4321, drop_bits=bits, gh=gh)
        return h2, h2l, gh, m, r, bits
    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    if p > 0:
        assert torch.equal(a[5], b[5])
    # (gamma reaches the kernel as hi + lo half planes with the dropout scale folded in)
    err = _check_mx_planes(b[1], M, Fp, b[0], a[1], f"ffmid_fwd_mx[{F},{nseq},{p}]", math.sqrt(F) * float((gph.float() + gpl.float()).abs().max()) / (1 - p))
    report(f"ffmid_fwd_mx[{F},{nseq},{p}]", lo8_rel_err=err)


def test_quant_rows_mx(ops, dev):
    """omlm_quant_rows_mx (the per-step fp8 re-pack of the FF weights): rows of an fp32 matrix -> hi8 / lo8 / scale at a row offset of the
    planes, two problems in one launch (the value / gate halves of W1), ragged C."""
    g = torch.Generator().manual_seed(9)
    F, Fp, D = 341, 384, 256
    w1 = (torch.randn(2 * F, D, generator=g) * 0.04).to(dev)
    w1[11] *= 50.0; w1[12] = 0.0
    w2 = (torch.randn(D, F, generator=g) * 0.02).to(dev)
    P1, P2 = ops.Fp8Planes(2 * Fp, D, dev), ops.Fp8Planes(D, Fp, dev)
    grp = ops.QuantRowsGroup()
    grp.add(w1, P1, 0, F, D, D)
    grp.add(w1[F:], P1, Fp, F, D, D)
    grp.add(w2, P2, 0, D, F, F)
    grp.flush()
    for P, rows0, src, K in ((P1, 0, w1[:F], D), (P1, Fp, w1[F:], D), (P2, 0, w2, F)):
        R = src.shape[0]
        hi = src.half()
        lo = src.double() - hi.double()
        e = P.scale[rows0:rows0 + R].cpu().to(torch.int32) - 127
        rowmax = hi.float().abs().amax(1).cpu()
        ok = rowmax > 0
        assert bool((torch.exp2((e + 8).float())[ok] >= rowmax[ok]).all()) and bool((torch.exp2((e + 8).float())[ok] <= 2.0 * rowmax[ok] * (1 + 1e-6)).all())
        assert torch.equal(P.planes[0, rows0:rows0 + R, :K].cpu(), _fp8_bytes(hi, e))
        assert torch.equal(P.planes[1, rows0:rows0 + R, :K].cpu(), _fp8_bytes(lo.float(), e - 11))
        assert float(P.planes[:, rows0:rows0 + R, K:].max()) == 0
    assert float(P1.planes[:, F:Fp].max()) == 0 and int(P1.scale[F]) == 127         # pad rows untouched


def test_cast_pad_group_lo_planes(ops, dev):
    """The grouped weight re-pack with lo set leaves rne16(v - rne16(v)) in the same padded / transposed layouts: hi + lo carries the fp32
    weight to the lo plane's rounding."""
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float16, torch.bfloat16):
        w = (torch.randn(300, 200, generator=g) * 0.03).to(dev)
        hi, lo = torch.zeros(300, 256, device=dev, dtype=dtype), torch.full((300, 256), 7.0, device=dev, dtype=dtype)
        thi, tlo = torch.zeros(200, 304, device=dev, dtype=dtype), torch.zeros(200, 304, device=dev, dtype=dtype)
        grp = ops.CastPadGroup()
        grp.add(w, hi, 300, 200, 200, 256)
        grp.add(w, lo, 300, 200, 200, 256, lo=True)
        grp.add(w, thi, 300, 200, 200, 304, transpose=True)
        grp.add(w, tlo, 300, 200, 200, 304, transpose=True, lo=True)
        grp.flush()
        assert torch.equal(hi[:, :200], w.to(dtype)) and torch.equal(lo[:, :200], (w - w.to(dtype).float()).to(dtype))
        assert float(lo[:, 200:].abs().max()) == 0.0                      # pad columns of a lo plane are zero like the hi plane's
        assert torch.equal(thi[:, :300], w.t().to(dtype)) and torch.equal(tlo[:, :300], (w - w.to(dtype).float()).t().to(dtype))
        e = relerr(hi[:, :200].double() + lo[:, :200].double(), w)
        assert e < (2e-6 if dtype == torch.float16 else 2e-5), e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_gemm_planes16_with_row_maps(ops, dev, dtype):
    """The logit-head form of omlm_gemm_planes16: logical row m reads A's planes at row a_map[m] and lands in C at row c_map[m]
    (open_musiclm.py:163-186: every hidden row is scored by one quantizer head), ragged N = 1025 at pitch 1032, rows beyond the maps untouched."""
    g = torch.Generator().manual_seed(3)
    R, K, N, ldc, M = 2500, 1024, 1025, 1032, 800
    A, B = torch.randn(R, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.05).to(dev)
    Ah, Al = hilo(A, dtype)
    Bh, Bl = hilo(B, dtype)
    a_map = torch.randperm(R, generator=g)[:M].to(torch.int32).to(dev)
    c_map = torch.randperm(1000, generator=g)[:M].to(torch.int32).to(dev)
    C = torch.full((1000, ldc), 7.0, device=dev)
    ops.gemm_planes16(Ah, Al, Bh, Bl, C, M=M, N=N, K=K, a_map=a_map, c_map=c_map, ldc=ldc, a_rows=R, b_rows=N)
    ref = (Ah.double() + Al.double())[a_map.long()] @ (Bh.double() + Bl.double()).t()
    e = relerr(C[c_map.long(), :N], ref)
    untouched = torch.ones(1000, dtype=torch.bool, device=dev)
    untouched[c_map.long()] = False
    report(f"gemm_planes16_maps[{dtype}]", err=e)
    assert e < (3e-6 if dtype == torch.float16 else 6e-5), e
    assert bool((C[untouched] == 7.0).all()) and bool((C[:, N:] == 7.0).all())


@pytest.mark.parametrize("case", ["fp16 B-kmajor 16-bit out", "bf16 fp32 + residual", "fp16 planes -> planes", "fp16 planes -> fp32 + residual"])
def test_gemm_peeled_tail_as_deterministic_split_k(ops, dev, case, monkeypatch):
    """The m-tile rows behind the last full round of 256 x 256 tiles (gemm_impl: tail peeling) as a split-K through the workspace + a fixed-order
    reduction (round 5): equal to the one-launch tail up to the summation order (bar: fp32 accumulation noise; 16-bit outputs equal except where
    that noise crosses a rounding boundary), bit-reproducible from launch to launch, and inside the fp64 bar of the one-launch form."""
    g = torch.Generator().manual_seed(17)
    planes = "planes" in case
    dtype = torch.bfloat16 if case.startswith("bf16") else torch.float16
    if case == "fp16 B-kmajor 16-bit out":
        M, N, K = 17920, 1024, 5504                 # d(xn2)'s form: 280 tiles = 64 m-tile rows + a 1536-row tail, 86 k-tiles
    elif case == "fp16 planes -> planes":
        M, N, K = 3428, 5504, 1024                  # 308 tiles: 11 m-tile rows on the ring + a 612-row tail, 48 loop tiles
    else:
        M, N, K = 17920, 1024, 2752                 # FF-out's form
    A32 = torch.randn(M, K, generator=g).to(dev)
    bk = case == "fp16 B-kmajor 16-bit out"
    B32 = ((torch.randn(K, N, generator=g) if bk else torch.randn(N, K, generator=g)) * 0.05).to(dev)
    A, B = A32.to(dtype), B32.to(dtype)
    Al, Bl = (A32 - A.float()).to(dtype), (B32 - B.float()).to(dtype)
    Cin = torch.randn(M, N, generator=g).to(dev) if "residual" in case else None
    out16 = case in ("fp16 B-kmajor 16-bit out", "fp16 planes -> planes")

    def run():
        C = torch.full((M, N), float("nan"), device=dev, dtype=dtype if out16 else torch.float32)
        Cl = torch.full((M, N), float("nan"), device=dev, dtype=dtype) if case == "fp16 planes -> planes" else None
        if planes:
            ops.gemm_planes16(A, Al, B, Bl, C, Cl, M=M, N=N, K=K, Cin=Cin)
        else:
            ops.gemm(A, B, C, M=M, N=N, K=K, b_kmajor=bk, Cin=Cin)
        return C, Cl

    monkeypatch.setenv("OMLM_GEMM_TAIL_SPLIT", "0")
    one, one_lo = run()
    monkeypatch.setenv("OMLM_GEMM_TAIL_SPLIT", "1")
    two, two_lo = run()
    again, again_lo = run()
    assert torch.equal(two, again) and (two_lo is None or torch.equal(two_lo, again_lo))          # deterministic
    Bm = B.double().t() if not bk else B.double()
    if planes:
        ref = (A.double() + Al.double()) @ ((B.double() + Bl.double()).t())
    else:
        ref = A.double() @ Bm
    if Cin is not None:
        ref = ref + Cin.double()
    val = (lambda c, l: c.double() + l.double()) if two_lo is not None else (lambda c, l: c.double())
    e_one, e_two = relerr(val(one, one_lo), ref), relerr(val(two, two_lo), ref)
    differ = float((one != two).float().mean())
    head = bool(torch.equal(one[:256], two[:256]))                      # the full-round rows never see the tail's path
    report(f"gemm_tail_split[{case}]", one_launch=e_one, split=e_two, differing_frac=differ)
    assert head and e_two <= 1.5 * e_one + 1e-7, (e_one, e_two)
    assert differ < (0.02 if out16 else 0.6)       # fp32 outputs: the last bit moves with the summation order; 16-bit: only across rounding boundaries
    assert differ > 0 or M * N == 0               # (the split form did run: some bit of some tail element differs)


def test_tail_split_gemms_on_two_streams_keep_their_own_scratch(ops, dev):
    """the split-K scratch of a peeled tail is an ARGUMENT of the call (include/omlm.h: workspace / workspace_bytes;
    ops.tail_workspace keeps one buffer per stream), not process-global state.  Two streams run tail-split GEMMs at the same time -- a
    single-product fp16 launch (d(xn2)'s form) on one, the MX launch (FF-in's form, tail slices inside the full-round grid) on the other,
    different operands every burst -- and every result must equal, bit for bit, what the same launch returns on an idle GPU."""
    g = torch.Generator().manual_seed(23)
    M1, N1, K1 = 17920, 1024, 5504
    A1 = torch.randn(M1, K1, generator=g).to(dev).half()
    B1 = (torch.randn(K1, N1, generator=g) * 0.05).to(dev).half()
    M2, N2, K2 = 3200, 5504, 1024
    A2, B2 = torch.randn(M2, K2, generator=g).to(dev), (torch.randn(N2, K2, generator=g) * 0.05).to(dev)
    A2h, A2l = hilo(A2, torch.float16)
    B2h, B2l = hilo(B2, torch.float16)
    A28, _, _ = torch_fp8_planes(ops, A2h, A2l, dev)
    B28, _, _ = torch_fp8_planes(ops, B2h, B2l, dev)

    def run1(out):
        ops.gemm(A1, B1, out, M=M1, N=N1, K=K1, b_kmajor=True)

    def run2(out, lo):
        ops.gemm_mx16(A2h, A28, B2h, B28, out, lo, M=M2, N=N2, K=K2)
    ref1 = torch.empty(M1, N1, device=dev, dtype=torch.float16)
    ref2, ref2l = torch.empty(M2, N2, device=dev, dtype=torch.float16), torch.empty(M2, N2, device=dev, dtype=torch.float16)
    run1(ref1); run2(ref2, ref2l)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs1 = [torch.empty_like(ref1) for _ in range(6)]
    outs2 = [(torch.empty_like(ref2), torch.empty_like(ref2l)) for _ in range(6)]
    bad = 0
    for rep in range(8):
        for o in outs1:
            o.fill_(float("nan"))
        for o, l in outs2:
            o.fill_(float("nan")); l.fill_(float("nan"))
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            for o in outs1:
                run1(o)
        with torch.cuda.stream(s2):
            for o, l in outs2:
                run2(o, l)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref1)) for o in outs1)
        bad += sum(int(not (torch.equal(o, ref2) and torch.equal(l, ref2l))) for o, l in outs2)
    report("gemm_tail_two_streams", bad_launches=bad, launches=8 * 12)
    assert bad == 0, bad


@pytest.mark.parametrize("half", [False, True])
def test_attn_bias_prepare_group_equals_per_layer_launches(ops, dev, half):
    """omlm_attn_bias_prepare_group (round 6): the tables of all layers of a forward in one launch -- every layer's table must equal, bit for bit,
    what omlm_attn_bias_prepare writes for that layer's scales (incl. a layer whose scales are too wide for the fixed reference point)."""
    g = torch.Generator().manual_seed(5)
    N, H, L = 333, 8, 5
    table = torch.zeros(N, 8)
    table[:, :H] = torch.randn(N, H, generator=g) * 0.7
    table = table.to(dev)
    qs = [(0.6 + 0.05 * torch.randn(64, generator=g)).to(dev) for _ in range(L)]      # (|q.k| <= ~0.5: inside half's exponent budget as well)
    ks = [(0.6 + 0.05 * torch.randn(64, generator=g)).to(dev) for _ in range(L)]
    qs[3] = qs[3] * 40.0                                       # beyond the exponent budget: flag 0, online softmax for this layer only
    grp = ops.AttnBias.group(table, N, H, dev, qs, ks, scale=8.0, half=half)
    flags = []
    for l in range(L):
        one = ops.AttnBias(table, N, H, dev, q_scale=qs[l], k_scale=ks[l], scale=8.0, half=half)
        assert torch.equal(one.tableT, grp[l].tableT), l
        ldT = one.tableT.numel() // 8
        flags.append(float(one.tableT.view(8, ldT)[0, ldT - 2]))
    assert flags[3] == 0.0 and flags[0] == 1.0, flags
