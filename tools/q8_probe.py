"""A/B of experimental 256x256 GEMM schedules against the production 256x256 tile kernel (8 waves of 128x64):

  q8 (OMLM_GEMM_Q8=1): quadrant-phased, two wave groups one slot apart, counted vmcnt (NT layouts only)
  w4 (OMLM_GEMM_W4=1): 4 waves of 128x128, accumulators in AGPRs (1/3 fewer LDS fragment bytes per flop)

Same operands through all kernels: the k order per accumulator is identical, so the results must be BIT-equal.
Each shape prints mismatches and the time of each variant.  Run under `timeout`: a barrier-count bug would hang."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
reps = int(os.environ.get("REPS", "10"))
os.environ["OMLM_GEMM_TILE"] = "256x256"
VARIANTS = [v for v in os.environ.get("VARIANTS", "tile,q8,w4").split(",")]

def run(fn, variant):
    os.environ["OMLM_GEMM_Q8"] = "1" if variant == "q8" else "0"
    os.environ["OMLM_GEMM_W4"] = "1" if variant == "w4" else "0"
    C = fn(None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn(C)
    e1.record(); torch.cuda.synchronize()
    return C, e0.elapsed_time(e1) * 1e3 / reps

# (M, N, K, out dtype, a_kmajor, b_kmajor, accumulate)
cases = [(300, 520, 64, torch.float32, 0, 0, 0), (300, 520, 128, torch.bfloat16, 0, 0, 0), (1000, 1000, 200, torch.float32, 0, 0, 0),
         (1000, 1000, 200, torch.float32, 1, 1, 0), (1000, 1000, 200, torch.bfloat16, 0, 1, 0),
         (4096, 4096, 4096, torch.bfloat16, 0, 0, 0), (35712, 5472, 1024, torch.bfloat16, 0, 0, 0), (35712, 1024, 5472, torch.float32, 0, 0, 0),
         (35712, 1024, 5472, torch.float32, 0, 1, 0), (5472, 1024, 35712, torch.float32, 1, 1, 1), (35712, 2752, 1024, torch.bfloat16, 0, 0, 0)]
if os.environ.get("SMALL_ONLY", "0") == "1": cases = cases[:5]
for (M, N, K, od, ak, bk, accum) in cases:
    A = torch.randn((K, M) if ak else (M, K), generator=g).to(dev).bfloat16()
    B = (torch.randn((K, N) if bk else (N, K), generator=g) * 0.05).to(dev).bfloat16()
    def fn(C, A=A, B=B):
        if C is None: C = torch.full((M, N), float("nan"), device=dev, dtype=od)
        if accum:
            C.zero_()
            ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=bool(ak), b_kmajor=bool(bk), Cin=C)
        else:
            ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=bool(ak), b_kmajor=bool(bk))
        return C
    tf = 2.0 * M * N * K / 1e6
    ref = None
    line = f"M={M:6d} N={N:5d} K={K:5d} {str(od)[6:]:8s} {'T' if ak else 'N'}{'N' if bk else 'T'}{'+' if accum else ' '}"
    for v in VARIANTS:
        if v == "q8" and (ak or bk or accum): continue
        C, t = run(fn, v)
        if ref is None: ref = C.clone(); bad = 0
        elif accum: bad = int(((C - ref).abs() > 1e-3 * ref.abs().max()).sum())     # split-K: atomic order differs
        else: bad = int((~(C == ref)).sum())          # NaN-aware: an unwritten (NaN) element counts as different
        line += f" | {v} {t:7.1f} us {tf / t:6.1f} TF bad={bad}"
    print(line, flush=True)
