"""Is the 256x256 GEMM bound by its schedule or by the chip (clock / power / L2)?  Same launch, different operand DATA
(all-zero, constant, random) and the OMLM_GEMM_DEBUG ablations, at 4096^3 (one workgroup per CU, a single round)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
os.environ["OMLM_GEMM_TILE"] = "256x256"
reps = int(os.environ.get("REPS", "20"))
M = N = K = int(os.environ.get("SIZE", "4096"))
g = torch.Generator().manual_seed(0)
C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

def timeit(A, B):
    ops.gemm(A, B, C, M=M, N=N, K=K); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.gemm(A, B, C, M=M, N=N, K=K)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

data = {
    "zeros": (torch.zeros(M, K, device=dev, dtype=torch.bfloat16), torch.zeros(N, K, device=dev, dtype=torch.bfloat16)),
    "ones": (torch.ones(M, K, device=dev, dtype=torch.bfloat16), torch.ones(N, K, device=dev, dtype=torch.bfloat16)),
    "randn": (torch.randn(M, K, generator=g).to(dev).bfloat16(), torch.randn(N, K, generator=g).to(dev).bfloat16()),
}
tf = 2.0 * M * N * K / 1e6
for name, (A, B) in data.items():
    t = timeit(A, B)
    print(f"data={name:6s} {t:8.1f} us {tf / t:7.1f} TF", flush=True)
A, B = data["randn"]
for dbg, label in ((0, "full"), (1, "no DMA (reads+MFMA+epi)"), (3, "no DMA, no MFMA (reads+epi)"), (2, "no MFMA (DMA+reads+epi)"),
                   (8, "DMA + barriers only + epi"), (9, "barriers + epi"), (4, "no epilogue")):
    os.environ["OMLM_GEMM_DEBUG"] = str(dbg)
    t = timeit(A, B)
    print(f"ablate randn {label:30s} {t:8.1f} us", flush=True)
A, B = data["zeros"]
for dbg, label in ((1, "no DMA (reads+MFMA+epi)"),):
    os.environ["OMLM_GEMM_DEBUG"] = str(dbg)
    t = timeit(A, B)
    print(f"ablate zeros {label:30s} {t:8.1f} us", flush=True)
