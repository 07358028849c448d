"""Split-K sweep of the weight-gradient GEMM shapes of a coarse-small micro-batch (K = 35712 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops
dev = torch.device("cuda:0")
K = 35712
g = torch.Generator().manual_seed(0)
def t(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, M, N in (("dW1", 5504, 1024), ("dW2", 1024, 2752), ("dWq", 512, 1024), ("dWo", 1024, 512), ("dWkv", 128, 1024)):
    A = torch.randn(K, M, generator=g).to(dev).bfloat16()
    B = torch.randn(K, N, generator=g).to(dev).bfloat16()
    C = torch.zeros(M, N, device=dev)
    res = []
    for sp in [0] + [int(x) for x in os.environ.get("SPLITS", "2,3,4,5,6,8,11,12,16,24,32").split(",")]:
        if sp: os.environ["OMLM_GEMM_SPLITS"] = str(sp)
        else: os.environ.pop("OMLM_GEMM_SPLITS", None)
        us = t(lambda: ops.gemm(A, B, C, M=M, N=N, K=K, a_kmajor=True, b_kmajor=True, Cin=C))
        res.append(f"{'auto' if not sp else sp}:{us:.0f}")
    print(f"{name} M={M} N={N}: " + "  ".join(res) + f"   (auto = {2.0 * M * N * K / float(res[0].split(':')[1]) / 1e6:.0f} TFLOP/s)", flush=True)
