"""Isolated GEMM probe for rocprofv3 PMC runs: the FF-in / dW1 / dX shapes of a coarse-small micro-batch of 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
M, D, F2 = 35712, 1024, 5472
g = torch.Generator().manual_seed(0)
X = torch.randn(M, D, generator=g).to(dev).bfloat16()
W = (torch.randn(F2, D, generator=g) * 0.03).to(dev).bfloat16()
H = torch.empty(M, F2, device=dev, dtype=torch.bfloat16)
dH = torch.randn(M, F2, generator=g).to(dev).bfloat16()
dX = torch.empty(M, D, device=dev)
dW = torch.zeros(F2, D, device=dev)
reps = int(os.environ.get("REPS", "10"))
shapes = {
    "ffin_NT_bf16out": lambda: ops.gemm(X, W, H, M=M, N=F2, K=D),
    "dX_NN_f32out": lambda: ops.gemm(dH, W, dX, M=M, N=D, K=F2, b_kmajor=True),
    "dW_TN_splitk": lambda: ops.gemm(dH, X, dW, M=F2, N=D, K=M, a_kmajor=True, b_kmajor=True, Cin=dW),
}
sel = os.environ.get("SHAPES", ",".join(shapes)).split(",")
for tile in os.environ.get("TILES", "128x128,256x256,256x128").split(","):
    os.environ["OMLM_GEMM_TILE"] = tile
    for name in sel:
        fn = shapes[name]
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print(f"{tile:8s} {name:18s} {us:8.1f} us  {2.0 * M * F2 * D / us / 1e6:7.1f} TFLOP/s", flush=True)
