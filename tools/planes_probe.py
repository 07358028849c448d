"""The step's two plane GEMMs (precision fp16ff: FF-in with plane output, FF-out with fp32 + residual) and their single-product forms, 10 launches
each -- the workload of tools/pmc_planes.sh (SQ counters of the 3-product half-tile-ring kernel next to the single-product kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
T = torch.float16
g = torch.Generator().manual_seed(0)


def planes(M, N, K, planes_out):
    A32, B32 = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.03).to(dev)
    A, B = A32.to(T), B32.to(T)
    Al, Bl = (A32 - A.float()).to(T), (B32 - B.float()).to(T)
    C = torch.empty(M, N, device=dev, dtype=T if planes_out else torch.float32)
    Cl = torch.empty(M, N, device=dev, dtype=T) if planes_out else None
    Cin = None if planes_out else torch.randn(M, N, generator=g).to(dev)
    for _ in range(10):
        ops.gemm_planes16(A, Al, B, Bl, C, Cl, M=M, N=N, K=K, Cin=Cin)
    for _ in range(10):
        ops.gemm(A, B, C, M=M, N=N, K=K, Cin=Cin)
    torch.cuda.synchronize()


planes(35712, 5504, 1024, True)
planes(35712, 1024, 2752, False)
