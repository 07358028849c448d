"""Timing probe (GPU box): the FF-in / FF-out forward GEMMs of precision fp16ff as one half product (omlm_gemm), three half products
(omlm_gemm_planes16) and half + two fp8 corrections (omlm_gemm_mx16), same operands, interleaved rounds.  python tools/mx_gemm_probe.py [B]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
M, D, Fp = B * 1116, 1024, 2752
g = torch.Generator(device=dev).manual_seed(1)


def planes(rows, K, scale):
    x = torch.randn(rows, K, device=dev, generator=g) * scale
    hi = x.half(); lo = (x - hi.float()).half()
    P = ops.Fp8Planes(rows, K, dev)
    b = torch.randint(0, 0x70, (2, rows, K), device=dev, generator=g, dtype=torch.uint8) | (torch.randint(0, 2, (2, rows, K), device=dev, generator=g, dtype=torch.uint8) << 7)
    P.planes[:, :rows, :K] = b
    P.scale[:rows] = 120
    return hi, lo, P


def timed(fn, reps=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(reps):
        ev[0].record(); fn(); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
    return statistics.median(ts)


for name, N, K, out_planes in (("FF-in", 2 * Fp, D, True), ("FF-out", D, Fp, False)):
    Ah, Al, A8 = planes(M, K, 1.0)
    Bh, Bl, B8 = planes(N, K, 0.03)
    if out_planes:
        C, Cl = torch.empty(M, N, device=dev, dtype=torch.float16), torch.empty(M, N, device=dev, dtype=torch.float16)
        f1 = lambda: ops.gemm(Ah, Bh, C, M=M, N=N, K=K)
        f3 = lambda: ops.gemm_planes16(Ah, Al, Bh, Bl, C, Cl, M=M, N=N, K=K)
        fx = lambda: ops.gemm_mx16(Ah, A8, Bh, B8, C, Cl, M=M, N=N, K=K)
    else:
        C, Cin = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
        f1 = lambda: ops.gemm(Ah, Bh, C, M=M, N=N, K=K, Cin=Cin)
        f3 = lambda: ops.gemm_planes16(Ah, Al, Bh, Bl, C, M=M, N=N, K=K, Cin=Cin)
        fx = lambda: ops.gemm_mx16(Ah, A8, Bh, B8, C, M=M, N=N, K=K, Cin=Cin)
    for f in (f1, f3, fx):
        f()
    torch.cuda.synchronize()
    r = {"one": [], "three": [], "mx": []}
    for _ in range(4):
        r["one"].append(timed(f1)); r["three"].append(timed(f3)); r["mx"].append(timed(fx))
    fl = 2.0 * M * N * K
    print(f"{name} M={M} N={N} K={K}: " + ", ".join(f"{k} {statistics.median(v):.0f} us ({fl / statistics.median(v) * 1e-6:.0f} TFLOP/s algorithmic)" for k, v in r.items()), flush=True)
