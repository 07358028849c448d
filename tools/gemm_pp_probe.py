"""Time and CHECK the large GEMM shapes of a coarse-small train step.  (Written for the A/B of the two-group ping-pong k-loop,
profiles/r02_gemm_pingpong_ab.md -- that loop was slower and is gone; $OMLM_GEMM_PP is only echoed now.)  Each shape is verified against a torch bf16 matmul (fp32 accumulate)
on several seeds -- a race in the LDS staging shows up as a wrong tile -- then timed with HIP events.
Run twice:  OMLM_GEMM_PP=0 python tools/gemm_pp_probe.py ; OMLM_GEMM_PP=1 python tools/gemm_pp_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
M, D, F2, Fp = int(os.environ.get("MROWS", "35712")), 1024, 5504, 2752
reps = int(os.environ.get("REPS", "10"))
nchk = int(os.environ.get("CHECKS", "3"))


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev).bfloat16()


def relerr(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def case(name, make, flops):
    worst = 0.0
    for sd in range(nchk):
        fn, check = make(sd)
        fn(); torch.cuda.synchronize()
        worst = max(worst, check())
    fn, _ = make(0)
    us = timed(fn)
    print(f"PP={os.environ.get('OMLM_GEMM_PP', 'default')}  {name:26s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   max rel err {worst:.2e}"
          f"{'   <-- WRONG' if worst > 2e-2 else ''}", flush=True)


def ffin(sd):
    X, W = rnd(M, D, seed=sd), rnd(F2, D, seed=100 + sd, scale=0.03)
    H = torch.empty(M, F2, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.gemm(X, W, H, M=M, N=F2, K=D)), (lambda: relerr(H, X.float() @ W.float().t()))


def ffout(sd):
    Hh, W, R = rnd(M, Fp, seed=sd), rnd(D, Fp, seed=100 + sd, scale=0.03), rnd(M, D, seed=200 + sd).float()
    O = torch.empty(M, D, device=dev)
    return (lambda: ops.gemm(Hh, W, O, M=M, N=D, K=Fp, Cin=R)), (lambda: relerr(O, Hh.float() @ W.float().t() + R))


def dx_ffin(sd):
    dH, W = rnd(M, F2, seed=sd), rnd(F2, D, seed=100 + sd, scale=0.03)
    dX = torch.empty(M, D, device=dev)
    return (lambda: ops.gemm(dH, W, dX, M=M, N=D, K=F2, b_kmajor=True)), (lambda: relerr(dX, dH.float() @ W.float()))


def dx_ffout(sd):
    dR, W = rnd(M, D, seed=sd), rnd(D, Fp, seed=100 + sd, scale=0.03)
    dH = torch.empty(M, Fp, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.gemm(dR, W, dH, M=M, N=Fp, K=D, b_kmajor=True)), (lambda: relerr(dH, dR.float() @ W.float()))


def wgrad_layer(sd):
    dH, X = rnd(M, F2, seed=sd), rnd(M, D, seed=100 + sd)
    dR, Hh = rnd(M, D, seed=200 + sd), rnd(M, Fp, seed=300 + sd)
    dW1, dW2 = torch.zeros(F2, D, device=dev), torch.zeros(D, Fp, device=dev)

    def run():
        dW1.zero_(); dW2.zero_()
        wg = ops.WgradGroup()
        wg.add(dH, X, dW1, M=F2, N=D, K=M)
        wg.add(dR, Hh, dW2, M=D, N=Fp, K=M)
        wg.flush()
    return run, (lambda: max(relerr(dW1, dH.float().t() @ X.float()), relerr(dW2, dR.float().t() @ Hh.float())))


case("FF-in  NT bf16 out K=1024", ffin, 2.0 * M * F2 * D)
case("FF-out NT f32+Cin K=2752", ffout, 2.0 * M * D * Fp)
case("dX FF-in  NN f32  K=5504", dx_ffin, 2.0 * M * D * F2)
case("dX FF-out NN bf16 K=1024", dx_ffout, 2.0 * M * Fp * D)
case("dW1+dW2 grouped   K=35712", wgrad_layer, 2.0 * M * D * (F2 + Fp))
