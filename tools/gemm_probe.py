"""Isolated GEMM probe for rocprofv3 PMC runs: the FF-in / dW1 / dX shapes of a coarse-small micro-batch of 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
M, D, F2 = int(os.environ.get("MROWS", "35712")), 1024, 5472
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
if os.environ.get("EXTRA", "0") == "1":      # layout experiments: same contractions with operands stored the other way round
    W1T = W.t().contiguous()                 # [D, F2]
    XT = X.t().contiguous()                  # [D, M]
    dHT = dH.t().contiguous()                # [F2, M]
    shapes["dX_NT_longK"] = lambda: ops.gemm(dH, W1T, dX, M=M, N=D, K=F2)
    shapes["dW_TN_Bnormal"] = lambda: ops.gemm(dH, XT, dW, M=F2, N=D, K=M, a_kmajor=True, Cin=dW)
    shapes["dW_NT_splitk"] = lambda: ops.gemm(dHT, XT, dW, M=F2, N=D, K=M, Cin=dW)
if os.environ.get("PITCHK", "0") != "0":     # k-major operands with a padded row pitch
    pad = int(os.environ["PITCHK"])
    Wp2 = torch.zeros(F2, D + pad, device=dev, dtype=torch.bfloat16); Wp2[:, :D] = W
    Xp2 = torch.zeros(M, D + pad, device=dev, dtype=torch.bfloat16); Xp2[:, :D] = X
    shapes["dX_NN_padB"] = lambda: ops.gemm(dH, Wp2, dX, M=M, N=D, K=F2, b_kmajor=True, ldb=D + pad)
    shapes["dW_TN_padB"] = lambda: ops.gemm(dH, Xp2, dW, M=F2, N=D, K=M, a_kmajor=True, b_kmajor=True, Cin=dW, ldb=D + pad)
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

# ---- A/B: same FF-in GEMM with padded row pitch (tests the power-of-two-stride L2 channel-conflict hypothesis)
if os.environ.get("PITCH_AB", "1") == "1":
    for pad in (0, 8, 32, 64, 128):
        Xp = torch.zeros(M, D + pad, device=dev, dtype=torch.bfloat16); Xp[:, :D] = X
        Wp = torch.zeros(F2, D + pad, device=dev, dtype=torch.bfloat16); Wp[:, :D] = W
        for tile in ("128x128", "256x256"):
            os.environ["OMLM_GEMM_TILE"] = tile
            fn = lambda: ops.gemm(Xp, Wp, H, M=M, N=F2, K=D, lda=D + pad, ldb=D + pad)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            print(f"pitch {D + pad:5d} {tile:8s} ffin_NT {us:8.1f} us  {2.0 * M * F2 * D / us / 1e6:7.1f} TFLOP/s", flush=True)

if os.environ.get("ABLATE", "1") == "1":
    for name in os.environ.get("ABLATE_SHAPES", "ffin_NT_bf16out").split(","):
        for tile in os.environ.get("ABLATE_TILES", "128x128,256x256").split(","):
            os.environ["OMLM_GEMM_TILE"] = tile
            for dbg, label in ((0, "full"), (3, "no-DMA no-MFMA"), (4, "no-epilogue"), (7, "no-DMA/MFMA/epi"), (15, "barriers only"),
                               (12, "DMA+barrier only"), (2, "no-MFMA"), (1, "no-DMA"), (12 + 32, "A-DMA only"), (12 + 16, "B-DMA only"), (64, "aux1"), (128, "aux2"), (192, "aux3")):
                os.environ["OMLM_GEMM_DEBUG"] = str(dbg)
                fn = shapes[name]
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): fn()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                print(f"ablate {name:16s} {tile:8s} {label:16s} {us:8.1f} us", flush=True)
    os.environ["OMLM_GEMM_DEBUG"] = "0"
