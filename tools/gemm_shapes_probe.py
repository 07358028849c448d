"""Per-shape GEMM probe: every distinct GEMM of a coarse-small micro-batch of 32 (M = 35 712 rows), timed alone under each forced tile
(OMLM_GEMM_TILE) and under the host's own choice.  Output: a markdown table (profiles/*_gemm_tile_probe.md).

    python tools/gemm_shapes_probe.py [out.md]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
M = int(os.environ.get("MROWS", "35712"))
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16


def rnd(r, c, dt=bf):
    return (torch.randn(r, c, generator=g) * 0.05).to(dev).to(dt)


# name, (N, K), output dtype, residual, b_kmajor
SHAPES = [("to_out   x1 = x + o Wo^T", 1024, 512, torch.float32, True, False),
          ("FF-out   x2 = x1 + h2 W2^T", 1024, 2752, torch.float32, True, False),
          ("q-proj   -> f32", 512, 1024, torch.float32, False, False),
          ("q-proj   -> bf16", 512, 1024, bf, False, False),
          ("kv-proj  -> f32", 128, 1024, torch.float32, False, False),
          ("kv-proj  -> bf16", 128, 1024, bf, False, False),
          ("d(o)     dx1 Wo (B k-major)", 512, 1024, bf, False, True),
          ("d(xn)    dq Wq (B k-major)", 1024, 512, bf, False, True),
          ("d(x) kv  dkv Wkv (B k-major)", 1024, 128, bf, False, True),
          ("FF-in    -> bf16", 5504, 1024, bf, False, False),
          ("d(h2)    dres W2 (B k-major)", 2752, 1024, bf, False, True),
          ("d(xn2)   dh1 W1 (B k-major)", 1024, 5504, bf, False, True)]
reps = int(os.environ.get("REPS", "8"))
rows = []
for name, N, K, odt, resid, bk in SHAPES:
    A = rnd(M, K)
    B = rnd(K, N) if bk else rnd(N, K)
    C = torch.empty(M, N, dtype=odt, device=dev)
    Cin = rnd(M, N, torch.float32) if resid else None
    res = {}
    for tile in ("", "128x128", "256x128", "256x256"):
        os.environ["OMLM_GEMM_TILE"] = tile
        fn = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, b_kmajor=bk, Cin=Cin)
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[tile or "host choice"] = e0.elapsed_time(e1) * 1e3 / reps
    fl = 2.0 * M * N * K
    byts = M * K * 2 + N * K * 2 + M * N * C.element_size() + (M * N * 4 if resid else 0)
    rows.append((name, N, K, fl, byts, res))
    print(name, {k: round(v, 1) for k, v in res.items()}, flush=True)
    del A, B, C, Cin
os.environ["OMLM_GEMM_TILE"] = ""
out = sys.argv[1] if len(sys.argv) > 1 else None
lines = [f"GEMM shapes of one coarse-small micro-batch (M = {M} rows), each timed alone ({reps} launches), us per launch (TFLOP/s)", "",
         "| GEMM | N | K | algorithmic MB | HBM floor us (5.5 TB/s) | host choice | 128x128 | 256x128 | 256x256 |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
for name, N, K, fl, byts, res in rows:
    cells = " | ".join(f"{res[k]:.1f} ({fl / res[k] / 1e6:.0f})" for k in ("host choice", "128x128", "256x128", "256x256"))
    lines.append(f"| {name} | {N} | {K} | {byts / 1e6:.0f} | {byts / 5.5e6:.0f} | {cells} |")
txt = "\n".join(lines) + "\n"
print(txt)
if out:
    open(out, "w").write(txt)
