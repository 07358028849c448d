"""Persistent walk of the 256x256 GEMM kernel against its one-tile grid (OMLM_GEMM_PERSIST=0 / 1 toggled inside one process): the wide GEMMs of
a coarse-small micro-batch of 32 (M = 35 712 rows), each timed alone, both ways, twice (a-b-a-b).

    python tools/gemm_persist_probe.py [out.md]
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
rnd = lambda r, c, dt=bf: (torch.randn(r, c, generator=g) * 0.05).to(dev).to(dt)
# name, N, K, output dtype, residual, b_kmajor
SHAPES = [("FF-in    -> bf16", 5504, 1024, bf, False, False),
          ("d(h2)    dres W2 (B k-major)", 2752, 1024, bf, False, True),
          ("d(xn2)   dh1 W1 (B k-major)", 1024, 5504, bf, False, True),
          ("FF-out   x2 = x1 + h2 W2^T", 1024, 2752, torch.float32, True, False),
          # 128x128 tiles (two workgroups per CU)
          ("q-proj   -> bf16 (plain epilogue)", 512, 1024, bf, False, False),
          ("q-proj   -> bf16 + l2-norm epilogue", 512, 1024, "qknorm", False, False),
          ("d(o)     dx1 Wo (B k-major)", 512, 1024, bf, False, True),
          ("to_out   x1 = x + o Wo^T", 1024, 512, torch.float32, True, False),
          ("d(xn)    dq Wq (B k-major)", 1024, 512, bf, False, True),
          ("d(x) kv  dkv Wkv (B k-major)", 1024, 128, bf, False, True)]
# the last three: "one-tile grid" column = OMLM_GEMM_PERSIST=0 (wide tiles, the host's choice without the walk), "persistent" = the default
reps = int(os.environ.get("REPS", "20"))
lines = ["| GEMM (M = %d) | N | K | one-tile grid us (TFLOP/s) | persistent walk us (TFLOP/s) | ratio |" % M, "|---|---:|---:|---:|---:|---:|"]
for name, N, K, odt, resid, bk in SHAPES:
    A = rnd(M, K)
    B = rnd(K, N) if bk else rnd(N, K)
    qk = odt == "qknorm"
    if qk:
        odt = bf
        scale, norms = torch.rand(64, device=dev) + 0.5, torch.empty(M, N // 64, device=dev)
    C = torch.empty(M, N, dtype=odt, device=dev)
    Cin = rnd(M, N, torch.float32) if resid else None
    os.environ["OMLM_GEMM_TILE"] = "128x128" if "forced 128x128" in name else ""
    t = {"0": [], "1": []}
    outs = {}
    for rnd_i in range(2):
        for mode in ("0", "1"):
            os.environ["OMLM_GEMM_PERSIST"] = mode
            if qk:
                fn = lambda: ops.gemm_qknorm(A, B, C, scale, norms, N // 64, M=M, N=N, K=K)
            else:
                fn = lambda: ops.gemm(A, B, C, M=M, N=N, K=K, b_kmajor=bk, Cin=Cin)
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) * 1e3 / reps)
            outs[mode] = C.clone()
    same = bool(torch.equal(outs["0"], outs["1"]))
    fl = 2.0 * M * N * K
    a, b = min(t["0"]), min(t["1"])
    print(name, {k: [round(x, 1) for x in v] for k, v in t.items()}, "bit-equal" if same else "DIFFERENT", flush=True)
    lines.append(f"| {name} | {N} | {K} | {a:.1f} ({fl / a / 1e6:.0f}) | {b:.1f} ({fl / b / 1e6:.0f}) | {b / a:.3f} |" + ("" if same else " DIFFERENT RESULTS"))
    del A, B, C, Cin
os.environ.pop("OMLM_GEMM_PERSIST", None)
os.environ.pop("OMLM_GEMM_TILE", None)
text = "\n".join(lines)
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(text + "\n")
