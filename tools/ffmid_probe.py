"""Time + cross-check the ConvFeedForward middle kernels at the bench shape (coarse musiclm_small, micro-batch 32).
OMLM_FFMID_FWD=row|strip selects the forward formulation; the probe runs both and compares them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops

dev = torch.device("cuda:0")
B, nseq, F = int(os.environ.get("BATCH", "32")), int(os.environ.get("NSEQ", "1116")), int(os.environ.get("F", "2730"))
Fp = (F + 63) // 64 * 64
M = B * nseq
reps = int(os.environ.get("REPS", "5"))
p = float(os.environ.get("PDROP", "0.1"))
g = torch.Generator().manual_seed(0)
dt = torch.bfloat16 if os.environ.get("DTYPE", "bf16") == "bf16" else torch.float32
h1 = torch.zeros(M, 2 * Fp, dtype=torch.float32)
h1[:, :F] = torch.randn(M, F, generator=g); h1[:, Fp:Fp + F] = torch.randn(M, F, generator=g)
h1 = h1.to(dev).to(dt)
convw = ops.pack_conv_taps((torch.randn(2 * F, 3, generator=g) * 0.5).to(dev), F, Fp).to(dt)
gamma = ops.pad_vector((1.0 + 0.1 * torch.randn(F, generator=g)).to(dev), Fp).to(dt)

def fwd(impl, strips=None):
    os.environ["OMLM_FFMID_FWD"] = impl
    if strips: os.environ["OMLM_FFMID_STRIPS"] = str(strips)
    h2 = torch.full((M, Fp), float("nan"), device=dev, dtype=dt)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    bits = torch.zeros(M, Fp // 8, device=dev, dtype=torch.uint8)
    run = lambda: ops.ffmid_fwd(h1, convw, gamma, h2, mean, rstd, nseq, F, Fp, p, 1234, drop_bits=bits)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return h2, mean, rstd, bits, e0.elapsed_time(e1) * 1e3 / reps

ref = fwd("row")
print(f"row    {ref[4]:8.1f} us", flush=True)
for variant in os.environ.get("VARIANTS", "r3,r2pf,r3pf").split(","):
    os.environ["OMLM_FFMID_STRIP_VARIANT"] = variant
    for strips in [int(x) for x in os.environ.get("STRIPS", "512,1024").split(",")]:
        out = fwd("strip", strips)
        d = (out[0].float() - ref[0].float()).abs()
        nbad = int((~(out[0] == ref[0])).sum())
        print(f"strip {variant:5s}/{strips:5d} {out[4]:8.1f} us | h2 differing {nbad} of {out[0].numel()} (max |d| {float(d.nan_to_num(nan=1e30).max()):.3e}), "
              f"mean rel {float(((out[1] - ref[1]).abs() / (ref[1].abs() + 1e-3)).max()):.2e}, rstd rel {float(((out[2] - ref[2]).abs() / ref[2]).max()):.2e}, "
              f"bits equal {bool((out[3] == ref[3]).all())}", flush=True)
