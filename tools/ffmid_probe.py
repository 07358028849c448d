"""Time the ConvFeedForward middle forward at the bench shape (coarse musiclm_small, micro-batch 32) and print checksums, so
that two builds of the library (OMLM_LIB_PATH) can be compared from two runs."""
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

def fwd():
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

h2, mean, rstd, bits, t = fwd()
print(f"ffmid_fwd {t:8.1f} us | sum h2 {float(h2.float().sum()):.6e} sum|h2| {float(h2.float().abs().sum()):.6e} "
      f"sum mean {float(mean.sum()):.6e} sum rstd {float(rstd.sum()):.6e} bits checksum {int(bits.long().sum())}", flush=True)

if os.environ.get("RUN_TESTS") == "1":      # the kernel-level parity tests against the same library build, same process
    import pytest
    del h1, h2, bits
    torch.cuda.empty_cache()
    rc = pytest.main(["-q", "-x", "-m", "gpu", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_kernels.py"), "-k", "ffmid"])
    print(f"pytest -k ffmid exit code {int(rc)}", flush=True)
