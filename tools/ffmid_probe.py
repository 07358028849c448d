"""Time the ConvFeedForward middle (forward, backward) at the bench shape (coarse musiclm_small, micro-batch 32) for both kernel
generations in ONE process (ops.ffmid_set_impl), with HIP events, and print achieved HBM GB/s against the algorithmic bytes."""
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
dh2 = torch.zeros(M, Fp)
dh2[:, :F] = torch.randn(M, F, generator=g)
dh2 = dh2.to(dev).to(dt)
esz = 2 if dt == torch.bfloat16 else 4


def timed(run):
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for impl in ([1, 0] if dt == torch.bfloat16 else [0]):
    ops.ffmid_set_impl(impl)
    h2 = torch.full((M, Fp), float("nan"), device=dev, dtype=dt)
    gh = torch.empty(M, Fp, device=dev, dtype=dt)
    mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
    bits = torch.zeros(M, Fp // 8, device=dev, dtype=torch.uint8)
    t_f = timed(lambda: ops.ffmid_fwd(h1, convw, gamma, h2, mean, rstd, nseq, F, Fp, p, 1234, drop_bits=bits, gh=gh))
    fwd_bytes = M * (2 * Fp + Fp + Fp) * esz + M * Fp // 8
    du = torch.empty(M, 2 * Fp, device=dev, dtype=dt)
    dh1 = torch.empty(M, 2 * Fp, device=dev, dtype=dt)
    dgamma, dconv = torch.zeros(F, device=dev), torch.zeros(2 * F * 3, device=dev)
    ws = torch.empty(ops.ffmid_bwd_workspace_floats(F, Fp), device=dev)
    t_b = timed(lambda: ops.ffmid_bwd(dh2, h1, convw, gamma, mean, rstd, du, dh1, dgamma, dconv, ws, nseq, F, Fp, p, 1234,
                                      drop_bits=bits, gh=gh))
    bwd_bytes = M * (Fp + Fp + 2 * Fp + 2 * Fp) * esz + M * Fp // 8       # dh2, gh, h1 in; dh1 out (no du, no second reads)
    print(f"impl {impl}: ffmid_fwd {t_f:7.1f} us ({fwd_bytes / t_f / 1e3:6.0f} GB/s of {fwd_bytes / 1e6:.0f} MB) | "
          f"ffmid_bwd {t_b:7.1f} us ({bwd_bytes / t_b / 1e3:6.0f} GB/s of {bwd_bytes / 1e6:.0f} MB) | "
          f"sum|h2| {float(h2.float().abs().sum()):.5e} sum|dh1| {float(dh1.float().abs().sum()):.5e} "
          f"sum|dgamma| {float(dgamma.abs().sum()) / reps:.5e} drop {1 - float((h2[:, :F] != 0).float().mean()):.4f}", flush=True)
ops.ffmid_set_impl(1)
