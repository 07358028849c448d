"""omlm_qk_norm_bwd2 at the bench shape (M = 35 712 rows, 8 heads), HIP events; OMLM_LIB_PATH selects the library (A/B of kernel forms)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops, hip

dev = torch.device("cuda:0")
M, H = int(os.environ.get("MROWS", "35712")), 8
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
q = torch.nn.functional.normalize(torch.randn(M, H, 64, generator=g), dim=-1).reshape(M, H * 64).to(dev).to(bf)
k = torch.nn.functional.normalize(torch.randn(M, 64, generator=g), dim=-1).to(dev).to(bf)
dq, dk, dv = torch.randn(M, H * 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev), torch.randn(M, 64, generator=g).to(dev)
qn, kn = (torch.rand(M, H, generator=g) + 0.5).to(dev), (torch.rand(M, generator=g) + 0.5).to(dev)
qs, ks = torch.ones(64, device=dev), torch.ones(64, device=dev)
dq_raw, dkv_raw = torch.empty(M, H * 64, device=dev, dtype=bf), torch.empty(M, 128, device=dev, dtype=bf)
dqs, dks = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
fn = lambda: ops.qk_norm_bwd2(dq, dk, dv, q, k, qn, kn, qs, ks, dq_raw, dkv_raw, dqs, dks, H)
fn(); fn(); torch.cuda.synchronize()
reps = int(os.environ.get("REPS", "30"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
byts = M * (H * 64 * (4 + 2 + 2) + 64 * (4 + 4 + 2) + 128 * 2 + (H + 1) * 4)
print(f"{os.path.basename(hip.LIB_PATH)}: qk_norm_bwd2 {us:6.1f} us ({byts / us / 1e3:5.0f} GB/s of {byts / 1e6:.0f} MB) | sum|dq_raw| {float(dq_raw.float().abs().sum()):.6e} "
      f"sum|dkv_raw| {float(dkv_raw.float().abs().sum()):.6e} dqs {float(dqs.sum()) / (reps + 2):.5e} dks {float(dks.sum()) / (reps + 2):.5e}")
