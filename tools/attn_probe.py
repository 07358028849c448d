"""Attention kernel timing probe (coarse-small micro-batch: B=32, N=1116, H=8; LARGE=1: fine-large B=8, N=1817, H=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops
dev = torch.device("cuda:0")
large = os.environ.get("LARGE") == "1"
B, N, H = (8, 1817, 16) if large else (int(os.environ.get("B", 32)), 1116, 8)
M = B * N
g = torch.Generator().manual_seed(0)
unit = lambda t: torch.nn.functional.normalize(t, dim=-1)
DT = torch.float16 if os.environ.get("DTYPE", "fp16") == "fp16" else torch.bfloat16
q = unit(torch.randn(B, N, H, 64, generator=g)).reshape(M, H * 64).to(dev).to(DT)
k = unit(torch.randn(M, 64, generator=g)).to(dev).to(DT)
v = torch.randn(M, 64, generator=g).to(dev).to(DT)
ld = (H + 7) // 8 * 8
bias = torch.zeros(N, ld); bias[:, :H] = torch.randn(N, H, generator=g) * 0.1; bias = bias.to(dev)
mask = (torch.rand(B, N, generator=g) > 0.15).to(torch.uint8).to(dev); mask[:, 0] = 1
out = torch.empty_like(q); lse = torch.empty(B, H, N, device=dev)
dout = torch.randn(M, H * 64, generator=g).to(dev).to(DT)
delta = torch.empty(B, H, N, device=dev)
dq = torch.empty(M, H * 64, device=dev); dk = torch.empty(M, 64, device=dev); dv = torch.empty(M, 64, device=dev)
dbias = torch.zeros(N, ld, device=dev)
ab = ops.AttnBias(bias, N, H, dev, qk_bound=float(os.environ.get('QKB', '1.0')), half=DT == torch.float16)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
flops = 4.0 * H * 64 * N * (N + 1) / 2 * B
tf = t(lambda: ops.attn_fwd(q, k, v, ab, mask, out, lse, B, N, H, 8.0))
print(f"fwd            {tf:8.1f} us  {flops / tf / 1e6:7.1f} TFLOP/s   checksum {float(out.float().abs().sum()):.6e} lse {float(lse.sum()):.6e}")
if os.environ.get("FWD_ONLY") != "1":
    tb = t(lambda: ops.attn_bwd(q, k, v, ab, mask, out, dout, lse, delta, dq, dk, dv, dbias, B, N, H, 8.0))
    print(f"bwd (dq+dkv)   {tb:8.1f} us  {2.5 * flops / tb / 1e6:7.1f} TFLOP/s (5 matmuls counted)")
    print(f"bwd no dbias   {t(lambda: ops.attn_bwd(q, k, v, ab, mask, out, dout, lse, delta, dq, dk, dv, None, B, N, H, 8.0)):8.1f} us")
    print(f"bwd no mask    {t(lambda: ops.attn_bwd(q, k, v, ab, None, out, dout, lse, delta, dq, dk, dv, dbias, B, N, H, 8.0)):8.1f} us")
