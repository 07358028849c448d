"""Attention kernel timing probe (coarse-small micro-batch: B=32, N=1116, H=8)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import ops
dev = torch.device("cuda:0")
B, N, H = int(os.environ.get("B", 32)), 1116, 8
M = B * N
g = torch.Generator().manual_seed(0)
q = torch.randn(M, H * 64, generator=g).to(dev).bfloat16()
k = torch.randn(M, 64, generator=g).to(dev).bfloat16()
v = torch.randn(M, 64, generator=g).to(dev).bfloat16()
bias = (torch.randn(N, 8, generator=g) * 0.1).to(dev)
mask = (torch.rand(B, N, generator=g) > 0.15).to(torch.uint8).to(dev); mask[:, 0] = 1
out = torch.empty_like(q); lse = torch.empty(B, H, N, device=dev)
dout = torch.randn(M, H * 64, generator=g).to(dev).bfloat16()
delta = torch.empty(B, H, N, device=dev)
dq = torch.empty(M, H * 64, device=dev); dk = torch.empty(M, 64, device=dev); dv = torch.empty(M, 64, device=dev)
dbias = torch.zeros(N, 8, device=dev)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"fwd            {t(lambda: ops.attn_fwd(q, k, v, bias, mask, out, lse, B, N, H, 8.0)):8.1f} us")
print(f"bwd (dq+dkv)   {t(lambda: ops.attn_bwd(q, k, v, bias, mask, out, dout, lse, delta, dq, dk, dv, dbias, B, N, H, 8.0)):8.1f} us")
print(f"bwd no dbias   {t(lambda: ops.attn_bwd(q, k, v, bias, mask, out, dout, lse, delta, dq, dk, dv, None, B, N, H, 8.0)):8.1f} us")
print(f"bwd no mask    {t(lambda: ops.attn_bwd(q, k, v, bias, None, out, dout, lse, delta, dq, dk, dv, dbias, B, N, H, 8.0)):8.1f} us")
