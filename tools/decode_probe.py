"""Decode-step timing probe: coarse-small, B from env (default 1), context rows from env CTX (default 216)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M, decode
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 1)); prec = os.environ.get("PREC", "bf16")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, precision=prec).to(dev).eval()
g = torch.Generator().manual_seed(1)
clap = torch.randint(0, 1024, (B, 13), generator=g).to(dev)
sem = torch.randint(0, 1024, (B, 200), generator=g).to(dev)
primed = int(os.environ.get("PRIMED", 0))
coarse = torch.randint(0, 1024, (B, primed), generator=g).to(dev)
steps = int(os.environ.get("STEPS", 200))
with torch.no_grad():
    dec = decode.CachedDecoder(model, B, 13 + 200 + 3 + primed + steps + 4, prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dec.prefill([clap, sem, coarse]); torch.cuda.synchronize()
    print(f"prefill {1e3 * (time.perf_counter() - t0):.2f} ms (rows {dec.rows})")
    ids = torch.randint(0, 1024, (B,), generator=g).to(dev)
    for k in range(5): dec.step(ids, primed + k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps - 5): dec.step(ids, primed + 5 + k)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    n = steps - 5
    print(f"B={B} {prec}: host {1e6 * (t1 - t0) / n:.1f} us/step, total {1e6 * (t2 - t0) / n:.1f} us/step -> {B * n / (t2 - t0):.0f} tok/s")
