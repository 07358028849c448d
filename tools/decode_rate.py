"""Per-step time of the KV-cached sampling loop as the stages run it (graph replays), prefill differenced out: two `generate` calls of
different lengths, (t_long - t_short) / extra ids.  env: B (1), PREC (fp16ff), REPS (5).  A/B of library builds: OMLM_LIB_PATH=tools/ab/libomlm_<x>.so"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M

dev = torch.device("cuda:0")
B = int(os.environ.get("B", 1)); prec = os.environ.get("PREC", "fp16ff"); reps = int(os.environ.get("REPS", 5))
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, precision=prec).to(dev)
stage = M.CoarseStage(coarse_transformer=model).eval()
g = torch.Generator().manual_seed(99)
kw = dict(clap_token_ids=torch.randint(0, 1024, (B, 12, 1), generator=g).to(dev),
          semantic_token_ids=torch.randint(0, 1024, (B, 199), generator=g).to(dev), use_cache=True)
short, long_ = 10, 110


def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    stage.generate(max_time_steps=n, **kw)
    torch.cuda.synchronize()
    return time.perf_counter() - t


run(2)
best = None
for _ in range(reps):
    ts, tl = run(short), run(long_)
    us = 1e6 * (tl - ts) / ((long_ - short) * 3)
    best = us if best is None else min(best, us)
print(f"B={B} {prec} lib={os.path.basename(os.environ.get('OMLM_LIB_PATH', 'default'))}: {best:.1f} us/step -> {B * 1e6 / best:.0f} ids/s "
      f"(short call {1e3 * ts:.1f} ms)")
