"""generate() throughput probe: coarse-small, B=1, 300 new ids, graph replay vs eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, precision="bf16").to(dev)
stage = M.CoarseStage(coarse_transformer=model).eval()
g = torch.Generator().manual_seed(99)
B = int(os.environ.get("B", 1))
clap = torch.randint(0, 1024, (B, 12, 1), generator=g).to(dev)
sem = torch.randint(0, 1024, (B, 199), generator=g).to(dev)
for use_graph in (True, False, True, False):
    kw = dict(clap_token_ids=clap, semantic_token_ids=sem, use_graph=use_graph)
    stage.generate(max_time_steps=4, **kw); torch.cuda.synchronize()
    t = time.perf_counter(); out = stage.generate(max_time_steps=100, **kw); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"B={B} use_graph={use_graph}: {B * 300 / dt:.0f} ids/s ({1e6 * dt / 300:.0f} us per step incl. prefill)")
