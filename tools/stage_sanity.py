import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from open_musiclm_amd import open_musiclm as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
g = torch.Generator().manual_seed(3)
sem = M.SemanticStage(semantic_transformer=M.create_semantic_transformer(dim=1024, depth=6, heads=8).to(dev)).eval()
clap = torch.randint(0, 1024, (2, 12, 1), generator=g).to(dev)
t = time.perf_counter(); s = sem.generate(clap_token_ids=clap, max_time_steps=499); torch.cuda.synchronize()
print("semantic", tuple(s.shape), f"{2 * 499 / (time.perf_counter() - t):.0f} ids/s", int(s.min()), int(s.max()))
prec = os.environ.get("PREC", "bf16")
fine = M.FineStage(fine_transformer=M.create_fine_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, num_fine_quantizers=5,
                                                               precision=prec).to(dev)).eval()
coarse = torch.randint(0, 1024, (2, 150, 3), generator=g).to(dev)
t = time.perf_counter(); f = fine.generate(clap_token_ids=clap, coarse_token_ids=coarse, max_time_steps=150); torch.cuda.synchronize()
print("fine", tuple(f.shape), f"{2 * 750 / (time.perf_counter() - t):.0f} ids/s", int(f.min()), int(f.max()))
a = fine.generate(clap_token_ids=clap, coarse_token_ids=coarse, max_time_steps=6, uniforms=torch.rand(30, 2, 1025, generator=g))
g2 = torch.Generator().manual_seed(3); torch.randint(0, 1024, (2, 12, 1), generator=g2); torch.randint(0, 1024, (2, 150, 3), generator=g2)
b = fine.generate(clap_token_ids=clap, coarse_token_ids=coarse, max_time_steps=6, uniforms=torch.rand(30, 2, 1025, generator=g2), use_cache=False)
print(prec, "fine cached == reforward:", bool(torch.equal(a, b)), "first mismatch at", (a != b).flatten().nonzero()[:1].tolist())
