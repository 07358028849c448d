"""Where one `generate` call spends its time (coarse-small, B = 1, 300 ids): decoder construction, prefill, uniforms, the sampling
loop, post-processing -- each bracketed by a device synchronise.  Answers whether the fixed cost per call (paid once per window by
MusicLM.generate) is worth caching."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M, decode

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, precision=os.environ.get("PREC", "fp16ff")).to(dev)
stage = M.CoarseStage(coarse_transformer=model).eval()
g = torch.Generator().manual_seed(99)
B = int(os.environ.get("B", 1))
clap = torch.randint(0, 1024, (B, 12, 1), generator=g).to(dev)
sem = torch.randint(0, 1024, (B, 199), generator=g).to(dev)
T = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[label] = T.get(label, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, timed)


wrap(decode.CachedDecoder, "__init__", "decoder construction")
wrap(decode.CachedDecoder, "prefill", "prefill")
wrap(decode.SamplingLoop, "run", "sampling loop")
kw = dict(clap_token_ids=clap, semantic_token_ids=sem)
for steps in (4, 100, 100):
    T.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    out = stage.generate(max_time_steps=steps, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    n = 3 * steps * B
    rest = dt - sum(T.values())
    print(f"B={B} {n} ids: total {1e3 * dt:7.2f} ms = " + ", ".join(f"{k} {1e3 * v:.2f}" for k, v in T.items()) +
          f", other {1e3 * rest:.2f} ms | loop {1e6 * T.get('sampling loop', 0) / n * B:.1f} us per step | {n / dt:.0f} ids/s", flush=True)
