"""CPU side of the fp16 seed sweep (round-3 VERDICT item 6): oracle logits of the final sequence for several weight / id seeds at
BASELINE config 2 (coarse-small, depth 6, N = 1116) and config 4 (musiclm_large fine, depth 24, N = 1817), B = 1, eval mode.  Written to
.bigfix/ (git-ignored, travels with gpurun); tools/fp16_seed_sweep.py measures the GPU modes against them.  Test infrastructure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M
from oracle import musiclm_oracle as O

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".bigfix")
os.makedirs(out, exist_ok=True)
cases = [("coarse6", dict(dim=1024, depth=6, heads=8), [1, 199, 300], O.coarse_spec, lambda **k: M.create_coarse_transformer(num_coarse_quantizers=3, ff_dropout=0.0, **k), int(sys.argv[1]) if len(sys.argv) > 1 else 5),
         ("fine24", dict(dim=1024, depth=24, heads=16), [1, 225, 225], O.fine_spec, lambda **k: M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, ff_dropout=0.0, **k), int(sys.argv[2]) if len(sys.argv) > 2 else 3)]
for name, kw, lens, spec_fn, mk, nseeds in cases:
    spec = spec_fn(**kw)
    for seed in range(nseeds):
        path = os.path.join(out, f"{name}_seed{seed}.pt")
        if os.path.exists(path):
            continue
        torch.manual_seed(100 + seed)
        model = mk(precision="bf16", **kw)
        sd = {k: v.detach() for k, v in model.state_dict().items()}
        ids = O.synthetic_ids(spec, 1, lens, seed=4321 + seed)
        t0 = time.time()
        with torch.no_grad():
            loss, logits, _ = O.wrapper_forward_loss(sd, spec, ids, [0., 0., 1.], forget_noise=None)
        torch.save(dict(ids=ids, logits=logits[-1].half() if False else logits[-1].float(), loss=float(loss), seed=seed), path)
        print(name, seed, f"{time.time() - t0:.1f}s", tuple(logits[-1].shape), flush=True)
