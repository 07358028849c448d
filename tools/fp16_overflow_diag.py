"""Diagnostic: fp16 training with an absurd initial loss scale -- per step: loss, scale block, which gradients are non-finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import engine, open_musiclm as M
from open_musiclm_amd.optimizer import get_optimizer

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0, precision="fp16").to(dev)
engine.loss_scale_state(model)[0] = float(2 ** int(os.environ.get("LOG2", "26")))
stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.]).train()
opt = get_optimizer(model.parameters(), lr=3e-3, wd=0.01)
opt.zero_grad()
g = torch.Generator().manual_seed(1)
names = [n for n, _ in model.named_parameters()]
for step in range(int(os.environ.get("STEPS", "14"))):
    kw = dict(clap_token_ids=torch.randint(0, 1024, (2, 12, 1), generator=g).to(dev), semantic_token_ids=torch.randint(0, 1024, (2, 99), generator=g).to(dev),
              coarse_token_ids=torch.randint(0, 1024, (2, 75, 3), generator=g).to(dev))
    opt.zero_grad()
    loss, _, _ = stage(**kw, return_loss=True, return_logits=False)
    loss.backward()
    opt.mark_grads_dirty()
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
    gmax = max(float(p.grad.abs().max()) for p in model.parameters() if torch.isfinite(p.grad).all()) if len(bad) < len(names) else float("nan")
    st_before = engine.loss_scale_state(model).tolist()
    opt.step(max_grad_norm=0.0)
    print(f"step {step}: loss {float(loss):.4f} scale_before {st_before[0]:g} non-finite grads {len(bad)}/{len(names)} {bad[:4]} max finite |g| {gmax:.3g} "
          f"gnorm_sq {float(opt._gnorm_sq):.4g} -> state {engine.loss_scale_state(model).tolist()}", flush=True)
