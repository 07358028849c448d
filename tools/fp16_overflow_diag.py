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

# ---- graph mode: does a replay read the LIVE loss scale? ----
if os.environ.get("GRAPH", "1") == "1":
    from open_musiclm_amd.graph import GraphedForwardBackward
    torch.manual_seed(0)
    model2 = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0, precision="fp16").to(dev)
    stage2 = M.CoarseStage(coarse_transformer=model2, cross_entropy_loss_weights=[0., 0., 1.]).train()
    opt2 = get_optimizer(model2.parameters(), lr=3e-3, wd=0.01)
    opt2.zero_grad()
    st = engine.loss_scale_state(model2)
    st[0] = 1024.0
    fb = GraphedForwardBackward(lambda **k: stage2(**k, return_loss=True, return_logits=False)[0])
    fb.prepare(kw, after_warmup=lambda: (opt2.mark_grads_dirty(), opt2.zero_grad()))
    print("captured:", fb.graph is not None, fb.capture_error, "state tensor id", st.data_ptr(), "opt state id", opt2._ls_state.data_ptr())
    for sc in (1024.0, 4096.0, 16384.0, 2.0 ** 26, 1024.0):
        st[0] = sc
        opt2.mark_grads_dirty(); opt2.zero_grad()
        loss = fb(**kw)
        torch.cuda.synchronize()
        G = opt2.flat_grad
        fin = torch.isfinite(G)
        print(f"graph replay at scale {sc:g}: loss {float(loss):.4f} finite {float(fin.float().mean()):.4f} max finite |G| {float(G[fin].abs().max()):.4g} "
              f"-> per unit scale {float(G[fin].abs().max()) / sc:.4g}", flush=True)
