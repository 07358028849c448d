import os, sys, torch, json
sys.path.insert(0, os.getcwd())
from open_musiclm_amd import open_musiclm as M
import open_musiclm_amd.open_musiclm as MM
import open_musiclm_amd.engine as E
from oracle import musiclm_oracle as O
dev = torch.device("cuda:0")
spec = O.coarse_spec(dim=1024, depth=6, heads=8)
ids = O.synthetic_ids(spec, 2, [1, 199, 300], seed=1234)
noise = torch.randn(2, 1116, generator=torch.Generator().manual_seed(7))
names = [f"transformer.rel_pos_bias.net.{i}.0.weight" for i in (0, 1, 2)] + ["transformer.rel_pos_bias.net.3.weight", "transformer.layers.5.0.to_q.weight"]
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.0, precision="bf16").to(dev)
sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
sdo = {k: v.clone().requires_grad_(k in names) for k, v in sd.items()}
o_loss, o_logits, _ = O.wrapper_forward_loss(sdo, spec, ids, [0., 0., 1.], forget_noise=noise)
o_grads = dict(zip(names, torch.autograd.grad(o_loss, [sdo[k] for k in names])))
MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
def run(precision, async_):
    E._RELPOS_ASYNC = async_
    model.precision = precision if hasattr(model, "precision") else None
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
    wrapper.train()
    for p in model.parameters(): p.grad = None
    loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
    loss.backward()
    torch.cuda.synchronize()
    out = {}
    P = dict(model.named_parameters())
    for k in names:
        g = P[k].grad.cpu(); r = o_grads[k]
        out[k.replace("transformer.", "")] = (round(float((g - r).abs().max() / r.abs().max()), 4), f"{float(r.abs().max()):.3e}", f"{float(g.abs().max()):.3e}")
    return out
for trial in range(3):
    for a in (True, False):
        print("async", a, "trial", trial, run("bf16", a), flush=True)
