"""Diagnostic: the fp16 overflow test's flow through SingleStageTrainer, with per-step state."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import engine, open_musiclm as M
from open_musiclm_amd.data import SyntheticTokenDataset
from open_musiclm_amd.trainer import SingleStageTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0, precision="fp16").to(dev)
engine.loss_scale_state(model)[0] = float(2 ** 26)
ds = SyntheticTokenDataset("coarse", length=8, coarse_window_seconds=1, semantic_window_seconds=2)
tr = SingleStageTrainer(model, "coarse", num_train_steps=60, batch_size=2, dataset=ds, lr=3e-3, lr_warmup=0, grad_accum_every=1, wd=0.01,
                        max_grad_norm=0.0, valid_frac=0.0, save_results_every=1000, save_model_every=1000,
                        results_folder=tempfile.mkdtemp(), save_predicted_tokens=False, save_reconstructed_wave=False,
                        use_hip_graph=os.environ.get("GRAPH", "1") == "1")
tr.optim.zero_grad()
print("ls tensors:", engine.loss_scale_state(model).data_ptr(), tr.optim._ls_state.data_ptr(), "graph", tr.use_hip_graph, "max_grad_norm", tr.max_grad_norm)
orig_step = tr.optim.step
def spy(*a, **k):
    G = tr.optim.flat_grad
    fin = torch.isfinite(G)
    names = [n for n, p in model.named_parameters() if not torch.isfinite(p.grad).all()]
    print(f"   before step: scale {float(tr.optim._ls_state[0]):g} finite frac {float(fin.float().mean()):.4f} bad {len(names)} kwargs {k}\n   bad: {names}\n   good: {[n for n, p in model.named_parameters() if torch.isfinite(p.grad).all()]}", flush=True)
    r = orig_step(*a, **k)
    print(f"   after step: gnorm_sq {float(tr.optim._gnorm_sq):.4g} state {tr.optim._ls_state.tolist()}", flush=True)
    return r
tr.optim.step = spy
for i in range(int(os.environ.get("STEPS", "10"))):
    if os.environ.get("FORCE_ZERO") == "1":
        tr.optim.mark_grads_dirty(); tr.optim.zero_grad()
    logs = tr.train_step()
    print(i, logs, flush=True)
