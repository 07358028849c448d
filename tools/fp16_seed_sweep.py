"""GPU side of the seed sweep: logits error (max |d| / max |ref|) of every precision mode against the oracle files of
tools/make_seed_oracles.py (.bigfix/), per seed; writes a markdown table (profiles/r04_seed_sweep.md).  No training: eval-mode forward with
return_loss=True (mask_prob off), B = 1."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M

dev = torch.device("cuda:0")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for name, kw, mk in (("coarse6", dict(dim=1024, depth=6, heads=8), lambda **k: M.create_coarse_transformer(num_coarse_quantizers=3, ff_dropout=0.0, **k)),
                     ("fine24", dict(dim=1024, depth=24, heads=16), lambda **k: M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, ff_dropout=0.0, **k))):
    for path in sorted(glob.glob(os.path.join(root, ".bigfix", f"{name}_seed*.pt"))):
        z = torch.load(path)
        for prec in ("bf16", "fp16", "fp16ff", "bf16x3"):
            torch.manual_seed(100 + z["seed"])
            model = mk(precision=prec, **kw).to(dev)
            wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.0)
            wrapper.eval()
            with torch.no_grad():
                loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in z["ids"]], return_loss=True)
            ref = z["logits"].double()
            got = logits[-1].double().cpu()
            e = float((got - ref).abs().max() / ref.abs().max())
            rows.append(dict(config=name, seed=z["seed"], precision=prec, logits_err=e, loss_err=abs(float(loss) - z["loss"]) / z["loss"]))
            print(rows[-1], flush=True)
            del model, wrapper
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "seed_sweep.md")
with open(out, "w") as fh:
    fh.write("Logits error max|d| / max|ref| against the CPU oracle, B = 1, eval forward, per weight / id seed (tools/make_seed_oracles.py, tools/fp16_seed_sweep.py)\n\n")
    fh.write("| config | precision | seeds | max | mean | per seed |\n|---|---|---:|---:|---:|---|\n")
    for cfg in ("coarse6", "fine24"):
        for prec in ("bf16", "fp16", "fp16ff", "bf16x3"):
            v = [r["logits_err"] for r in rows if r["config"] == cfg and r["precision"] == prec]
            if v:
                fh.write(f"| {cfg} | {prec} | {len(v)} | {max(v):.2e} | {sum(v) / len(v):.2e} | {', '.join(f'{x:.2e}' for x in v)} |\n")
json.dump(rows, open(out.replace(".md", ".json"), "w"), indent=1)
print(open(out).read())
