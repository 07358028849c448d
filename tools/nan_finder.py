"""Which kernel reads memory it (or its producer) never wrote?  Every torch.empty / empty_like of the step is filled with NaN, every ops.*
call is followed by a scan of its tensor arguments, and the first calls after which an argument still / newly holds a non-finite value
are listed with the argument's position, shape and the fraction of bad elements.  Tiny fp16 coarse model by default (ragged everything:
N = 341, F = 341 -> Fp = 384, 2 heads), eager mode, loss scale 1024.

    python tools/nan_finder.py            [DIM=128 DEPTH=2 HEADS=2 PREC=fp16 FUSED=1]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import engine, ops, open_musiclm as M

dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "fp16")
torch.manual_seed(0)
model = M.create_coarse_transformer(dim=int(os.environ.get("DIM", "128")), depth=int(os.environ.get("DEPTH", "2")), heads=int(os.environ.get("HEADS", "2")),
                                    num_coarse_quantizers=3, ff_dropout=0.0, precision=prec).to(dev)
if prec == "fp16":
    engine.loss_scale_state(model)[0] = 1024.0
stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.]).train()
g = torch.Generator().manual_seed(1)
kw = dict(clap_token_ids=torch.randint(0, 1024, (2, 12, 1), generator=g).to(dev), semantic_token_ids=torch.randint(0, 1024, (2, 99), generator=g).to(dev),
          coarse_token_ids=torch.randint(0, 1024, (2, 75, 3), generator=g).to(dev))
if os.environ.get("FUSED", "1") == "0":
    os.environ["OMLM_FUSED_PREP"] = "0"

# ---- NaN-filled allocations ----
_empty, _empty_like = torch.empty, torch.empty_like


def nan_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def nan_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    return t


torch.empty, torch.empty_like = nan_empty, nan_empty_like

# ---- scan after every ops call ----
seen_bad = {}
events = []


def wrap(name, fn):
    def inner(*a, **k):
        r = fn(*a, **k)
        items = list(enumerate(a)) + list(k.items())
        for pos, t in items:
            ts = t if isinstance(t, (list, tuple)) else [t]
            for t_ in ts:
                if isinstance(t_, torch.Tensor) and t_.is_cuda and t_.is_floating_point() and t_.numel() > 0:
                    bad = float((~torch.isfinite(t_)).float().mean())
                    key = (t_.data_ptr(), t_.numel())
                    if bad > 0 and seen_bad.get(key) != name:
                        events.append((len(events), name, pos, tuple(t_.shape), str(t_.dtype)[6:], round(bad, 5)))
                        seen_bad[key] = name
        return r
    return inner


for n in dir(ops):
    f = getattr(ops, n)
    if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__ and n not in ("dcode", "tdtype", "planes_begin", "planes_end", "operand_planes", "index_error_flag", "raise_on_index_error"):
        if isinstance(f, type):
            continue
        setattr(ops, n, wrap(n, f))
loss, _, _ = stage(**kw, return_loss=True, return_logits=False)
loss.backward()
torch.cuda.synchronize()
torch.empty, torch.empty_like = _empty, _empty_like
bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print(f"loss {float(loss):.4f}; parameters with non-finite gradients: {len(bad)} {bad[:8]}")
print("first calls that left a non-finite value in one of their tensor arguments (call #, op, argument, shape, dtype, bad fraction):")
for e in events[:40]:
    print("  ", e)
