"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on CPU.

Run in the build container only (the GPU box has no /root/reference):

    python oracle/make_golden.py            # writes tests/golden/, asserts oracle == reference

The reference package cannot be imported raw (beartype / torchaudio / torchvision /
vector_quantize_pytorch / encodec / librosa are not installed — SURVEY.md §8c), so a stub
shim registers empty stand-ins for those modules.  None of them touches the arithmetic of
transformer.py / open_musiclm.py / utils.py, which then execute unmodified on torch CPU.

Every fixture stores: the reference's own randomly-initialised state_dict (so the weights
are the reference's init stream, not ours), the seeded token ids, the injected RNG draws,
and the reference outputs.  The oracle restatement is asserted against each one here, and
again (from the stored file) in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import os
import sys
import types
import typing

sys.dont_write_bytecode = True          # never drop __pycache__ into the read-only reference

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, ROOT)
from oracle import musiclm_oracle as O  # noqa: E402


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import transformers  # noqa: F401  (real)
    _stub("beartype", beartype=lambda f: f)
    _stub("beartype.typing", **{k: getattr(typing, k) for k in
                                ("List", "Optional", "Union", "Dict", "Tuple", "Literal")})
    _stub("beartype.door", is_bearable=lambda *a, **k: True)
    _stub("beartype.vale", Is=object)
    _stub("torchaudio", save=None, load=None)
    _stub("torchaudio.functional", resample=lambda x, *a, **k: x)
    _stub("torchvision")
    _stub("torchvision.transforms")
    _stub("vector_quantize_pytorch", ResidualVQ=object)
    _stub("encodec", EncodecModel=object)
    sys.path.insert(0, REF)
    _stub("open_musiclm.laion_clap", CLAP_Module=object)
    # the package __init__ of the reference is empty; register it so relative imports resolve
    pkg = types.ModuleType("open_musiclm")
    pkg.__path__ = [os.path.join(REF, "open_musiclm")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("open_musiclm", None, is_package=True)
    sys.modules["open_musiclm"] = pkg
    return importlib.import_module("open_musiclm.open_musiclm")


def to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def spec_of(model, **kw) -> O.ModelSpec:
    seqs = [O.SeqInfo(s.codebook_size, s.num_quantizers) for s in model.token_sequences]
    l0 = model.transformer.layers[0]
    return O.ModelSpec(seqs, dim=model.start_tokens[0].shape[0], depth=len(model.transformer.layers),
                       heads=l0[0].heads, **kw)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def make_train_case(ref, name, stage, kwargs, lengths, batch, seed, loss_weights, spec_kw=None):
    torch.manual_seed(seed)
    create = getattr(ref, f"create_{stage}_transformer")
    model = create(**kwargs)
    model.train()
    spec = spec_of(model, **(spec_kw or {}))
    ids = O.synthetic_ids(spec, batch, lengths, seed=1234 + seed)
    wrapper = ref.TokenConditionedTransformerWrapper(
        transformer=model, unique_consecutive=False,
        cross_entropy_loss_weights=list(loss_weights), mask_prob=0.15)
    wrapper.train()
    n_total = sum(int(np.prod(t.shape[1:])) + 2 for t in ids) - 1
    # the only RNG consumer with all dropouts at 0 is generate_mask_with_prob's randn (utils.py:51)
    torch.manual_seed(777 + seed)
    noise = torch.randn(batch, n_total)
    torch.manual_seed(777 + seed)
    loss, logits, labels = wrapper(all_token_ids=[t.clone() for t in ids], return_loss=True)
    loss.backward()
    grads = {"grad." + k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}

    wrapper.eval()
    with torch.no_grad():
        ev_logits = wrapper(all_token_ids=[t.clone() for t in ids], return_loss=False)

    # oracle check -----------------------------------------------------------------
    sdo = {k: v.clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("beta")) for k, v in sd.items()}
    o_loss, o_logits, o_labels = O.wrapper_forward_loss(sdo, spec, ids, loss_weights, forget_noise=noise)
    names = [k for k, v in sdo.items() if v.requires_grad]
    o_grads = torch.autograd.grad(o_loss, [sdo[k] for k in names], allow_unused=True)
    assert abs(float(o_loss.detach()) - float(loss.detach())) <= 2e-5 * abs(float(loss)), (float(o_loss), float(loss))
    for a, b in zip(o_logits, logits):
        assert rel_err(a.detach(), b.detach()) < 2e-5, rel_err(a.detach(), b.detach())
    for a, b in zip(o_labels, labels):
        assert torch.equal(a, b)
    worst = 0.0
    for k, g in zip(names, o_grads):
        rg = grads.get("grad." + k)
        if rg is None:
            assert g is None or float(g.abs().max()) == 0.0, k
            continue
        if float(rg.abs().max()) < 1e-5 and float(g.abs().max()) < 1e-5:
            continue    # analytically-zero grads (e.g. the per-head constant net.3.bias cancels in softmax): fp noise only
        e = rel_err(g, rg)
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    with torch.no_grad():
        ids_e, _, mask_e = O.build_training_inputs(ids, spec)
        # eval mode, return_loss False: eos appended to every sequence, nothing dropped (:346-371)
        b = ids[0].shape[0]
        ids_full = [O.append_eos(t.reshape(b, -1).long(), e) for t, e in zip(ids, spec.eos_ids)]
        parts = []
        for k in range(len(ids_full) - 1):
            m = (ids_full[k] != -1) & (ids_full[k] != spec.eos_ids[k])
            ids_full[k] = ids_full[k].masked_fill(~m, 0)
            parts.append(torch.nn.functional.pad(m, (1, 0), value=True))
        mk = torch.nn.functional.pad(torch.cat(parts, -1), (0, ids_full[-1].shape[-1] + 1), value=True)
        o_ev = O.token_conditioned_forward({k: v.detach() for k, v in sdo.items()}, spec, ids_full, mk)
    for a, b in zip(o_ev, ev_logits):
        assert rel_err(a, b) < 2e-5
    print(f"[{name}] loss {float(loss):.6f}  oracle-vs-reference worst grad rel err {worst:.2e}  N={n_total}")

    out = {"sd." + k: v for k, v in to_np(sd).items()}
    out.update(to_np(grads))
    for i, t in enumerate(ids):
        out[f"ids.{i}"] = t.numpy()
    for i, (lg, lb, ev) in enumerate(zip(logits, labels, ev_logits)):
        out[f"logits.{i}"] = lg.detach().numpy()          # [b, c, n] (training, forgetful mask on)
        out[f"labels.{i}"] = lb.numpy()
        out[f"eval_logits.{i}"] = ev.numpy()               # [b, n, c]
    out["forget_noise"] = noise.numpy()
    out["loss"] = np.float64(float(loss))
    out["loss_weights"] = np.asarray(loss_weights, np.float64)
    out["meta.stage"] = np.array(stage)
    out["meta.kwargs"] = np.array(repr(kwargs))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    return model, spec, ids


def make_generate_case(ref, name, model, spec, ids, max_time_steps, temperature, seed):
    wrapper = ref.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    b = ids[0].shape[0]
    q = spec.token_sequences[-1].num_quantizers
    v1 = spec.token_sequences[-1].codebook_size + 1
    n_steps = max_time_steps * q
    torch.manual_seed(seed)
    uniforms = torch.stack([torch.zeros(b, v1).uniform_(0, 1) for _ in range(n_steps)])
    torch.manual_seed(seed)
    out_ids = wrapper.generate(conditioning_token_ids=[t.clone() for t in ids[:-1]],
                               max_time_steps=max_time_steps, temperature=temperature)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    with torch.no_grad():
        o_ids = O.generate(sd, spec, ids[:-1], max_time_steps, uniforms, temperature=temperature)
    assert torch.equal(o_ids, out_ids), (o_ids, out_ids)
    # primed continuation (pred_token_ids given): :272-275
    prime = out_ids[:, :1]
    torch.manual_seed(seed + 1)
    uniforms2 = torch.stack([torch.zeros(b, v1).uniform_(0, 1) for _ in range((max_time_steps - 1) * q)])
    torch.manual_seed(seed + 1)
    out2 = wrapper.generate(conditioning_token_ids=[t.clone() for t in ids[:-1]], pred_token_ids=prime.clone(),
                            max_time_steps=max_time_steps, temperature=temperature)
    with torch.no_grad():
        o2 = O.generate(sd, spec, ids[:-1], max_time_steps, uniforms2, pred_ids=prime, temperature=temperature)
    assert torch.equal(o2, out2)
    print(f"[{name}] generated {tuple(out_ids.shape)} ids, oracle bit-exact")
    out = {"sd." + k: v for k, v in to_np(sd).items()}
    for i, t in enumerate(ids[:-1]):
        out[f"cond.{i}"] = t.numpy()
    out["uniforms"] = uniforms.numpy()
    out["uniforms_primed"] = uniforms2.numpy()
    out["generated"] = out_ids.numpy()
    out["generated_primed"] = out2.numpy()
    out["prime"] = prime.numpy()
    out["temperature"] = np.float64(temperature)
    out["max_time_steps"] = np.int64(max_time_steps)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def make_kmeans_case(name):
    from sklearn.cluster import MiniBatchKMeans
    rng = np.random.RandomState(0)
    feats = rng.randn(4096, 64).astype(np.float32)
    km = MiniBatchKMeans(n_clusters=32, batch_size=1024, n_init=1, random_state=0, max_iter=5).fit(feats)
    x = rng.randn(512, 64).astype(np.float32)
    ref = km.predict(x).astype(np.int64)
    mine = O.kmeans_assign(x, km.cluster_centers_.astype(np.float32))
    mism = int((ref != mine).sum())
    print(f"[{name}] sklearn predict vs oracle nearest_code: {mism} mismatches / {len(x)}")
    assert mism == 0
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x,
                        centroids=km.cluster_centers_.astype(np.float32), assign=ref)


def make_rvq_case(name):
    """RVQ pin: the reference's arithmetic lives in the un-vendored vector-quantize-pytorch (`argmax(-cdist(x, embed))`), so the
    fixture's ids are written by torch.cdist ITSELF on exactly representable inputs (tests/rvq_cases.py: on those every summation
    order gives the same bits, so the ids are a property of the distance form)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    import rvq_cases as RC
    x, cb, info = RC.exact_rvq_case(24, 64, 24, 4, seed=1)       # both sides <= 25 rows: cdist's direct path, root-merged row included
    idx = RC.cdist_chain(x, cb)
    RC.check_engineered(idx, info)
    assert np.array_equal(O.rvq_encode(x, cb), idx)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x, codebooks=cb, indices=idx)
    print(f"[{name}] rvq fixture written: ids by torch.cdist, oracle agrees bit for bit")


def check_full_size(ref):
    """BASELINE config 0: musiclm_small semantic stage, B=2, N=514, one fwd+bwd on CPU.
    Weights are too large to store; assert oracle == reference here and record the scalars."""
    torch.manual_seed(0)
    model = ref.create_semantic_transformer(dim=1024, depth=6, heads=8, ff_dropout=0.0)
    spec = spec_of(model)
    ids = O.synthetic_ids(spec, 2, [1, 499], seed=1234)
    wrapper = ref.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                     cross_entropy_loss_weights=[0., 1.], mask_prob=0.15)
    wrapper.train()
    torch.manual_seed(5)
    noise = torch.randn(2, 514)
    torch.manual_seed(5)
    loss, logits, _ = wrapper(all_token_ids=[t.clone() for t in ids], return_loss=True)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    with torch.no_grad():
        o_loss, o_logits, _ = O.wrapper_forward_loss(sd, spec, ids, [0., 1.], forget_noise=noise)
    e = rel_err(o_logits[-1], logits[-1].detach())
    print(f"[full-size semantic small] ref loss {float(loss):.5f} oracle loss {float(o_loss):.5f} logits rel err {e:.2e}")
    assert e < 1e-4
    with open(os.path.join(OUT, "full_size_semantic_small.txt"), "w") as f:
        f.write(f"reference_loss {float(loss):.6f}\noracle_loss {float(o_loss):.6f}\nlogits_rel_err {e:.3e}\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    tiny = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.0)
    # small, mutually different codebooks keep the fixtures small and catch table/offset mix-ups
    cbs = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    cbf = dict(clap_codebook_size=32, acoustic_codebook_size=40)
    cbm = dict(clap_codebook_size=32, semantic_codebook_size=48)
    m, spec, ids = make_train_case(ref, "tiny_coarse", "coarse", dict(tiny, num_coarse_quantizers=3, **cbs),
                                   lengths=[1, 7, 5], batch=2, seed=1, loss_weights=[0., 0., 1.])
    make_generate_case(ref, "tiny_coarse_generate", m, spec, ids, max_time_steps=3, temperature=0.95, seed=11)
    make_train_case(ref, "tiny_fine_allweights", "fine",
                    dict(tiny, num_coarse_quantizers=3, num_fine_quantizers=5, **cbf),
                    lengths=[1, 4, 4], batch=2, seed=2, loss_weights=[0.5, 1., 2.])
    make_train_case(ref, "tiny_semantic_t5_plainff", "semantic",
                    dict(tiny, use_conv_ff=False, relative_position_bias_type="t5", **cbm),
                    lengths=[1, 21], batch=3, seed=3, loss_weights=[0., 1.],
                    spec_kw=dict(use_conv_ff=False, relative_position_bias_type="t5"))
    make_kmeans_case("kmeans_assign")
    make_rvq_case("rvq_cdist_pin")
    if os.environ.get("GOLDEN_FULL", "1") == "1":
        check_full_size(ref)


if __name__ == "__main__":
    main()
