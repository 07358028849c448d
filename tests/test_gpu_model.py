"""GPU: end-to-end parity of the MI355X path (through the C ABI) against
  (a) the golden vectors produced by the REAL reference (tests/golden/*.npz), and
  (b) the CPU oracle (oracle/musiclm_oracle.py) at BASELINE.json's full sizes.

Tolerances (written here, per north_star "logits within 1e-3 rel of CPU reference"):
  precision "bf16x3" (fp32 operands split hi/lo on the bf16 matrix cores):
      logits  max|d| / max|ref| <= 1e-3   (the north-star bar; measured values are ~1e-5)
      loss    <= 1e-4 relative;  parameter grads  max|d| / max(max|ref_k|, 1e-3 * max_all|ref|) <= 6e-2 per tensor
      (gradients that are analytically zero -- the per-head constant of the rel-pos bias cancels in softmax -- are pure
      rounding noise in the reference too, hence the global-scale floor in the denominator)
      (attention backward: S and dP are formed from hi/lo splits in this mode, the dQ/dK/dV contractions are single bf16
      passes -- stated in DESIGN.md section 2)
      sampled token ids: bit-exact against the reference's golden ids (same injected uniforms)
  precision "bf16" (single-pass bf16 operands, fp32 accumulation / residual / statistics):
      logits  <= 1.2e-2 (measured 6-7e-3 at full size)  -- bf16 operand rounding (2^-9 per element) cannot meet 1e-3; SURVEY.md measured 1.25e-2
      for the reference itself under bf16 autocast.  Throughput numbers quoted in bf16 carry THIS parity figure.
"""
import ast
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "model_report.json")

# Bars = about 2x the worst value measured over every case of this file (profiles/r03c_model_report.json holds the measurements):
#   bf16x3  logits <= 2.9e-4 (depth 24), loss <= 1.1e-5, per-tensor gradients <= 7.4e-3 of the tensor's largest entry
#   bf16    logits <= 7.7e-3 at depth 6 (1.6e-2 at depth 24: its own bar below), loss <= 4.3e-4, gradients <= 7.2e-2
#   fp16    logits <= 8.9e-4 at depth 6 (2.0e-3 at depth 24), loss <= 4.3e-5, gradients <= 7.2e-3
# `invariant`: the analytically-zero gradient directions of the rel-pos bias (see the train test), in the same units.
TOL = {"bf16x3": dict(logits=1e-3, loss=4e-5, grad=1.5e-2, invariant=1e-2), "bf16": dict(logits=1.2e-2, loss=1e-3, grad=1.5e-1, invariant=1.5),
       "fp16": dict(logits=1e-3, loss=1e-4, grad=1.5e-2, invariant=0.3),       # invariant: pure rounding noise, 3x its measured size (round 5: <= 0.086; bf16 <= 0.43 against 1.5)
       # "fp16ff" (round 5: the ConvFeedForward forward on hi/lo half planes): VERDICT round 4's bars -- 5e-4 at depth 6, 1e-3 at depth 24 (own bar
       # in the depth-24 test); its backward is fp16's
       # (gradients measured <= 4.8e-3 over every tensor at full size: bar 1e-2)
       "fp16ff": dict(logits=5e-4, loss=1e-4, grad=1e-2, invariant=0.3)}


def grad_unscale(precision):
    """1 / loss scale the backward of `precision` leaves in param.grad (engine.loss_scale: 1 except for fp16)."""
    from open_musiclm_amd import engine
    return 1.0 / engine.loss_scale(precision)


def report(name, **metrics):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    data = {}
    if os.path.exists(REPORT):
        try:
            data = json.load(open(REPORT))
        except Exception:
            data = {}
    data[name] = metrics
    json.dump(data, open(REPORT, "w"), indent=1, default=float)


def relerr(a, b, floor=0.0):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


RELPOS_TENSORS = [f"transformer.rel_pos_bias.net.{i}.0.{wb}" for i in range(3) for wb in ("weight", "bias")] + \
                 ["transformer.rel_pos_bias.net.3.weight", "transformer.rel_pos_bias.net.3.bias"]


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from open_musiclm_amd import hip
    hip.lib()
    return torch.device("cuda:0")


def build_from_golden(golden_dir, name, dev, precision):
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    stage, kwargs = str(z["meta.stage"]), ast.literal_eval(str(z["meta.kwargs"]))
    model = getattr(M, f"create_{stage}_transformer")(**kwargs, precision=precision)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)                       # reference checkpoints load strictly
    return z, model.to(dev)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
@pytest.mark.parametrize("name", ["tiny_coarse", "tiny_fine_allweights", "tiny_semantic_t5_plainff"])
def test_training_step_matches_reference_golden(golden_dir, dev, monkeypatch, name, precision):
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    z, model = build_from_golden(golden_dir, name, dev, precision)
    noise = torch.from_numpy(z["forget_noise"])
    monkeypatch.setattr(M, "generate_mask_with_prob",
                        lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device))
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                   cross_entropy_loss_weights=list(z["loss_weights"]), mask_prob=0.15)
    wrapper.train()
    ids = [torch.from_numpy(z[f"ids.{i}"]).to(dev) for i in range(len(model.token_sequences))]
    loss, logits, labels = wrapper(all_token_ids=ids, return_loss=True)
    loss.backward()
    tol = TOL[precision]
    e_loss = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    e_logits = [relerr(l, torch.from_numpy(z[f"logits.{i}"])) for i, l in enumerate(logits)]
    for i, lb in enumerate(labels):
        assert torch.equal(lb.cpu(), torch.from_numpy(z[f"labels.{i}"]))
    grads = {}
    gscale = max(float(np.abs(z[k]).max()) for k in z.files if k.startswith("grad."))
    for k, p in model.named_parameters():
        gk = "grad." + k
        if gk not in z.files:
            continue
        ref = torch.from_numpy(z[gk])
        got = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu() * grad_unscale(precision)
        grads[k] = float((got.double() - ref.double()).abs().max() / max(float(ref.abs().max()), 1e-3 * gscale))
    # The softmax is invariant to a per-head constant added to the rel-pos bias, so the gradient along that direction
    # (net.3.bias; for the causal T5 variant every distance maps to bucket 0, so its whole table) is analytically ZERO:
    # the reference holds ~1e-8 of rounding noise there, a flash-style backward holds its own (row sums of dS cancel
    # only to the precision of delta = sum(dO * O) vs sum(P * dP)).  Those entries are bounded in absolute terms only.
    invariant = {k: v for k, v in grads.items() if k.endswith("rel_pos_bias.net.3.bias") or k.endswith("relative_attention_bias.weight")}
    for k in invariant:
        grads.pop(k)
    # measured: <= 3e-3 (bf16x3), 0.1 (fp16), 0.43 (bf16) of 1e-3 x the largest gradient entry in the model
    assert all(v < tol["invariant"] for v in invariant.values()), invariant
    worst = max(grads.items(), key=lambda kv: kv[1])
    report(f"train[{name},{precision}]", loss=e_loss, logits=e_logits, worst_grad=worst, grads=grads, invariant=invariant)
    assert e_loss < tol["loss"], e_loss
    assert max(e_logits) < tol["logits"], e_logits
    assert worst[1] < tol["grad"], worst
    # eval-mode forward (no forgetful mask), return_loss=False path
    wrapper.eval()
    with torch.no_grad():
        ev = wrapper(all_token_ids=ids, return_loss=False)
    e_ev = [relerr(l, torch.from_numpy(z[f"eval_logits.{i}"])) for i, l in enumerate(ev)]
    assert max(e_ev) < tol["logits"], e_ev


def test_loss_only_step_skips_zero_weight_heads(golden_dir, dev, monkeypatch):
    """return_logits=False (what the trainers' optimizer steps use): same loss and gradients, bit for bit, as the full call; the
    logits of zero-weight sequences are not evaluated."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    z, model = build_from_golden(golden_dir, "tiny_coarse", dev, "bf16x3")
    noise = torch.from_numpy(z["forget_noise"])
    monkeypatch.setattr(M, "generate_mask_with_prob",
                        lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device))
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                   cross_entropy_loss_weights=[0, 0, 1], mask_prob=0.15)
    wrapper.train()
    ids = [torch.from_numpy(z[f"ids.{i}"]).to(dev) for i in range(3)]
    out = []
    for full in (True, False):
        model.zero_grad(set_to_none=True)
        loss, logits, _ = wrapper(all_token_ids=ids, return_loss=True, return_logits=full)
        loss.backward()
        out.append((float(loss.detach()), logits, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert abs(out[0][0] - out[1][0]) <= 2e-6 * abs(out[0][0])      # the row losses meet in one fp32 atomic sum: order-free to an ulp
    assert out[1][1][0] is None and out[1][1][1] is None and torch.equal(out[0][1][2], out[1][1][2])
    assert all(v is not None for v in out[0][1])
    # weight gradients meet in fp32 atomics (grouped split-K launches): equal to summation-order noise (tensors that are analytically
    # zero -- the rel-pos bias's constant direction -- hold only that noise: measured against the model's largest gradient entry)
    gscale = max(float(g.abs().max()) for g in out[0][2].values())
    for k, g in out[0][2].items():
        err = float((out[1][2][k] - g).abs().max()) / max(float(g.abs().max()), 1e-3 * gscale)
        invariant = k.endswith("rel_pos_bias.net.3.bias") or k.endswith("relative_attention_bias.weight")     # pure rounding noise (see the train test)
        assert err < (1e-3 if invariant else 1e-5), (k, err)
    loss = float(out[0][0])


@pytest.mark.parametrize("precision", ["fp16", "fp16ff"])
def test_train_mode_forward_without_autograd_keeps_dropout_on(golden_dir, dev, precision):
    """A train-mode forward under no_grad (validation loops that forget .eval(); the reference applies dropout there too, transformer.py:147):
    nothing is saved for a backward, FF dropout is still drawn -- fp16 takes the kernels that regenerate the mask, fp16ff (whose plane forward
    exists on the strip kernels only) hands the mask through keep bits of its own.  The loss moves with the draw and stays finite; with
    dropout off the call equals the grad-enabled one bit for bit."""
    from open_musiclm_amd import open_musiclm as M
    z, model = build_from_golden(golden_dir, "tiny_coarse", dev, precision)
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0, 0, 1], mask_prob=0.0)
    wrapper.train()
    ids = [torch.from_numpy(z[f"ids.{i}"]).to(dev) for i in range(3)]
    loss_g, logits_g, _ = wrapper(all_token_ids=ids, return_loss=True)
    with torch.no_grad():
        loss_n, logits_n, _ = wrapper(all_token_ids=ids, return_loss=True)
    # ff_dropout is 0 in the golden model: same logits bit for bit; the row losses meet in one fp32 atomic sum (order-free to an ulp)
    assert torch.equal(logits_g[-1], logits_n[-1]) and abs(float(loss_g) - float(loss_n)) <= 2e-6 * abs(float(loss_n))
    for _, _, ff in model.transformer.layers:
        ff[ff._idx["dropout"]].p = 0.3
    with torch.no_grad():
        a = float(wrapper(all_token_ids=ids, return_loss=True)[0])
        b = float(wrapper(all_token_ids=ids, return_loss=True)[0])
    assert np.isfinite(a) and np.isfinite(b) and a != b and a != float(loss_n)              # a new mask per forward
    assert abs(a - float(loss_n)) < 0.5 * float(loss_n)


def test_logits_path_autograd_matches_fused_loss(golden_dir, dev, monkeypatch):
    """TokenConditionedTransformer.forward (logits with grad) + torch CE == the fused loss path."""
    from open_musiclm_amd import open_musiclm as M
    z, model = build_from_golden(golden_dir, "tiny_coarse", dev, "bf16x3")
    model.eval()
    ids = [torch.from_numpy(z[f"ids.{i}"]).to(dev) for i in range(3)]
    full = [M.append_eos_id(i.reshape(i.shape[0], -1).long(), e) for i, e in zip(ids, model.eos_ids)]
    logits = model(all_token_ids=[full[0], full[1], full[2][:, :-1]])
    loss = torch.nn.functional.cross_entropy(logits[-1].transpose(1, 2), full[2])
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                   cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.0)
    wrapper.eval()
    # eval-mode wrapper masks the conditioning eos tokens; replicate by calling the transformer directly above
    # without a mask, so compare against the fused path under the same (no-mask) conditions:
    l2, *_ = model.loss_and_logits([full[0], full[1], full[2][:, :-1]], full, None, [0., 0., 1.])
    l2.backward()
    assert abs(float(l2) - float(loss)) < 1e-4 * abs(float(loss))
    gscale = max(float(v.abs().max()) for v in g1.values())
    assert {k for k, p in model.named_parameters() if p.grad is not None} == set(g1)
    errs = {k: float((p.grad.double() - g1[k].double()).abs().max() / max(float(g1[k].abs().max()), 1e-3 * gscale))
            for k, p in model.named_parameters() if k in g1 and not k.endswith("rel_pos_bias.net.3.bias")}
    worst = max(errs.items(), key=lambda kv: kv[1])
    report("logits_vs_fused", worst=worst)
    assert worst[1] < 2e-2, worst      # both paths share the kernels; they differ only in how dlogits is rounded


def test_fused_loss_with_unique_consecutive_padding(dev):
    """unique_consecutive sequences (open_musiclm.py:349-353): runs of equal ids collapse, the tail is padded with pad_id, the
    loss ignores those labels and the normaliser counts only the rest (:396-404) -- fused path vs logits + torch CE."""
    from open_musiclm_amd import open_musiclm as M
    torch.manual_seed(0)
    seqs = [M.TokenSequenceInfo(32, 2, False), M.TokenSequenceInfo(32, 1, True)]
    model = M.TokenConditionedTransformer(token_sequences=seqs, dim=64, depth=2, heads=2, ff_dropout=0.0,
                                          precision="bf16x3").to(dev)
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=True,
                                                   cross_entropy_loss_weights=[0.5, 1.0], mask_prob=0.0)
    wrapper.train()
    g = torch.Generator().manual_seed(2)
    a = torch.randint(0, 32, (3, 4, 2), generator=g).to(dev)
    b = torch.randint(0, 4, (3, 12), generator=g).to(dev)           # few distinct values: plenty of repeated runs
    loss, logits, labels = wrapper(all_token_ids=[a, b], return_loss=True)
    assert int((labels[1] == -1).sum()) > 0                         # padding really occurred
    loss.backward()
    g1 = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    ids, lbs, mask = wrapper._prepare([a, b], True, False)
    lg = model(all_token_ids=ids, self_attn_mask=mask)
    n0, n1 = lbs[0].numel(), int((lbs[1] != -1).sum())
    ce0 = torch.nn.functional.cross_entropy(lg[0].transpose(1, 2), lbs[0])
    ce1 = torch.nn.functional.cross_entropy(lg[1].transpose(1, 2), lbs[1], ignore_index=-1)
    ref = (ce0 * n0 * 0.5 + ce1 * n1 * 1.0) / (n0 + n1)
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref)), (float(loss), float(ref))
    gscale = max(float(v.abs().max()) for v in g1.values())
    errs = {k: float((p.grad.double() - g1[k].double()).abs().max() / max(float(g1[k].abs().max()), 1e-3 * gscale))
            for k, p in model.named_parameters() if k in g1 and p.grad is not None}
    worst = max(errs.items(), key=lambda kv: kv[1])
    report("unique_consecutive_loss", worst=worst, padded=int((labels[1] == -1).sum()))
    assert worst[1] < 2e-2, worst


@pytest.mark.parametrize("precision", ["bf16x3"])
def test_generate_matches_reference_golden_ids(golden_dir, dev, precision):
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "tiny_coarse_generate.npz"))
    zt, model = build_from_golden(golden_dir, "tiny_coarse", dev, precision)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    cond = [torch.from_numpy(z["cond.0"]).to(dev), torch.from_numpy(z["cond.1"]).to(dev)]
    out = wrapper.generate(conditioning_token_ids=cond, max_time_steps=int(z["max_time_steps"]),
                           temperature=float(z["temperature"]), uniforms=torch.from_numpy(z["uniforms"]))
    out2 = wrapper.generate(conditioning_token_ids=cond, pred_token_ids=torch.from_numpy(z["prime"]).to(dev),
                            max_time_steps=int(z["max_time_steps"]), temperature=float(z["temperature"]),
                            uniforms=torch.from_numpy(z["uniforms_primed"]))
    report("generate_golden", equal=bool(np.array_equal(out.cpu().numpy(), z["generated"])),
           got=out.cpu().numpy().tolist(), want=z["generated"].tolist())
    assert np.array_equal(out.cpu().numpy(), z["generated"])               # bit-exact token ids
    assert np.array_equal(out2.cpu().numpy(), z["generated_primed"])


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
def test_cached_decode_matches_full_reforward(golden_dir, dev, precision):
    """KV-cached single-row decode (csrc/decode.hip) vs the reference-style full re-forward of the same model: the logits
    of every step agree to the precision mode's tolerance, and the sampled ids are identical for the same uniforms."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "tiny_coarse_generate.npz"))
    zt, model = build_from_golden(golden_dir, "tiny_coarse", dev, precision)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    model.eval()
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    cond = [torch.from_numpy(z["cond.0"]).to(dev), torch.from_numpy(z["cond.1"]).to(dev)]
    kw = dict(conditioning_token_ids=cond, max_time_steps=int(z["max_time_steps"]), temperature=float(z["temperature"]),
              uniforms=torch.from_numpy(z["uniforms"]))
    a = wrapper.generate(use_cache=True, **kw)
    b = wrapper.generate(use_cache=False, **kw)
    c = wrapper.generate(use_cache=True, use_graph=True, **kw)      # cycles captured into HIP graphs and replayed
    assert torch.equal(a, b) and torch.equal(a, c)
    # step-by-step logits: teacher-forced on the ids of the run above
    with torch.no_grad():
        from open_musiclm_amd.utils import append_eos_id
        condx = [append_eos_id(t.reshape(t.shape[0], -1).long(), e) for t, e in zip(cond, wrapper.eos_ids)]
        flat = a.reshape(a.shape[0], -1)
        B, n = flat.shape
        rows = sum(t.shape[-1] + 1 for t in condx) + 1 + n
        dec = decode.CachedDecoder(model, B, rows, precision)
        got = [dec.prefill(condx + [flat[:, :0]]).clone()]
        for k in range(n - 1):
            got.append(dec.step(flat[:, k].contiguous(), k).clone())
        want = [model.last_logits(condx + [flat[:, :k]]).clone() for k in range(n)]
    V1 = dec.V1
    err = max(relerr(g[:, :V1], w[:, :V1]) for g, w in zip(got, want))
    report("cached_decode_" + precision, max_rel_err=err, steps=n)
    assert err < TOL[precision]["logits"], err       # (fp16ff: the steps read hi + lo weight planes and keep LN outputs / h1 un-rounded)


def test_cached_decode_batches_beyond_the_kernel_group(golden_dir, dev):
    """generate() with more samples than one decode call holds (MAX_DECODE_BATCH) runs groups of samples through the KV-cached
    path; samples are independent, so the ids equal the full re-forward path's for the same uniforms."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "tiny_coarse_generate.npz"))
    _, model = build_from_golden(golden_dir, "tiny_coarse", dev, "bf16x3")
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}
    model.load_state_dict(sd, strict=True)
    model.eval()
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    B = decode.MAX_DECODE_BATCH + 3
    g = torch.Generator().manual_seed(17)
    c0, c1 = torch.from_numpy(z["cond.0"]), torch.from_numpy(z["cond.1"])
    cond = [torch.randint(0, int(c0.max()) + 1, (B,) + tuple(c0.shape[1:]), generator=g).to(dev),
            torch.randint(0, int(c1.max()) + 1, (B,) + tuple(c1.shape[1:]), generator=g).to(dev)]
    steps, Q, V1 = 3, model.token_sequences[-1].num_quantizers, model.token_sequences[-1].codebook_size + 1
    U = torch.rand(steps * Q, B, V1, generator=g)
    kw = dict(conditioning_token_ids=cond, max_time_steps=steps, temperature=float(z["temperature"]), uniforms=U)
    a = wrapper.generate(use_cache=True, **kw)
    b = wrapper.generate(use_cache=False, **kw)
    assert a.shape == (B, steps, Q) and torch.equal(a, b)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("name", ["tiny_fine_allweights", "tiny_semantic_t5_plainff"])
def test_cached_decode_other_stages(golden_dir, dev, name, precision):
    """The cached decode on the other trunk variants: fine stage (two conditioning sequences, Q > 1 heads / offsets) and the
    semantic stage with the T5 bucket bias and the plain FeedForward (identity conv taps, Q = 1)."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.utils import append_eos_id
    z, model = build_from_golden(golden_dir, name, dev, precision)
    model.eval()
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    nseq = len(model.token_sequences)
    cond = [torch.from_numpy(z[f"ids.{i}"]).to(dev) for i in range(nseq - 1)]
    B = cond[0].shape[0]
    seq = model.token_sequences[-1]
    V1, Q, steps = seq.codebook_size + 1, seq.num_quantizers, 4
    g = torch.Generator().manual_seed(5)
    U = torch.rand(steps * Q, B, V1, generator=g)
    kw = dict(conditioning_token_ids=cond, max_time_steps=steps, uniforms=U)
    a = wrapper.generate(use_cache=True, **kw)
    b = wrapper.generate(use_cache=False, **kw)
    if precision == "bf16x3":            # bf16: the two paths round differently (fp32 cached rows vs bf16 batched rows), a near-tie
        assert torch.equal(a, b), (a.tolist(), b.tolist())      # may flip an id; their logits are compared below instead
    with torch.no_grad():
        condx = [append_eos_id(t.reshape(t.shape[0], -1).long(), e) for t, e in zip(cond, wrapper.eos_ids)]
        flat = a.reshape(B, -1)
        n = flat.shape[1]
        dec = decode.CachedDecoder(model, B, sum(t.shape[-1] + 1 for t in condx) + 1 + n, precision)
        got = [dec.prefill(condx + [flat[:, :0]]).clone()]
        for k in range(n - 1):
            got.append(dec.step(flat[:, k].contiguous(), k).clone())
        want = [model.last_logits(condx + [flat[:, :k]]).clone() for k in range(n)]
    err = max(relerr(x[:, :V1], y[:, :V1]) for x, y in zip(got, want))
    report(f"cached_decode_{name}_{precision}", max_rel_err=err, steps=n)
    assert err < TOL[precision]["logits"], err


@pytest.mark.parametrize("precision,B", [("bf16", 1), ("bf16", 4), ("bf16", 8), ("fp16", 8), ("bf16x3", 3), ("bf16", 16), ("fp16", 11),
                                         ("fp16ff", 1), ("fp16ff", 5), ("fp16ff", 16)])
def test_cached_decode_at_full_width(dev, precision, B):
    """The step kernels the bench runs (dim 1024, 8 heads, F = 2730: dec3 at B = 1, the matrix-core dec4 kernels with their LayerNorm
    partial sums at 2 <= B <= 16 in the 16-bit modes, the VALU dec2 kernels for fp32 weights) against the re-forward of the growing
    sequence: same bars as the batched forward itself."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.utils import append_eos_id
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=1024, depth=2, heads=8, ff_dropout=0.0, num_coarse_quantizers=3, precision=precision).to(dev)
    assert decode.supports(model, B, precision)
    model.eval()
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    g = torch.Generator().manual_seed(3)
    cond = [torch.randint(0, 1024, (B, 12, 1), generator=g).to(dev), torch.randint(0, 1024, (B, 40), generator=g).to(dev)]
    n, V1 = 9, 1025
    flat = torch.randint(0, 1024, (B, n), generator=g).to(dev)
    with torch.no_grad():
        condx = [append_eos_id(t.reshape(t.shape[0], -1).long(), e) for t, e in zip(cond, wrapper.eos_ids)]
        dec = decode.CachedDecoder(model, B, sum(t.shape[-1] + 1 for t in condx) + 1 + n, precision)
        got = [dec.prefill(condx + [flat[:, :0]]).clone()]
        for k in range(n - 1):
            got.append(dec.step(flat[:, k].contiguous(), k).clone())
        want = [model.last_logits(condx + [flat[:, :k]]).clone() for k in range(n)]
    err = max(relerr(x[:, :V1], y[:, :V1]) for x, y in zip(got, want))
    report(f"cached_decode_full_width[{precision},B={B}]", max_rel_err=err, steps=n)
    assert err < TOL[precision]["logits"], err


def _cached_steps_vs_oracle(dev, stage, depth, heads, B, lens_prompt, n_steps, precision, oracle_samples):
    """Teacher-forced KV-cached steps of a full-width model against the CPU ORACLE's forward of the same sequence (not against the product's
    own re-forward): prefill over the conditioning sequences + `lens_prompt[-1]` known ids of the predicted sequence, then n_steps cached
    steps; row j of the oracle's final-sequence logits predicts id j.  Returns the worst max-abs / max-ref over the checked steps."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.utils import append_eos_id
    from oracle import musiclm_oracle as O
    torch.manual_seed(0)
    if stage == "coarse":
        model = M.create_coarse_transformer(dim=1024, depth=depth, heads=heads, num_coarse_quantizers=3, ff_dropout=0.0, precision=precision).to(dev)
        spec = O.coarse_spec(dim=1024, depth=depth, heads=heads)
    else:
        model = M.create_fine_transformer(dim=1024, depth=depth, heads=heads, num_coarse_quantizers=3, num_fine_quantizers=5, ff_dropout=0.0,
                                          precision=precision).to(dev)
        spec = O.fine_spec(dim=1024, depth=depth, heads=heads)
    model.eval()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    g = torch.Generator().manual_seed(11)
    Q = spec.token_sequences[-1].num_quantizers
    cond = [torch.randint(0, 1024, (B, lens_prompt[0], 12), generator=g),
            torch.randint(0, 1024, (B, lens_prompt[1]) if spec.token_sequences[1].num_quantizers == 1 else (B, lens_prompt[1], spec.token_sequences[1].num_quantizers), generator=g)]
    n0 = lens_prompt[2]                                         # ids of the predicted sequence already known at the prefill
    flat = torch.randint(0, 1024, (B, n0 + n_steps), generator=g)
    with torch.no_grad():
        condx = [append_eos_id(t.reshape(B, -1).long(), e) for t, e in zip(cond, wrapper.eos_ids)]
        rows = sum(t.shape[-1] + 1 for t in condx) + 1 + n0 + n_steps
        dec = decode.CachedDecoder(model, B, rows, precision)
        fd = flat.to(dev)
        got = [dec.prefill([t.to(dev) for t in condx] + [fd[:, :n0]]).clone()]
        for k in range(n0, n0 + n_steps - 1):
            got.append(dec.step(fd[:, k].contiguous(), k).clone())
        # oracle: one forward of the whole teacher-forced sequence for a few of the samples; its final-sequence row j predicts id j
        sel = list(range(B))[:oracle_samples] if B <= oracle_samples else [0, B - 1][:oracle_samples]
        o = O.token_conditioned_forward(sd, spec, [t[sel] for t in condx] + [flat[sel][:, :n0 + n_steps - 1]], only_final=True)[-1]
    V1, worst = 1025, 0.0
    for i, lg in enumerate(got):
        worst = max(worst, relerr(lg[sel][:, :V1], o[:, n0 + i]))
    return worst, rows


@pytest.mark.parametrize("B", [1, 16])
def test_cached_steps_at_full_context_vs_oracle(dev, B):
    """the cached step of the headline precision against the ORACLE at depth 6 with the context the bench decodes at --
    prefill 214 conditioning rows (+ 830 known coarse ids in the deep variant) then 64 teacher-forced steps up to N = 1116 -- bar: the mode's own 5e-4."""
    e0, rows0 = _cached_steps_vs_oracle(dev, "coarse", 6, 8, B, [1, 199, 0], 64, "fp16ff", 2)
    e1, rows1 = _cached_steps_vs_oracle(dev, "coarse", 6, 8, B, [1, 199, 836], 64, "fp16ff", 2)
    report(f"cached_steps_vs_oracle[depth6,B={B}]", after_prefill=e0, rows_after_prefill=rows0, deep_context=e1, rows_deep=rows1)
    assert rows1 == 1116
    assert max(e0, e1) < TOL["fp16ff"]["logits"], (e0, e1)


@pytest.mark.parametrize("B", [1, 16])
def test_cached_steps_at_musiclm_large_depth_vs_oracle(dev, B):
    """The same at musiclm_large's depth (fine stage, 24 layers, 16 heads): 8 cached steps behind a 1 + 13 + 676 + 64 row prefill, bar 1e-3."""
    e, rows = _cached_steps_vs_oracle(dev, "fine", 24, 16, B, [1, 225, 64], 8, "fp16ff", 2)
    report(f"cached_steps_vs_oracle[depth24,B={B}]", err=e, rows=rows)
    assert e < 1e-3, e


def test_cached_decode_with_absolute_position_embeddings(dev):
    """use_absolute_position_embeddings=True (open_musiclm.py:81-82,134-136): the KV-cached single-row steps add the position row of the
    id they embed, so cached ids equal the re-forward ids (bf16x3) and the step logits equal last_logits of the growing sequence."""
    from open_musiclm_amd import decode
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.utils import append_eos_id
    torch.manual_seed(0)
    kw = dict(dim=128, depth=2, heads=2, ff_dropout=0.0, num_coarse_quantizers=3, clap_codebook_size=64, semantic_codebook_size=64,
              acoustic_codebook_size=64, use_absolute_position_embeddings=True, max_absolute_position_embeddings=64)
    model = M.create_coarse_transformer(precision="bf16x3", **kw).to(dev)
    assert decode.supports(model, 2)
    model.eval()
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    g = torch.Generator().manual_seed(3)
    cond = [torch.randint(0, 64, (2, 12, 1), generator=g).to(dev), torch.randint(0, 64, (2, 9), generator=g).to(dev)]
    Q, V1, steps = 3, 65, 4
    U = torch.rand(steps * Q, 2, V1, generator=g)
    kwg = dict(conditioning_token_ids=cond, max_time_steps=steps, uniforms=U)
    a = wrapper.generate(use_cache=True, **kwg)
    b = wrapper.generate(use_cache=False, **kwg)
    assert torch.equal(a, b), (a.tolist(), b.tolist())
    with torch.no_grad():
        condx = [append_eos_id(t.reshape(t.shape[0], -1).long(), e) for t, e in zip(cond, wrapper.eos_ids)]
        flat = a.reshape(2, -1)
        n = flat.shape[1]
        dec = decode.CachedDecoder(model, 2, sum(t.shape[-1] + 1 for t in condx) + 1 + n, "bf16x3")
        got = [dec.prefill(condx + [flat[:, :0]]).clone()]
        for k in range(n - 1):
            got.append(dec.step(flat[:, k].contiguous(), k).clone())
        want = [model.last_logits(condx + [flat[:, :k]]).clone() for k in range(n)]
    err = max(relerr(x[:, :V1], y[:, :V1]) for x, y in zip(got, want))
    report("cached_decode_abs_pos", max_rel_err=err, steps=n)
    assert err < TOL["bf16x3"]["logits"], err
    # and the position rows matter: zeroing them changes the logits
    with torch.no_grad():
        model.absolute_position_embeddings[-1].weight.zero_()
        other = model.last_logits(condx + [flat[:, :3]])
    assert relerr(other[:, :V1], want[3][:, :V1]) > 1e-3


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
def test_full_size_coarse_small_vs_oracle(dev, precision):
    """BASELINE config 2 shapes: musiclm_small coarse stage, N = 1116, B = 2; logits, loss and grads vs the CPU oracle."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.0,
                                        precision=precision).to(dev)
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ids = O.synthetic_ids(spec, 2, [1, 199, 300], seed=1234)
    noise = torch.randn(2, 1116, generator=torch.Generator().manual_seed(7))
    sdo = {k: v.clone().requires_grad_(k in ("transformer.layers.0.2.1.weight", "transformer.layers.5.0.to_q.weight",
                                             "embeddings.2.weight", "logit_weights.2",
                                             "transformer.rel_pos_bias.net.1.0.weight",
                                             "transformer.layers.3.0.to_kv.weight",
                                             "transformer.layers.2.2.2.ds_conv.weight")) for k, v in sd.items()}
    for k in RELPOS_TENSORS:                      # every tensor of the rel-pos MLP (round 3's defect lived in one of eight, and one was checked)
        sdo[k].requires_grad_(True)
    o_loss, o_logits, _ = O.wrapper_forward_loss(sdo, spec, ids, [0., 0., 1.], forget_noise=noise)
    names = [k for k, v in sdo.items() if v.requires_grad]
    o_grads = dict(zip(names, torch.autograd.grad(o_loss, [sdo[k] for k in names])))

    import open_musiclm_amd.open_musiclm as MM
    orig = MM.generate_mask_with_prob
    MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
    try:
        wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                       cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
        wrapper.train()
        loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
        loss.backward()
    finally:
        MM.generate_mask_with_prob = orig
    e_inf = relerr(logits[-1], o_logits[-1])
    e_l2 = rel_l2(logits[-1], o_logits[-1])
    e_loss = abs(float(loss) - float(o_loss)) / float(o_loss)
    top1 = float((logits[-1].argmax(1).cpu() == o_logits[-1].argmax(1)).float().mean())
    gmax = max(float(v.abs().max()) for v in o_grads.values())
    # the MLP's biases are near-invariant directions (a per-head constant of the bias cancels in the softmax: their gradients are 1e-3 of
    # the weights'), judged against 1e-2 of the largest gradient instead of their own size
    g = {k: relerr(dict(model.named_parameters())[k].grad * grad_unscale(precision), o_grads[k],
                   floor=1e-2 * gmax if (k in RELPOS_TENSORS and k.endswith("bias")) else 0.0) for k in names}
    print(g)
    report(f"full_coarse_small[{precision}]", logits_inf=e_inf, logits_l2=e_l2, loss=e_loss, top1_agree=top1, grads=g,
           loss_value=float(loss))
    tol = TOL[precision]
    assert e_inf < tol["logits"], (e_inf, e_l2)
    assert e_loss < tol["loss"]
    assert max(g.values()) < tol["grad"], g


@pytest.mark.parametrize("precision,bar", [("bf16x3", 1e-3), ("bf16", 2e-2)])
def test_full_size_backward_is_reproducible(dev, precision, bar):
    """Ten forward + backward passes of the full-size coarse step on the same inputs: every rel-pos gradient and three trunk gradients
    agree run to run.  fp32-operand mode: the order-of-addition noise of the fp32 atomics (split-K / d(bias) / dK, dV: ~1e-7 on d(table))
    reaches 3e-5 of a tensor's largest entry after the hi/lo splits downstream (measured); bf16 mode: that noise passes through bf16
    roundings of dK / dV / dq (one flipped 2^-9 ulp somewhere upstream) and reaches ~2e-3 (tests/hammer_relpos.py) -- the bars are
    10-30x those, far below round 3's 30-70 % defect."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.0, precision=precision).to(dev)
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    ids = [t.to(dev) for t in O.synthetic_ids(spec, 2, [1, 199, 300], seed=1234)]
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.],
                                                   mask_prob=0.0)
    wrapper.train()
    names = RELPOS_TENSORS + ["transformer.layers.0.2.1.weight", "transformer.layers.5.0.to_q.weight", "transformer.layers.3.0.to_kv.weight"]
    params = dict(model.named_parameters())
    first, worst = None, 0.0
    for rep in range(10):
        for p in model.parameters():
            p.grad = None
        if rep % 3 == 2:                     # recycled, non-zero memory for every torch.empty of the step
            junk = [torch.full((n,), float("nan"), device=dev) for n in (571392, 1142784, 262144, 1 << 22, 1 << 24)]
            del junk
        loss, _, _ = wrapper(all_token_ids=ids, return_loss=True)
        loss.backward()
        got = {k: params[k].grad.detach().clone() for k in names}
        if first is None:
            first = got
            gmax = max(float(v.abs().max()) for v in got.values())
            continue
        for k in names:
            worst = max(worst, relerr(got[k], first[k], floor=1e-2 * gmax))
    report(f"backward_reproducible[{precision}]", worst_run_to_run=worst)
    assert worst < bar, worst


def test_gemm_epilogue_survives_concurrent_streams(dev):
    """The GEMM tail race of round 3: non-split GEMMs launched on a second stream next to HBM-bound copies must
    return bit for bit what the same launch returns on an idle GPU (tests/stress_gemm_tail.py; the library built with
    -DOMLM_GEMM_TAIL_WAIT=0 fails ~45 % of these launches, profiles/r04_stress_gemm_tail_unfixed.json)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("stress_gemm_tail", os.path.join(ROOT, "tests", "stress_gemm_tail.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run(iters=150, burst=10)
    report("gemm_tail_stress", **{c["name"]: c["bad_launches"] for c in res["cases"]})
    assert res["bad_launches_total"] == 0, res["cases"]


@pytest.mark.parametrize("stage,lengths,N,kw", [("semantic", [1, 499], 514, {}),
                                                 ("fine", [1, 150, 150], 1217, dict(num_coarse_quantizers=3, num_fine_quantizers=5))])
def test_full_size_other_stages_forward_loss_vs_oracle(dev, stage, lengths, N, kw):
    """BASELINE sequence shapes of the other two musiclm_small stages (semantic N = 514, fine N = 1217, 5 fine quantizers):
    training-mode loss and final-sequence logits vs the CPU oracle, B = 1, default bf16 mode."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    torch.manual_seed(1)
    # musiclm_small quantizer counts (configs/model/musiclm_small.json): 3 coarse, 5 fine; the factory defaults are 4 / 8
    model = getattr(M, f"create_{stage}_transformer")(dim=1024, depth=6, heads=8, ff_dropout=0.0, precision="bf16", **kw).to(dev)
    spec = getattr(O, f"{stage}_spec")(dim=1024, depth=6, heads=8)
    assert [(s.codebook_size, s.num_quantizers) for s in spec.token_sequences] == \
           [(s.codebook_size, s.num_quantizers) for s in model.token_sequences]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ids = O.synthetic_ids(spec, 1, lengths, seed=4321)
    nseq = len(ids)
    w = [0.] * (nseq - 1) + [1.]
    noise = torch.randn(1, N, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        o_loss, o_logits, _ = O.wrapper_forward_loss(sd, spec, ids, w, forget_noise=noise)
    import open_musiclm_amd.open_musiclm as MM
    orig = MM.generate_mask_with_prob
    MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
    try:
        wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                       cross_entropy_loss_weights=w, mask_prob=0.15)
        wrapper.train()
        loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
        loss.backward()
    finally:
        MM.generate_mask_with_prob = orig
    assert logits[-1].shape == o_logits[-1].shape
    e_inf, e_loss = relerr(logits[-1], o_logits[-1]), abs(float(loss) - float(o_loss)) / float(o_loss)
    gfinite = all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    report(f"full_{stage}_small[bf16]", logits_inf=e_inf, loss=e_loss, N=N, grads_finite=gfinite)
    assert e_inf < TOL["bf16"]["logits"] and e_loss < TOL["bf16"]["loss"] and gfinite


_LARGE_ORACLE = {}


def _large_fine(dev, depth, precision, with_grads):
    """musiclm_large fine stage (configs/model/musiclm_large.json:53-63: dim 1024, heads 16; 3 s windows -> clap 12x1,
    coarse 225x3, fine 225x5 ids -> N = 1817), B = 1, against the CPU oracle."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    torch.manual_seed(2)
    model = M.create_fine_transformer(dim=1024, depth=depth, heads=16, ff_dropout=0.0, num_coarse_quantizers=3,
                                      num_fine_quantizers=5, precision=precision).to(dev)
    spec = O.fine_spec(dim=1024, depth=depth, heads=16)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ids = O.synthetic_ids(spec, 1, [1, 225, 225], seed=777)
    N = 1817
    noise = torch.randn(1, N, generator=torch.Generator().manual_seed(13))
    gnames = ["transformer.layers.0.0.to_q.weight", f"transformer.layers.{depth - 1}.2.1.weight", "logit_weights.2",
              "transformer.layers.0.0.to_kv.weight"] + [k for k in RELPOS_TENSORS if k.endswith("weight")] if with_grads else []
    if depth not in _LARGE_ORACLE:                   # same seed -> same weights for both precisions: one oracle run per depth
        sdo = {k: v.clone().requires_grad_(k in gnames) for k, v in sd.items()}
        with torch.set_grad_enabled(with_grads):
            o_loss, o_logits, _ = O.wrapper_forward_loss(sdo, spec, ids, [0., 0., 1.], forget_noise=noise)
        o_grads = dict(zip(gnames, torch.autograd.grad(o_loss, [sdo[k] for k in gnames]))) if with_grads else {}
        _LARGE_ORACLE[depth] = (o_loss.detach(), [l.detach() if l is not None else None for l in o_logits], o_grads)
    o_loss, o_logits, o_grads = _LARGE_ORACLE[depth]
    import open_musiclm_amd.open_musiclm as MM
    orig = MM.generate_mask_with_prob
    MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
    try:
        wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                       cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
        wrapper.train()
        loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
        loss.backward()
    finally:
        MM.generate_mask_with_prob = orig
    assert logits[-1].shape == o_logits[-1].shape and logits[-1].shape[-1] == 1126
    e_inf, e_l2 = relerr(logits[-1], o_logits[-1].detach()), rel_l2(logits[-1], o_logits[-1].detach())
    e_loss = abs(float(loss) - float(o_loss)) / float(o_loss)
    params = dict(model.named_parameters())
    g = {k: relerr(params[k].grad * grad_unscale(precision), o_grads[k]) for k in gnames}
    gfinite = all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    report(f"large_fine[depth={depth},{precision}]", logits_inf=e_inf, logits_l2=e_l2, loss=e_loss, grads=g, N=N,
           grads_finite=gfinite)
    return e_inf, e_loss, g, gfinite


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
def test_large_fine_stage_full_depth_forward_loss_vs_oracle(dev, precision):
    """BASELINE config 4 (musiclm_large fine stage): all 24 layers, 16 heads, N = 1817: loss and logits vs the oracle."""
    e_inf, e_loss, _, gfinite = _large_fine(dev, 24, precision, with_grads=False)
    # bf16 operand rounding accumulates with depth: 24 layers measured 1.55e-2 (6 layers: 7e-3); the bar for this depth is 2.5e-2
    # fp16ff: FF-in + FF-out exact removes 88 % of fp16's error variance at this depth (profiles/r05_error_budget.md: 1.7e-3 -> 6.1e-4); bar 1e-3
    bar = {"bf16": 2.5e-2, "fp16": 4e-3, "fp16ff": 1e-3}.get(precision, TOL[precision]["logits"])
    assert e_inf < bar and e_loss < TOL[precision]["loss"] and gfinite, (e_inf, e_loss)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
def test_large_fine_stage_gradients_vs_oracle(dev, precision):
    """Same shapes (16 heads, N = 1817, 5 fine quantizers) at depth 2 so that the oracle's autograd fits a test: grads."""
    e_inf, e_loss, g, _ = _large_fine(dev, 2, precision, with_grads=True)
    assert e_inf < TOL[precision]["logits"] and e_loss < TOL[precision]["loss"], (e_inf, e_loss)
    assert max(g.values()) < TOL[precision]["grad"], g


def test_trainer_steps_and_checkpoint_roundtrip(dev, tmp_path):
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd.trainer import SingleStageTrainer
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.1,
                                        precision="bf16").to(dev)
    ds = SyntheticTokenDataset("coarse", length=8, coarse_window_seconds=1, semantic_window_seconds=2)
    tr = SingleStageTrainer(model, "coarse", num_train_steps=30, batch_size=2, dataset=ds, lr=3e-3, lr_warmup=5,
                            grad_accum_every=2, wd=0.01, max_grad_norm=0.5, valid_frac=0.0, save_results_every=1000,
                            save_model_every=1000, results_folder=str(tmp_path / "res"), save_predicted_tokens=False,
                            save_reconstructed_wave=False)
    losses = [tr.train_step()["loss"] for _ in range(30)]
    report("trainer", first=losses[0], last=losses[-1])
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 0.9          # memorises 8 samples
    mp, op, sp = (str(tmp_path / f"coarse.{n}.5.pt") for n in ("transformer", "optimizer", "scheduler"))
    tr.save(mp, op, sp)
    sd = torch.load(mp)
    assert list(sd.keys()) == list(model.state_dict().keys())
    before = {k: v.clone() for k, v in model.state_dict().items()}
    tr.train_step()
    tr.steps.zero_()
    tr.load(mp, op, sp, steps=6)
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    assert int(tr.steps.item()) == 6


def test_fp16_overflow_is_skipped_and_the_loss_scale_backs_off(dev, tmp_path):
    """precision "fp16" through SingleStageTrainer with an absurd initial loss scale (2^26): the first backward passes overflow half's range,
    so the fused optimizer must (a) leave weights, both Adam moments and its Adam clock untouched, (b) clear the gradients, (c) halve the
    device-side scale and count the skip -- every step, with no host read inside the captured micro-step -- until the scale is workable;
    then training proceeds (finite, falling loss), the checkpoint's `step` is the number of APPLIED steps, and clipping switched off
    (max_grad_norm = 0) still guards (the guard used to exist only with clipping on, and the clock advanced on skips)."""
    from open_musiclm_amd import engine
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd.trainer import SingleStageTrainer
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0, precision="fp16").to(dev)
    engine.loss_scale_state(model)[0] = float(2 ** 26)
    ds = SyntheticTokenDataset("coarse", length=8, coarse_window_seconds=1, semantic_window_seconds=2)
    tr = SingleStageTrainer(model, "coarse", num_train_steps=60, batch_size=2, dataset=ds, lr=3e-3, lr_warmup=0, grad_accum_every=1, wd=0.01,
                            max_grad_norm=0.0, valid_frac=0.0, save_results_every=1000, save_model_every=1000,
                            results_folder=str(tmp_path / "res"), save_predicted_tokens=False, save_reconstructed_wave=False)
    tr.optim.zero_grad()
    f = tr.optim._flat
    P0, M0, V0 = f["P"].clone(), f["M"].clone(), f["V"].clone()
    logs = tr.train_step()
    rep = tr.optim.loss_scale_report()
    assert rep["skipped_steps"] == 1 and rep["applied_steps"] == 0 and rep["scale"] == 2 ** 25, rep
    assert torch.equal(f["P"], P0) and torch.equal(f["M"], M0) and torch.equal(f["V"], V0)            # nothing reached the state
    assert float(f["G"].abs().max()) == 0.0                                                            # the poisoned gradients were cleared
    assert np.isfinite(logs["loss"]) and logs["skipped_steps"] == 1
    losses = [tr.train_step()["loss"] for _ in range(45)]
    rep = tr.optim.loss_scale_report()
    assert 1 < rep["skipped_steps"] < 26 and rep["applied_steps"] == 46 - rep["skipped_steps"], rep
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] * 0.9, (losses[0], losses[-1])
    assert torch.isfinite(f["P"]).all() and not torch.equal(f["P"], P0)
    sd = tr.optim.state_dict()
    assert int(float(sd["state"][0]["step"])) == rep["applied_steps"] and sd["omlm_loss_scale"]["scale"] == rep["scale"]
    report("fp16_overflow", **rep, first_loss=losses[0], last_loss=losses[-1])
    # resume: the restored loss-scale block carries the cumulative skip count; the first step of the resumed trainer must
    # not report those skips again (it used to rewind the LR warm-up a second time)
    tr.save(str(tmp_path / "m.pt"), str(tmp_path / "o.pt"))
    tr.load(str(tmp_path / "m.pt"), str(tmp_path / "o.pt"))
    assert tr._skipped_seen == rep["skipped_steps"]
    logs = tr.train_step()
    assert np.isfinite(logs["loss"]) and logs["skipped_steps"] - rep["skipped_steps"] in (0, 1)          # (a fresh overflow may still be skipped)


@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp16", "fp16ff"])
def test_graphed_bf16x3_training_tracks_eager_over_optimizer_steps(dev, precision):
    """bf16x3 at a size where the fp32 GEMMs take the hi/lo-plane route (M N K >= ops._X3_MIN_MACS), captured into a HIP graph,
    over several optimizer steps: the replayed micro-step must read the CURRENT weights -- the planes of the persistent Parameter
    objects (Wq / Wkv / Wo) are split inside the graph, not left over from the eager warm-up -- so its losses and final weights
    track an eager run from the same start.  (lr is large so that stale projection weights would show within two steps.)"""
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd import ops
    from open_musiclm_amd.graph import GraphedForwardBackward
    from open_musiclm_amd.optimizer import get_optimizer
    B, n_sem, n_coarse = 2, 199, 128
    g = torch.Generator().manual_seed(3)
    batches = [[torch.randint(0, 1024, (B, 12, 1), generator=g).to(dev), torch.randint(0, 1024, (B, n_sem), generator=g).to(dev),
                torch.randint(0, 1024, (B, n_coarse, 3), generator=g).to(dev)] for _ in range(4)]
    keys = ("clap_token_ids", "semantic_token_ids", "coarse_token_ids")
    assert B * (3 + 13 + n_sem + 1 + 3 * n_coarse) * 256 * 256 >= ops._X3_MIN_MACS        # the to_q / to_out GEMMs are on the plane route

    def run(use_graph):
        torch.manual_seed(0)
        model = M.create_coarse_transformer(dim=256, depth=2, heads=4, num_coarse_quantizers=3, ff_dropout=0.0,
                                            precision=precision).to(dev)
        stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.0)
        stage.train()
        opt = get_optimizer(model.parameters(), lr=3e-3, wd=0.01)
        opt.zero_grad()
        fb = GraphedForwardBackward(lambda **kw: stage(**kw, return_loss=True)[0], enabled=use_graph, instances=1)

        def discard():
            opt.mark_grads_dirty()
            opt.zero_grad()
        fb.prepare(dict(zip(keys, batches[0])), after_warmup=discard)
        assert (fb.graph is not None) == use_graph, fb.capture_error
        losses = []
        for k in range(4):
            opt.zero_grad()
            loss = fb(**dict(zip(keys, batches[k])))
            opt.mark_grads_dirty()
            opt.step(max_grad_norm=0.5)
            losses.append(float(loss))
        wq, wo = model.transformer.layers[0][0].to_q.weight, model.transformer.layers[1][0].to_out[0].weight
        out = (losses, wq.detach().clone(), wo.detach().clone())
        if use_graph:
            # the decisive check: rewrite a projection weight through its storage (what the fused optimizer does: no version bump) and
            # replay -- the captured micro-step must see it exactly like an eager run on the same weights does
            with torch.no_grad():
                wq.data.mul_(-0.5)
                wo.data.mul_(0.25)
            kw = dict(zip(keys, batches[1]))
            l_replay = float(fb(**kw))
            l_eager = float(fb._eager({k: v.clone() for k, v in kw.items()}))
            assert abs(l_replay - l_eager) <= (1e-5 if precision == "bf16x3" else 1e-4) * abs(l_eager), (l_replay, l_eager)
            assert abs(l_replay - losses[1]) > 1e-3 * abs(l_eager), "the rewritten weights must matter for this check to mean anything"
        import gc
        gc.unfreeze()
        return out

    le, wq_e, wo_e = run(False)
    lg, wq_g, wo_g = run(True)
    report(f"graph_{precision}_vs_eager", eager=le, graphed=lg, wq=relerr(wq_g, wq_e), wo=relerr(wo_g, wo_e))
    assert le[0] != le[-1]
    # 16-bit modes (round 4: they are the ones whose attention backward carried a memset node in the captured step): the order-of-addition
    # noise of the atomics passes through operand roundings, so two runs agree to ~1e-3 in the loss after three Adam steps at lr 3e-3
    for a, b in zip(le, lg):
        assert abs(a - b) <= (2e-4 if precision == "bf16x3" else 5e-3) * abs(a), (le, lg)
    # Adam's normalised update amplifies the atomics' summation-order noise on near-zero gradients: weights agree to ~lr, not to rounding
    wbar = 3e-2 if precision == "bf16x3" else 0.15
    assert relerr(wq_g, wq_e) < wbar and relerr(wo_g, wo_e) < wbar


@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_graph_replay_is_complete_when_the_next_kernel_starts(dev, precision):
    """Stream order across a HIP-graph replay: the squared gradient norm taken by a kernel enqueued RIGHT AFTER the replay (what the fused
    optimizer does) must equal the one taken after a device-wide synchronise.  Round 4 found the 16-bit attention backward's hipMemsetAsync
    (a memset node in the captured micro-step) detaching the rest of the backward from the graph's completion: the optimizer's kernels
    overlapped the backward's tail (seen as a permanently non-finite fp16 run after one overflow).  The fill is a kernel node now."""
    from open_musiclm_amd import ops
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.graph import GraphedForwardBackward
    from open_musiclm_amd.optimizer import get_optimizer
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=1024, depth=2, heads=8, num_coarse_quantizers=3, ff_dropout=0.1, precision=precision).to(dev)
    stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.]).train()
    opt = get_optimizer(model.parameters(), lr=1e-4, wd=0.01)
    opt.zero_grad()
    g = torch.Generator().manual_seed(4)
    kw = dict(clap_token_ids=torch.randint(0, 1024, (4, 12, 1), generator=g).to(dev), semantic_token_ids=torch.randint(0, 1024, (4, 199), generator=g).to(dev),
              coarse_token_ids=torch.randint(0, 1024, (4, 300, 3), generator=g).to(dev))
    fb = GraphedForwardBackward(lambda **k: stage(**k, return_loss=True, return_logits=False)[0])
    fb.prepare(kw, after_warmup=lambda: (opt.mark_grads_dirty(), opt.zero_grad()))
    assert fb.graph is not None, fb.capture_error
    G = opt.flat_grad
    n1, n2 = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
    part = torch.empty(2048, device=dev)
    worst = 0.0
    for rep in range(6):
        G.zero_()
        torch.cuda.synchronize()
        fb(**kw)
        n1.zero_()
        ops.sumsq_accumulate(G, n1, part)                   # enqueued behind the replay, no host wait in between
        torch.cuda.synchronize()
        n2.zero_()
        ops.sumsq_accumulate(G, n2, part)
        torch.cuda.synchronize()
        a, b = float(n1), float(n2)
        assert np.isfinite(a) and np.isfinite(b) and b > 0
        worst = max(worst, abs(a - b) / b)
    report(f"graph_replay_complete[{precision}]", worst_rel_diff=worst)
    assert worst == 0.0, worst


def test_train_coarse_stage_script_flow_on_a_preprocessed_store(dev, golden_dir, tmp_path):
    """The call sequence of the reference's scripts/train_coarse_stage.py (:33-75) -- JSON configs -> load_model_config /
    load_training_config -> create_coarse_transformer_from_config -> create_single_stage_trainer_from_config(..., accelerate_kwargs with
    log_with / logging_dir, config_paths) -> trainer.train() -- on a sqlite token store written with the reference's own adapters
    (tests/golden/preprocessed, oracle/make_golden_r2.py), `use_preprocessed_data` as in configs/training/train_fma_preprocess.json.
    The script file itself cannot run here: it lives in /root/reference (absent on the GPU box) and constructs the `encodec`
    package's model unconditionally (:51-52; not installed) -- a stand-in codec object takes that one slot; everything else is the
    script's own sequence, through checkpoint files and a resume."""
    import json
    from open_musiclm_amd.config import (create_coarse_transformer_from_config, create_single_stage_trainer_from_config,
                                         load_model_config, load_training_config)
    stage_cfg = dict(dim=128, depth=2, heads=2, attn_dropout=0.0, ff_dropout=0.1, grad_shrink_alpha=0.1, non_causal_prefix_size=0,
                     relative_position_bias_type="continuous", use_memory_efficient_attention=False)
    model_cfg = {
        "global_cfg": dict(semantic_audio_length_seconds=4.0, coarse_audio_length_seconds=2.0, fine_audio_length_seconds=1.0,
                           clap_audio_length_seconds=4.0, num_coarse_quantizers=3, num_fine_quantizers=5),
        "clap_rvq_cfg": dict(enable_fusion=False, rq_num_quantizers=12, codebook_size=1024, rq_ema_decay=0.95, threshold_ema_dead_code=0.5),
        "hubert_kmeans_cfg": dict(model_name="m-a-p/MERT-v0", normalize_embeds=True, embed_layer=7, target_sample_hz=16000,
                                  seq_len_multiple_of=320, codebook_size=1024, output_hz=5),
        "encodec_cfg": dict(bandwidth=6.0, codebook_size=1024, output_hz=3),
        "semantic_cfg": stage_cfg, "coarse_cfg": stage_cfg, "fine_cfg": stage_cfg}
    stage_train = dict(folder=os.path.join(golden_dir, "preprocessed"), valid_frac=0.0, lr=3e-3, lr_warmup=2, batch_size=2,
                       grad_accum_every=2, wd=0.01, max_grad_norm=0.5, num_train_steps=4, save_results_every=2, save_model_every=2,
                       save_predicted_tokens=True, save_reconstructed_wave=False, use_preprocessed_data=True)
    train_cfg = {
        "clap_rvq_trainer_cfg": dict(folder="./data", num_train_steps=1, batch_size=2, accumulate_batches=1, save_model_every=10, save_results_every=5),
        "hubert_kmeans_trainer_cfg": dict(folder="./data", feature_extraction_num_steps=1, feature_extraction_batch_size=2),
        "semantic_trainer_cfg": dict(stage="semantic", cross_entropy_loss_weights=[0.0, 1.0], **stage_train),
        "coarse_trainer_cfg": dict(stage="coarse", cross_entropy_loss_weights=[0.0, 0.0, 1.0], **stage_train),
        "fine_trainer_cfg": dict(stage="fine", cross_entropy_loss_weights=[0.0, 0.0, 1.0], **stage_train),
        "data_preprocessor_cfg": dict(folder="./data", metadata_folder="./meta", results_folder="./pre", max_audio_length_seconds=30,
                                      random_crop=True, num_crops=1, clap_batch_size=32)}
    mp, tp = tmp_path / "model.json", tmp_path / "train.json"
    mp.write_text(json.dumps(model_cfg)); tp.write_text(json.dumps(train_cfg))
    model_config, training_config = load_model_config(str(mp)), load_training_config(str(tp))
    assert training_config.coarse_trainer_cfg.use_preprocessed_data

    class Codec:                                  # the one slot the script fills from the `encodec` package
        codebook_size, sample_rate, output_hz = 1024, 24000, 3

        def to(self, device):
            return self
    torch.manual_seed(0)
    results = tmp_path / "results" / "coarse"
    coarse = create_coarse_transformer_from_config(model_config, None, dev)
    trainer = create_single_stage_trainer_from_config(
        model_config=model_config, training_config=training_config, stage="coarse", results_folder=str(results), transformer=coarse,
        clap=None, wav2vec=None, encodec_wrapper=Codec(), device=dev,
        accelerate_kwargs={"log_with": "tensorboard", "logging_dir": str(tmp_path / "logs")}, config_paths=[str(mp), str(tp)])
    seen = []
    trainer.train(log_fn=lambda logs: seen.append(dict(logs)))
    assert int(trainer.steps.item()) == 4 and len(seen) == 4 and all(np.isfinite(l["loss"]) for l in seen)
    assert seen[-1]["loss"] < seen[0]["loss"]
    files = sorted(os.path.basename(str(f)) for f in results.glob("coarse.*.pt"))
    assert "coarse.transformer.2.pt" in files and "coarse.optimizer.2.pt" in files, files
    assert (results / "configs" / "model.json").exists()
    # resume the way scripts/train_utils.load_checkpoint_from_args does: trainer.load(model, optim, scheduler, steps = step + 1)
    sd_path = {n: str(results / f"coarse.{n}.2.pt") for n in ("transformer", "optimizer", "scheduler")}
    coarse2 = create_coarse_transformer_from_config(model_config, None, dev)
    trainer2 = create_single_stage_trainer_from_config(
        model_config=model_config, training_config=training_config, stage="coarse", results_folder=str(tmp_path / "results2"),
        transformer=coarse2, clap=None, wav2vec=None, encodec_wrapper=Codec(), device=dev, accelerate_kwargs={}, config_paths=None)
    trainer2.load(sd_path["transformer"], sd_path["optimizer"], sd_path["scheduler"], steps=3)
    assert int(trainer2.steps.item()) == 3
    saved = torch.load(sd_path["transformer"], map_location="cpu")
    for k, v in coarse2.state_dict().items():
        assert torch.equal(v.cpu(), saved[k]), k
    assert np.isfinite(trainer2.train_step()["loss"]) and int(trainer2.steps.item()) == 4


def test_musiclm_hierarchical_decode_tokens(dev):
    """MusicLM.forward window stitching on tiny stages: shapes of the 3-level token hierarchy."""
    from open_musiclm_amd import open_musiclm as M
    torch.manual_seed(0)
    kw = dict(dim=64, depth=1, heads=1, precision="bf16")
    sem = M.create_semantic_transformer(**kw).to(dev)
    coarse = M.create_coarse_transformer(num_coarse_quantizers=3, **kw).to(dev)
    fine = M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, **kw).to(dev)
    mlm = M.MusicLM(wav2vec=None, clap=None, neural_codec=None, semantic_transformer=sem, coarse_transformer=coarse,
                    fine_transformer=fine)
    clap_ids = torch.randint(0, 1024, (1, 12, 1), device=dev)
    s, c, f = mlm.generate(clap_token_ids=clap_ids, output_seconds=2, semantic_window_seconds=1, coarse_window_seconds=1,
                           fine_window_seconds=1, semantic_steps_per_second=10, acoustic_steps_per_second=6,
                           return_tokens=True)
    report("musiclm_tokens", sem=list(s.shape), coarse=list(c.shape), fine=list(f.shape))
    assert s.shape == (1, 20, 1) and c.shape[0] == 1 and c.shape[2] == 3 and f.shape[2] == 5
    assert c.shape[1] == f.shape[1]
    assert int(c.max()) < 1024 and int(c.min()) >= 0


def test_data_parallel_step_equals_single_process_accumulation(dev, tmp_path):
    """DP equivalence (flat-buffer all-reduce + grad_scale + clip order, trainer.py:270-275 here / :428-447 in the reference):
    two ranks x one micro-batch == one rank x two accumulated micro-batches, incl. the start-up parameter broadcast (rank 1
    is built from a different seed).  Both ranks share cuda:0 and exchange through gloo, so this runs on a 1-GPU box."""
    import subprocess
    import sys
    worker = os.path.join(ROOT, "tests", "dp_equiv_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", OMLM_DP_BACKEND="gloo", LOCAL_RANK="0")
    outs = [str(tmp_path / f"dp{r}.pt") for r in range(2)]
    import time

    def run_pair(port):
        procs = [subprocess.Popen([sys.executable, worker, "dp", outs[r]], env=dict(env, RANK=str(r), WORLD_SIZE="2", MASTER_PORT=str(port)),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
        t_end = time.time() + 200                 # both ranks normally finish in ~25 s; the worker dumps its stacks and exits after 150 s
        while time.time() < t_end and any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                time.sleep(3)                      # a rank died: give the other a moment, then stop waiting for its collective
                break
            time.sleep(0.5)
        for p in procs:
            if p.poll() is None:
                p.kill()
        return [p.returncode for p in procs], [p.communicate()[0].decode() for p in procs]
    rcs, logs = run_pair(29731)
    if any(rc != 0 for rc in rcs):
        # Two processes time-sharing ONE GPU with a gloo exchange is a test-only arrangement (the product runs one rank per GPU over RCCL);
        # round 4 saw it stall once in ~20 runs right after the rendezvous.  One retry on a fresh port; the first attempt's stacks are kept.
        report("dp_equivalence_first_attempt", returncodes=str(rcs), tail0=logs[0][-1500:], tail1=logs[1][-1500:])
        rcs, logs = run_pair(29741)
    assert all(rc == 0 for rc in rcs), "\n----- rank log -----\n".join(l[-3000:] for l in logs)
    single = str(tmp_path / "single.pt")
    r = subprocess.run([sys.executable, worker, "single", single], env=dict(env, RANK="0", WORLD_SIZE="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    assert r.returncode == 0, r.stdout.decode()
    a, b, c = torch.load(outs[0]), torch.load(outs[1]), torch.load(single)
    assert a["world"] == 2 and c["world"] == 1
    assert torch.equal(a["param"], b["param"]), "replicas diverged"
    e_g = relerr(a["grad"], c["grad"])
    dp = (a["param"] - c["param"]).abs()
    e_p, e_mean, frac = float(dp.max()), float(dp.mean()), float((dp > 2e-4).float().mean())
    report("dp_equivalence", grad_rel=e_g, param_abs=e_p, param_mean_abs=e_mean, frac_above_2e4=frac)
    assert e_g < 1e-4, e_g                       # same arithmetic, different summation order (atomics, all-reduce)
    # Two AdamW steps at lr 1e-3.  The first Adam update is lr * sign(g): an element whose gradient is at the rounding level of
    # the two summation orders may take opposite signs (up to 2 * lr apart per step), so the maximum is bounded by 4 * lr and the
    # comparison that carries information is how RARE such elements are and how small the mean difference is.
    assert e_p <= 4.1e-3 and frac < 1e-3 and e_mean < 2e-5, (e_p, frac, e_mean)


def test_bench_two_rank_dry_run_on_one_gpu(dev):
    """`bench.py --gpus 2` the way the driver launches it (torch.distributed.run, one rank per GPU) -- here both ranks share cuda:0 and
    exchange through gloo (RCCL refuses two ranks on one device): rendezvous, per-rank data, the flat gradient all-reduce, the barrier /
    max-over-ranks timing and ONE JSON line from rank 0 with n_gpus = 2 and the whole-job rate.  Says nothing about RCCL speed."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, OMLM_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29677", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "2"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-2 * out["value"] and np.isfinite(out["final_loss"])
    assert "legs" not in out and "cpu_baseline" not in out


def test_bench_self_launches_without_a_launcher(dev):
    """`python bench.py --gpus 2 ...` with NO launcher environment (the form the round-3 driver used): bench.py re-executes itself under
    torch.distributed.run; on this 1-GPU box the ranks share cuda:0 and DataParallel falls back to gloo by itself (and says so in the line)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "OMLM_DP_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-legs", "--batch", "2"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    if torch.cuda.device_count() < 2:
        assert "DRY RUN" in out["config"]["exchange"] and "gloo" in out["config"]["exchange"]
    else:
        assert "nccl" in out["config"]["exchange"]
    assert np.isfinite(out["final_loss"])


@pytest.mark.parametrize("stage,B", [("coarse", 5), ("fine", 2), ("semantic", 3)])
def test_fused_batch_preparation_equals_the_torch_path(dev, stage, B, monkeypatch):
    """omlm_prepare_train_batch (one launch) against the wrapper's torch construction (open_musiclm.py:340-376 + :116-130 + utils.py:49-56
    restated in _prepare / engine.build_ids / generate_mask_with_prob): ids32, key mask (with the forgetful mask from the SAME randn
    draw, pads in the conditioning sequences, engineered ties in the scores) and labels bit for bit; then the training loss of the two
    paths from the same RNG state is the same number."""
    from open_musiclm_amd import engine, ops
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.utils import generate_mask_with_prob
    torch.manual_seed(0)
    kw = dict(dim=64, depth=1, heads=1, ff_dropout=0.0, precision="bf16")
    if stage == "coarse":
        model = M.create_coarse_transformer(num_coarse_quantizers=3, **kw).to(dev)
        shapes, weights = [(B, 12, 1), (B, 37), (B, 40, 3)], [0., 0., 1.]
    elif stage == "fine":
        model = M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, **kw).to(dev)
        shapes, weights = [(B, 12, 1), (B, 21, 3), (B, 21, 5)], [0., 0., 1.]
    else:
        model = M.create_semantic_transformer(**kw).to(dev)
        shapes, weights = [(B, 12, 1), (B, 131)], [0., 1.]
    g = torch.Generator().manual_seed(11)
    raw = [torch.randint(0, 1024, sh, generator=g).to(dev) for sh in shapes]
    raw[0][0, 3:] = -1                                     # padded conditioning ids
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=weights, mask_prob=0.15)
    wrapper.train()
    flat = [t.reshape(B, -1).long().contiguous() for t in raw]
    N = sum(t.shape[1] + 1 for t in flat) + len(flat) - 1
    n_drop = min(int(N * 0.15), N - 1)
    scores = torch.randn(B, N, generator=g).to(dev)
    scores[1, 5:9] = scores[1].topk(n_drop).values[-1]     # ties at the threshold: exactly n_drop positions must still go
    seqs = model.token_sequences
    ids32, keymask, labels, lens = ops.prepare_train_batch(flat, [int(e) for e in wrapper.eos_ids], [s_.num_quantizers for s_ in seqs],
                                                           [s_.codebook_size for s_ in seqs], wrapper.pad_id, scores, n_drop, [True] * len(flat))
    # torch path on the same scores
    monkeypatch.setattr(M, "generate_mask_with_prob", lambda shape, p, device: torch.ones(shape, device=device, dtype=torch.bool))
    t_ids, t_labels, t_mask = wrapper._prepare(raw, True, False)
    t_ids32, t_lens = engine.build_ids(model, t_ids)
    sc = scores.clone(); sc[:, 0] = torch.finfo(sc.dtype).min
    assert torch.equal(ids32, t_ids32) and list(lens) == list(t_lens)
    for a_, b_ in zip(labels, t_labels):
        assert torch.equal(a_.long(), b_)
    assert int((keymask == 0).sum()) >= B * n_drop and torch.equal(keymask[:, 0], torch.ones(B, dtype=torch.uint8, device=dev))
    for b in range(B):
        dropped = (~keymask[b].bool()) & t_mask[b]         # positions the forgetful mask removed
        assert int(dropped.sum()) + int((~t_mask[b]).sum() - ((~t_mask[b]) & keymask[b].bool()).sum()) >= 0
        kth = sc[b].topk(n_drop).values[-1]
        forgot = torch.zeros(N, dtype=torch.bool, device=dev)
        forgot[(sc[b] > kth).nonzero().flatten()] = True
        eq = (sc[b] == kth).nonzero().flatten()
        forgot[eq[: n_drop - int(forgot.sum())]] = True     # ties: lowest indices first
        assert int(forgot.sum()) == n_drop
        assert torch.equal(keymask[b].bool(), t_mask[b] & ~forgot), b
    monkeypatch.undo()
    # end to end: both paths draw the same randn -> the same loss
    torch.manual_seed(123)
    loss_f, logits_f, labels_f = wrapper(all_token_ids=raw, return_loss=True, return_logits=False)
    monkeypatch.setenv("OMLM_FUSED_PREP", "0")
    torch.manual_seed(123)
    loss_t, _, _ = wrapper(all_token_ids=raw, return_loss=True, return_logits=False)
    # (the loss sums its rows with atomics: equal up to the order of addition)
    assert labels_f[-1].dtype == torch.int32 and abs(float(loss_f) - float(loss_t)) < 2e-6 * abs(float(loss_t)), (float(loss_f), float(loss_t))
    loss_f.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)


def test_generate_top_match_ranks_samples_by_clap_similarity(dev, monkeypatch):
    """MusicLM.generate_top_match (open_musiclm.py:1039-1071): num_samples waves per prompt through the hierarchical decode, re-embedded
    by CLAP and ranked by cosine similarity with the prompt's text embedding.  The towers are stand-ins (weights unobtainable offline):
    a text -> ids / text -> embedding / audio -> embedding object with the reference's call signature, a codec that turns ids into a
    waveform, and a pass-through `torchaudio.functional.resample` (equal rates).  Checked: shapes, the returned similarities are the
    top-k of ALL samples' similarities recomputed here from the returned order, sorted descending, and the waves are the matching rows."""
    import sys
    import types
    from open_musiclm_amd import open_musiclm as M
    ta, taf = types.ModuleType("torchaudio"), types.ModuleType("torchaudio.functional")
    taf.resample = lambda wave, a, b: wave if a == b else torch.nn.functional.interpolate(wave[:, None], scale_factor=b / a, mode="linear")[:, 0]
    ta.functional = taf
    monkeypatch.setitem(sys.modules, "torchaudio", ta)
    monkeypatch.setitem(sys.modules, "torchaudio.functional", taf)
    torch.manual_seed(0)
    tiny = dict(dim=64, depth=1, heads=1, ff_dropout=0.0)
    cb = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    sem = M.create_semantic_transformer(**tiny, clap_codebook_size=32, semantic_codebook_size=48, precision="bf16x3").to(dev)
    coarse = M.create_coarse_transformer(**tiny, num_coarse_quantizers=3, precision="bf16x3", **cb).to(dev)
    fine = M.create_fine_transformer(**tiny, num_coarse_quantizers=3, num_fine_quantizers=5, clap_codebook_size=32, acoustic_codebook_size=40,
                                     precision="bf16x3").to(dev)

    class Clap:                                     # ClapQuantized's call signature (clap_quantized.py:57-87)
        sample_rate = 16000
        seen_audio = []

        def __call__(self, *, text_input=None, audio_input=None, return_embedding=False):
            if text_input is not None:
                h = torch.tensor([[sum(map(ord, t)) % 29 + i for i in range(12)] for t in text_input], device=dev)
                if not return_embedding:
                    return (h % 32)[:, :, None]                                  # [B, 12, 1] conditioning ids
                return torch.nn.functional.normalize(h.float(), dim=-1)
            assert return_embedding and audio_input.dim() == 2
            Clap.seen_audio.append(audio_input.clone())
            feats = torch.stack([audio_input[:, i::12].mean(-1) for i in range(12)], dim=-1)
            return feats + 0.01

    class Codec:
        sample_rate = 16000

        def decode_from_codebook_indices(self, ids):                             # [B, T, Q] -> [B, 1, T * 8]: any deterministic map will do
            w = (ids.float().mean(-1) / 40.0 - 0.5 + 0.05 * ids[..., 0].float().sin())
            return w.repeat_interleave(8, dim=-1)[:, None]
    mlm = M.MusicLM(wav2vec=None, clap=Clap(), neural_codec=Codec(), semantic_transformer=sem, coarse_transformer=coarse, fine_transformer=fine)
    kw = dict(output_seconds=4, semantic_window_seconds=2, coarse_window_seconds=2, fine_window_seconds=1, semantic_steps_per_second=6,
              acoustic_steps_per_second=4)                   # the window arithmetic of the golden MusicLM.forward case
    waves, sims = mlm.generate_top_match(text=["a slow piano", "drums"], num_samples=4, num_top_matches=2, **kw)
    assert len(waves) == len(sims) == 2 and len(Clap.seen_audio) == 2
    for p, (w, sm, audio) in enumerate(zip(waves, sims, Clap.seen_audio)):
        assert w.shape[0] == 2 and sm.shape == (2,) and w.shape[1] == audio.shape[1]
        text_lat = Clap()(text_input=[["a slow piano", "drums"][p]], return_embedding=True).repeat(4, 1)
        all_sim = torch.nn.functional.cosine_similarity(text_lat, Clap()(audio_input=audio, return_embedding=True), dim=-1).cpu()
        Clap.seen_audio.pop()                                                    # the recomputation above logged itself
        top = all_sim.topk(2, sorted=True)
        assert torch.allclose(sm, top.values, atol=1e-6) and sm[0] >= sm[1]
        # int16 round trip of the reference (utils.int16_to_float32(float32_to_int16(.))) is what CLAP saw; the returned waves are the raw rows
        assert w.shape == (2, audio.shape[1])
    report("generate_top_match", sims=[[float(v) for v in s_] for s_ in sims])


@pytest.mark.parametrize("use_cache", [True, False])
def test_musiclm_forward_matches_reference_golden_tokens(golden_dir, dev, monkeypatch, use_cache):
    """MusicLM.forward (open_musiclm.py:864-1035 of the reference) token-level parity: the reference ran on tiny stages with
    injected conditioning ids (oracle/make_golden_r2.py); every stage.generate output of its sliding-window decode and the final
    [coarse | fine] ids must be reproduced bit for bit when the same uniform draws are replayed (bf16x3)."""
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "musiclm_forward.npz"))
    tiny, kw = ast.literal_eval(str(z["meta.tiny"])), ast.literal_eval(str(z["meta.kwargs"]))
    cb = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    sem = M.create_semantic_transformer(**tiny, clap_codebook_size=32, semantic_codebook_size=48, precision="bf16x3")
    coarse = M.create_coarse_transformer(**tiny, num_coarse_quantizers=3, precision="bf16x3", **cb)
    fine = M.create_fine_transformer(**tiny, num_coarse_quantizers=3, num_fine_quantizers=5, clap_codebook_size=32,
                                     acoustic_codebook_size=40, precision="bf16x3")
    for pfx, m in (("sem", sem), ("coarse", coarse), ("fine", fine)):
        m.load_state_dict({k[len(f"sd.{pfx}."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"sd.{pfx}.")}, strict=True)
        m.to(dev)
    mlm = M.MusicLM(wav2vec=None, clap=None, neural_codec=None, semantic_transformer=sem, coarse_transformer=coarse,
                    fine_transformer=fine)
    n_calls = int(z["n_calls"])
    state = dict(i=0)
    got_calls = []

    def source(n, batch, v1):                      # the draws of the reference's i-th stage.generate call, in order
        u = torch.from_numpy(z[f"call.{state['i']}.uniforms"])
        assert u.shape == (n, batch, v1), (state["i"], tuple(u.shape), (n, batch, v1))
        return u
    monkeypatch.setattr(M, "UNIFORM_SOURCE", source)
    for name in ("semantic", "coarse", "fine"):
        stage = getattr(mlm, name)
        orig = stage.generate

        def wrapped(*a, _orig=orig, _name=name, **k):
            assert str(z[f"call.{state['i']}.stage"]) == _name
            out = _orig(*a, use_cache=use_cache, **k)
            got_calls.append(out.cpu().numpy())
            state["i"] += 1
            return out
        monkeypatch.setattr(stage, "generate", wrapped)
    s, c, f = mlm.generate(clap_token_ids=torch.from_numpy(z["clap_ids"]).to(dev), return_tokens=True, **kw)
    assert state["i"] == n_calls
    for i, g in enumerate(got_calls):
        assert np.array_equal(g, z[f"call.{i}.ids"]), (i, str(z[f"call.{i}.stage"]))
    acoustic = torch.cat([c, f], dim=-1).cpu().numpy()
    report(f"musiclm_forward_golden[cache={use_cache}]", calls=n_calls, acoustic_shape=list(acoustic.shape))
    assert np.array_equal(acoustic, z["acoustic"])


@pytest.mark.parametrize("use_cache", [True, False])
def test_generate_with_eos_flags_matches_reference_golden_ids(golden_dir, dev, use_cache):
    """wrapper.generate(allow_eos_in_output=True, include_eos_in_output False / True) (open_musiclm.py:309-313, :321-322): the reference's
    ids for near-uniform sampling on a 6-code acoustic book (oracle/make_golden_r5.py) -- eos ids are sampled on last-quantizer steps,
    EMBEDDED by the steps that follow (offset aliasing :126-130: the cached decode's sampler gathers that row too) and masked at the end.
    Bit for bit, KV-cached and re-forward paths (bf16x3)."""
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "generate_eos.npz"))
    kwargs = ast.literal_eval(str(z["meta.kwargs"]))
    model = M.create_coarse_transformer(**kwargs, precision="bf16x3")
    model.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}, strict=True)
    model.to(dev)
    wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False)
    cond = [torch.from_numpy(z["cond.0"]).to(dev), torch.from_numpy(z["cond.1"]).to(dev)]
    kw = dict(conditioning_token_ids=cond, max_time_steps=int(z["max_time_steps"]), temperature=float(z["temperature"]),
              filter_thres=float(z["filter_thres"]), uniforms=torch.from_numpy(z["uniforms"]), use_cache=use_cache)
    a = wrapper.generate(allow_eos_in_output=True, **kw).cpu().numpy()
    b = wrapper.generate(allow_eos_in_output=True, include_eos_in_output=True, **kw).cpu().numpy()
    c = wrapper.generate(**kw).cpu().numpy()
    eos = model.eos_ids[-1]
    assert (z["generated_allow_include"] == eos).any() and (z["generated_allow"] == -1).any()      # the fixture does exercise both rules
    report(f"generate_eos_golden[cache={use_cache}]", allow=bool(np.array_equal(a, z["generated_allow"])),
           include=bool(np.array_equal(b, z["generated_allow_include"])), default=bool(np.array_equal(c, z["generated_default"])))
    assert np.array_equal(a, z["generated_allow"])
    assert np.array_equal(b, z["generated_allow_include"])
    assert np.array_equal(c, z["generated_default"])


@pytest.mark.parametrize("use_cache", [True, False])
def test_musiclm_forward_with_prime_wave_matches_reference_golden(golden_dir, dev, monkeypatch, use_cache):
    """MusicLM.forward's audio-continuation branch (open_musiclm.py:896-926): the reference ran with a stereo 3 s prime on tiny stages and
    stand-in tokenizers (oracle/make_golden_r5.py).  Pinned here: the two prepare_audio results handed to the tokenizers (channel fold,
    normalisation for wav2vec only, truncation to the semantic window, int16 round trip), the conditioning slices and token adjustments of
    all three stages (every stage.generate output), and the final [prime | generated] acoustic ids -- bit for bit, same draws (bf16x3)."""
    from open_musiclm_amd import open_musiclm as M
    z = np.load(os.path.join(golden_dir, "musiclm_forward_prime.npz"))
    tiny, kw = ast.literal_eval(str(z["meta.tiny"])), ast.literal_eval(str(z["meta.kwargs"]))
    cb = dict(clap_codebook_size=32, semantic_codebook_size=48, acoustic_codebook_size=40)
    sem = M.create_semantic_transformer(**tiny, clap_codebook_size=32, semantic_codebook_size=48, precision="bf16x3")
    coarse = M.create_coarse_transformer(**tiny, num_coarse_quantizers=3, precision="bf16x3", **cb)
    fine = M.create_fine_transformer(**tiny, num_coarse_quantizers=3, num_fine_quantizers=5, clap_codebook_size=32,
                                     acoustic_codebook_size=40, precision="bf16x3")
    for pfx, m in (("sem", sem), ("coarse", coarse), ("fine", fine)):
        m.load_state_dict({k[len(f"sd.{pfx}."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"sd.{pfx}.")}, strict=True)
        m.to(dev)
    rate = int(z["rate"])
    seen, captured = {}, {}

    class Clap:
        def __call__(self, *, text_input=None, audio_input=None, **k):
            return torch.from_numpy(z["clap_ids"]).to(dev)

    class Wav2vec:
        target_sample_hz, codebook_size = rate, 48

        def __call__(self, wav, flatten=False, **k):
            seen["wav2vec_in"] = wav.detach().cpu().numpy()
            return torch.from_numpy(z["sem_prime"]).to(dev)

    class Codec:
        sample_rate = rate

        def eval(self):
            return self

        def __call__(self, wav, return_encoded=True, **k):
            seen["codec_in"] = wav.detach().cpu().numpy()
            return None, torch.from_numpy(z["ac_prime"]).to(dev), None

        def decode_from_codebook_indices(self, ids):
            captured["acoustic"] = ids.cpu().numpy()
            return torch.zeros(ids.shape[0], 1, 8)

    mlm = M.MusicLM(wav2vec=Wav2vec(), clap=Clap(), neural_codec=Codec(), semantic_transformer=sem, coarse_transformer=coarse,
                    fine_transformer=fine)
    n_calls = int(z["n_calls"])
    state, got_calls = dict(i=0), []

    def source(n, batch, v1):
        u = torch.from_numpy(z[f"call.{state['i']}.uniforms"])
        assert u.shape == (n, batch, v1), (state["i"], tuple(u.shape), (n, batch, v1))
        return u
    monkeypatch.setattr(M, "UNIFORM_SOURCE", source)
    for name in ("semantic", "coarse", "fine"):
        stage = getattr(mlm, name)
        orig = stage.generate

        def wrapped(*a, _orig=orig, _name=name, **k):
            assert str(z[f"call.{state['i']}.stage"]) == _name
            out = _orig(*a, use_cache=use_cache, **k)
            got_calls.append(out.cpu().numpy())
            state["i"] += 1
            return out
        monkeypatch.setattr(stage, "generate", wrapped)
    mlm(text=["x"], prime_wave=torch.from_numpy(z["prime_wave"]), prime_wave_sample_hz=rate, **kw)
    assert np.array_equal(seen["wav2vec_in"], z["wav2vec_in"]) and np.array_equal(seen["codec_in"], z["codec_in"])
    assert state["i"] == n_calls
    for i, g in enumerate(got_calls):
        assert np.array_equal(g, z[f"call.{i}.ids"]), (i, str(z[f"call.{i}.stage"]))
    report(f"musiclm_forward_prime_golden[cache={use_cache}]", calls=n_calls, acoustic_shape=list(captured["acoustic"].shape))
    assert np.array_equal(captured["acoustic"], z["acoustic"])


def test_rccl_allreduce_of_the_flat_gradient_buffer_after_a_graph_replay(dev, tmp_path):
    """RCCL on hardware: a 1-rank `backend="nccl"` process group -- library load, HSA_ENABLE_IPC_MODE_LEGACY=0,
    communicator creation -- and the product's exchange call on the product's buffer (366.6 MB flat fp32 gradients of coarse-small) right
    behind a replay of the captured micro-step, on the current stream: the reduced buffer equals the replay's gradients (sum over one rank),
    the fused optimizer consumes it.  What stays unmeasured on a 1-GPU lease: xGMI transport and multi-rank timing."""
    import subprocess
    import sys
    out = str(tmp_path / "rccl.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29761", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("OMLM_DP_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=420)
    assert r.returncode == 0, r.stdout.decode()[-4000:]
    rep = json.load(open(out))
    report("rccl_single_rank", **rep)
    assert rep["equal"] == [True, True, True], rep                # (the third exchange is the bucketed form)
    assert rep["buckets"] >= 5
    assert rep["flat_mb"] > 360 and rep["reduce_mean"] == 2.5 and rep["gather_shape"] == [2, 3]
    assert np.isfinite(rep["loss"]) and np.isfinite(rep["grad_norm_sq"]) and rep["grad_norm_sq"] > 0


HEADLINE_MODES = ["bf16", "fp16", "fp16ff"]


@pytest.mark.parametrize("precision", HEADLINE_MODES)
def test_benchmarked_batch_forward_vs_oracle(dev, precision):
    """Model-level parity AT THE BENCHMARKED BATCH: B = 32, N = 1116 -- M = 35 712 token rows, where the
    persistent GEMM walk, the 256 x 256 tiles, the tail peeling and the attention / ConvFeedForward grids of bench.py are the ones live
    (the B = 2 / B = 1 tests never reach them) -- forward with the forgetful mask injected, loss and final-sequence logits against the
    CPU oracle (forward only: ~20 s of host time)."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    B = 32
    torch.manual_seed(0)
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.0, precision=precision).to(dev)
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ids = O.synthetic_ids(spec, B, [1, 199, 300], seed=77)
    noise = torch.randn(B, 1116, generator=torch.Generator().manual_seed(9))
    import open_musiclm_amd.open_musiclm as MM
    orig = MM.generate_mask_with_prob
    MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
    try:
        wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                       cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
        wrapper.train()
        with torch.no_grad():
            loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
        got = logits[-1].float().cpu()
        loss = float(loss)
    finally:
        MM.generate_mask_with_prob = orig
    del logits
    torch.cuda.empty_cache()
    e_inf, e_l2, o_losses, top1 = 0.0, [], [], []
    with torch.no_grad():
        for b0 in range(0, B, 8):                   # samples are independent: the oracle runs in chunks of 8 (its [b, h, N, N] scores are 320 MB each)
            o_loss, o_logits, _ = O.wrapper_forward_loss(sd, spec, [t[b0:b0 + 8] for t in ids], [0., 0., 1.], forget_noise=noise[b0:b0 + 8])
            ref = o_logits[-1]
            o_losses.append(float(o_loss))
            e_inf = max(e_inf, relerr(got[b0:b0 + 8], ref))
            e_l2.append(rel_l2(got[b0:b0 + 8], ref))
            top1.append(float((got[b0:b0 + 8].argmax(1) == ref.argmax(1)).float().mean()))
    o_mean = sum(o_losses) / len(o_losses)          # equal chunk sizes: the batch loss is the mean of the chunk losses (:407-410)
    e_loss = abs(loss - o_mean) / o_mean
    report(f"bench_batch_forward[{precision}]", logits_inf=e_inf, logits_l2=max(e_l2), loss=e_loss, top1_agree=min(top1))
    tol = TOL[precision]
    assert e_inf < tol["logits"], (e_inf, max(e_l2))
    assert e_loss < tol["loss"]


@pytest.mark.parametrize("precision", ["fp16ff", "bf16"])    # the headline mode (its backward is fp16's, reading the hi planes of the forward) and the dtype BASELINE config 2 names
def test_full_size_gradients_of_every_parameter_vs_oracle(dev, precision):
    """Every parameter tensor of the full-size coarse-small model (the B = 2 test checks 15 of them): B = 1,
    N = 1116, forgetful mask injected, all 82 non-zero gradients against the oracle's autograd.  Bars: TOL's per-tensor bar; the rel-pos MLP's
    biases (near-invariant directions of the softmax) against 1e-2 of the largest gradient, like the B = 2 test."""
    from open_musiclm_amd import open_musiclm as M
    from oracle import musiclm_oracle as O
    torch.manual_seed(3)
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, num_coarse_quantizers=3, ff_dropout=0.0, precision=precision).to(dev)
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ids = O.synthetic_ids(spec, 1, [1, 199, 300], seed=55)
    noise = torch.randn(1, 1116, generator=torch.Generator().manual_seed(11))
    pnames = [k for k, _ in model.named_parameters()]
    sdo = {k: v.clone().requires_grad_(k in pnames) for k, v in sd.items()}
    o_loss, _, _ = O.wrapper_forward_loss(sdo, spec, ids, [0., 0., 1.], forget_noise=noise)
    o_grads = dict(zip(pnames, torch.autograd.grad(o_loss, [sdo[k] for k in pnames], allow_unused=True)))
    import open_musiclm_amd.open_musiclm as MM
    orig = MM.generate_mask_with_prob
    MM.generate_mask_with_prob = lambda shape, p, device: O.forgetful_mask_from_noise(noise, p).to(device)
    try:
        wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False,
                                                       cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.15)
        wrapper.train()
        loss, _, _ = wrapper(all_token_ids=[t.to(dev) for t in ids], return_loss=True)
        loss.backward()
    finally:
        MM.generate_mask_with_prob = orig
    params = dict(model.named_parameters())
    gmax = max(float(g.abs().max()) for g in o_grads.values() if g is not None)
    errs, zero = {}, []
    for k in pnames:
        og, g = o_grads[k], params[k].grad
        if og is None or float(og.abs().max()) == 0.0:          # conditioning heads (loss weight 0): no gradient on either side
            assert g is None or float(g.abs().max()) == 0.0, k
            zero.append(k)
            continue
        assert g is not None, k
        floor = 1e-2 * gmax if (k in RELPOS_TENSORS and k.endswith("bias")) else 0.0
        errs[k] = relerr(g * grad_unscale(precision), og, floor=floor)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    report(f"all_parameter_gradients[{precision}]", tensors=len(errs), zero_grad_tensors=len(zero), worst=worst,
           loss=abs(float(loss) - float(o_loss)) / float(o_loss))
    assert len(errs) + len(zero) == len(pnames) and len(errs) >= 80          # coarse-small: 84 tensors, 2 of them zero-weight heads
    # bar: ~2-3 x the worst values measured at B = 1 in fp16ff (rel-pos MLP weights 4.8e-3 / 4.6e-3, profiles/r05g_model_report.json; in plain fp16 the
    # head logit_weights.2 led with 1.48e-2 -- its forward now runs on planes)
    # bf16 (8 significand bits instead of 11: 8 x the 16-bit bars; the 15-tensor check at B = 2 measured <= 7.2e-2 per tensor)
    assert worst[0][1] < (1.5e-2 if precision == "fp16ff" else 1.5e-1) and worst[1][1] < TOL[precision]["grad"], worst


@pytest.mark.parametrize("use_graph", [True, False])
def test_relpos_once_per_optimizer_step_equals_once_per_micro_batch(dev, tmp_path, monkeypatch, use_graph):
    """engine.RelposStepCache (round 6): with gradient accumulation the trainer evaluates the rel-pos MLP once per
    optimizer step (table before the micro-batches, backward once on their summed d(table)) instead of once per micro-batch.  Same
    weights, same batches: every parameter after four optimizer steps of three micro-batches must agree with the per-micro-batch form
    (OMLM_RELPOS_CACHE=0) up to the order of fp32 additions, and the captured micro-step must not contain the MLP any more."""
    from open_musiclm_amd import engine
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd.trainer import SingleStageTrainer

    def run(cache: str):
        monkeypatch.setenv("OMLM_RELPOS_CACHE", cache)
        torch.manual_seed(0)
        model = M.create_coarse_transformer(dim=128, depth=2, heads=2, num_coarse_quantizers=3, ff_dropout=0.0, precision="bf16x3").to(dev)
        ds = SyntheticTokenDataset("coarse", length=12, coarse_window_seconds=1, semantic_window_seconds=2)
        tr = SingleStageTrainer(model, "coarse", num_train_steps=10, batch_size=2, dataset=ds, lr=1e-3, lr_warmup=0,
                                grad_accum_every=3, wd=0.01, max_grad_norm=0.5, valid_frac=0.0, save_results_every=1000,
                                save_model_every=1000, results_folder=str(tmp_path / f"res{cache}{int(use_graph)}"), save_predicted_tokens=False,
                                save_reconstructed_wave=False, use_hip_graph=use_graph)
        calls = {"n": 0}
        orig = engine.relpos_backward

        def counted(*a, **k):
            calls["n"] += 1
            return orig(*a, **k)
        monkeypatch.setattr(engine, "relpos_backward", counted)
        losses = [tr.train_step()["loss"] for _ in range(4)]
        monkeypatch.setattr(engine, "relpos_backward", orig)
        return {k: v.detach().clone() for k, v in model.state_dict().items()}, losses, calls["n"]
    p0, l0, n0 = run("0")
    p1, l1, n1 = run("1")
    worst = max(float((p1[k].double() - p0[k].double()).abs().max() / (p0[k].double().abs().max() + 1e-12)) for k in p0)
    report(f"relpos_step_cache[graph={use_graph}]", worst_param_rel=worst, backward_calls=(n0, n1), losses=(l0[-1], l1[-1]))
    # (parameters: the bias of the MLP's last layer moves every head's scores by a constant -- a direction the softmax does not see, whose
    # gradient is rounding noise on both sides and which Adam turns into +- lr steps; everything else agrees to ~1e-3 after four steps)
    assert worst < 5e-2 and all(abs(a - b) < 1e-5 * abs(a) for a, b in zip(l0, l1)), (worst, l0, l1)
    assert n1 == 4, n1                                   # one MLP backward per optimizer step (Python-side count: eager calls only)
