"""Per-op error budget of the 16-bit forward (round-5 VERDICT item 1).

Runs the full-size eval forwards of tools/make_seed_oracles.py (coarse-small depth 6, musiclm_large fine depth 24; B = 1) with every
contraction class of the trunk in IEEE half EXCEPT the ones named in a configuration, which take the hi/lo-plane ("three products",
fp32-grade) route, and reports the logits error of each configuration against the CPU oracle -- i.e. what each class's operand rounding
contributes to the fp16 mode's ~9e-4 (depth 6) / ~1.6e-3 (depth 24).

Contraction classes (transformer.py:214-333, :140-161; open_musiclm.py:163-186):
  qproj   to_q           (A = LayerNorm(x),  W = to_q.weight)
  kvproj  to_kv          (A = x un-normalised, W = to_kv.weight)
  attn    S = Q K^T and O = P V inside the attention kernel (q, k, v, o operands)
  toout   to_out         (A = o,             W = to_out.weight)
  ffin    FF-in          (A = LayerNorm(x1), W = w_in)      [its output h1 then stays fp32 into conv + GEGLU + LayerNorm(F)]
  ffout   FF-out         (A = h2,            W = w_out)     [h2 is then taken un-rounded from the conv-GEGLU-LN kernel]
  heads   logit heads    (A = final LayerNorm, W = logit_weights)
Modes per class: "16" both operands rounded to half (the fp16 mode), "32" both exact (hi/lo planes), "A" only the activation exact
(weights rounded to half first), "W" only the weights exact -- the two-product variants.

It drives the product's own kernels through ops.* (a diagnostic re-sequencing of engine.trunk_forward; nothing here is on the product path).
Usage (GPU box):  python tools/error_budget.py [out.md] [max seeds per config]
"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_musiclm_amd import open_musiclm as M, engine as E, ops

CLASSES = ("qproj", "kvproj", "attn", "toout", "ffin", "ffout", "heads")
H = torch.float16
CFG = {}            # class -> mode, set per configuration


def mode(c):
    return CFG.get(c, "16")


def operands(c, a32, w32, w16):
    """(A, W) for class c: both half ("16"), both fp32 ("32"), or one of them rounded to half and widened again ("A" / "W")."""
    m = mode(c)
    if m == "16":
        return a32.to(H), w16
    if m == "32":
        return a32, w32
    if m == "A":                                   # activation exact, weight rounded
        return a32, w16.float()
    if m == "W":
        return a32.to(H).float(), w32
    raise ValueError(m)


def lin(c, a32, w32, w16, out, **kw):
    A, W = operands(c, a32, w32, w16)
    ops.gemm(A.contiguous(), W, out, **kw)


def diag_trunk_forward(tr, pw16, x, keymask, B, N, save, training):
    model = tr.__dict__["_omlm_owner"]
    pw32 = model.__dict__.get("_diag_pw32")
    if pw32 is None:
        pw32 = model.__dict__["_diag_pw32"] = E.PreparedWeights(model, "bf16x3", persistent=False)
    dev = x.device
    Mr, D = x.shape
    Hh = tr.heads
    table, _ = E.relpos_forward(tr, N, False)
    for (attn, _, ff), w16, w32 in zip(tr.layers, pw16.layers, pw32.layers):
        F_, Fp = w16["F"], w16["Fp"]
        m1 = torch.empty(Mr, device=dev); r1 = torch.empty(Mr, device=dev)
        xn = torch.empty(Mr, D, device=dev)
        ops.layernorm_fwd(x, attn.norm.gamma.detach(), xn, None, m1, r1)
        q_raw = torch.empty(Mr, Hh * 64, device=dev)
        kv_raw = torch.empty(Mr, 128, device=dev)
        lin("qproj", xn, w32["Wq"], w16["Wq"], q_raw, M=Mr, N=Hh * 64, K=D)
        lin("kvproj", x, w32["Wkv"], w16["Wkv"], kv_raw, M=Mr, N=128, K=D)
        Ta = torch.float32 if mode("attn") == "32" else H
        q = torch.empty(Mr, Hh * 64, dtype=Ta, device=dev)
        k = torch.empty(Mr, 64, dtype=Ta, device=dev)
        v = torch.empty(Mr, 64, dtype=Ta, device=dev)
        ops.qk_norm_fwd(q_raw, kv_raw, attn.q_scale.detach(), attn.k_scale.detach(), q, k, v, Hh)
        o = torch.empty(Mr, Hh * 64, dtype=Ta, device=dev)
        lse = torch.empty(B, Hh, N, device=dev)
        fixed_ok = Ta != H
        abias = ops.AttnBias(table, N, Hh, dev, q_scale=attn.q_scale.detach() if fixed_ok else None,
                             k_scale=attn.k_scale.detach() if fixed_ok else None, scale=E.ATTN_SCALE)
        ops.attn_fwd(q, k, v, abias, keymask, o, lse, B, N, Hh, E.ATTN_SCALE)
        x1 = torch.empty(Mr, D, device=dev)
        lin("toout", o.float(), w32["Wo"], w16["Wo"], x1, M=Mr, N=D, K=Hh * 64, Cin=x)
        m2 = torch.empty(Mr, device=dev); r2 = torch.empty(Mr, device=dev)
        xn2 = torch.empty(Mr, D, device=dev)
        ops.layernorm_fwd(x1, ff.norm_in.gamma.detach(), xn2, None, m2, r2)
        h1 = torch.empty(Mr, 2 * Fp, device=dev)
        lin("ffin", xn2, w32["W1p"], w16["W1p"], h1, M=Mr, N=2 * Fp, K=D)
        # conv + GEGLU + LayerNorm(F): fp32 in / out when either neighbour is exact (an fp16 h1 widened to fp32 is the same values)
        wide = mode("ffin") != "16" or mode("ffout") != "16"
        h1m = h1 if mode("ffin") != "16" else h1.to(H)
        if wide:
            h1m = h1m.float()
        wm = w32 if wide else w16
        h2 = torch.empty(Mr, Fp, dtype=h1m.dtype, device=dev)
        m3 = torch.empty(Mr, device=dev); r3 = torch.empty(Mr, device=dev)
        ops.ffmid_fwd(h1m.contiguous(), wm["convw"], wm["gamma_mid"], h2, m3, r3, N, F_, Fp, 0.0, 0)
        x2 = torch.empty(Mr, D, device=dev)
        lin("ffout", h2.float(), w32["W2p"], w16["W2p"], x2, M=Mr, N=D, K=Fp, Cin=x1)
        x = x2
    mf = torch.empty(Mr, device=dev); rf = torch.empty(Mr, device=dev)
    y = torch.empty(Mr, D, device=dev)
    ops.layernorm_fwd(x, tr.norm.gamma.detach(), y, None, mf, rf)
    return y, None


def diag_heads_forward(model, pw16, y32, lay, want):
    pw32 = model.__dict__["_diag_pw32"]
    out = []
    for s, seq in enumerate(model.token_sequences):
        if not want[s]:
            out.append(None)
            continue
        V1 = seq.codebook_size + 1
        ldV = E.ceil_to(V1, 8)
        n_s = lay.n_out[s]
        buf = torch.empty(lay.B * n_s, ldV, device=y32.device)
        for qq in range(seq.num_quantizers):
            ent = lay.head_maps.get((s, qq))
            if ent is None:
                continue
            a_map, c_map, rows = ent
            A, W = operands("heads", y32, pw32.heads[s][qq].detach(), pw16.heads[s][qq])
            ops.gemm(A.contiguous(), W.contiguous(), buf, M=rows, N=V1, K=model.dim, a_map=a_map, c_map=c_map, ldc=ldV,
                     a_rows=A.shape[0], b_rows=V1)
        out.append(buf)
    return out


def configurations():
    yield "all fp16 (= precision fp16)", {}
    for c in CLASSES:
        yield f"{c} exact", {c: "32"}
    yield "attention branch exact (qproj kvproj attn toout)", {c: "32" for c in ("qproj", "kvproj", "attn", "toout")}
    yield "ffin + ffout exact", {"ffin": "32", "ffout": "32"}
    yield "ffin + ffout + heads exact", {"ffin": "32", "ffout": "32", "heads": "32"}
    yield "ffin + ffout + heads: activations exact only", {"ffin": "A", "ffout": "A", "heads": "A"}
    yield "ffin + ffout + heads: weights exact only", {"ffin": "W", "ffout": "W", "heads": "W"}
    yield "all but ffin + ffout exact", {c: "32" for c in CLASSES if c not in ("ffin", "ffout")}
    yield "all exact (= bf16x3 arithmetic)", {c: "32" for c in CLASSES}


def main():
    dev = torch.device("cuda:0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "error_budget.md")
    max_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    E.trunk_forward = diag_trunk_forward
    E.heads_forward = diag_heads_forward
    rows = []
    for name, kw, mk in (("coarse6", dict(dim=1024, depth=6, heads=8), lambda **k: M.create_coarse_transformer(num_coarse_quantizers=3, ff_dropout=0.0, **k)),
                         ("fine24", dict(dim=1024, depth=24, heads=16), lambda **k: M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, ff_dropout=0.0, **k))):
        for path in sorted(glob.glob(os.path.join(root, ".bigfix", f"{name}_seed*.pt")))[:max_seeds]:
            z = torch.load(path)
            torch.manual_seed(100 + z["seed"])
            model = mk(precision="fp16", **kw).to(dev)
            wrapper = M.TokenConditionedTransformerWrapper(transformer=model, unique_consecutive=False, cross_entropy_loss_weights=[0., 0., 1.], mask_prob=0.0)
            wrapper.eval()
            ref = z["logits"].double()
            for label, cfg in configurations():
                CFG.clear(); CFG.update(cfg)
                with torch.no_grad():
                    loss, logits, _ = wrapper(all_token_ids=[t.to(dev) for t in z["ids"]], return_loss=True)
                got = logits[-1].double().cpu()
                d = got - ref
                e = float(d.abs().max() / ref.abs().max())
                rms = float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
                rows.append(dict(config=name, seed=z["seed"], what=label, logits_err=e, rel_rms=rms))
                print(rows[-1], flush=True)
            del model, wrapper
            torch.cuda.empty_cache()
    with open(out, "w") as fh:
        fh.write("Per-op error budget of the fp16 forward: logits error max|d| / max|ref| (and relative rms) against the CPU oracle, B = 1, eval forward;\n"
                 "every contraction class in IEEE half except the ones named, which run on hi/lo planes (tools/error_budget.py).  `removed` = the share of the\n"
                 "all-fp16 error VARIANCE a configuration takes away: 1 - (err / err_fp16)^2.\n\n")
        for cfg in ("coarse6", "fine24"):
            seeds = sorted({r["seed"] for r in rows if r["config"] == cfg})
            if not seeds:
                continue
            fh.write(f"### {cfg} (seeds {seeds})\n\n| configuration | max-err per seed | mean max-err | mean rel-rms | removed |\n|---|---|---:|---:|---:|\n")
            base = None
            for label, _ in configurations():
                v = [r["logits_err"] for r in rows if r["config"] == cfg and r["what"] == label]
                rm = [r["rel_rms"] for r in rows if r["config"] == cfg and r["what"] == label]
                mean = sum(v) / len(v)
                if base is None:
                    base = mean
                fh.write(f"| {label} | {', '.join(f'{x:.2e}' for x in v)} | {mean:.2e} | {sum(rm) / len(rm):.2e} | {1 - (mean / base) ** 2:+.2f} |\n")
            fh.write("\n")
    json.dump(rows, open(out.replace(".md", ".json"), "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    main()
