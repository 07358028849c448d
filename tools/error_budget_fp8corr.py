"""Error budget of the fp8-corrected ConvFeedForward forward -- a CPU simulation, no GPU needed.

The `fp16ff` forward runs FF-in / FF-out / the logit heads as three half products (hi*hi + hi*lo + lo*hi).  The two correction
products only need a few significant bits; this tool measures what is left of the logits when they run on fp8 (e4m3) operands with one
power-of-two scale per operand ROW (what `v_mfma_scale_f32_32x32x64_f8f6f4` consumes at twice the half rate):

    y = A_hi B_hi^T  +  2^(ea + eb - 11) [ q8(A_hi 2^-ea) q8(B_lo 2^-(eb-11))^T + q8(A_lo 2^-(ea-11)) q8(B_hi 2^-eb)^T ]

Everything else of the forward is kept in exact fp32 (the oracle), so each figure is the contribution of the FF / head contractions alone;
it adds in quadrature to the rest of the mode's error (measured on the GPU: profiles/r05_seed_sweep.md).
Usage:  python tools/error_budget_fp8corr.py [out.md] [seeds_small] [seeds_large]
"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import musiclm_oracle as O

H = torch.float16
F8 = torch.float8_e4m3fn


def row_exp(bound):
    """power-of-two row scale: values / 2^e <= 256 (inside e4m3's 448)"""
    return torch.ceil(torch.log2(bound.clamp(min=1e-30))) - 8.0


def q8(v, e):
    return (v * torch.exp2(-e)).clamp(-448.0, 448.0).to(F8).to(torch.float32) * torch.exp2(e)


def ff_linear(x, w, mode, xbound=None):
    """x [.., K] fp32, w [N, K] fp32."""
    if mode == "exact":
        return F.linear(x, w)
    xh, wh = x.to(H).float(), w.to(H).float()
    if mode == "fp16":
        return F.linear(xh, wh)
    xl, wl = (x - xh).to(H).float(), (w - wh).to(H).float()
    if mode == "planes3":
        return F.linear(xh, wh) + F.linear(xh, wl) + F.linear(xl, wh)
    if mode.startswith("mx8"):
        xb = xbound if xbound is not None else xh.abs().amax(-1, keepdim=True)
        ea = row_exp(xb)
        eb = row_exp(wh.abs().amax(-1, keepdim=True))
        if mode == "mx8t":                       # one scale per TENSOR instead of per row
            ea = ea.amax().expand_as(ea); eb = eb.amax().expand_as(eb)
        return (F.linear(xh, wh) + F.linear(q8(xh, ea), q8(wl, eb - 11.0)) + F.linear(q8(xl, ea - 11.0), q8(wh, eb)))
    raise ValueError(mode)


def ln_with_bound(x, gamma, eps=1e-5):
    """LayerNorm and the row bound its kernel can form from ONE pass of (sum, sum of squares, max, min): max |y| <= max(xmax - mean, mean - xmin) rstd max|gamma|"""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    bound = torch.maximum(x.amax(-1, keepdim=True) - mu, mu - x.amin(-1, keepdim=True)) * rstd * gamma.abs().max()
    return (x - mu) * rstd * gamma, bound


def forward(sd, spec, ids, mode, heads_mode):
    x, split_at = O.embed_sequences(sd, ids, spec)
    n = x.shape[1]
    bias = O.rel_pos_bias_matrix(O.rel_pos_table_continuous(sd, "transformer.rel_pos_bias.", n), n)
    for l in range(spec.depth):
        lp = f"transformer.layers.{l}."
        x = O.attention(sd, lp + "0.", x, bias, None, spec) + x
        p = lp + "2."
        xn, b1 = ln_with_bound(x, sd[p + "0.gamma"])
        hdn = ff_linear(xn, sd[p + "1.weight"], mode, b1)
        hdn = O.causal_dwconv3(hdn, sd[p + "2.ds_conv.weight"])
        a, gate = hdn.chunk(2, dim=-1)
        g, b2 = ln_with_bound(O.gelu_erf(gate) * a, sd[p + "4.gamma"])
        x = ff_linear(g, sd[p + "6.weight"], mode, b2) + x
    hidden, bh = ln_with_bound(x, sd["transformer.norm.gamma"])
    pieces = list(torch.tensor_split(hidden, split_at, dim=1))
    bounds = list(torch.tensor_split(bh, split_at, dim=1))
    pz, bz = pieces[-1], bounds[-1]
    w = sd[f"logit_weights.{len(pieces) - 1}"]
    q = spec.token_sequences[-1].num_quantizers
    out = torch.empty(pz.shape[0], pz.shape[1], w.shape[1])
    for qq in range(q):
        out[:, qq::q] = ff_linear(pz[:, qq::q], w[qq], heads_mode, bz[:, qq::q])
    return out


def main():
    out_md = sys.argv[1] if len(sys.argv) > 1 else None
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    nl = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    torch.set_num_threads(os.cpu_count())
    rows = []
    cfgs = [("coarse-small depth 6, N = 1116", O.coarse_spec(), [1, 199, 300], ns),
            ("musiclm_large fine depth 24, N = 1817", O.fine_spec(depth=24, heads=16), [1, 225, 225], nl)]
    modes = ["fp16", "planes3", "mx8", "mx8t"]
    for name, spec, lens, nseed in cfgs:
        for seed in range(nseed):
            t0 = time.time()
            sd = O.init_state_dict(spec, seed=seed)
            ids = O.synthetic_ids(spec, 1, lens, seed=100 + seed)
            with torch.no_grad():
                ref = forward(sd, spec, ids, "exact", "exact")
                errs = {}
                for m in modes:
                    y = forward(sd, spec, ids, m, m)
                    errs[m] = float((y - ref).abs().max() / ref.abs().max())
            rows.append((name, seed, errs))
            print(name, "seed", seed, {k: f"{v:.2e}" for k, v in errs.items()}, f"{time.time() - t0:.0f} s", flush=True)
    lines = ["# fp8 (e4m3, one power-of-two scale per operand row) correction products: what they leave in the logits",
             "",
             "CPU simulation (`tools/error_budget_fp8corr.py`): the full-size eval forward with EVERYTHING in exact fp32 except FF-in, FF-out and the",
             "final sequence's logit heads, which run in the named arithmetic.  Figure = max |logits - exact| / max |exact| (the tests' metric), i.e. the",
             "contribution of those contractions alone; it adds in quadrature to the rest of the `fp16ff` forward (attention branch etc. in plain half:",
             "1.3e-4 ... 1.8e-4 at depth 6, 4.6e-4 ... 5.5e-4 at depth 24, `profiles/r05_seed_sweep.md`).",
             "",
             "| model | seed | one half product (`fp16`) | three half products (`fp16ff`, round 5) | half + two fp8 corrections, row scales | the same, one scale per tensor |",
             "|---|---:|---:|---:|---:|---:|"]
    for name, seed, e in rows:
        lines.append(f"| {name} | {seed} | {e['fp16']:.2e} | {e['planes3']:.2e} | {e['mx8']:.2e} | {e['mx8t']:.2e} |")
    txt = "\n".join(lines) + "\n"
    print(txt)
    if out_md:
        with open(out_md, "w") as f:
            f.write(txt)


if __name__ == "__main__":
    main()
