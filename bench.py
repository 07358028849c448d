#!/usr/bin/env python
"""Benchmark of the MI355X TokenConditionedTransformer hot path (BASELINE.json metric:
"train steps/sec + AR tokens/sec, coarse-stage musiclm_small, 1/2/4/8 MI355X").

    python bench.py --gpus N --steps K --warmup W [--precision fp16|bf16|bf16x3] [--batch B] [--accum A]

One "step" = one optimizer step of the coarse stage of musiclm_small (dim 1024, depth 6, heads 8, conv-GEGLU FF,
N = 1116 positions = 3 start + 13 clap + 200 semantic + 900 coarse ids -- the exact length the reference's data
pipeline yields; BASELINE's "seq_len 1024" is nominal) on --batch samples per GPU split in --accum micro-batches:
forward + backward (forgetful mask on, FF dropout 0.1 on) + ONE gradient all-reduce + fused clip/AdamW.  Token ids
are synthetic (seeded U{0..1023}) and already resident in HBM when the timed region starts; weights are the
reference's random init (no checkpoints offline).  For N > 1 launch with torch.distributed.run (one rank per GPU).

`value` = whole-job training samples/s (global batch * steps / s, max over ranks) of BASELINE config 2 in precision "fp16ff" since
round 5: IEEE-half operands everywhere, with the forward of the two ConvFeedForward linears and of the logit heads on hi/lo half planes (three
products; the two linears carry 86-88 % of the fp16 logits-error variance) -- the mode whose logits meet north_star's 1e-3 against the CPU reference
WITH MARGIN at both model depths (1.3e-4 .. 1.8e-4 over 5 seeds at this model, 1.6e-4 at the benchmarked batch; 4.6e-4 .. 5.5e-4 at musiclm_large
depth 24), at 1.3x the fp16 step.  Plain fp16 measures 8.7e-4 .. 9.8e-4 here (inside 1e-3 by a few per cent) and 1.6e-3 at depth 24; bf16, the dtype the
config names, 7e-3 .. 8e-3.  At N = 1 the same JSON line carries, under "legs", the other modes and configurations of BASELINE.json measured
in the same run:
  legs.fp16         the same train step in plain fp16 (logits 8.7e-4 .. 9.8e-4: inside the tolerance at this depth, without margin)
  legs.bf16         the same train step with bf16 operands (the dtype BASELINE config 2 names; logits 7e-3 .. 8e-3: outside the tolerance)
  legs.bf16x3       the same train step with fp32 operands split hi/lo on the bf16 matrix cores (fp32-grade products)
  legs.large_fine   BASELINE config 4: musiclm_large fine stage (depth 24, heads 16, N = 1817, 5 fine quantizers), fp16ff (+ the bf16 step time)
  legs.e2e_generate BASELINE config 5: MusicLM.generate, 10 s (RVQ + 500 semantic + 2250 coarse + 3750 fine ids)
plus `roofline` (HIP events around every MFMA GEMM launch of extra, instrumented steps), `ar_tokens_per_sec`
(coarse-stage decode) and `cpu_baseline` (the CPU oracle on this box's host cores, timed BEFORE the GPU regions).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_SEQ = 1116
PEAK_TFLOPS = 2500.0                                    # dense bf16 MFMA peak (MI355X_MICROARCH.md); bf16x3 issues 3 MFMAs per product
HBM_ACHIEVABLE_TBS = 6.3                                # measured streaming rate (MI355X_MICROARCH.md)
P_LIN = {8: 9_566_208, 16: 10_614_784}                  # linear MACs / token / layer by head count


def algorithmic_flops_per_sample(N=N_SEQ, L=6, h=8, dh=64, n_out=901):
    """F_fwd = 2 [L (N P_lin + h dh N (N + 1)) + N_out 1,049,600]; fwd + bwd = 3 F_fwd.  N_out counts the logit rows the
    timed step COMPUTES: the trainers' call (return_logits=False) evaluates only the heads of sequences with a loss weight -- the predicted
    sequence's 1 + 900 rows of the 1114 (the two conditioning sequences' zero-weight heads add nothing to loss or gradients and are skipped)."""
    fwd = 2 * (L * (N * P_LIN[h] + h * dh * N * (N + 1)) + n_out * 1_049_600)
    return 3 * fwd


T_START = time.perf_counter()


def progress(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


class TrainLeg:
    """One training configuration: model + stage + optimizer + captured micro-step, and its timed loop."""

    def __init__(self, dev, dp, *, stage, dim, depth, heads, precision, batch, accum, use_graph, ds_kwargs=None, seed_rank=0):
        from open_musiclm_amd import open_musiclm as M
        from open_musiclm_amd.data import SyntheticTokenDataset
        from open_musiclm_amd.graph import GraphedForwardBackward
        from open_musiclm_amd.optimizer import get_linear_scheduler, get_optimizer
        self.dev, self.dp, self.batch, self.accum, self.precision = dev, dp, batch, accum, precision
        torch.manual_seed(0)                               # identical replicas on every rank
        if stage == "coarse":
            self.model = M.create_coarse_transformer(dim=dim, depth=depth, heads=heads, attn_dropout=0.0, ff_dropout=0.1,
                                                     num_coarse_quantizers=3, precision=precision).to(dev)
            self.stage = M.CoarseStage(coarse_transformer=self.model, cross_entropy_loss_weights=[0., 0., 1.])
            self.keys = ("clap_token_ids", "semantic_token_ids", "coarse_token_ids")
        elif stage == "fine":
            self.model = M.create_fine_transformer(dim=dim, depth=depth, heads=heads, attn_dropout=0.0, ff_dropout=0.1,
                                                   num_coarse_quantizers=3, num_fine_quantizers=5, precision=precision).to(dev)
            self.stage = M.FineStage(fine_transformer=self.model, cross_entropy_loss_weights=[0., 0., 1.])
            self.keys = ("clap_token_ids", "coarse_token_ids", "fine_token_ids")
        else:
            raise ValueError(stage)
        self.stage.train()
        self.optim = get_optimizer(self.model.parameters(), lr=3e-4, wd=0.01)
        self.sched = get_linear_scheduler(self.optim, total_iters=6000)
        assert batch % accum == 0
        micro = batch // accum
        ds = SyntheticTokenDataset(stage, length=1 << 20, seed=1234 + seed_rank, **(ds_kwargs or {}))

        def make_batch(i):
            items = [ds[i * micro + j] for j in range(micro)]
            return [torch.cat([it[f] for it in items], 0).to(dev) for f in range(len(self.keys))]
        self.batches = [make_batch(i) for i in range(4 * accum)]       # resident in HBM before timing
        self.fb = GraphedForwardBackward(lambda **kw: self.stage(**kw, return_loss=True, return_logits=False)[0], loss_scale=1.0 / accum,
                                         enabled=use_graph)
        self.optim.zero_grad()                                 # adopt the flat parameter / gradient buffers before capture
        # gradient accumulation (--accum > 1): the rel-pos MLP once per optimizer step, like SingleStageTrainer (engine.RelposStepCache)
        self.rc = None
        trunk = getattr(self.model, "transformer", None)
        if (accum > 1 and os.environ.get("OMLM_RELPOS_CACHE", "1") != "0" and trunk is not None
                and getattr(trunk, "rel_pos_bias", None) is not None and getattr(trunk, "relative_position_bias_type", "") != "t5"):
            from open_musiclm_amd import engine
            self.rc = engine.relpos_step_cache(trunk)
            self.rc.enabled = True

        def discard():
            self.optim.mark_grads_dirty()
            self.optim.zero_grad()
            if self.rc is not None:
                self.rc.reset_accum()
        self.fb.prepare(dict(zip(self.keys, self.batches[0])), after_warmup=discard)

    def step(self, k, eager=False, exchange=True):
        self.optim.zero_grad()
        if self.rc is not None:
            self.rc.refresh()
        for a in range(self.accum):
            kw = dict(zip(self.keys, self.batches[(k * self.accum + a) % len(self.batches)]))
            loss = self.fb._eager(kw) if eager else self.fb(**kw)
            self.optim.mark_grads_dirty()
        if self.rc is not None:
            self.rc.flush()
        if exchange:
            self.dp.allreduce_sum_(self.optim.flat_grad)
        self.optim.step(max_grad_norm=0.5, grad_scale=self.dp.grad_scale())
        self.sched.step()
        return loss

    def timed(self, steps, warmup):
        for k in range(warmup):
            loss = self.step(k)
            torch.cuda.synchronize()
        self.dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            loss = self.step(warmup + k)
        self.dp.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device=self.dev)
        if self.dp.is_distributed:
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        return float(tmax.item()), float(loss.item())

    def gemm_roofline(self, k0, traffic=None, tag=""):
        """HIP events (torch events on the stream the kernels are launched on) around every MFMA GEMM launch of two extra,
        eager, un-exchanged steps -- single GEMMs and the grouped weight-gradient launches alike: algorithmic flops of the
        launches / their summed durations."""
        from open_musiclm_amd import ops
        import open_musiclm_amd.engine as E
        ops_gemm, rec = ops.gemm, []

        def timed_gemm(A, B, C_, *, M, N, K, **kw):
            if A.dtype == torch.float32 and ops._X3_PLANES and kw.get("planes", True) and M * N * K >= ops._X3_MIN_MACS:
                # bf16x3: the hi/lo plane split of the operands is its own (HBM-bound) kernel; the events bracket the GEMM launch
                lda, ldb = kw.get("lda") or A.shape[-1], kw.get("ldb") or B.shape[-1]
                ops.operand_planes(A, kw.get("a_rows") or A.numel() // lda, lda)
                ops.operand_planes(B, kw.get("b_rows") or B.numel() // ldb, ldb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops_gemm(A, B, C_, M=M, N=N, K=K, **kw)
            e1.record()
            rec.append((e0, e1, 2.0 * M * N * K, (M, N, K, str(A.dtype)[6:], str(C_.dtype)[6:], int(bool(kw.get('a_kmajor'))), int(bool(kw.get('b_kmajor'))))))
        ops_p16 = ops.gemm_planes16

        def timed_p16(A, A_lo, B, B_lo, C_, C_lo=None, *, M, N, K, **kw):       # fp16ff: the FF forward GEMMs on hi/lo planes (3 products issued)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops_p16(A, A_lo, B, B_lo, C_, C_lo, M=M, N=N, K=K, **kw)
            e1.record()
            rec.append((e0, e1, 2.0 * M * N * K, (M, N, K, str(A.dtype)[6:] + " hi/lo planes", (str(C_.dtype)[6:] + " planes") if C_lo is not None else str(C_.dtype)[6:], 0, 0), 3))
        ops_mx = ops.gemm_mx16

        def timed_mx(A, A8, B, B8, C_, C_lo=None, *, M, N, K, **kw):           # fp16ff: one half product + two fp8 correction products at twice the rate
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops_mx(A, A8, B, B8, C_, C_lo, M=M, N=N, K=K, **kw)
            e1.record()
            rec.append((e0, e1, 2.0 * M * N * K, (M, N, K, str(A.dtype)[6:] + " + fp8 planes (mx16)", (str(C_.dtype)[6:] + " planes") if C_lo is not None else str(C_.dtype)[6:], 0, 0), 2))
        ops_qkn = ops.gemm_qknorm

        def timed_qkn(A, B, C_, scale, norm_out, groups, *, M, N, K, **kw):      # q / k projections with the l2-norm epilogue: GEMM launches too
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops_qkn(A, B, C_, scale, norm_out, groups, M=M, N=N, K=K, **kw)
            e1.record()
            rec.append((e0, e1, 2.0 * M * N * K, (M, N, K, str(A.dtype)[6:], str(C_.dtype)[6:] + "+l2norm", 0, 0)))
        # the weight-gradient contractions of a backward go out as grouped launches (ops.WgradGroup.flush): same bookkeeping
        group_flush = ops.WgradGroup.flush

        def timed_flush(wg, splits=0):
            if not wg.items:
                return
            fl = sum(2.0 * it[4] * it[5] * it[6] for it in wg.items)       # (dY, X, dW, c_map, M, N, K, ...)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            group_flush(wg, splits)
            e1.record()
            rec.append((e0, e1, fl, ('wgrad_group', len(wg.items), splits)))
        E.ops.gemm = timed_gemm
        E.ops.gemm_planes16 = timed_p16
        E.ops.gemm_mx16 = timed_mx
        E.ops.gemm_qknorm = timed_qkn
        ops.WgradGroup.flush = timed_flush
        try:
            for k in range(2):
                self.step(k0 + k, eager=True, exchange=False)
            torch.cuda.synchronize()
        finally:
            E.ops.gemm = ops_gemm
            E.ops.gemm_planes16 = ops_p16
            E.ops.gemm_mx16 = ops_mx
            E.ops.gemm_qknorm = ops_qkn
            ops.WgradGroup.flush = group_flush
        tot_ms = sum(r[0].elapsed_time(r[1]) for r in rec)
        tot_fl = sum(r[2] for r in rec)
        mult = 3 if self.precision == "bf16x3" else 1
        issued_fl = sum(r[2] * (r[4] if len(r) > 4 else mult) for r in rec)      # products the matrix cores are issued, in half-rate equivalents (hi/lo half planes: 3 per algorithmic one; mx16: 1 half + 2 fp8 at twice the rate = 2)
        big = [(r[0].elapsed_time(r[1]), r[2]) for r in rec if r[2] > 1e11]
        if os.environ.get("OMLM_BENCH_GEMM_TABLE"):        # per-shape table of the same launches (tools: profiles/*_gemm_calls.md)
            agg = {}
            for r in rec:
                a, b, f, key = r[:4]
                t = agg.setdefault(key, [0, 0.0, 0.0]); t[0] += 1; t[1] += a.elapsed_time(b); t[2] += f
            rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
            tpath = os.environ["OMLM_BENCH_GEMM_TABLE"]
            with open(tpath if not tag else tpath.replace(".md", f"_{tag}.md"), "w") as fh:
                fh.write("| M, N, K, operands, out, A k-major, B k-major | launches / step | us / launch | TFLOP/s | ms / step |\n|---|---:|---:|---:|---:|\n")
                for key, (n, ms, fl) in rows:
                    fh.write(f"| {key} | {n / 2:g} | {ms / n * 1e3:.1f} | {fl / (ms * 1e-3) / 1e12:.0f} | {ms / 2:.3f} |\n")
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        # the launch family that takes the most time of a step (fp16ff: FF-in forward on omlm_gemm_mx16), per launch
        fam = {}
        for r in rec:
            t = fam.setdefault(r[3], [0, 0.0, 0.0, r[4] if len(r) > 4 else mult]); t[0] += 1; t[1] += r[0].elapsed_time(r[1]); t[2] += r[2]
        dk, (dn, dms, dfl, dmult) = max(fam.items(), key=lambda kv: kv[1][1])
        dominant = {"launch": str(dk), "launches_per_step": dn / 2, "us_per_launch": round(dms / dn * 1e3, 1), "flops_per_launch": dfl / dn,
                    "achieved": round(dfl / (dms * 1e-3) / 1e12, 1), "unit": "TFLOP/s", "frac": round(dfl / (dms * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
                    "issue_frac": round(dmult * dfl / (dms * 1e-3) / 1e12 / PEAK_TFLOPS, 4), "ms_per_step": round(dms / 2, 3)}
        return {"bound": "mfma", "kernel": "gemm kernels (all layouts, every GEMM launch of a train step)", "dominant_launch": dominant,
                "achieved": round(ach, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_TFLOPS, 4),
                "traffic": traffic, "launches": len(rec) // 2, "gemm_ms_per_step": round(tot_ms / 2, 3),
                "mfma_issue_frac": round(issued_fl / (tot_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4),
                "large_gemm_achieved": round(sum(f for _, f in big) / (sum(t for t, _ in big) * 1e-3) / 1e12, 2) if big else None,
                "single_product_achieved": (round(sum(r[2] for r in rec if len(r) <= 4) / (sum(r[0].elapsed_time(r[1]) for r in rec if len(r) <= 4) * 1e-3) / 1e12, 2)
                                            if any(len(r) > 4 for r in rec) else None),
                "note": ("achieved / frac count ALGORITHMIC flops (2 M N K per linear); the FF-in / FF-out forward launches of fp16ff issue one half product + two fp8 "
                         "correction products at twice the matrix rate (omlm_gemm_mx16: 2 half-rate equivalents), the head launches three half products "
                         "(omlm_gemm_planes16): mfma_issue_frac counts those; single_product_achieved = every other GEMM launch of the step")
                        if any(len(r) > 4 for r in rec) else "achieved / frac count algorithmic flops (2 M N K per linear)"}

    def free(self):
        self.fb = self.model = self.stage = self.optim = self.batches = None
        import gc
        gc.unfreeze()
        gc.collect()
        torch.cuda.empty_cache()


def gemm_source_sha16():
    """Identity of the GEMM kernels a committed traffic measurement belongs to: the first 16 hex digits of sha256 over csrc/gemm.hip +
    common.h + gemm_common.h + gemm_mx.hip (tools/pmc_gemm_traffic.sh records it; roofline.traffic is only attached while it matches the tree)."""
    import hashlib
    m = hashlib.sha256()
    for f in ("gemm.hip", "common.h", "gemm_common.h", "gemm_mx.hip"):
        with open(os.path.join(ROOT, "open_musiclm_amd", "csrc", f), "rb") as fh:
            m.update(fh.read())
    return m.hexdigest()[:16]


def cpu_baseline(progress_fn):
    """The oracle (port of the reference arithmetic, torch CPU kernels) on this box's host cores: one warm-up + median
    of 3 full micro-steps (forward + backward + global-norm clip + AdamW) at B = 2, N = 1116, fp32: a bounded
    sample of the same workload (~60 s)."""
    from oracle import musiclm_oracle as O
    spec = O.coarse_spec(dim=1024, depth=6, heads=8)
    sd = O.init_state_dict(spec, seed=0)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith("beta")]
    opt = torch.optim.AdamW(params, lr=3e-4, betas=(0.9, 0.99), weight_decay=0.01)
    CB = 2                                                 # the configs' own batch (train_musiclm_fma.json:39)
    ids = O.synthetic_ids(spec, CB, [1, 199, 300], seed=1234)
    noise = torch.randn(CB, N_SEQ, generator=torch.Generator().manual_seed(1))
    times = []
    for it in range(4):
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        l, _, _ = O.wrapper_forward_loss(sd, spec, ids, [0., 0., 1.], forget_noise=noise)
        l.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 0.5)
        opt.step()
        times.append(time.perf_counter() - t1)
        progress_fn(f"cpu baseline iteration {it}{' (warm-up)' if it == 0 else ''}: {times[-1]:.2f}s")
    med = statistics.median(times[1:])
    return {"value": round(CB / med, 4), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "host_cpu_count": os.cpu_count(),
            "sample": f"1 warm-up + median of 3 micro-steps (fwd + bwd + clip + AdamW) of B={CB}, N=1116, fp32, torch CPU kernels; "
                      f"times {[round(t, 2) for t in times]} s; threads = torch's default ({torch.get_num_threads()} of the box's "
                      f"{os.cpu_count()} hardware threads: one per physical core); kind 'port' because /root/reference does not "
                      "exist on the GPU box (the port is pinned to the reference by tests/golden)"}


def decode_leg(stage, dev, decode_ids):
    """AR decode of the coarse stage, B = 1 and B = 8: KV-cached single-row steps (decode.py) and, for comparison, the
    reference's own scheme (full re-forward per id, open_musiclm.py:299-319).  Rates include the prompt prefill."""
    stage.eval()
    g = torch.Generator().manual_seed(99)
    clap = torch.randint(0, 1024, (1, 12, 1), generator=g).to(dev)
    sem = torch.randint(0, 1024, (1, 199), generator=g).to(dev)
    res = {}
    for mode, use_cache, steps_new in (("kv_cache", True, 100), ("reforward", False, max(decode_ids // 3, 1))):
        for label, primed in (("empty_context", 0), ("full_context", 300 - steps_new)):
            prime = torch.randint(0, 1024, (1, primed, 3), generator=g).to(dev) if primed else None
            tgt = primed + steps_new
            kw = dict(clap_token_ids=clap, semantic_token_ids=sem, coarse_token_ids=prime, use_cache=use_cache)
            stage.generate(max_time_steps=primed + 2, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stage.generate(max_time_steps=tgt, **kw)
            torch.cuda.synchronize()
            res[f"{mode}_{label}"] = round(steps_new * 3 / (time.perf_counter() - t1), 2)
    clap8 = torch.randint(0, 1024, (8, 12, 1), generator=g).to(dev)
    sem8 = torch.randint(0, 1024, (8, 199), generator=g).to(dev)
    kw8 = dict(clap_token_ids=clap8, semantic_token_ids=sem8, use_cache=True)
    stage.generate(max_time_steps=2, **kw8)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stage.generate(max_time_steps=50, **kw8)
    torch.cuda.synchronize()
    res["kv_cache_batch8_empty_context"] = round(8 * 50 * 3 / (time.perf_counter() - t1), 2)
    kw16 = dict(clap_token_ids=torch.randint(0, 1024, (16, 12, 1), generator=g).to(dev),
                semantic_token_ids=torch.randint(0, 1024, (16, 199), generator=g).to(dev), use_cache=True)
    stage.generate(max_time_steps=2, **kw16)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    stage.generate(max_time_steps=50, **kw16)
    torch.cuda.synchronize()
    res["kv_cache_batch16_empty_context"] = round(16 * 50 * 3 / (time.perf_counter() - t1), 2)
    stage.train()
    return res


def e2e_generate_leg(dev, seconds=10, batch=1, precision="fp16ff"):
    """BASELINE config 5: MusicLM.generate for `seconds` of audio with the three musiclm_small stages (random init) --
    seeded synthetic 512-d conditioning embedding -> RVQ kernel (12 x 1024 x 512 seeded codebooks) -> semantic (50 Hz) ->
    coarse (75 Hz x 3) -> fine (75 Hz x 5) sliding-window AR decode.  Encodec / CLAP towers are outside the path (weights
    unobtainable offline): the rate covers the AR stack + RVQ only.  Floor: every sampled id streams its stage's trunk + one
    head once (16-bit weights; the headline precision fp16ff additionally streams the lo planes of the FF-in / FF-out / head weights)."""
    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd.clap_quantized import ClapQuantized
    torch.manual_seed(0)
    kw = dict(dim=1024, depth=6, heads=8, precision=precision)
    sem = M.create_semantic_transformer(**kw).to(dev)
    coarse = M.create_coarse_transformer(num_coarse_quantizers=3, **kw).to(dev)
    fine = M.create_fine_transformer(num_coarse_quantizers=3, num_fine_quantizers=5, **kw).to(dev)
    mlm = M.MusicLM(wav2vec=None, clap=None, neural_codec=None, semantic_transformer=sem, coarse_transformer=coarse,
                    fine_transformer=fine)
    cq = ClapQuantized(clap=None, codebook_size=1024, rq_num_quantizers=12, embed_dim=512).to(dev)
    g = torch.Generator().manual_seed(5)
    cq.rq.codebooks.copy_(torch.randn(12, 1024, 512, generator=g))
    emb = torch.randn(batch, 512, generator=g).to(dev)      # `batch` prompts generated together (one weight stream per step serves all)

    def run(secs):
        clap_ids = cq.quantize(emb)
        return mlm.generate(clap_token_ids=clap_ids, output_seconds=secs, return_tokens=True)
    run(4)                                                  # warm-up (one coarse window, two fine windows): weight prep, allocator
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s, c, f = run(seconds)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    assert s.shape[0] == batch
    n_steps = s.shape[1] + c.shape[1] * c.shape[2] + f.shape[1] * f.shape[2]      # decode steps = ids per prompt
    n_ids = n_steps * batch
    trunk_bytes = 58.06e6 * 2 + 1025 * 1024 * 2            # 16-bit trunk + one logit head per decode step
    if precision == "fp16ff":                              # + the lo planes of W1 / W2 (6 x 8.39 M entries) and of the head
        trunk_bytes += 6 * 8_386_560 * 2 + 1025 * 1024 * 2
    floor_s = n_steps * trunk_bytes / (HBM_ACHIEVABLE_TBS * 1e12)
    return {"workload": f"MusicLM.generate output_seconds={seconds}, musiclm_small stages, B={batch}, {precision}, KV-cached windows",
            "precision": precision,
            "batch": batch, "sampled_ids": int(n_ids),
            "ids_per_prompt": {"semantic": int(s.shape[1]), "coarse": int(c.shape[1] * c.shape[2]), "fine": int(f.shape[1] * f.shape[2])},
            "seconds": round(dt, 3), "ids_per_sec": round(n_ids / dt, 1), "audio_seconds_per_sec": round(batch * seconds / dt, 4),
            "roofline": {"bound": "hbm", "achieved": round(n_steps * trunk_bytes / dt / 1e9, 1), "peak": HBM_ACHIEVABLE_TBS * 1e3,
                         "unit": "GB/s", "frac": round(floor_s / dt, 4),
                         "note": "weight-streaming floor: the 16-bit trunk (+ the lo planes fp16ff reads) + one head per decode step (a step serves the whole batch) over the achievable 6.3 TB/s"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("OMLM_BENCH_PRECISION", "fp16ff"), choices=["bf16", "fp16", "fp16ff", "bf16x3"])
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per optimizer step")
    ap.add_argument("--accum", type=int, default=1, help="micro-batches per optimizer step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the fp16 / bf16x3 / large_fine / e2e_generate legs")
    ap.add_argument("--legs", default="fp16,bf16,bf16x3,large_fine,e2e_generate")
    ap.add_argument("--large-batch", type=int, default=16, help="samples per step of the large_fine leg")
    ap.add_argument("--decode-ids", type=int, default=48)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU over RCCL; rank 0 prints the JSON line)
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("OMP_NUM_THREADS", "8")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        progress(f"no launcher environment: re-executing under torch.distributed.run with {args.gpus} ranks (port {port})")
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, cmd)

    from open_musiclm_amd.parallel import DataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks "
                         f"(use --nproc-per-node {args.gpus}, or run `python bench.py --gpus {args.gpus}` without a launcher)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = local_rank % max(torch.cuda.device_count(), 1)       # == local_rank on a real node (one GPU per rank)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dp = DataParallel(device=dev)
    rank = dp.rank
    legs = [] if (args.no_legs or world > 1) else [l for l in args.legs.split(",") if l]

    # ---- CPU baseline first (rank 0, N = 1): the GPU regions then run back to back at the end of the process
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        progress(f"cpu baseline on {torch.get_num_threads()} torch threads (os.cpu_count()={os.cpu_count()})")
        cpu = cpu_baseline(progress)

    main_leg = TrainLeg(dev, dp, stage="coarse", dim=1024, depth=6, heads=8, precision=args.precision, batch=args.batch,
                        accum=args.accum, use_graph=not args.no_graph, seed_rank=rank)
    progress(f"micro-step captured into a HIP graph: {main_leg.fb.graph is not None}" +
             (f" ({main_leg.fb.capture_error})" if main_leg.fb.capture_error else ""))
    dt, final_loss = main_leg.timed(args.steps, args.warmup)
    progress(f"timed region: {args.steps} steps in {dt:.3f}s")

    ms_per_step = 1e3 * dt / args.steps
    steps_per_s = args.steps / dt
    samples_per_s = steps_per_s * args.batch * world
    flops_step = algorithmic_flops_per_sample() * args.batch
    model_tflops_per_gpu = flops_step / (dt / args.steps) / 1e12
    parity = {"bf16": "logits <= 1.2e-2 rel vs CPU reference (measured 7.1e-3 at B = 2, 7.8e-3 at the benchmarked B = 32; bf16 operands cannot meet 1e-3)",
              "fp16": "logits <= 1e-3 rel vs CPU reference (measured 8.1e-4 .. 9.8e-4 over 5 seeds, 9.1e-4 at the benchmarked B = 32; "
                      "every parameter gradient <= 1.5e-2 of its tensor's max; musiclm_large depth 24: 1.5e-3 .. 1.8e-3, profiles/r05_error_budget.md)",
              "fp16ff": "logits <= 5e-4 rel vs CPU reference at musiclm_small depth (measured 1.3e-4 .. 1.8e-4 over 5 seeds, 1.6e-4 at the benchmarked "
                        "B = 32) and <= 1e-3 at musiclm_large depth 24 (4.6e-4 .. 5.5e-4 over 3 seeds); every checked gradient <= 1.5e-2 of its "
                        "tensor's max (the ConvFeedForward forward on hi/lo half planes, everything else and the backward as fp16)",
              "bf16x3": "logits <= 1e-3 rel vs CPU reference (measured 7.1e-5; 2.9e-4 at musiclm_large depth 24)"}
    out = {
        "metric": "train steps/sec + AR tokens/sec, coarse-stage musiclm_small",
        "value": round(samples_per_s, 3), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "musiclm_small coarse-stage train step (fwd+bwd+allreduce+clip+AdamW), N=1116 "
                               "(3 start + 13 clap + 200 semantic + 900 coarse), forgetful mask 0.15, ff_dropout 0.1",
                   "global_batch": args.batch * world, "per_gpu_batch": args.batch, "grad_accum": args.accum,
                   "seq_len": N_SEQ, "parallelism": f"dp{world}", "precision": args.precision,
                   "hip_graph": main_leg.fb.graph is not None,
                   "exchange": ("none (single GPU)" if world == 1 else
                                f"one flat fp32 SUM all-reduce per step, torch.distributed backend {torch.distributed.get_backend()}"
                                + (" -- DRY RUN: ranks share a GPU, not an RCCL / xGMI measurement" if dp.shared_gpu else " (RCCL over xGMI)")),
                   "parity": parity[args.precision] + " (tests/test_gpu_model.py)",
                   "other_modes": "legs.fp16 (~0.78x this step; logits 8.7e-4 .. 9.8e-4: inside 1e-3 without margin, 1.6e-3 at depth 24), "
                                  "legs.bf16 (the dtype BASELINE config 2 names: ~0.75x this step, logits 7e-3 .. 8e-3 -- outside north_star's 1e-3), "
                                  "legs.bf16x3 (fp32-grade products everywhere, ~2x this step)"},
        "steps_per_sec": round(steps_per_s, 4),
        "model_tflops_per_gpu": round(model_tflops_per_gpu, 2),
        "model_flops_frac_of_bf16_peak": round(model_tflops_per_gpu / PEAK_TFLOPS, 4),
        "final_loss": round(final_loss, 4),
    }

    if os.environ.get("BENCH_STEP_TIMES") == "1":
        for k in range(3):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            main_leg.step(args.warmup + args.steps + 10 + k)
            tb = time.perf_counter()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            progress(f"diagnostic step {k}: host enqueue {1e3 * (tb - ta):.1f} ms, until GPU idle {1e3 * (tc - ta):.1f} ms")

    if rank == 0:
        # HBM-side bytes of the same launches (all GEMMs of one step) from the committed PMC passes of tools/pmc_gemm_traffic.sh
        # (separate FETCH_SIZE / WRITE_SIZE runs; counters cannot be read inside this process): B = 32 bf16 only
        traffic = detail = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath) and args.precision in ("bf16", "fp16", "fp16ff") and args.batch == 32 and args.accum == 1:
            try:
                detail = json.load(open(tpath))
                traffic = detail["fetch_bytes"] + detail["write_bytes"]
                if detail.get("gemm_source_sha16") != gemm_source_sha16() or detail.get("precision", "bf16") != args.precision:
                    # the GEMM kernels changed since the PMC passes were taken: a stale byte count is worse than none
                    progress(f"profiles/gemm_traffic.json belongs to GEMM sources {detail.get('gemm_source_sha16')}, the tree has "
                             f"{gemm_source_sha16()}: roofline.traffic left null (re-run tools/pmc_gemm_traffic.sh)")
                    traffic = detail = None
            except Exception:
                traffic = detail = None
        out["roofline"] = main_leg.gemm_roofline(args.warmup + args.steps, traffic)
        if detail:
            out["roofline"]["traffic_detail"] = detail
            out["roofline"]["traffic_source"] = ("committed PMC passes (profiles/gemm_traffic.json, tools/pmc_gemm_traffic.sh over this bench.py's train step; "
                                                 "counters cannot be read inside the timed process), build: " + str(detail.get("build", "see profiles/")))
        progress(f"roofline probe: {out['roofline']['launches']} GEMM launches, {out['roofline']['gemm_ms_per_step']} ms per step")
        if not args.no_decode:
            out["ar_tokens_per_sec"] = decode_leg(main_leg.stage, dev, args.decode_ids)
            out["ar_tokens_per_sec"]["precision"] = args.precision + (" (cached steps: FF-in / FF-out / head on hi + lo weight planes, activations un-rounded)"
                                                                       if args.precision == "fp16ff" else "")
            progress(f"decode {out['ar_tokens_per_sec']}")
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if legs:
            out["legs"] = {}
            main_leg.free()
        leg_notes = {
            "fp16": "same train step, precision fp16: IEEE-half operands on v_mfma_f32_32x32x16_f16 (the bf16 MFMA rate, 11 instead of 8 "
                    "significand bits), device-side dynamic loss scale divided out inside the fused AdamW kernel: the 16-bit mode that meets "
                    "the north-star 1e-3 logits tolerance at musiclm_small depth",
            "bf16": "same train step, precision bf16 (the dtype BASELINE config 2 names): bf16 operands on v_mfma_f32_32x32x16_bf16 -- same rate, "
                    "same bytes, 8 significand bits: logits 7e-3 .. 8e-3 against the CPU reference",
            "fp16ff": "same train step, precision fp16ff: fp16 whose two ConvFeedForward linears run the forward on hi/lo half planes (three "
                      "products, h1 / h2 un-rounded in between) -- the mode that meets the north-star 1e-3 logits tolerance with margin at "
                      "musiclm_small AND musiclm_large depth",
            "bf16x3": "same train step, precision bf16x3 (fp32 operands split hi/lo on the bf16 matrix cores: fp32-grade products at a "
                      "third of the MFMA rate)"}
        for prec in ("bf16", "fp16", "fp16ff", "bf16x3"):
            if prec not in legs or args.precision == prec:
                continue
            leg = TrainLeg(dev, dp, stage="coarse", dim=1024, depth=6, heads=8, precision=prec, batch=args.batch,
                           accum=args.accum, use_graph=not args.no_graph)
            k, w = (min(args.steps, 10), 2) if prec in ("fp16", "bf16", "fp16ff") else (min(args.steps, 5), 1)
            dt3, loss3 = leg.timed(k, w)
            tf3 = flops_step / (dt3 / k) / 1e12
            out["legs"][prec] = {
                "workload": leg_notes[prec], "dtype": prec, "parity": parity[prec], "value": round(args.batch * k / dt3, 3),
                "unit": "samples/s", "steps": k, "warmup": w, "ms_per_step": round(1e3 * dt3 / k, 3),
                "vs_headline_step": round((1e3 * dt3 / k) / ms_per_step, 4), "model_tflops_per_gpu": round(tf3, 2),
                "model_flops_frac_of_bf16_peak": round(tf3 / PEAK_TFLOPS, 4), "final_loss": round(loss3, 4),
                "hip_graph": leg.fb.graph is not None, "roofline": leg.gemm_roofline(k + w, tag=prec)}
            progress(f"{prec} leg: {out['legs'][prec]['ms_per_step']} ms/step")
            leg.free()
        if "large_fine" in legs:
            Bl = args.large_batch
            torch.cuda.reset_peak_memory_stats()
            k, w = 3, 1
            # the dtype the config names first (its step time only: bf16 logits are 1.2e-2 .. 1.4e-2 at this depth), then the mode that meets the
            # tolerance at depth 24
            leg = TrainLeg(dev, dp, stage="fine", dim=1024, depth=24, heads=16, precision="bf16", batch=Bl, accum=1,
                           use_graph=not args.no_graph, ds_kwargs=dict(fine_window_seconds=3))
            dtb, _ = leg.timed(k, w)
            leg.free()
            torch.cuda.reset_peak_memory_stats()
            leg = TrainLeg(dev, dp, stage="fine", dim=1024, depth=24, heads=16, precision="fp16ff", batch=Bl, accum=1,
                           use_graph=not args.no_graph, ds_kwargs=dict(fine_window_seconds=3))
            dtl, lossl = leg.timed(k, w)
            fl = algorithmic_flops_per_sample(N=1817, L=24, h=16, n_out=1126) * Bl
            tfl = fl / (dtl / k) / 1e12
            out["legs"]["large_fine"] = {
                "workload": "musiclm_large fine-stage train step (BASELINE config 4): dim 1024, depth 24, heads 16, N=1817 "
                            "(3 start + 13 clap + 676 coarse + 1125 fine), 5 fine quantizers, forgetful mask, ff_dropout 0.1",
                "dtype": "fp16ff", "parity": "logits 4.6e-4 .. 5.5e-4 vs CPU reference over 3 seeds at this depth (profiles/r05_seed_sweep.md; fp16 1.6e-3, bf16 1.3e-2)",
                "per_gpu_batch": Bl, "value": round(Bl * k / dtl, 3), "unit": "samples/s", "steps": k,
                "warmup": w, "ms_per_step": round(1e3 * dtl / k, 3), "bf16_ms_per_step": round(1e3 * dtb / k, 3), "model_tflops_per_gpu": round(tfl, 2),
                "model_flops_frac_of_bf16_peak": round(tfl / PEAK_TFLOPS, 4), "final_loss": round(lossl, 4),
                "peak_hbm_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                "hip_graph": leg.fb.graph is not None, "roofline": leg.gemm_roofline(k + w, tag="large_fine")}
            progress(f"large_fine leg: {out['legs']['large_fine']['ms_per_step']} ms/step, {out['legs']['large_fine']['peak_hbm_gb']} GB")
            leg.free()
        if "e2e_generate" in legs:
            out["legs"]["e2e_generate"] = e2e_generate_leg(dev, precision=args.precision)
            progress(f"e2e generate leg: {out['legs']['e2e_generate']['ids_per_sec']} ids/s")
            out["legs"]["e2e_generate_b8"] = e2e_generate_leg(dev, batch=8, precision=args.precision)
            progress(f"e2e generate leg, 8 prompts: {out['legs']['e2e_generate_b8']['ids_per_sec']} ids/s")
            out["legs"]["e2e_generate_b16"] = e2e_generate_leg(dev, batch=16, precision=args.precision)
            progress(f"e2e generate leg, 16 prompts: {out['legs']['e2e_generate_b16']['ids_per_sec']} ids/s")
            if args.precision != "bf16":                   # the dtype BASELINE config 5's sibling configs name, for comparison (outside the tolerance)
                out["legs"]["e2e_generate_bf16"] = e2e_generate_leg(dev, precision="bf16")
                progress(f"e2e generate leg, bf16: {out['legs']['e2e_generate_bf16']['ids_per_sec']} ids/s")
        print(json.dumps(out), flush=True)
    dp.barrier()
    dp.shutdown()


if __name__ == "__main__":
    main()
