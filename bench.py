#!/usr/bin/env python
"""Benchmark of the MI355X TokenConditionedTransformer hot path (BASELINE.json metric:
"train steps/sec + AR tokens/sec, coarse-stage musiclm_small, 1/2/4/8 MI355X").

    python bench.py --gpus N --steps K --warmup W [--precision bf16|bf16x3] [--batch B] [--accum A]

One "step" = one optimizer step of the coarse stage of musiclm_small (dim 1024, depth 6, heads 8, conv-GEGLU FF,
N = 1116 positions = 3 start + 13 clap + 200 semantic + 900 coarse ids -- the exact length the reference's data
pipeline yields; BASELINE's "seq_len 1024" is nominal) on --batch samples per GPU split in --accum micro-batches:
forward + backward (forgetful mask on, FF dropout 0.1 on) + ONE gradient all-reduce + fused clip/AdamW.  Token ids
are synthetic (seeded U{0..1023}) and already resident in HBM when the timed region starts; weights are the
reference's random init (no checkpoints offline).  For N > 1 launch with torch.distributed.run (one rank per GPU).

`value` = whole-job training samples/s (global batch * steps / s, max over ranks); steps/s and AR tokens/s are
reported next to it, as are the live GEMM roofline (HIP events around every MFMA GEMM launch of extra, untimed-for-
`value` instrumented steps) and the CPU oracle timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_SEQ = 1116
PEAK_TFLOPS = {"bf16": 2500.0, "bf16x3": 2500.0}      # dense bf16 MFMA peak (MI355X_MICROARCH.md); bf16x3 issues 3 MFMAs per algorithmic product
P_LIN = 9_566_208                                       # linear MACs / token / layer, musiclm_small


def algorithmic_flops_per_sample(N=N_SEQ, L=6, h=8, dh=64, n_out=1114):
    fwd = 2 * (L * (N * P_LIN + h * dh * N * (N + 1)) + n_out * 1_049_600)
    return 3 * fwd                                       # fwd + bwd


T_START = time.perf_counter()


def progress(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default=os.environ.get("OMLM_PRECISION", "bf16"), choices=["bf16", "bf16x3"])
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU per optimizer step")
    ap.add_argument("--accum", type=int, default=1, help="micro-batches per optimizer step")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying the captured HIP graph")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode-ids", type=int, default=48)
    args = ap.parse_args()

    from open_musiclm_amd import open_musiclm as M
    from open_musiclm_amd import ops
    from open_musiclm_amd.data import SyntheticTokenDataset
    from open_musiclm_amd.optimizer import get_linear_scheduler, get_optimizer
    from open_musiclm_amd.parallel import DataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dp = DataParallel(device=dev)
    rank = dp.rank

    torch.manual_seed(0)                                   # identical replicas on every rank
    model = M.create_coarse_transformer(dim=1024, depth=6, heads=8, attn_dropout=0.0, ff_dropout=0.1,
                                        num_coarse_quantizers=3, precision=args.precision).to(dev)
    stage = M.CoarseStage(coarse_transformer=model, cross_entropy_loss_weights=[0., 0., 1.])
    stage.train()
    optim = get_optimizer(model.parameters(), lr=3e-4, wd=0.01)
    sched = get_linear_scheduler(optim, total_iters=6000)

    assert args.batch % args.accum == 0
    micro = args.batch // args.accum
    ds = SyntheticTokenDataset("coarse", length=1 << 20, seed=1234 + rank)
    n_micro = (args.steps + args.warmup + 2) * args.accum

    def make_batch(i):
        items = [ds[i * micro + j] for j in range(micro)]
        return [torch.cat([it[f] for it in items], 0).to(dev) for f in range(3)]
    batches = [make_batch(i) for i in range(min(n_micro, 8))]       # resident in HBM before timing

    from open_musiclm_amd.graph import GraphedForwardBackward
    fb = GraphedForwardBackward(lambda **kw: stage(**kw, return_loss=True)[0], loss_scale=1.0 / args.accum,
                                enabled=not args.no_graph)
    optim.zero_grad()                                      # adopt the flat parameter / gradient buffers before capture

    def discard():
        optim.mark_grads_dirty()
        optim.zero_grad()
    fb.prepare(dict(clap_token_ids=batches[0][0], semantic_token_ids=batches[0][1], coarse_token_ids=batches[0][2]),
               after_warmup=discard)
    progress(f"micro-step captured into a HIP graph: {fb.graph is not None}" + (f" ({fb.capture_error})" if fb.capture_error else ""))

    def one_step(k):
        optim.zero_grad()
        for a in range(args.accum):
            clap, sem, coarse = batches[(k * args.accum + a) % len(batches)]
            loss = fb(clap_token_ids=clap, semantic_token_ids=sem, coarse_token_ids=coarse)
            optim.mark_grads_dirty()
        dp.allreduce_sum_(optim.flat_grad)
        optim.step(max_grad_norm=0.5, grad_scale=dp.grad_scale())
        sched.step()
        return loss

    progress(f"model + {len(batches)} resident batches ready (micro-batch {micro} x accum {args.accum})")
    for k in range(args.warmup):
        loss = one_step(k)
        torch.cuda.synchronize()
        progress(f"warmup step {k} done")
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        loss = one_step(args.warmup + k)
    dp.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev)
    if dp.is_distributed:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    final_loss = float(loss.item())
    progress(f"timed region: {args.steps} steps in {dt:.3f}s")

    ms_per_step = 1e3 * dt / args.steps
    steps_per_s = args.steps / dt
    samples_per_s = steps_per_s * args.batch * world
    flops_step = algorithmic_flops_per_sample() * args.batch
    model_tflops_per_gpu = flops_step / (dt / args.steps) / 1e12

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
                   "hip_graph": fb.graph is not None,
                   "parity": "bf16x3: logits <=1e-3 vs CPU reference; bf16: <=3e-2 (tests/test_gpu_model.py)"},
        "steps_per_sec": round(steps_per_s, 4),
        "model_tflops_per_gpu": round(model_tflops_per_gpu, 2),
        "model_flops_frac_of_bf16_peak": round(model_tflops_per_gpu / PEAK_TFLOPS[args.precision], 4),
        "final_loss": round(final_loss, 4),
    }

    if os.environ.get("BENCH_HOST_PROFILE") == "1":
        import cProfile, pstats, io
        batch = dict(clap_token_ids=batches[0][0], semantic_token_ids=batches[0][1], coarse_token_ids=batches[0][2])
        fb._eager(batch); torch.cuda.synchronize()
        pr = cProfile.Profile(); pr.enable()
        fb._eager(batch)
        pr.disable(); torch.cuda.synchronize()
        sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(22)
        print(sio.getvalue(), file=sys.stderr)

    if os.environ.get("BENCH_HOST_TIMELINE") == "1":
        # host-side phase stamps over consecutive un-synchronised steps (steady state, queues full)
        import gc
        stamps = []
        torch.cuda.synchronize()
        gc_was = gc.isenabled()
        if os.environ.get("BENCH_GC_OFF") == "1":
            gc.disable()
        for k in range(6):
            kk = args.warmup + args.steps + 20 + k
            t = [time.perf_counter()]
            optim.zero_grad(); t.append(time.perf_counter())
            clap, sem, coarse = batches[kk % len(batches)]
            loss = fb(clap_token_ids=clap, semantic_token_ids=sem, coarse_token_ids=coarse); t.append(time.perf_counter())
            optim.mark_grads_dirty()
            dp.allreduce_sum_(optim.flat_grad); t.append(time.perf_counter())
            optim.step(max_grad_norm=0.5, grad_scale=dp.grad_scale()); t.append(time.perf_counter())
            sched.step(); t.append(time.perf_counter())
            del loss; t.append(time.perf_counter())
            stamps.append(t)
        torch.cuda.synchronize()
        tend = time.perf_counter()
        if gc_was:
            gc.enable()
        names = ["zero_grad", "fwd+bwd", "allreduce", "optim.step", "sched.step", "del loss"]
        for k, t in enumerate(stamps):
            progress(f"host timeline step {k}: start {1e3 * (t[0] - stamps[0][0]):7.1f} ms | " +
                     " ".join(f"{n} {1e3 * (t[i + 1] - t[i]):.1f}" for i, n in enumerate(names)))
        progress(f"host timeline: all 6 steps issued at {1e3 * (stamps[-1][-1] - stamps[0][0]):.1f} ms, GPU idle at {1e3 * (tend - stamps[0][0]):.1f} ms")

    if os.environ.get("BENCH_STEP_TIMES") == "1":
        for k in range(3):
            torch.cuda.synchronize()
            ta = time.perf_counter()
            one_step(args.warmup + args.steps + 10 + k)
            tb = time.perf_counter()
            torch.cuda.synchronize()
            tc = time.perf_counter()
            progress(f"diagnostic step {k}: host enqueue {1e3 * (tb - ta):.1f} ms, until GPU idle {1e3 * (tc - ta):.1f} ms")

    if rank == 0:
        # ---- live roofline of the dominant kernel (the MFMA GEMM): HIP events around every GEMM launch of 2 extra steps
        ops_gemm = ops.gemm
        rec = []

        def timed_gemm(A, B, C_, *, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops_gemm(A, B, C_, M=M, N=N, K=K, **kw)
            e1.record()
            rec.append((e0, e1, 2.0 * M * N * K))
        import open_musiclm_amd.engine as E
        E.ops.gemm = timed_gemm
        def eager_step(k):                                # the graph replays recorded launches: instrument eager ones
            optim.zero_grad()
            for a in range(args.accum):
                clap, sem, coarse = batches[(k * args.accum + a) % len(batches)]
                fb._eager(dict(clap_token_ids=clap, semantic_token_ids=sem, coarse_token_ids=coarse))
                optim.mark_grads_dirty()
            optim.step(max_grad_norm=0.5, grad_scale=dp.grad_scale())
        try:
            for k in range(2):
                eager_step(args.warmup + args.steps + k)
            torch.cuda.synchronize()
        finally:
            E.ops.gemm = ops_gemm
        tot_ms = sum(a.elapsed_time(b) for a, b, _ in rec)
        progress(f"roofline probe: {len(rec)} GEMM launches, {tot_ms:.1f} ms")
        tot_fl = sum(f for _, _, f in rec)
        big = [(a.elapsed_time(b), f) for a, b, f in rec if f > 1e11]
        ach = tot_fl / (tot_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_kernel (all layouts, all GEMM launches of a train step)",
                           "achieved": round(ach, 2), "peak": PEAK_TFLOPS[args.precision], "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_TFLOPS[args.precision], 4), "traffic": None,
                           "launches": len(rec) // 2, "gemm_ms_per_step": round(tot_ms / 2, 3),
                           "large_gemm_achieved": round(sum(f for _, f in big) / (sum(t for t, _ in big) * 1e-3) / 1e12, 2) if big else None}

        # ---- AR decode, B = 1: KV-cached single-row steps (decode.py) and, for comparison, the reference's own scheme
        #      (full re-forward per id, open_musiclm.py:299-319).  Rates include the prompt prefill.
        if not args.no_decode:
            stage.eval()
            g = torch.Generator().manual_seed(99)
            clap = torch.randint(0, 1024, (1, 12, 1), generator=g).to(dev)
            sem = torch.randint(0, 1024, (1, 199), generator=g).to(dev)
            res = {}
            for mode, use_cache, steps_new in (("kv_cache", True, 100), ("reforward", False, max(args.decode_ids // 3, 1))):
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
            # batched decode (8 independent samples per step share every weight read): aggregate ids/s
            clap8 = torch.randint(0, 1024, (8, 12, 1), generator=g).to(dev)
            sem8 = torch.randint(0, 1024, (8, 199), generator=g).to(dev)
            kw8 = dict(clap_token_ids=clap8, semantic_token_ids=sem8, use_cache=True)
            stage.generate(max_time_steps=2, **kw8)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            stage.generate(max_time_steps=50, **kw8)
            torch.cuda.synchronize()
            res["kv_cache_batch8_empty_context"] = round(8 * 50 * 3 / (time.perf_counter() - t1), 2)
            out["ar_tokens_per_sec"] = res
            progress(f"decode {res}")
            stage.train()

        # ---- CPU baseline: the oracle (port of the reference arithmetic) on this box's host cores, bounded sample
        if world == 1 and not args.no_cpu_baseline:
            from oracle import musiclm_oracle as O
            progress(f"cpu baseline on {torch.get_num_threads()} torch threads (os.cpu_count()={os.cpu_count()})")
            spec = O.coarse_spec(dim=1024, depth=6, heads=8)
            sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point and not k.endswith("beta"))
                  for k, v in model.state_dict().items()}
            ids = O.synthetic_ids(spec, 2, [1, 199, 300], seed=1234)
            noise = torch.randn(2, N_SEQ, generator=torch.Generator().manual_seed(1))
            times = []
            for it in range(2):                              # ~30 s of CPU work on this box
                t1 = time.perf_counter()
                l, _, _ = O.wrapper_forward_loss(sd, spec, ids, [0., 0., 1.], forget_noise=noise)
                torch.autograd.grad(l, [v for v in sd.values() if v.requires_grad], allow_unused=True)
                times.append(time.perf_counter() - t1)
                progress(f"cpu baseline iteration {it}: {times[-1]:.2f}s")
            best = min(times)
            out["cpu_baseline"] = {"value": round(2 / best, 4), "unit": "samples/s", "cores": torch.get_num_threads(),
                                   "kind": "port",
                                   "sample": f"{len(times)} fwd+bwd micro-steps of B=2, N=1116, fp32, torch CPU kernels (best; no optimizer step)"}
        print(json.dumps(out), flush=True)
    dp.barrier()
    dp.shutdown()


if __name__ == "__main__":
    main()
