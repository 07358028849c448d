#!/bin/bash
# per-shape GEMM table (HIP events, eager) + kernel trace of a train step at HEAD
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r3c16; mkdir -p $out
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 300 python bench.py --no-decode --no-cpu-baseline --no-legs > $out/bench.log 2>$out/bench.err; tail -1 $out/bench.log | cut -c1-400
cat $out/gemm_calls.md
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $out/prof.log 2>&1 ) || true
python tools/prof_summary.py stats /tmp/pf/rf_results.db $out/kernel_stats.md --steps 5 || true
python tools/prof_summary.py shapes /tmp/pf/rf_results.db $out/gemm_shapes.md gemm || true
