#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r3c23; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 | tee $out/pytest.log
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 600 python bench.py --no-cpu-baseline --no-legs > $out/bench.log 2>$out/bench.err; tail -1 $out/bench.log | cut -c1-2500
head -12 $out/gemm_calls.md
