#!/bin/bash
# round 4, eighth GPU call (short): fp16 graph-mode scale diagnostic, the DP equivalence test with rank logs, the two tests behind it
cd "$(dirname "$0")/.."; out=gpurun_out/r4c8; mkdir -p $out
STEPS=2 timeout 150 python tools/fp16_overflow_diag.py > $out/fp16_diag.log 2>&1; tail -8 $out/fp16_diag.log | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "data_parallel_step or fused_batch or top_match or bench_self or bench_two" > $out/pytest_sel.log 2>&1; tail -40 $out/pytest_sel.log | cut -c1-400
