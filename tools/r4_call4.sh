#!/bin/bash
# round 4, fourth GPU call: suite at HEAD, bench line (bf16 + fp16 leg), GEMM tile probe after the epilogue change, decode breakdown
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c4; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 600 python bench.py --no-cpu-baseline --legs fp16 > $out/bench.log 2> $out/bench.err; tail -1 $out/bench.log | cut -c1-400; grep -E "decode|fp16 leg|roofline" $out/bench.err | cut -c1-400
SHAPES_ONLY=1 timeout 300 python tools/gemm_shapes_probe.py $out/gemm_tile_probe.md > $out/gemm_tile_probe.log 2>&1; sed -n 1,8p $out/gemm_tile_probe.md
timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown.log 2>&1; tail -2 $out/decode_breakdown.log
B=16 timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown_b16.log 2>&1; tail -1 $out/decode_breakdown_b16.log
cat $out/gemm_calls.md | head -14
