#!/bin/bash
# round 4, third GPU call: the extended suite at HEAD, the per-shape GEMM tile probe, a kernel trace of a B = 1 generate call
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c3; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 300 python tools/gemm_shapes_probe.py $out/gemm_tile_probe.md > $out/gemm_tile_probe.log 2>&1; tail -16 $out/gemm_tile_probe.log
timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown.log 2>&1; tail -3 $out/decode_breakdown.log
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python tools/decode_breakdown.py > $out/decode_prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pd/d_results.db $out/decode_b1_kernels.md --steps 1 > /dev/null; head -45 $out/decode_b1_kernels.md
