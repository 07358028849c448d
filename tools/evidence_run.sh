#!/bin/bash
# The evidence run of a round, one gpurun call:  gpurun --timeout 3000 -- tools/evidence_run.sh [tag]
#   full GPU suite, the default bench line (all legs), a kernel trace of the train step at HEAD (profiles/<tag>_kernel_stats.md,
#   <tag>_gemm_shapes.md), the per-shape GEMM table, the HBM traffic PMC passes of the GEMM launches (profiles/gemm_traffic.json) and
#   the attention kernels timed through tools/lib_ab.  Everything lands in gpurun_out/<tag>/; copy what is to be judged into profiles/.
cd "$(dirname "$0")/.."; ROOT=$PWD; tag=${1:-evidence}; out=gpurun_out/$tag; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 900 python bench.py > $out/bench.log 2> $out/bench.err; tail -1 $out/bench.log | cut -c1-600
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $out/prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pf/rf_results.db $out/kernel_stats.md --steps 5 > /dev/null
python tools/prof_summary.py shapes /tmp/pf/rf_results.db $out/gemm_shapes.md gemm > /dev/null
if [ -z "$EVIDENCE_LIGHT" ]; then      # EVIDENCE_LIGHT=1: without the PMC traffic passes and the attention / decode lib_ab lines (~4 GPU-minutes)
timeout 1300 tools/pmc_gemm_traffic.sh > $out/gemm_traffic.log 2>&1; cp gpurun_out/gemm_traffic.json gpurun_out/gemm_traffic.md $out/ 2>/dev/null
timeout 120 tools/lib_ab open_musiclm_amd/libomlm_hip.so open_musiclm_amd/libomlm_hip.so -- attn attn_large attn32 decode 2>&1 | grep -v "^  d" > $out/lib_ab.log
tail -4 $out/gemm_traffic.log | cut -c1-400
fi
