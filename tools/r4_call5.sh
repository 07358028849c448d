#!/bin/bash
# round 4, fifth GPU call: suite at HEAD; same-box A/B of the rel-pos GEMM route and of the fused q/k-norm epilogue; decode; kernel trace
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c5; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
for cfg in "" "OMLM_RELPOS_PLANES=1" "OMLM_QKNORM_FUSED=0" "OMLM_RELPOS_ASYNC=0"; do
  env $cfg timeout 300 $B > $out/bench_${cfg:-head}.log 2> $out/bench_${cfg:-head}.err
  echo "${cfg:-HEAD}: $(python -c "import json,sys; d=json.loads(open('$out/bench_${cfg:-head}.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])")"
done
timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown.log 2>&1; tail -1 $out/decode_breakdown.log
B=16 timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown_b16.log 2>&1; tail -1 $out/decode_breakdown_b16.log
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $out/prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pf/rf_results.db $out/kernel_stats.md --steps 5 > /dev/null; head -50 $out/kernel_stats.md
