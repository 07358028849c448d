#!/bin/bash
# round 4, sixth GPU call: suite at HEAD (fused batch preparation, sampler), same-box A/B of the rel-pos route in line, decode trace
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c6; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
for cfg in "" "OMLM_RELPOS_PLANES=1" "OMLM_FUSED_PREP=0" "OMLM_RELPOS_ASYNC=1"; do
  env $cfg timeout 300 $B > $out/bench_${cfg:-head}.log 2> $out/bench_${cfg:-head}.err
  echo "${cfg:-HEAD}: $(python -c "import json,sys; d=json.loads(open('$out/bench_${cfg:-head}.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])")"
done
timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown.log 2>&1; tail -1 $out/decode_breakdown.log
B=16 timeout 200 python tools/decode_breakdown.py > $out/decode_breakdown_b16.log 2>&1; tail -1 $out/decode_breakdown_b16.log
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pd && cd "$ROOT" && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python tools/decode_breakdown.py > $out/decode_prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pd/d_results.db $out/decode_b1_kernels.md --steps 1 > /dev/null; head -18 $out/decode_b1_kernels.md
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $out/prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pf/rf_results.db $out/kernel_stats.md --steps 5 > /dev/null; sed -n 8,70p $out/kernel_stats.md | cut -c1-150
