#!/bin/bash
# The last GPU call of a round:  gpurun --timeout 1500 -- tools/final_check.sh [tag]
#   suite as the driver runs it, the hammer + GEMM stress at HEAD, one bench line -- everything into gpurun_out/<tag>/
cd "$(dirname "$0")/.."; tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python tests/hammer_relpos.py --iters 200 --out $out/hammer.json > $out/hammer.log 2>&1; grep -E "^phase|HAMMER" $out/hammer.log | cut -c1-300
timeout 400 python tests/stress_gemm_tail.py --iters 280 --out $out/stress.json > $out/stress.log 2>&1; grep bad_launches_total $out/stress.log
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err
python -c "
import json; d=json.loads(open('$out/bench.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'], d['ar_tokens_per_sec'])"
# optional same-box A/B of library variants on one probe:  FINAL_AB="probe.py libA.so libB.so ..."  (paths relative to the repo root)
if [ -n "$FINAL_AB" ]; then set -- $FINAL_AB; probe=$1; shift; for L in "$@" "$@"; do OMLM_LIB_PATH=$PWD/$L timeout 120 python $probe 2>&1 | tail -1 | tee -a $out/variant_ab.log; done; fi
