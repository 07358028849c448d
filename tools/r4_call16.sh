#!/bin/bash
# suite + hammer at the final HEAD on a fresh lease, HBM counters of the non-GEMM kernels, a bench line
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c16; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python tests/hammer_relpos.py --iters 200 --out $out/hammer.json > $out/hammer.log 2>&1; grep -E "^phase|HAMMER" $out/hammer.log | cut -c1-300
timeout 300 python tests/stress_gemm_tail.py --iters 400 --out $out/stress.json > $out/stress.log 2>&1; grep bad_launches_total $out/stress.log
timeout 900 tools/pmc_hbm_kernels.sh > $out/hbm.log 2>&1; cp gpurun_out/hbm_kernels.md $out/ 2>/dev/null; head -24 $out/hbm_kernels.md | cut -c1-200
OMLM_BENCH_GEMM_TABLE=$out/gemm_calls.md timeout 600 python bench.py --no-cpu-baseline --no-legs --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; python -c "
import json; d=json.loads(open('$out/bench.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'], d['ar_tokens_per_sec'])"; grep -E "float32', 'float32'" $out/gemm_calls.md | cut -c1-120
