#!/bin/bash
# round 4, second GPU call: the GEMM tail race -- reproducer against the library without the fix and with it, the hammer in the round-3
# configuration (plane-route MLP GEMMs) on the unfixed library, then suite + hammer + a bench line at HEAD
cd "$(dirname "$0")/.."; out=gpurun_out/r4c2; mkdir -p $out
OMLM_LIB_PATH=$PWD/.variants/libomlm_notailwait.so timeout 300 python tests/stress_gemm_tail.py --iters 400 --out $out/stress_old.json > $out/stress_old.log 2>&1; echo "stress old rc=$?"; grep -E "bad_launches_total|name|\"bad_launches\"" $out/stress_old.log | paste - - | head -8
timeout 300 python tests/stress_gemm_tail.py --iters 400 --out $out/stress_new.json > $out/stress_new.log 2>&1; echo "stress new rc=$?"; grep -E "bad_launches_total" $out/stress_new.log
OMLM_RELPOS_PLANES=1 OMLM_LIB_PATH=$PWD/.variants/libomlm_notailwait.so timeout 400 python tests/hammer_relpos.py --iters 240 --phases A --out $out/hammer_old.json > $out/hammer_old.log 2>&1; echo "hammer old lib + planes: $(grep -c FAIL $out/hammer_old.log) failures"; tail -2 $out/hammer_old.log | cut -c1-300
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 600 python tests/hammer_relpos.py --iters 200 --out $out/hammer.json > $out/hammer.log 2>&1; echo "hammer HEAD: $(grep -c FAIL $out/hammer.log) failures"; grep -E "^phase|HAMMER" $out/hammer.log | cut -c1-400
timeout 600 python bench.py --no-cpu-baseline --legs fp16 > $out/bench.log 2> $out/bench.err; tail -1 $out/bench.log | cut -c1-900
