#!/bin/bash
# round 4, first GPU call: the suite as the driver runs it, then the hammer in the production configuration and (if it is red) with each
# bisecting lever
cd "$(dirname "$0")/.."; out=gpurun_out/r4c1; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 900 python tests/hammer_relpos.py --iters 150 --out $out/hammer.json > $out/hammer.log 2>&1; rc=$?; grep -c FAIL $out/hammer.log; tail -4 $out/hammer.log
if [ $rc -ne 0 ]; then
  for lever in OMLM_RELPOS_ASYNC=0 OMLM_X3_PLANES=0 OMLM_GEMM_SPLITS=1; do
    env $lever timeout 400 python tests/hammer_relpos.py --iters 90 --phases A --out $out/hammer_$lever.json > $out/hammer_$lever.log 2>&1
    echo "$lever: $(grep -c FAIL $out/hammer_$lever.log) failures"; tail -2 $out/hammer_$lever.log
  done
fi
