#!/bin/bash
# full GPU suite + bench line at the d(bias)-workspace / third-generation dK/dV build
cd "$(dirname "$0")/.."; out=gpurun_out/r3c15; mkdir -p $out
timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -5 | tee $out/pytest.log
timeout 600 python bench.py --legs none > $out/bench.log 2>&1; tail -1 $out/bench.log
