#!/bin/bash
# full GPU evidence run: tests, smoke, bench (+ optional trace with TRACE=1)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/${OUT:-full}; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-4000; grep "bench +" $O/bench.err | tail -12 | cut -c1-300
if [ "$TRACE" = "1" ]; then
  rm -rf /tmp/pf; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $O/prof.log 2>&1
  python tools/prof_summary.py stats /tmp/pf/rf_results.db $O/kernel_stats.md --steps 5; head -30 $O/kernel_stats.md | cut -c1-150
  python tools/prof_summary.py shapes /tmp/pf/rf_results.db $O/gemm_shapes.md gemm
fi
