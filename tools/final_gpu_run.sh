#!/bin/bash
# Round-end GPU evidence run (on the GPU box, from the repo root): tests, smoke, bench, kernel trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1; tail -2 gpurun_out/final_smoke.log
timeout 900 python bench.py > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err; tail -1 gpurun_out/final_bench.log | cut -c1-1500; tail -4 gpurun_out/final_bench.err | cut -c1-300
rm -rf /tmp/pf; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-graph > gpurun_out/final_prof.log 2>&1
python tools/prof_summary.py stats /tmp/pf/rf_results.db gpurun_out/final_kernel_stats.md --steps 5; head -30 gpurun_out/final_kernel_stats.md | cut -c1-160
