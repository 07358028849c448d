#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c9; mkdir -p $O
rm -rf /tmp/pf3; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf3 -o rf -- python bench.py --precision bf16x3 --steps 3 --warmup 1 --no-decode --no-cpu-baseline --no-legs --no-graph > $O/prof.log 2>&1
python tools/prof_summary.py stats /tmp/pf3/rf_results.db $O/kernel_stats_bf16x3.md --steps 3; head -34 $O/kernel_stats_bf16x3.md | cut -c1-150
