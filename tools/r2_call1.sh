#!/bin/bash
# Round-2 GPU call 1: baseline of the tree (tests incl. the new musiclm_large cases), AT_LEAN A/B, full bench line, kernel trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
O=gpurun_out/c1; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
bash tools/ab_variant.sh run attention tools/attn_probe.py > $O/at_lean_ab.log 2>&1; cat $O/at_lean_ab.log
timeout 900 python bench.py > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-3000; tail -5 $O/bench.err | cut -c1-300
rm -rf /tmp/pf; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $O/prof.log 2>&1
python tools/prof_summary.py stats /tmp/pf/rf_results.db $O/kernel_stats.md --steps 5; head -24 $O/kernel_stats.md | cut -c1-150
python tools/prof_summary.py shapes /tmp/pf/rf_results.db $O/gemm_shapes.md gemm; cat $O/gemm_shapes.md | cut -c1-200
