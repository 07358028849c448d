#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c2; mkdir -p $O
for v in "" fr2o2 fr2o1; do
  lib=""; [ -n "$v" ] && lib=$R/.variants/libomlm_$v.so
  echo "=== ${v:-default}"
  OMLM_LIB_PATH=$lib timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1"
done
OMLM_LIB_PATH=$R/.variants/libomlm_fr2o2.so timeout 200 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k ffmid 2>&1 | tail -2
rm -rf /tmp/pf; REPS=3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python tools/ffmid_probe.py > $O/prof.log 2>&1
python tools/prof_summary.py stats /tmp/pf/rf_results.db $O/ffmid_probe_stats.md --steps 1; grep -E "ffmid|colsum" $O/ffmid_probe_stats.md | cut -c1-140
