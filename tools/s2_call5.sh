#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c5; mkdir -p $O
for v in "" fr2o2; do
  lib=""; [ -n "$v" ] && lib=$R/.variants/libomlm_$v.so
  echo "=== ${v:-default}"; OMLM_LIB_PATH=$lib timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1"
done
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k ffmid 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu 2>&1 | tail -3 | cut -c1-200
for a in 1 0; do OMLM_RELPOS_ASYNC=$a timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs 2> $O/bench$a.err | tail -1 | cut -c1-300; grep "bench +" $O/bench$a.err | tail -1; done
