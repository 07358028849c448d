#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c4; mkdir -p $O
for pp in 0 1; do OMLM_GEMM_PP=$pp timeout 300 python tools/gemm_pp_probe.py 2>&1 | grep "PP="; done | tee $O/pp_probe.txt
OMLM_GEMM_PP=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k gemm 2>&1 | tail -2
OMLM_GEMM_PP=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs 2> $O/bench.err | tail -1 | cut -c1-330; grep "bench +" $O/bench.err | tail -2
