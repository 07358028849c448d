#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c7; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm" 2>&1 | tail -4 | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "bf16x3" 2>&1 | tail -4 | cut -c1-220
for x in 1 0; do
  OMLM_X3_PLANES=$x timeout 400 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs 2> $O/b$x.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('X3_PLANES=$x', d['ms_per_step'], 'ms/step', d['value'], 'samples/s', 'gemm', d['roofline']['achieved'], 'TF', d['roofline']['gemm_ms_per_step'], 'ms', 'loss', d['final_loss'])"
done
