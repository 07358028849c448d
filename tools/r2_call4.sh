#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm" 2>&1 | tail -2
for bal in 0 1; do
  echo "=== OMLM_GEMM_BAL=$bal"
  OMLM_GEMM_BAL=$bal PITCH_AB=0 ABLATE=0 TILES=256x256 SHAPES=dW_TN_splitk REPS=10 timeout 200 python tools/gemm_probe.py 2>&1 | tail -1
  OMLM_GEMM_BAL=$bal timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
done
OMLM_GEMM_BAL=1 timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or full_size_coarse" 2>&1 | tail -2
