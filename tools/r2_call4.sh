#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "OMLM_WT=1 OMLM_GEMM_BAL=0" "OMLM_WT=0 OMLM_GEMM_BAL=0" "OMLM_WT=0 OMLM_GEMM_BAL=1"; do
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'gemm TF', d['roofline']['achieved'], 'gemm ms', d['roofline']['gemm_ms_per_step'])"
done
OMLM_GEMM_BAL=1 PITCH_AB=0 ABLATE=0 TILES=256x256 REPS=10 SHAPES=dW_TN_splitk timeout 200 python tools/gemm_probe.py 2>&1 | grep -v amdgpu | tail -1
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or full_size" 2>&1 | tail -2
