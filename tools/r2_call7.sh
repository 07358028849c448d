#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention" 2>&1 | tail -2
echo "== small default"; python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -3 | head -2
echo "== large default (dq2 auto, windowed dkv)"; LARGE=1 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -3 | head -2
echo "== large DQ2=0"; OMLM_ATTN_DQ2=0 LARGE=1 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -3 | head -2
echo "== large v1 (old dkv staging, old fwd)"; OMLM_ATTN_V1=1 LARGE=1 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -3 | head -2
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or large_fine" 2>&1 | tail -2
