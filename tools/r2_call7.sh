#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention or ffmid" 2>&1 | tail -3
echo "== v2"; python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -4
echo "== v2 large"; LARGE=1 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -2
echo "== v1"; OMLM_ATTN_V1=1 python tools/attn_probe.py 2>&1 | grep -v amdgpu | tail -4
python tools/ffmid_probe.py 2>&1 | tail -1 | cut -c1-60
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or full_size or large_fine" 2>&1 | tail -2
