#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "rvq or clap or index_guards or embed or cross_entropy" 2>&1 | tail -12 | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "unique_consecutive or logits_path or training_step or trainer" 2>&1 | tail -6 | cut -c1-220
timeout 200 python tools/decode_breakdown.py 2>&1 | grep "ids:" 
