#!/bin/bash
# ffmid generation-2 check: kernel tests, probe A/B, model-level tests, short bench
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c1; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k ffmid 2>&1 | tail -15 | cut -c1-250
timeout 120 python tools/ffmid_probe.py 2>&1 | grep -v amdgpu | tail -4
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or full_coarse or trainer" 2>&1 | tail -4 | cut -c1-250
timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs > $O/bench.log 2> $O/bench.err; tail -1 $O/bench.log | cut -c1-400; grep "bench +" $O/bench.err | tail -4
