#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "ffmid or layernorm" 2>&1 | tail -3 | cut -c1-200
timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1"
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "training_step or full_size_coarse or batches_beyond or large_fine_stage_grad or trainer" 2>&1 | tail -3 | cut -c1-200
for a in 1 0; do OMLM_BF16_LN_GRAD=$a timeout 300 python bench.py --steps 10 --warmup 3 --no-decode --no-cpu-baseline --no-legs 2> $O/bench$a.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BF16_LN_GRAD=$a', d['ms_per_step'], 'ms', d['value'], 'samples/s, gemm', d['roofline']['achieved'], 'TF', d['roofline']['gemm_ms_per_step'], 'ms, loss', d['final_loss'])"; done
