#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "ffmid" 2>&1 | tail -4 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "bf16x3" 2>&1 | tail -3 | cut -c1-250
for x in 1 0; do
  OMLM_FFMID_IMPL=$x timeout 400 python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs 2> $O/b$x.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FFMID_IMPL=$x', d['ms_per_step'], 'ms/step', d['value'], 'samples/s', 'loss', d['final_loss'])"
done
DTYPE=f32 timeout 200 python tools/ffmid_probe.py 2>&1 | grep "impl"
