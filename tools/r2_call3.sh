#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k attention 2>&1 | tail -2
echo "full kernel:"; FWD_ONLY=1 python tools/attn_probe.py 2>&1 | tail -1
for v in 1 2 3 4; do echo "ablate $v (1 no arithmetic, 2 no steady DMA, 3 both, 4 no exp2):"; OMLM_LIB_PATH=$R/.variants/libomlm_abl$v.so FWD_ONLY=1 python tools/attn_probe.py 2>&1 | tail -1; done
