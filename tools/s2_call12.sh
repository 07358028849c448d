#!/bin/bash
# d(bias) diagonal sums with immediate-offset permutes: parity first, then A/B, then (only if green) the full evidence run
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c12; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention" > $O/attn_tests.log 2>&1; tail -2 $O/attn_tests.log
if ! grep -q " passed" $O/attn_tests.log || grep -q "failed" $O/attn_tests.log; then echo "ATTENTION TESTS FAILED -- stopping"; tail -30 $O/attn_tests.log | cut -c1-200; exit 0; fi
echo "== asm permutes (default)"; python tools/attn_probe.py 2>&1 | grep -E "^bwd" | head -2
echo "== __shfl permutes";        OMLM_LIB_PATH=$R/.variants/libomlm_diag0.so python tools/attn_probe.py 2>&1 | grep -E "^bwd" | head -2
echo "== large, asm";             LARGE=1 python tools/attn_probe.py 2>&1 | grep -E "^bwd" | head -2
cd $R; TRACE=1 OUT=s2_final3 bash tools/r2_full.sh
