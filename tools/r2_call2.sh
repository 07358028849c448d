#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/c2; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k attention 2>&1 | tail -4
for qkb in 1.0 0; do
  echo "=== attention2 forward, qk_bound=$qkb (0 = online softmax)"
  QKB=$qkb FWD_ONLY=1 timeout 120 python tools/attn_probe.py 2>&1 | tail -1
  QKB=$qkb FWD_ONLY=1 LARGE=1 timeout 120 python tools/attn_probe.py 2>&1 | tail -1
done
echo "=== v1 forward"
OMLM_ATTN_V1=1 FWD_ONLY=1 timeout 120 python tools/attn_probe.py 2>&1 | tail -1
QKB=1.0 FWD_ONLY=1 bash tools/pmc_kernel.sh attn2_fwd python tools/attn_probe.py 2>&1 | grep -E "WAVE_CYCLES|WAIT|ACTIVE_INST_(ANY|VALU|LDS)|MFMA|dur_ns|INSTS_VALU|GUI"
