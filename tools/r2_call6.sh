#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "cached_decode or generate_matches or musiclm" 2>&1 | tail -3
for cpw in 4 2 1; do echo "CPW=$cpw"; OMLM_DECODE_CPW=$cpw python tools/decode_probe.py 2>&1 | tail -1; done
PRIMED=290 STEPS=20 python tools/decode_probe.py 2>&1 | tail -1
B=8 python tools/decode_probe.py 2>&1 | tail -1
