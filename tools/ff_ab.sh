#!/bin/bash
# A/B of the ConvFeedForward forward variants on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4ff; mkdir -p $O
for v in HEAD d1 d2 d4 d7 d8 d15 HEAD; do
  if [ $v = HEAD ]; then L=open_musiclm_amd/libomlm_hip.so; else L=.variants/libomlm_$v.so; fi
  echo "== $v" | tee -a $O/ffmid_ab.log
  OMLM_LIB_PATH=$PWD/$L REPS=20 timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1" | tee -a $O/ffmid_ab.log
done
# correctness of the new forward: kernel tests that cover ffmid
true
