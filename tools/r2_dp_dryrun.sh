#!/bin/bash
# Dry run of the multi-process bench path on a 1-GPU box: 2 ranks share cuda:0, gradients exchanged through gloo (RCCL refuses two
# ranks per GPU).  Validates rendezvous, per-rank data, the flat all-reduce, max-over-ranks timing and the JSON line -- not speed.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OMLM_DP_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 \
  bench.py --gpus 2 --steps 3 --warmup 1 --batch 4 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700
