#!/bin/bash
# same-box A/B of the training step under environment levers:  tools/env_ab.sh <outdir> VAR=a VAR=b ...   (each setting twice, interleaved)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/$1; shift; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
for rep in 1 2; do for kv in "$@"; do
  echo -n "$kv  " | tee -a $O/env_ab.log
  env $kv timeout 300 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'issue', r.get('mfma_issue_frac'), 'loss', d.get('final_loss'))
" | tee -a $O/env_ab.log
done; done
