#!/bin/bash
# persistent GEMM: kernel tests, A/B probe, concurrent-stream stress, step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r4gp}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or qk_norm" 2>&1 | tail -8 | tee $O/pytest_gemm.log
timeout 300 python tools/gemm_persist_probe.py $O/gemm_persist_probe.md 2>&1 | tail -14 | tee $O/probe.log
timeout 300 python tests/stress_gemm_tail.py --iters 140 2>&1 | tail -5 | tee $O/stress.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
for m in 0 1 0 1; do echo "== OMLM_GEMM_PERSIST=$m" | tee -a $O/step_ab.log; OMLM_GEMM_PERSIST=$m timeout 300 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'large', r.get('large_gemm_achieved'), 'loss', d.get('final_loss'))
" | tee -a $O/step_ab.log; done
