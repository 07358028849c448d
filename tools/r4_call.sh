#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "layernorm or qk_norm" 2>&1 | tail -4 | tee $O/pytest_ln.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
run() { name=$1; shift; echo "== $name" | tee -a $O/step_ab.log; env "$@" timeout 300 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'large', r.get('large_gemm_achieved'), 'loss', d.get('final_loss'))
" | tee -a $O/step_ab.log; }
run ln_old OMLM_LIB_PATH=$PWD/.variants/libomlm_lnold.so
run new X=1
run ln_old OMLM_LIB_PATH=$PWD/.variants/libomlm_lnold.so
run new X=1
