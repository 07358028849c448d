#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r4c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm or qk_norm" 2>&1 | tail -8 | tee $O/pytest_gemm.log
for L in .variants/libomlm_r4head.so open_musiclm_amd/libomlm_hip.so .variants/libomlm_r4head.so open_musiclm_amd/libomlm_hip.so; do OMLM_LIB_PATH=$PWD/$L timeout 120 python tools/qknorm_bwd_probe.py 2>&1 | tail -1 | tee -a $O/qknorm_bwd.log; done
timeout 300 python tools/gemm_persist_probe.py $O/gemm_persist_probe.md 2>&1 | tail -16 | tee $O/probe.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
run() { name=$1; shift; echo "== $name" | tee -a $O/step_ab.log; env "$@" timeout 300 $B 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('ms_per_step', d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'large', r.get('large_gemm_achieved'), 'loss', d.get('final_loss'))
" | tee -a $O/step_ab.log; }
run head_lib OMLM_LIB_PATH=$PWD/.variants/libomlm_r4head.so
run new X=1
run new_smallk0 OMLM_GEMM_SMALLK=0
run new X=1
