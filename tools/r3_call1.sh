#!/bin/bash
# Round 3, first GPU call: parity + same-box A/B of the attention-backward scheduling changes made blind at the end of round 2
# (ISA-verified only: AT_DKV_FENCE, AT_DQ_BATCH in attention.hip; A2_DQ_BATCH in attention2.hip), the rotated GEMM k-loop
# (OMLM_GEMM_ROTATE, off by default) against the in-tree kernel, then the full GPU suite and the bench line.
#   here (no GPU):   tools/r3_call1.sh build      -> .variants/libomlm_attn_old.so  (attention.hip with the three switches off)
#   gpurun:          tools/r3_call1.sh run        -> gpurun_out/r3c1/*
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
case "$1" in
build)
    VARIANT=attn_old "$ROOT/tools/ab_variant.sh" build attention -DAT_DKV_FENCE=0 -DAT_DQ_BATCH=0
    VARIANT=gemm_rot "$ROOT/tools/ab_variant.sh" build gemm -DOMLM_GEMM_ROTATE=1      # rotated k-loop (gemm.hip), default off
    ;;
run)
    cd "$ROOT"; out=gpurun_out/r3c1; mkdir -p $out
    timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attn or attention" > $out/attn_tests.log 2>&1 || true
    tail -3 $out/attn_tests.log
    for lib in "" "$ROOT/.variants/libomlm_attn_old.so"; do
        echo "=== library: ${lib:-in-tree}" >> $out/attn_probe.log
        OMLM_LIB_PATH=$lib timeout 120 python tools/attn_probe.py >> $out/attn_probe.log 2>&1 || true
    done
    tail -20 $out/attn_probe.log
    for lib in "" "$ROOT/.variants/libomlm_gemm_rot.so"; do
        echo "=== library: ${lib:-in-tree}" >> $out/gemm_ab.log
        OMLM_LIB_PATH=$lib timeout 200 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "gemm" >> $out/gemm_ab.log 2>&1 || true
        OMLM_LIB_PATH=$lib timeout 200 python tools/gemm_probe.py >> $out/gemm_ab.log 2>&1 || true
    done
    tail -30 $out/gemm_ab.log
    timeout 600 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1 || true
    tail -3 $out/pytest.log
    timeout 300 python bench.py > $out/bench.log 2> $out/bench.err || true
    tail -1 $out/bench.log
    ;;
*) echo "usage: $0 build | run"; exit 2;;
esac
