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
    # the same with attention2.hip's dQ kernel as it was (A2_DQ_BATCH=0): both attention sources of the previous build in one library
    CS=$ROOT/open_musiclm_amd/csrc
    for f in attention attention2; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -DAT_DKV_FENCE=0 -DAT_DQ_BATCH=0 -DA2_DQ_BATCH=0 -c $CS/$f.hip -o /tmp/${f}_old.o
    done
    objs=""; for o in gemm norm ffmid ffmid2 embed_ce optim_misc decode vq_fit err; do objs="$objs $CS/$o.o"; done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/attention_old.o /tmp/attention2_old.o -o "$ROOT/.variants/libomlm_attn_old.so"
    hipcc -O2 "$ROOT/tools/lib_ab.cpp" -o "$ROOT/tools/lib_ab" -ldl               # torch-free A/B harness (seconds per run)
    VARIANT=gemm_rot "$ROOT/tools/ab_variant.sh" build gemm -DOMLM_GEMM_ROTATE=1      # rotated k-loop (gemm.hip), default off
    ;;
run)
    cd "$ROOT"; out=gpurun_out/r3c1; mkdir -p $out
    # seconds, no Python: outputs of the two builds compared bit for bit, both timed (attention backward at both shapes, 3 GEMM shapes)
    timeout 120 tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_attn_old.so attn attn_large > $out/lib_ab_attn.log 2>&1 || true
    cat $out/lib_ab_attn.log
    timeout 120 tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_gemm_rot.so gemm > $out/lib_ab_gemm.log 2>&1 || true
    cat $out/lib_ab_gemm.log
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
