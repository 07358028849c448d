#!/bin/bash
# A/B of one build option against the in-tree library, in one GPU call.
#   here (no GPU):   tools/ab_variant.sh build attention -DAT_LEAN=1     -> .variants/libomlm_variant.so (travels with gpurun)
#   on the GPU box:  tools/ab_variant.sh run attention tools/attn_probe.py
# `run` executes the matching kernel parity tests and the probe against BOTH libraries (OMLM_LIB_PATH selects the build).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/open_musiclm_amd/csrc
case "$1" in
build)
    src=$2; shift 2
    name=${VARIANT:-variant}
    mkdir -p "$ROOT/.variants"
    make -C "$CS" >/dev/null
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value "$@" -c "$CS/$src.hip" -o "/tmp/${src}_variant.o"
    objs=""
    for o in gemm attention attention2 norm ffmid ffmid2 embed_ce optim_misc decode vq_fit err; do
        if [ "$o" = "$src" ]; then objs="$objs /tmp/${src}_variant.o"; else objs="$objs $CS/$o.o"; fi
    done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$ROOT/.variants/libomlm_$name.so"
    echo "built $ROOT/.variants/libomlm_$name.so ($src.hip with $*)"
    ;;
run)
    sel=$2; probe=$3
    cd "$ROOT"
    for lib in "" "$ROOT/.variants/libomlm_variant.so"; do
        echo "=== library: ${lib:-in-tree default}"
        OMLM_LIB_PATH=$lib timeout 120 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "$sel" 2>&1 | tail -2
        [ -n "$probe" ] && OMLM_LIB_PATH=$lib timeout 120 python "$probe" 2>&1 | tail -6
    done
    ;;
*) echo "usage: $0 build <source> <flags...> | run <pytest -k expr> [probe.py]"; exit 2;;
esac
