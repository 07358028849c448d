#!/bin/bash
# Builds the library with extra compile flags for the named source files into .variants/ (travels with gpurun), for an A/B through
# tools/lib_ab or OMLM_LIB_PATH:
#   tools/ab_variant.sh NAME "attention attention2" -DOMLM_DIAG_HORNER=1     -> .variants/libomlm_NAME.so
# Every other object is taken from the in-tree build (made up to date first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/open_musiclm_amd/csrc
name=$1; srcs=$2; shift 2
make -C "$CS" -j16 >/dev/null
tmp=/tmp/omlm_variant_$name; mkdir -p "$tmp" "$ROOT/.variants"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-inline-asm"
objs=""
for o in gemm attention attention2 attention3 norm ffmid ffmid2 embed_ce optim_misc decode vq_fit; do
    if [[ " $srcs " == *" $o "* ]]; then
        hipcc $F "$@" -c "$CS/$o.hip" -o "$tmp/$o.o" &
        objs="$objs $tmp/$o.o"
    else objs="$objs $CS/$o.o"; fi
done
for o in gemm attention attention2 attention3 norm ffmid ffmid2 decode; do
    if [[ " $srcs " == *" $o "* ]]; then
        hipcc $F "$@" -DOMLM_FP16=1 -c "$CS/$o.hip" -o "$tmp/${o}_h.o" &
        objs="$objs $tmp/${o}_h.o"
    else objs="$objs $CS/${o}_h.o"; fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs "$CS/err.o" -o "$ROOT/.variants/libomlm_$name.so"
echo "built .variants/libomlm_$name.so ($srcs with $*)"
