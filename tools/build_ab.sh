#!/bin/bash
# A/B builds of one source of libomlm_hip.so: tools/build_ab.sh <name> <source.hip> [extra hipcc flags...]  ->  tools/ab/libomlm_<name>.so
# (both objects of <source> are rebuilt with the flags, everything else is taken from the csrc build; run `make -C open_musiclm_amd/csrc` first)
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../open_musiclm_amd/csrc"
mkdir -p ../../tools/ab
base=${src%.hip}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-inline-asm"
hipcc $FLAGS -DOMLM_FP16=1 "$@" -c $src -o /tmp/ab_${name}_h.o
hipcc $FLAGS "$@" -c $src -o /tmp/ab_${name}.o
objs=""
for o in *.o; do
  case $o in
    ${base}_h.o) objs="$objs /tmp/ab_${name}_h.o";;
    ${base}.o) objs="$objs /tmp/ab_${name}.o";;
    *) objs="$objs $o";;
  esac
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../tools/ab/libomlm_${name}.so
echo built tools/ab/libomlm_${name}.so
