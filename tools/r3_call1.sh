#!/bin/bash
# Round 3, first GPU call: the end-of-round-2 kernels (attention-backward scheduling, rotated GEMM k-loop) were validated and
# timed kernel by kernel through tools/lib_ab only (profiles/r02d_lib_ab.md); this call runs the whole GPU suite, the bench line and
# a kernel-trace profile with them in, plus the quick library A/B against the previous builds on the same box.
#   here (no GPU):   tools/r3_call1.sh build      -> .variants/libomlm_{attn_old,gemm_old}.so, tools/lib_ab
#   gpurun:          tools/r3_call1.sh run        -> gpurun_out/r3c1/*
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/open_musiclm_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value"
case "$1" in
build)
    make -C "$CS" >/dev/null
    mkdir -p "$ROOT/.variants"
    rest() { o=""; for f in gemm attention attention2 norm ffmid ffmid2 embed_ce optim_misc decode vq_fit err; do case " $1 " in *" $f "*) ;; *) o="$o $CS/$f.o";; esac; done; echo $o; }
    for f in attention attention2; do hipcc $FL -DAT_DKV_FENCE=0 -DAT_DQ_BATCH=0 -DA2_DQ_BATCH=0 -c $CS/$f.hip -o /tmp/${f}_old.o; done
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest "attention attention2") /tmp/attention_old.o /tmp/attention2_old.o -o "$ROOT/.variants/libomlm_attn_old.so"
    # dQ kernel register diet: dS * scale formed after the diagonal sums (scratch 84 -> 40 B, same arithmetic); and at one wave per SIMD
    hipcc $FL -DAT_DQ_LATE_SCALE=1 -c $CS/attention.hip -o /tmp/attention_ls.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest attention) /tmp/attention_ls.o -o "$ROOT/.variants/libomlm_attn_ls.so"
    hipcc $FL -DAT_DQ_WPE=1 -DAT_DKV_WPE=1 -DAT_DQP_WPE=1 -c $CS/attention.hip -o /tmp/attention_w1.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest attention) /tmp/attention_w1.o -o "$ROOT/.variants/libomlm_attn_w1.so"
    hipcc $FL -DOMLM_GEMM_ROTATE=0 -c $CS/gemm.hip -o /tmp/gemm_old.o
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest gemm) /tmp/gemm_old.o -o "$ROOT/.variants/libomlm_gemm_old.so"
    # experiment: the 256x256 tile on four waves of 128x128 (one per SIMD; ~2 min to compile) -- rejected in round 1, before the
    # DMA-wait fix and the rotated loop; tools/lib_ab makes the re-test a few seconds
    hipcc $FL -DOMLM_GEMM_W4=1 -c $CS/gemm.hip -o /tmp/gemm_w4.o &
    # lean LDS-DMA issue: 1 = one wait state after the M0 write instead of five, 2 = also without saving / restoring M0
    for v in 1 2; do ( hipcc $FL -DOMLM_DMA_LEAN=$v -c $CS/gemm.hip -o /tmp/gemm_lean$v.o ) & done
    ( hipcc $FL -DOMLM_DMA_SPREAD_ROT=1 -c $CS/gemm.hip -o /tmp/gemm_sp1.o ) &      # next tile's DMA all inside the deferred step (earliest possible)
    ( hipcc $FL -DOMLM_GEMM_BK32=1 -c $CS/gemm.hip -o /tmp/gemm_bk32.o ) &            # 32-deep k-tiles, four-stage ring (three tiles in flight): never run
    wait
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest gemm) /tmp/gemm_bk32.o -o "$ROOT/.variants/libomlm_gemm_bk32.so"
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest gemm) /tmp/gemm_sp1.o -o "$ROOT/.variants/libomlm_gemm_sp1.so"
    hipcc --offload-arch=gfx950 -shared -fPIC $(rest gemm) /tmp/gemm_w4.o -o "$ROOT/.variants/libomlm_gemm_w4.so"
    for v in 1 2; do hipcc --offload-arch=gfx950 -shared -fPIC $(rest gemm) /tmp/gemm_lean$v.o -o "$ROOT/.variants/libomlm_gemm_lean$v.so"; done
    hipcc -O2 "$ROOT/tools/lib_ab.cpp" -o "$ROOT/tools/lib_ab" -ldl
    echo "built .variants/libomlm_attn_old.so, .variants/libomlm_gemm_old.so, tools/lib_ab"
    ;;
run)
    cd "$ROOT"; out=gpurun_out/r3c1; mkdir -p $out
    timeout 60 tools/lib_ab .variants/libomlm_attn_old.so open_musiclm_amd/libomlm_hip.so -- attn attn_large attn32 ffmid ln decode > $out/lib_ab_attn.log 2>&1 || true
    timeout 60 tools/lib_ab .variants/libomlm_gemm_old.so open_musiclm_amd/libomlm_hip.so -- gemm_edge gemm wgrad > $out/lib_ab_gemm.log 2>&1 || true
    timeout 60 tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_attn_ls.so .variants/libomlm_attn_w1.so -- attn attn32 > $out/lib_ab_attn_regs.log 2>&1 || true
    cat $out/lib_ab_attn_regs.log
    # hypothesis test: the same kernels with the per-tile DMA off (OMLM_GEMM_DEBUG=1: compute on stale LDS, results are
    # garbage, time is what the k-loop costs without memory) and with the MFMAs off (=2: what the data movement alone costs)
    cp open_musiclm_amd/libomlm_hip.so .variants/libomlm_dbg1.so; cp open_musiclm_amd/libomlm_hip.so .variants/libomlm_dbg2.so
    timeout 60 tools/lib_ab open_musiclm_amd/libomlm_hip.so OMLM_GEMM_DEBUG=1@.variants/libomlm_dbg1.so OMLM_GEMM_DEBUG=2@.variants/libomlm_dbg2.so -- gemm > $out/lib_ab_gemm_ablate.log 2>&1 || true
    grep -v "differ\|identical" $out/lib_ab_gemm_ablate.log
    timeout 90 tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_gemm_w4.so .variants/libomlm_gemm_lean1.so .variants/libomlm_gemm_lean2.so .variants/libomlm_gemm_sp1.so .variants/libomlm_gemm_bk32.so -- gemm_edge gemm wgrad > $out/lib_ab_w4.log 2>&1 || true
    cat $out/lib_ab_attn.log $out/lib_ab_gemm.log $out/lib_ab_w4.log
    # SQ / LDS counters of the attention backward kernels through the torch-free harness (4 short passes): where the ~7 k cycles per
    # 32x32 block per wave go now (waits, LDS bank conflicts, VALU / MFMA busy)
    timeout 200 tools/pmc_kernel.sh attn_bwd tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_attn_old.so -- attn > $out/pmc_attn_bwd.log 2>&1 || true
    tail -60 $out/pmc_attn_bwd.log
    timeout 600 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1 || true
    tail -3 $out/pytest.log
    timeout 300 python bench.py > $out/bench.log 2> $out/bench.err || true
    tail -1 $out/bench.log
    # kernel trace of the shipped library (profiles/r03a_kernel_stats.md, r03a_gemm_shapes.md)
    ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pf && cd "$ROOT" && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf -o rf -- python bench.py --steps 5 --warmup 2 --no-decode --no-cpu-baseline --no-legs --no-graph > $out/prof.log 2>&1 ) || true
    python tools/prof_summary.py stats /tmp/pf/rf_results.db $out/kernel_stats.md --steps 5 || true
    python tools/prof_summary.py shapes /tmp/pf/rf_results.db $out/gemm_shapes.md gemm || true
    head -40 $out/kernel_stats.md | cut -c1-160
    ;;
*) echo "usage: $0 build | run"; exit 2;;
esac
