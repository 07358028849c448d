#!/bin/bash
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r3c21; mkdir -p $out
timeout 120 tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_w3.so .variants/libomlm_prev.so -- attn attn_large 2>&1 | tee $out/lib_ab.log | grep -v "^  d"
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pa && cd "$ROOT" && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pa -o a -- tools/lib_ab open_musiclm_amd/libomlm_hip.so .variants/libomlm_w3.so -- attn > $out/prof.log 2>&1 )
python tools/prof_summary.py stats /tmp/pa/a_results.db $out/kernel_stats.md --steps 1 | head -5; head -22 $out/kernel_stats.md | cut -c1-140
