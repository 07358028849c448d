#!/bin/bash
# d(bias) partial-buffer flush: kernel parity + A/B against the same build with a null workspace (atomics) and the Horner diagonal sums
cd "$(dirname "$0")/.."; out=gpurun_out/r3c14; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention" 2>&1 | tail -3
timeout 120 tools/lib_ab LIB_AB_NO_DBIAS_WS=1@.variants/libomlm_atomics.so open_musiclm_amd/libomlm_hip.so .variants/libomlm_horner.so -- attn attn_large attn32 2>&1 | tee $out/lib_ab.log | tail -40
