#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r3c20; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention" 2>&1 | tail -2
timeout 120 tools/lib_ab .variants/libomlm_prev.so open_musiclm_amd/libomlm_hip.so -- attn attn_large 2>&1 | tee $out/lib_ab.log | grep -v "^  d"
