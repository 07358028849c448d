#!/bin/bash
# packed-math dQ kernel vs the previous build; PMC passes over the two backward kernels in use
cd "$(dirname "$0")/.."; out=gpurun_out/r3c19; mkdir -p $out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention" 2>&1 | tail -2
timeout 120 tools/lib_ab OMLM_ATTN_DQ2=1@.variants/libomlm_dq2.so open_musiclm_amd/libomlm_hip.so -- attn attn_large 2>&1 | tee $out/lib_ab.log | grep -v "^  d"
timeout 400 tools/pmc_kernel.sh bwd tools/lib_ab open_musiclm_amd/libomlm_hip.so open_musiclm_amd/libomlm_hip.so -- attn > $out/pmc_attn_bwd.log 2>&1
cat $out/pmc_attn_bwd.log | tail -70
