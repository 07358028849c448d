#!/bin/bash
# dQ kernel generations at the bench shape (OMLM_ATTN_DQ2=1 forces the LDS-DMA kernel), new bench-shape kernel tests, loss-only step
cd "$(dirname "$0")/.."; out=gpurun_out/r3c17; mkdir -p $out
cp open_musiclm_amd/libomlm_hip.so .variants/libomlm_dq2.so
timeout 120 tools/lib_ab open_musiclm_amd/libomlm_hip.so OMLM_ATTN_DQ2=0@.variants/libomlm_dq2.so -- attn 2>&1 | tee $out/lib_ab_dq2.log | grep -v "^  d"
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -m gpu -k "attention or wgrad_group" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "loss_only or training_step or golden" 2>&1 | tail -3
timeout 300 python bench.py --no-decode --no-cpu-baseline --no-legs > $out/bench.log 2>$out/bench.err; tail -1 $out/bench.log | cut -c1-300
