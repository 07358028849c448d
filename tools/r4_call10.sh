#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4c10; mkdir -p $out
for cfg in "X=1" "OMLM_QKNORM_FUSED=0" "OMLM_FUSED_PREP=0" "OMLM_LIB_PATH=$PWD/.variants/libomlm_fs_late.so"; do
  env $cfg STEPS=9 timeout 100 python tools/fp16_trainer_diag.py > $out/diag_$cfg.log 2>&1
  echo "== $cfg"; grep -E "bad:|good:|^8 " "$out/diag_$cfg.log" | tail -3 | cut -c1-900
done
