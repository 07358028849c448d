#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4c12; mkdir -p $out
for cfg in "FORCE_ZERO=1" "OMLM_ATTN_DKV3=0" "OMLM_ATTN_DQ2=0" "GRAPH=0"; do
  env $cfg STEPS=9 timeout 100 python tools/fp16_trainer_diag.py > $out/diag_$cfg.log 2>&1
  echo "== $cfg"; grep -E "^8 |bad [0-9]+ kwargs" "$out/diag_$cfg.log" | tail -2 | cut -c1-200
done
