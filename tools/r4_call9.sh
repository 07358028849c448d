#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4c9; mkdir -p $out
STEPS=10 timeout 150 python tools/fp16_trainer_diag.py > $out/fp16_trainer_diag.log 2>&1; grep -v Warning $out/fp16_trainer_diag.log | tail -34 | cut -c1-260
