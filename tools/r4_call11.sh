#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4c11; mkdir -p $out
FUSED=1 timeout 120 python tools/nan_finder.py > $out/nan_fused.log 2>&1; grep -v Warn $out/nan_fused.log | tail -45 | cut -c1-200
echo ======; FUSED=0 timeout 120 python tools/nan_finder.py > $out/nan_unfused.log 2>&1; grep -v Warn $out/nan_unfused.log | tail -30 | cut -c1-200
