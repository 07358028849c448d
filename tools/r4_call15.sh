#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r4c15; mkdir -p $out
timeout 500 python tools/fp16_seed_sweep.py $out/seed_sweep.md > $out/seed_sweep.log 2>&1; tail -12 $out/seed_sweep.log | cut -c1-250
