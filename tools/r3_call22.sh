#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r3c22; mkdir -p $out
timeout 300 python tools/torch_ops_probe.py 2>&1 | tail -50 | tee $out/torch_ops.log
