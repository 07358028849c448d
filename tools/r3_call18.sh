#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r3c18; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "loss_only" 2>&1 | tail -30
