#!/bin/bash
cd "$(dirname "$0")/.."; out=gpurun_out/r3dec; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "full_width" 2>&1 | tail -12
python - <<'PY'
import json
d=json.load(open('gpurun_out/model_report.json'))
for k,v in d.items():
    if 'full_width' in k: print(k, v)
PY
