#!/bin/bash
cd "$(dirname "$0")/.."
for v in 0 1 0 1; do OMLM_PACK_GROUP=$v timeout 300 python bench.py --no-decode --no-cpu-baseline --no-legs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PACK_GROUP=$v', d['ms_per_step'])"; done
