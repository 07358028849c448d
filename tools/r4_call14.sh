#!/bin/bash
# suite at HEAD; the graph-vs-eager test against the library that still has the memset node (for the record); bench line
cd "$(dirname "$0")/.."; out=gpurun_out/r4c14; mkdir -p $out
timeout 900 python -m pytest tests -q -x -m gpu > $out/pytest.log 2>&1; tail -3 $out/pytest.log
echo "== memset-node library, graph vs eager"; OMLM_LIB_PATH=$PWD/.variants/libomlm_fs_late.so timeout 200 python -m pytest tests/test_gpu_model.py -q -m gpu -k "tracks_eager" > $out/graph_old.log 2>&1; grep -E "passed|failed|^E  +Assert|^E  +assert" $out/graph_old.log | head -8 | cut -c1-250
timeout 600 python bench.py --no-cpu-baseline --legs fp16 --steps 20 --warmup 5 > $out/bench.log 2> $out/bench.err; python -c "
import json; d=json.loads(open('$out/bench.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['gemm_ms_per_step'], d['roofline']['frac'], d['ar_tokens_per_sec'], d['legs']['fp16']['ms_per_step'])"
