#!/bin/bash
# graph-ordering test against the library that still has the memset node, then HEAD: trainer diag, the new tests, the fp16 overflow test
cd "$(dirname "$0")/.."; out=gpurun_out/r4c13; mkdir -p $out
echo "== memset-node library"; OMLM_LIB_PATH=$PWD/.variants/libomlm_fs_late.so timeout 200 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "graph_replay_is_complete" > $out/order_old.log 2>&1; grep -E "passed|failed|assert worst|^E  " $out/order_old.log | head -6 | cut -c1-200
echo "== HEAD"; STEPS=9 timeout 100 python tools/fp16_trainer_diag.py > $out/diag_head.log 2>&1; grep -E "^8 |bad [0-9]+ kwargs" $out/diag_head.log | tail -2 | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -m gpu -k "graph_replay_is_complete or fp16_overflow or fused_batch or top_match or data_parallel_step or trainer_steps" > $out/pytest_sel.log 2>&1; tail -25 $out/pytest_sel.log | cut -c1-300
