#!/bin/bash
# round 4, seventh GPU call: fp16 overflow diagnostic, ffmid A/B (FS_EARLY / pipelined row sums), suite, bench
cd "$(dirname "$0")/.."; ROOT=$PWD; out=gpurun_out/r4c7; mkdir -p $out
timeout 120 python tools/fp16_overflow_diag.py > $out/fp16_diag.log 2>&1; tail -15 $out/fp16_diag.log | cut -c1-330
echo "--- ffmid: HEAD"; timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1" | cut -c1-150
echo "--- ffmid: FS_EARLY=0, round-3 row sums"; OMLM_LIB_PATH=$PWD/.variants/libomlm_fs_late.so timeout 120 python tools/ffmid_probe.py 2>&1 | grep "impl 1" | cut -c1-150
timeout 900 python -m pytest tests -q -x -m gpu --deselect tests/test_gpu_model.py::test_fp16_overflow_is_skipped_and_the_loss_scale_backs_off > $out/pytest.log 2>&1; tail -3 $out/pytest.log
B="python bench.py --no-cpu-baseline --no-legs --no-decode --steps 20 --warmup 5"
for cfg in "" "OMLM_LIB_PATH=$PWD/.variants/libomlm_fs_late.so"; do
  env $cfg timeout 300 $B > $out/bench_x.log 2> $out/bench_x.err
  echo "${cfg:-HEAD}: $(python -c "import json,sys; d=json.loads(open('$out/bench_x.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])")"
done
