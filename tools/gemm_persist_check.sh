#!/bin/bash
# persistent GEMM: kernel tests, A/B probe, concurrent-stream stress
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r4gp; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm" 2>&1 | tail -8 | tee $O/pytest_gemm.log
timeout 300 python tools/gemm_persist_probe.py $O/gemm_persist_probe.md 2>&1 | tail -12 | tee $O/probe.log
timeout 300 python tests/stress_gemm_tail.py 2>&1 | tail -5 | tee $O/stress.log
