#!/bin/bash
# round-2 final evidence: full GPU tests, smoke, bench (+ trace), GEMM HBM traffic (PMC)
TRACE=1 OUT=s2_final bash tools/r2_full.sh
bash tools/pmc_gemm_traffic.sh > gpurun_out/s2_final/traffic.log 2>&1; tail -3 gpurun_out/s2_final/traffic.log | cut -c1-400
