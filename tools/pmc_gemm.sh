#!/bin/bash
# PMC passes over the GEMM probe (one counter group per pass); run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc2
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_THRASHING_STALL_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" "WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  ( cd $R && REPS=3 PITCH_AB=0 ABLATE=0 TILES=${TILES:-256x256} timeout 200 rocprofv3 --pmc $grp -d $OUT/g$i -o p --output-format csv -- python tools/gemm_probe.py > $OUT/g$i.log 2>&1 )
done
cd $R && python tools/prof_summary.py pmc $OUT $OUT/summary.md && cat $OUT/summary.md
