#!/bin/bash
# SQ counters of the two GEMM schedules side by side (round 5): tools/lib_ab runs the step's large shapes through the rotated loop
# (OMLM_GEMM_T8=0) and through the half-tile ring (OMLM_GEMM_T8=1) in ONE process; one PMC pass per counter group (no trace domains next
# to --pmc).  Writes gpurun_out/pmc_t8.md: per kernel instantiation, counters per launch and the derived MFMA-busy / wait fractions.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
L=open_musiclm_amd/libomlm_hip.so
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pt8_$i
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pt8_$i -o p --output-format csv -- tools/lib_ab $L OMLM_GEMM_T8=1@$L -- gemm5 wgrad5 > /tmp/pt8_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pt8_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm" not in n: continue
        n = re.sub(r"omlm_(bf16|f16)::", "", re.sub(r"^void ", "", n)).split("(")[0]
        key = (n[:90], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": agg[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
        "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum"]
lines = ["# SQ counters per launch: rotated loop / persistent walk (production, OMLM_GEMM_T8=0) vs half-tile ring (gemm_tile8 / wgrad_group<true, true>)", "",
         "| kernel | grid | launches | avg us (under PMC) | " + " | ".join(c.replace("SQ_", "").replace("_sum", "") for c in cols) + " | wait / wave | inst-wait / wave | MFMA busy of 4 x busy cycles |",
         "|---|---:|---:|---:|" + "---:|" * (len(cols) + 3)]
for (k, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_us", [0]))):
    avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else float("nan")) for c in cols}
    n = len(d.get("SQ_WAVE_CYCLES", []))
    if n < 2: continue
    us = sum(d["dur_us"]) / len(d["dur_us"])
    w = avg["SQ_WAVE_CYCLES"]
    lines.append(f"| `{k}` | {grid} | {n} | {us:.1f} | " + " | ".join(f"{avg[c]:.3g}" for c in cols) +
                 f" | {avg['SQ_WAIT_ANY'] / w:.2f} | {avg['SQ_WAIT_INST_ANY'] / w:.2f} | {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_BUSY_CYCLES']):.2f} |")
open("gpurun_out/pmc_t8.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
