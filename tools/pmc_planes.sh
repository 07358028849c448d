#!/bin/bash
# SQ counters of the 3-product plane GEMMs of precision fp16ff next to their single-product forms (tools/planes_probe.py): one PMC pass per
# counter group (no trace domains next to --pmc).  Writes gpurun_out/pmc_planes.md.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1)); rm -rf /tmp/ppl_$i
  timeout 300 rocprofv3 --pmc $grp -d /tmp/ppl_$i -o p --output-format csv -- python tools/planes_probe.py > /tmp/ppl_$i.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ppl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gemm" not in n: continue
        n = re.sub(r"omlm_(bf16|f16)::", "", re.sub(r"^void ", "", n)).split("(")[0]
        key = (n[:100], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": agg[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cols = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE",
        "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"]
lines = ["# SQ counters per launch: the plane GEMMs of fp16ff (gemm_tile8_kernel<.., true> = 3 products on the half-tile ring) and their single-product forms", "",
         "| kernel | grid | launches | avg us (under PMC) | " + " | ".join(c.replace("SQ_", "") for c in cols) + " | wait / wave | inst-wait / wave | MFMA busy of 4 x busy cycles |",
         "|---|---:|---:|---:|" + "---:|" * (len(cols) + 3)]
for (k, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_us", [0]))):
    avg = {c: (sum(d[c]) / len(d[c]) if d.get(c) else float("nan")) for c in cols}
    n = len(d.get("SQ_WAVE_CYCLES", []))
    if n < 2: continue
    us = sum(d["dur_us"]) / len(d["dur_us"])
    w = avg["SQ_WAVE_CYCLES"]
    lines.append(f"| `{k}` | {grid} | {n} | {us:.1f} | " + " | ".join(f"{avg[c]:.3g}" for c in cols) +
                 f" | {avg['SQ_WAIT_ANY'] / w:.2f} | {avg['SQ_WAIT_INST_ANY'] / w:.2f} | {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * avg['SQ_BUSY_CYCLES']):.2f} |")
open("gpurun_out/pmc_planes.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
