#!/bin/bash
# HBM-side bytes and achieved rate of every kernel of a training step: FETCH_SIZE and WRITE_SIZE in separate PMC passes (no trace
# domains next to --pmc); FETCH_SIZE doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md),
# WRITE_SIZE as reported.  Durations under PMC collection run a few % long.  Writes gpurun_out/hbm_kernels.md.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ph_$c; timeout 600 rocprofv3 --pmc $c -d /tmp/ph_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-legs --no-graph > /tmp/ph_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/ph_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void ", "", r["Kernel_Name"])
            n = re.sub(r"omlm_(bf16|f16)::", "", n)
            key = n.split("(")[0][:70]
            agg[key][c].append(float(r["Counter_Value"]))
            agg[key]["dur_" + c].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, d in agg.items():
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", []); du = d.get("dur_FETCH_SIZE", []) or [0]
    if len(f) < 2 or "gemm" in k: continue
    fb = 2 * sum(f) / len(f) * 1024; wb = (sum(w) / len(w) * 1024) if w else 0.0; us = sum(du) / len(du) / 1e3
    rows.append((sum(du), k, len(f), fb, wb, us))
rows.sort(reverse=True)
lines = ["# HBM-side bytes per launch of the non-GEMM kernels of a training step (PMC passes, B = 32, N = 1116, bf16 mode)", "",
         "| kernel | launches | fetch MB (x2-corrected) | write MB | avg us | achieved TB/s (fetch + write) |", "|---|---:|---:|---:|---:|---:|"]
for _, k, n, fb, wb, us in rows[:28]:
    lines.append(f"| `{k}` | {n} | {fb / 1e6:.1f} | {wb / 1e6:.1f} | {us:.1f} | {(fb + wb) / us / 1e6:.2f} |")
open("gpurun_out/hbm_kernels.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
