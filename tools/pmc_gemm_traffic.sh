#!/bin/bash
# HBM-side traffic of the GEMM launches of a training step: FETCH_SIZE and WRITE_SIZE in separate passes (guide: they do not
# fit one pass).  FETCH_SIZE is doubled (gfx950 counts 128-B requests at 64 B for wide coalesced reads), WRITE_SIZE as is.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c; timeout 600 rocprofv3 --pmc $c -d /tmp/pt_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-graph > /tmp/pt_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pt_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm" not in n: continue
            m = re.search(r"ILi(\d+)ELi(\d+)ELi\d+ELi\d+ELb(\d)ELb(\d)E(DF16b|f)", n) or re.search(r"<(\d+), (\d+), \d+, \d+, (true|false), (true|false), (\w+)", n)
            key = "x".join(m.groups()[:2]) + (" kmajor" if m.group(3) in ("1", "true") else "") + (" bf16out" if m.group(5) in ("DF16b", "bf16") else " f32out") if m else n[:40]
            agg[(key, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[(key, r["Grid_Size"])]["dur_" + c].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| tile / layout | grid | launches | fetch MB (x2-corrected) | write MB | avg us |")
print("|---|---:|---:|---:|---:|---:|")
for (k, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_FETCH_SIZE", [0]))):
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", []); du = d.get("dur_FETCH_SIZE", [0])
    if len(f) < 2: continue
    print(f"| {k} | {grid} | {len(f)} | {2 * sum(f) / len(f) / 1e3:.1f} | {(sum(w) / len(w) / 1e3) if w else float('nan'):.1f} | {sum(du) / len(du) / 1e3:.1f} |")
PY
