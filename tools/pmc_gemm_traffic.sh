#!/bin/bash
# HBM-side traffic of the GEMM launches of a training step: FETCH_SIZE and WRITE_SIZE in separate passes (guide: they do not
# fit one pass; no trace domains next to --pmc).  FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at
# 64 B, MI355X_MICROARCH.md §HBM), WRITE_SIZE is taken as reported.  Writes gpurun_out/gemm_traffic.{md,json}; the JSON (bytes per
# optimizer step summed over all GEMM launches) is what bench.py reports as roofline.traffic once copied to profiles/gemm_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c; timeout 600 rocprofv3 --pmc $c -d /tmp/pt_$c -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-legs --no-graph > /tmp/pt_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, re, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pt_{c}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gemm" not in n: continue
            m = re.search(r"<(\d+), (\d+), \d+, \d+, (true|false), (true|false), (\w+)", n) or re.search(r"ILi(\d+)ELi(\d+)ELi\d+ELi\d+ELb(\d)ELb(\d)E(DF16b|f)", n)
            if "wgrad_group" in n: m = None; n = "gemm_wgrad_group_kernel (all weight gradients of a backward)"
            key = ("x".join(m.groups()[:2]) + (" A-kmajor" if m.group(3) in ("1", "true") else "") + (" B-kmajor" if m.group(4) in ("1", "true") else "") +
                   (" bf16out" if m.group(5) in ("DF16b", "bf16") else " f32out") + (" walk" if "persist" in n else "")) if m else n[:70]
            agg[(key, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[(key, r["Grid_Size"])]["dur_" + c].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
# steps in the trace: the grouped weight-gradient kernel is launched once per micro-step (the FF-in grid no longer identifies a launch:
# the wide tiles run as a persistent walk of #CUs workgroups since round 4)
wg = [v for (k, g), v in agg.items() if "wgrad_group" in k]
steps = float(sum(len(v["FETCH_SIZE"]) for v in wg)) if wg else 3.0
lines = ["| tile / layout | grid | launches | fetch MB (x2-corrected) | write MB | avg us |", "|---|---:|---:|---:|---:|---:|"]
tot_f = tot_w = 0.0
for (k, grid), d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("dur_FETCH_SIZE", [0]))):
    f = d.get("FETCH_SIZE", []); w = d.get("WRITE_SIZE", []); du = d.get("dur_FETCH_SIZE", [0])
    tot_f += 2 * sum(f) * 1024; tot_w += sum(w) * 1024          # counters are KB
    if len(f) < 2: continue
    lines.append(f"| {k} | {grid} | {len(f)} | {2 * sum(f) / len(f) / 1e3:.1f} | {(sum(w) / len(w) / 1e3) if w else float('nan'):.1f} | {sum(du) / len(du) / 1e3:.1f} |")
out = {"unit": "bytes per optimizer step, all GEMM launches", "fetch_bytes": round(tot_f / steps), "write_bytes": round(tot_w / steps),
       "steps_in_trace": steps, "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB); FETCH x2 (gfx950 counts 128-B requests at 64 B), WRITE as reported",
       "command": "tools/pmc_gemm_traffic.sh", "precision": __import__("os").environ.get("OMLM_BENCH_PRECISION", "fp16ff"),
       "gemm_source_sha16": __import__("hashlib").sha256(b"".join(open("open_musiclm_amd/csrc/" + f, "rb").read() for f in ("gemm.hip", "common.h", "gemm_common.h", "gemm_mx.hip"))).hexdigest()[:16]}
open("gpurun_out/gemm_traffic.json", "w").write(json.dumps(out))
open("gpurun_out/gemm_traffic.md", "w").write("# HBM-side traffic of the GEMM launches of a training step\n\n" + "\n".join(lines) + f"\n\nper step: fetch {tot_f / steps / 1e9:.2f} GB (x2-corrected), write {tot_w / steps / 1e9:.2f} GB over {steps:.0f} traced steps\n")
print("\n".join(lines)); print(json.dumps(out))
PY
