#!/bin/bash
# inter-kernel gaps of the captured decode step (B from env): kernel trace of tools/decode_rate.py, per-kernel duration and the idle time in front of it
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/dg; REPS=1 rocprofv3 --kernel-trace -d /tmp/dg -o p --output-format csv -- python tools/decode_rate.py > /tmp/dg.log 2>&1 || tail -3 /tmp/dg.log
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/dg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the long call's steps
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    n = b["Kernel_Name"][:70]
    gap[n].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])); dur[n].append(int(b["End_Timestamp"]) - int(b["Start_Timestamp"]))
tot = 0
for n in sorted(dur, key=lambda k: -len(dur[k]))[:12]:
    g = sorted(gap[n]); d = sorted(dur[n])
    print(f"{n:72s} n={len(d):5d} dur med {d[len(d)//2]/1e3:6.2f} us | gap before: med {g[len(g)//2]/1e3:6.2f} us")
PY
