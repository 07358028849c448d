#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c11; mkdir -p $O
rm -rf /tmp/pf4; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf4 -o rf -- python bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline --legs large_fine --no-graph > $O/prof.log 2>&1
python - <<'PY'
import sqlite3, glob, re, collections
db = glob.glob('/tmp/pf4/rf_results.db')[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = con.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the large leg = launches with N=1817-sized grids come last; take the last 60% of the timeline by time after the biggest gap
agg = collections.defaultdict(lambda: [0, 0.0])
t_split = None
# find attention kernels with H=16: name-based is not possible; use the time of the first ffmid2 fwd launch with grid of B=16 leg: simply take dispatches in the final 45% of kernels by index
n = len(rows); sub = rows[int(n * 0.55):]
for name, s, e, g in sub:
    nm = re.sub(r'\(.*$', '', name)[:70]
    agg[nm][0] += 1; agg[nm][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"last 45% of dispatches: {tot/1e3:.1f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{v[1]/1e3:9.2f} ms {v[0]:6d} calls {v[1]/v[0]:9.1f} us  {k}")
PY
