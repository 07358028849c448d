#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/c5; mkdir -p $O
python tools/decode_probe.py 2>&1 | tail -4
rm -rf /tmp/pd; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o d -- python tools/decode_probe.py > $O/prof.log 2>&1
python tools/prof_summary.py shapes /tmp/pd/d_results.db $O/decode_shapes.md dec; cat $O/decode_shapes.md | cut -c1-170
python tools/prof_summary.py shapes /tmp/pd/d_results.db $O/sample_shapes.md sample; tail -3 $O/sample_shapes.md | cut -c1-170
python - <<'PY'
import sqlite3
c = sqlite3.connect('/tmp/pd/d_results.db')
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = rows[len(rows)//2:]
# gaps between consecutive kernels in steady state
gaps = [(rows[i+1][1]-rows[i][2])/1e3 for i in range(len(rows)-1)]
gaps.sort()
print("kernel-to-kernel gaps (us): median %.2f p10 %.2f p90 %.2f; mean kernel %.2f us" % (gaps[len(gaps)//2], gaps[len(gaps)//10], gaps[9*len(gaps)//10], sum((r[2]-r[1]) for r in rows)/len(rows)/1e3))
PY
