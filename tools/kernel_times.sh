#!/bin/bash
# per-kernel average durations of one command under rocprofv3 --kernel-trace --stats:  tools/kernel_times.sh <rows> <command...>
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rows=$1; shift
rm -rf /tmp/kt; rocprofv3 --kernel-trace --stats -d /tmp/kt -o p --output-format csv -- "$@" > /tmp/kt.log 2>&1 || tail -5 /tmp/kt.log
python - $rows <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[1])]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} us")
PY
