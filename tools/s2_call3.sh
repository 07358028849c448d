#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/s2c3; mkdir -p $O
export REPS=1
bash tools/pmc_kernel.sh ffmid2_ python tools/ffmid_probe.py > $O/pmc_ffmid2.txt 2>&1
cat $O/pmc_ffmid2.txt | cut -c1-120
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt; timeout 200 rocprofv3 --pmc $c -d /tmp/pt -o p --output-format csv -- python tools/ffmid_probe.py > /tmp/pt.log 2>&1
  python - $c <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob('/tmp/pt/*counter_collection.csv') + glob.glob('/tmp/pt/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'ffmid' in r['Kernel_Name']: agg[r['Kernel_Name'][:40]].append(float(r['Counter_Value']))
for k, v in agg.items(): print(sys.argv[1], k, 'mean', sum(v) / len(v), 'n', len(v))
PY
done 2>&1 | tee $O/traffic.txt
