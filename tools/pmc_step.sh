#!/bin/bash
# one SQ counter pass over a short eager training run; prints per-kernel means for the non-GEMM kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf /tmp/pmcs; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d /tmp/pmcs -o p --output-format csv -- python bench.py --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-graph > /tmp/pmcs.log 2>&1
python - <<'PY'
import csv, glob, collections
rows = []
for f in glob.glob('/tmp/pmcs/*counter_collection.csv'):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name']
    key = None
    for k in ('ffmid_fwd', 'ffmid_bwd1', 'ffmid_bwd2', 'ln_bwd', 'ln_fwd', 'attn_bwd_dq', 'attn_bwd_dkv', 'attn_fwd', 'qk_norm_bwd'):
        if k in n and 'DF16b' in n or (k in n and 'bf16' in n.lower()):
            key = k
    if key is None: continue
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[key]['dur'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(f"{k:14s} dur {m['dur']/1e3:7.1f} us | VALU insts {m.get('SQ_INSTS_VALU',0)/1e6:8.1f} M | of wave-cycles: VALU-active {100*m.get('SQ_ACTIVE_INST_VALU',0)/wc:5.1f}% LDS-active {100*m.get('SQ_ACTIVE_INST_LDS',0)/wc:5.1f}% wait-inst {100*m.get('SQ_WAIT_INST_ANY',0)/wc:5.1f}% wait-any {100*m.get('SQ_WAIT_ANY',0)/wc:5.1f}% | busy {m.get('SQ_BUSY_CYCLES',0)/1e6:.1f} M wave {wc/1e6:.0f} M")
PY
