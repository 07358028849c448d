#!/bin/bash
# SQ / LDS counter passes over one probe command; prints per-kernel means.   usage: tools/pmc_kernel.sh <kernel-name-substring> <probe command...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
sel=$1; shift
rm -rf /tmp/pk; mkdir -p /tmp/pk; i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" \
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
  "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d /tmp/pk/g$i -o p --output-format csv -- "$@" > /tmp/pk/g$i.log 2>&1 || tail -3 /tmp/pk/g$i.log
done
python - "$sel" <<'PY'
import csv, glob, collections, sys
sel = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pk/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if sel not in n: continue
        key = n[:60] + ' grid ' + r['Grid_Size']
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
        agg[key]['dur_ns'].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    wc = m.get('SQ_WAVE_CYCLES', 0)
    for c in sorted(m):
        extra = f"  ({100 * m[c] / wc:5.1f}% of wave-cycles)" if wc and c.startswith('SQ_') and c not in ('SQ_WAVE_CYCLES',) and ('CYCLES' in c or 'WAIT' in c or 'ACTIVE' in c or 'CONFLICT' in c or 'STALL' in c) else ""
        print(f"   {c:28s} {m[c]:16.4g}{extra}")
PY
