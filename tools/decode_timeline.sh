cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/dg; B=1 REPS=1 rocprofv3 --kernel-trace -d /tmp/dg -o p --output-format csv -- python tools/decode_rate.py > /tmp/dg.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/dg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
print(list(rows[0].keys()))
i0 = len(rows) - 200
prev = None
for r in rows[i0:i0 + 45]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{r['Kernel_Name'][:48]:50s} q={r.get('Queue_Id','?')} grid={r.get('Grid_Size','?'):>8s} wg={r.get('Workgroup_Size','?'):>5s} lds={r.get('LDS_Block_Size','?'):>6s} gap {((s - prev) / 1e3) if prev else 0:6.2f} dur {(e - s) / 1e3:6.2f}")
    prev = e
PY
