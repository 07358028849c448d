"""Host-side view of a rocprofv3 --hip-trace --kernel-trace database: where does the launching thread spend its time
between optimizer steps?  Prints a small text report (the db itself is far too large to keep).

    python tools/hip_trace_report.py <results.db> [n_last_steps]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 4
kern = list(db.execute("select name, start, end from kernels order by start"))
adam = [k for k in kern if "adamw" in k[0]]
# one optimizer step = 2 adamw launches (decay / no-decay group): take the second of each pair
marks = [adam[i][2] for i in range(1, len(adam), 2)]
print(f"{len(kern)} kernels, {len(marks)} optimizer steps in trace")
if len(marks) < nlast + 1:
    raise SystemExit("not enough steps")
t0, t1 = marks[-nlast - 1], marks[-1]
print(f"window: last {nlast} steps, {(t1 - t0) / 1e6 / nlast:.2f} ms per step (adamw end -> adamw end)")
busy = sum(min(e, t1) - max(s, t0) for _, s, e in kern if e > t0 and s < t1)
print(f"kernel busy in window: {busy / 1e6 / nlast:.2f} ms per step")
regs = list(db.execute("select name, tid, start, end from regions where end > ? and start < ? order by start", (t0, t1)))
bytid = collections.Counter(r[1] for r in regs)
print("threads:", dict(bytid))
for tid in bytid:
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, t, s, e in regs:
        if t != tid:
            continue
        a = agg[name]
        a[0] += 1
        a[1] += (e - s) / 1e6
        a[2] = max(a[2], (e - s) / 1e6)
    print(f"--- thread {tid}: HIP API time per step (ms), calls per step, longest single call (ms)")
    for name, (n, tot, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"   {name:40s} {tot / nlast:8.2f} {n / nlast:8.1f} {mx:8.2f}")
# the 12 longest individual API calls with their offset inside the step
print("--- longest individual calls in window")
for name, tid, s, e in sorted(regs, key=lambda r: -(r[3] - r[2]))[:12]:
    step = max(i for i, m in enumerate(marks) if m <= max(s, marks[0])) if s >= marks[0] else -1
    print(f"   {name:36s} tid {tid} dur {(e - s) / 1e6:8.2f} ms, starts {(s - marks[step]) / 1e6:8.2f} ms after adamw-end of step {step}")
# GPU idle gaps > 1 ms in window
print("--- GPU idle gaps > 0.5 ms in window")
prev_end = None
for name, s, e in kern:
    if e <= t0 or s >= t1:
        prev_end = max(prev_end or e, e)
        continue
    if prev_end is not None and s - prev_end > 5e5:
        print(f"   gap {(s - prev_end) / 1e6:7.2f} ms before {name[:60]}")
    prev_end = max(prev_end or e, e)

# ---- host timeline around one step boundary (argv[3] = index of the optimizer step, default: middle of the trace)
bi = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
tb = marks[bi]
nxt = min((s for _, s, e in kern if s > tb), default=tb)
print(f"=== host timeline around the end of step {bi}: adamw kernel ends at t=0, next kernel starts at +{(nxt - tb) / 1e6:.2f} ms")
win = list(db.execute("select name, tid, start, end from regions where end > ? and start < ? order by start", (tb - 70e6, nxt + 5e6)))
last_end = {}
lines = 0
for name, tid, s, e in win:
    gap = (s - last_end.get(tid, s)) / 1e6
    dur = (e - s) / 1e6
    if gap > 0.3 or dur > 0.3:
        print(f"   t={(s - tb) / 1e6:8.2f} ms tid {tid} {name:28s} dur {dur:6.2f} ms  (host gap before: {gap:6.2f} ms)")
        lines += 1
        if lines > 60:
            break
    last_end[tid] = e
# first/last launch per thread inside the window -> who launches when
for tid in sorted({w[1] for w in win}):
    ls = [w for w in win if w[1] == tid and "Launch" in w[0]]
    if ls:
        print(f"   tid {tid}: {len(ls)} launches, first at t={(ls[0][2] - tb) / 1e6:.2f}, last at t={(ls[-1][2] - tb) / 1e6:.2f} ms")
