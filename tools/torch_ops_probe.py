"""Which torch (non-omlm) kernels run inside one eager train step, by the CPU operator that launched them (torch.profiler)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from open_musiclm_amd.parallel import DataParallel

dev = torch.device("cuda:0")
dp = DataParallel()
leg = bench.TrainLeg(dev, dp, stage="coarse", dim=1024, depth=6, heads=8, precision="bf16", batch=32, accum=1, use_graph=False)
for k in range(2):
    leg.step(k, eager=True)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    leg.step(2, eager=True)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_stack_n=6):
    if ev.device_time_total > 0 and ev.key.startswith("aten::"):
        rows.append((ev.device_time_total, ev.count, ev.key, [s for s in ev.stack if "open_musiclm_amd" in s or "bench.py" in s][:2]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with device time: {tot:.0f} us in {sum(r[1] for r in rows)} calls")
for t, n, k, st in rows[:45]:
    print(f"{t:8.1f} us  x{n:<3d} {k:28s} {' | '.join(s.split('/')[-1] for s in st)}")
