"""Turn rocprofv3 output (scratch, under gpurun_out/) into the small tracked summaries under profiles/.

    python tools/prof_summary.py stats gpurun_out/prof8/r8_results.db profiles/r01_kernel_stats.md --steps 5
    python tools/prof_summary.py pmc   gpurun_out/pmc profiles/r01_gemm_pmc.md
    python tools/prof_summary.py shapes gpurun_out/prof/x_results.db profiles/r02_gemm_shapes.md   (per kernel x grid)
"""
import collections
import csv
import glob
import os
import re
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True,
                             text=True, check=True).stdout.splitlines()
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(name):
    m = re.match(r"_Z(\d+)", name)
    if m:  # llvm-cxxfilt does not know the bf16 mangling (DF16b): keep the plain function name + a dtype hint
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        tmpl = re.search(r"ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
        extra = "<%s, %s, %s, %s, %s, %s, bf16>" % tuple(
            (g if i < 4 else ("true" if g == "1" else "false")) for i, g in enumerate(tmpl.groups())) if tmpl else "<bf16>"
        return base + extra
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    if name.startswith("at::native::"):
        name = "torch:" + re.sub(r"<.*$", "", name[len("at::native::"):])
    return name[:88]


def stats(db, out, steps):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    dm = demangle([r[0] for r in rows])
    total = sum(r[2] for r in rows)
    busy_span = list(c.execute("select min(start), max(end) from kernels"))[0]
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{db}` ({steps} timed optimizer steps + warmup/setup in the trace)\n\n")
        f.write(f"total kernel time {total / 1e3:.1f} ms; first-to-last kernel span {(busy_span[1] - busy_span[0]) / 1e6:.1f} ms\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, tot, avg, pct in rows:
            if pct < 0.02:
                continue
            f.write(f"| `{short(dm[name])}` | {calls} | {tot / 1e3:.2f} | {avg:.1f} | {pct:.2f} |\n")
        fam = collections.OrderedDict()
        for name, calls, tot, avg, pct in rows:
            n = dm[name]
            key = ("GEMM (MFMA)" if "gemm" in n else "attention" if "attn" in n else "ffmid (conv-GEGLU-LN-dropout)" if "ffmid" in n
                   else "layernorm / qk-norm" if ("ln_" in n or "qk_norm" in n) else "optimizer" if ("adamw" in n or "sumsq" in n)
                   else "torch / copies" if ("at::native" in n or "rocclr" in n) else "other omlm kernels")
            a = fam.setdefault(key, [0, 0.0])
            a[0] += calls
            a[1] += tot
        f.write("\n## by family\n\n| family | calls | total ms | % |\n|---|---:|---:|---:|\n")
        for k, (calls, tot) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            f.write(f"| {k} | {calls} | {tot / 1e3:.2f} | {100 * tot / total:.1f} |\n")


def pmc(root, out):
    agg = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "*", "*counter_collection.csv"))):
        for r in csv.DictReader(open(path)):
            if "gemm" not in r["Kernel_Name"]:
                continue
            agg[(r["Kernel_Name"], r["Grid_Size"], r["Counter_Name"])].append(
                (float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    dm = demangle(sorted({k[0] for k in agg}))
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --pmc passes (one counter group per pass), source `{root}`\n\n")
        f.write("command: `python tools/gemm_probe.py` (FF-in / dX / dW1 GEMMs of a coarse-small micro-batch of 32: M=35712, d=1024, 2Fp=5472)\n\n")
        f.write("FETCH_SIZE / WRITE_SIZE are KB as reported; per the microarch guide FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced reads "
                "(multiply by 2), WRITE_SIZE is taken as is.\n\n")
        f.write("| kernel | grid | counter | mean value | mean duration us | n |\n|---|---:|---|---:|---:|---:|\n")
        for (name, grid, ctr), v in sorted(agg.items()):
            f.write(f"| `{short(dm[name])}` | {grid} | {ctr} | {sum(x[0] for x in v) / len(v):.4g} | {sum(x[1] for x in v) / len(v) / 1e3:.1f} | {len(v)} |\n")


def shapes(db, out, match="gemm"):
    """Per (kernel instantiation, grid) durations: separates the GEMM shapes of a training step that share one kernel."""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, grid_x, grid_y, workgroup_x, start, end, lds_size, vgpr_count, accum_vgpr_count "
                          "from kernels order by start"))
    rows = rows[len(rows) // 3:]                     # skip warm-up / setup launches
    dm = demangle(sorted({r[0] for r in rows}))
    agg = collections.OrderedDict()
    for n, gx, gy, wx, s, e, lds, vg, ag in rows:
        if match not in n:
            continue
        agg.setdefault((short(dm[n]), gx // max(wx, 1), gy, lds, vg, ag), []).append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    with open(out, "w") as f:
        f.write(f"# per-launch-shape durations of `{match}` kernels (steady-state part of `{db}`)\n\n")
        f.write("| kernel | workgroups | grid.y | LDS B | VGPR | AGPR | launches | avg us | min us | share % |\n|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
        for (name, wg, gy, lds, vg, ag), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{name}` | {wg} | {gy} | {lds} | {vg} | {ag} | {len(v)} | {sum(v) / len(v):.1f} | {min(v):.1f} | {100 * sum(v) / tot:.1f} |\n")


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
        stats(sys.argv[2], sys.argv[3], steps)
    elif mode == "pmc":
        pmc(sys.argv[2], sys.argv[3])
    elif mode == "shapes":
        shapes(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "gemm")
    else:
        raise SystemExit(__doc__)
