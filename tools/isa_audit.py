#!/usr/bin/env python3
"""Static audit of the gfx950 ISA hipcc produces for the kernels in open_musiclm_amd/csrc (no GPU needed).

For every kernel: registers, spills, occupancy, and per loop the instruction mix plus the patterns that cost the attention
backward kernels most of their time before they were found by reading the ISA (profiles/r02d_isa_audit.md):
  * DRAIN   an `s_waitcnt vmcnt(<=3)` inside a loop that follows global loads issued in the same iteration with no MFMA in
            between -- a prefetch that is waited for as soon as it is issued (hipcc's wait-count pass merges the pre-loop state
            into the loop header; pin pre-loop loads with an empty asm that takes them as "+v");
  * SERIAL  LDS reads retired one by one (`ds_read` ... `s_waitcnt lgkmcnt(0|1)` four or more times in a row with at most one
            read in flight) -- per-element `cond ? f(lds[...]) : const`, or fragments fed through one register set;
  * SPILL   scratch traffic inside a loop;
  * BRANCHY more than 8 exec-mask branches in a loop body.

usage: tools/isa_audit.py [--all] [file.hip | file.s ...] [-D...]
       default: every .hip in csrc, loops that feed the matrix cores and kernels with spills only (--all: every loop);
       -D flags go to hipcc
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, "open_musiclm_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value", "-S", "--cuda-device-only"]


def compile_to_asm(src, defs):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    src = os.path.abspath(src)
    subprocess.run(["hipcc", *FLAGS, *defs, src, "-o", out], check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(src))
    return out


def demangle(names):
    try:
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return dict(zip(names, r.stdout.split("\n")))
    except Exception:
        return {n: n for n in names}


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"): return "perm"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_exp") or op.startswith("v_rcp") or op.startswith("v_rsq") or op.startswith("v_log") or op.startswith("v_sqrt"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return None


def parse(asm_path):
    lines = open(asm_path).read().split("\n")
    kernels, cur = [], None
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = dict(name=m.group(1), body=[], meta={})
            kernels.append(cur)
            continue
        if cur is None:
            continue
        mm = re.match(r"^; (NumVgprs|ScratchSize|Occupancy|NumSgprs|NumAgprs): (\d+)", ln)
        if mm and mm.group(1) not in cur["meta"]:
            cur["meta"][mm.group(1)] = int(mm.group(2))
        cur["body"].append(ln)
    return [k for k in kernels if "NumVgprs" in k["meta"]]


def loops_of(body):
    """header label -> list of instruction lines of the blocks that belong to the loop (by hipcc's block comments)."""
    loops = collections.OrderedDict()
    owner = None
    for ln in body:
        m = re.match(r"^(\.LBB\w+):\s*;\s*(.*)$", ln)
        if m:
            label, com = m.group(1), m.group(2)
            if "Loop Header" in com:
                owner = label.replace(".L", "")
                loops.setdefault(owner, [])
            else:
                mh = re.search(r"in Loop: Header=(\w+)", com)
                owner = mh.group(1) if mh else None
                if owner:
                    loops.setdefault(owner, [])
            continue
        if re.match(r"^\.LBB\w+:", ln):          # label without a loop comment: outside
            owner = None
            continue
        mb = re.match(r"^; %bb\.\d+:\s*;\s*in Loop: Header=(\w+)", ln)
        if mb:
            owner = mb.group(1)
            loops.setdefault(owner, [])
            continue
        if re.match(r"^; %bb\.\d+:", ln):
            owner = None
            continue
        if owner:
            t = ln.strip()
            if t and not t.startswith(";") and not t.startswith("."):
                loops[owner].append(t)
    return loops


def audit_loop(ins):
    mix = collections.Counter()
    flags = []
    since_load_mfma = None        # number of MFMAs since the last global load in this iteration (None: no load yet)
    drains = 0
    serial_run, best_serial, inflight = 0, 0, 0
    for t in ins:
        op = t.split()[0]
        k = kind(op)
        if k:
            mix[k] += 1
        if k == "vmem" and "load" in op:
            since_load_mfma = 0
        elif k == "mfma" and since_load_mfma is not None:
            since_load_mfma += 1
        elif k == "wait":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) <= 3 and since_load_mfma == 0:
                drains += 1
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                if int(m.group(1)) <= 1 and inflight <= 2:
                    serial_run += 1
                    best_serial = max(best_serial, serial_run)
                else:
                    serial_run = 0
                inflight = int(m.group(1))
        if k == "lds" and "read" in op:
            inflight += 1
    # both patterns are the normal shape of a streaming kernel (load -> use, shuffle ladders); they are findings only where a
    # loop feeds the matrix cores and is meant to run ahead of them
    if drains and mix["mfma"]: flags.append(f"DRAIN x{drains}")
    if best_serial >= 4 and mix["mfma"]: flags.append(f"SERIAL x{best_serial}")
    if mix["scratch"]: flags.append(f"SPILL x{mix['scratch']}")
    nb = sum(1 for t in ins if t.startswith("s_cbranch_execz") or t.startswith("s_cbranch_execnz"))
    if nb > 8: flags.append(f"BRANCHY x{nb}")
    return mix, flags


def main():
    show_all = "--all" in sys.argv[1:]
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    files = args or sorted(os.path.join(CS, f) for f in os.listdir(CS) if f.endswith(".hip"))
    print("| file | kernel | VGPR | scratch B | occ | loop | instr | mfma | valu | trans | lds | perm | vmem | wait | branch | flags |")
    print("|---|---|---:|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---|")
    for f in files:
        asm = f if f.endswith(".s") else compile_to_asm(f, defs)
        ks = parse(asm)
        names = demangle([k["name"] for k in ks])
        for k in ks:
            nm = re.sub(r"\(.*$", "", names[k["name"]]).replace("void ", "")
            if len(nm) > 70: nm = nm[:67] + "..."
            meta = k["meta"]
            loops = loops_of(k["body"])
            rows = []
            for h, ins in loops.items():
                if len(ins) < 24:
                    continue
                mix, flags = audit_loop(ins)
                rows.append((h, len(ins), mix, flags))
            if not rows:
                rows = [("-", 0, collections.Counter(), ["SPILL (outside loops)"] if meta.get("ScratchSize") else [])]
            for h, n, mix, flags in rows:
                if not show_all and not mix["mfma"] and not meta.get("ScratchSize"):
                    continue
                if meta.get("ScratchSize") and not any(x.startswith("SPILL") for x in flags):
                    flags = flags + ["spill outside this loop"]
                print(f"| {os.path.basename(f)} | `{nm}` | {meta.get('NumVgprs')} | {meta.get('ScratchSize')} | {meta.get('Occupancy')} | {h} | {n} | "
                      f"{mix['mfma']} | {mix['valu']} | {mix['trans']} | {mix['lds']} | {mix['perm']} | {mix['vmem']} | {mix['wait']} | {mix['branch']} | "
                      f"{', '.join(flags)} |")


if __name__ == "__main__":
    main()
