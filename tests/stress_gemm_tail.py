"""Reproducer / regression stress for the GEMM tail race (round 4).

    python tests/stress_gemm_tail.py [--iters 300] [--out gpurun_out/stress.json]      # OMLM_LIB_PATH selects the library under test

The LDS-DMA tile kernel (csrc/gemm.hip: gemm_tile_body) issues the "next tile" DMA pieces of its LAST k-tile with out-of-bounds
offsets; the hardware answers them with zeros written into the other LDS stage, which the epilogue re-uses as its transpose patch.
Before the fix nothing ordered those zero-fills before the epilogue's LDS writes: when the pieces were late (other streams' memory
traffic on the same CU) they wiped staged output values.  Non-split GEMMs are deterministic, so ANY bit that differs from the result of
the same launch on an otherwise idle GPU is corruption.  This script runs the rel-pos MLP's GEMM shapes (fp32 operands through the hi/lo
plane route, 128x128 tiles) and two 16-bit trunk shapes on a second stream while the first stream runs HBM-bound copies, and counts
differing elements.  `.variants/libomlm_notailwait.so` (tools/ab_variant.sh notailwait gemm -DOMLM_GEMM_TAIL_WAIT=0) is the library
without the fix.  Test infrastructure only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from open_musiclm_amd import hip, ops            # noqa: E402


def run(iters=300, burst=12):
    import types
    args = types.SimpleNamespace(iters=iters, burst=burst)
    dev = torch.device("cuda:0")
    hip.lib()
    g = torch.Generator().manual_seed(0)
    main_st = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)

    def case_planes(M, N, K, b_kmajor):
        A = torch.randn(M, K, generator=g).to(dev)
        B = (torch.randn(K, N, generator=g) if b_kmajor else torch.randn(N, K, generator=g)).to(dev)
        return dict(name=f"fp32 planes M={M} N={N} K={K} b_kmajor={int(b_kmajor)}", A=A, B=B, M=M, N=N, K=K, kw=dict(b_kmajor=b_kmajor),
                    out_dtype=torch.float32)

    def case_h16(M, N, K, out_dtype, cin):
        A = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16)
        B = torch.randn(N, K, generator=g).to(dev).to(torch.bfloat16)
        c = dict(name=f"bf16 M={M} N={N} K={K} out={str(out_dtype)[6:]} cin={int(cin)}", A=A, B=B, M=M, N=N, K=K, kw={}, out_dtype=out_dtype)
        if cin:
            c["Cin"] = torch.randn(M, N, generator=g).to(dev)
        return c

    def case_p16(M, N, K, planes_out):
        """omlm_gemm_planes16 (precision fp16ff): fp16 hi/lo operand planes, three products; plane output (ref = the hi plane, the lo plane is
        checked through `extra`) or fp32 + residual"""
        T = torch.float16
        A32, B32 = torch.randn(M, K, generator=g).to(dev), (torch.randn(N, K, generator=g) * 0.05).to(dev)
        A, B = A32.to(T), B32.to(T)
        c = dict(name=f"fp16 hi/lo planes M={M} N={N} K={K} out={'planes' if planes_out else 'float32 + residual'}", A=A, B=B, M=M, N=N, K=K, kw={},
                 out_dtype=T if planes_out else torch.float32, A_lo=(A32 - A.float()).to(T), B_lo=(B32 - B.float()).to(T), planes_out=planes_out)
        if not planes_out:
            c["Cin"] = torch.randn(M, N, generator=g).to(dev)
        return c

    cases = [case_planes(1116, 512, 512, True),           # dz = ds W of the rel-pos MLP's backward (24 k-tiles: even)
             case_planes(1116, 512, 512, False),          # its forward a = z W^T
             case_h16(1116, 512, 512, torch.float32, False),
             case_h16(2048, 1024, 1024, torch.float32, True),      # 256x256 tiles, 16 k-tiles, fp32 + residual epilogue
             case_h16(2048, 2048, 1024, torch.bfloat16, False),
             # more than one round of 256x256 tiles: the persistent walk (gemm_bf16_tile_persist_kernel) -- the next tile's first k-tile lands
             # in one LDS stage while the epilogue patches the other; late pieces (second stream's traffic) must not meet a patch
             case_h16(8192, 4096, 1024, torch.bfloat16, False),
             case_h16(8192, 4096, 576, torch.float32, True),         # 9 k-tiles (odd: the stage parity alternates between tiles), fp32 + residual
             # long contractions on more than one round of tiles: the half-tile-ring schedule (gemm_tile8_body, round 5) by default -- five
             # half-tiles in flight behind counted waits; a slot re-requested too early or a short wait would meet late pieces here
             case_h16(8192, 4096, 2752, torch.float32, True),
             case_h16(9000, 4096, 2048, torch.bfloat16, False),       # ragged last tile row
             # the hi/lo-plane route of precision fp16ff (round 5, second half): 3 x the k-tiles on the half-tile ring with a descriptor per plane,
             # the plane-output epilogue (two stores per piece), and the peeled tail of m-tiles on the rotated SPLIT3 loop
             case_p16(3428, 5504, 1024, True),             # 308 tiles: 11 full-round m-tile rows on the ring + a 612-row tail
             case_p16(17920, 1024, 2752, False)]           # 280 tiles: 64 m-tile rows + a 1536-row tail

    def launch(c, out):
        if "A_lo" in c:
            if c["planes_out"]:
                ops.gemm_planes16(c["A"], c["A_lo"], c["B"], c["B_lo"], out, c["lo_out"], M=c["M"], N=c["N"], K=c["K"])
                out.view(torch.int16).bitwise_xor_(c["lo_out"].view(torch.int16))     # one tensor to compare: hi XOR lo (both planes deterministic)
            else:
                ops.gemm_planes16(c["A"], c["A_lo"], c["B"], c["B_lo"], out, M=c["M"], N=c["N"], K=c["K"], Cin=c["Cin"])
            return
        ops.gemm(c["A"], c["B"], out, M=c["M"], N=c["N"], K=c["K"], Cin=c.get("Cin"), **c["kw"])

    # reference results on an idle GPU
    for c in cases:
        ref = torch.empty(c["M"], c["N"], dtype=c["out_dtype"], device=dev)
        if c.get("planes_out"):
            c["lo_out"] = torch.empty_like(ref)      # (shared by the burst's launches: stream-ordered, folded into `out` by the launch itself)
        launch(c, ref)
        torch.cuda.synchronize()
        ref2 = torch.empty_like(ref)
        launch(c, ref2)
        torch.cuda.synchronize()
        assert torch.equal(ref, ref2), "idle-GPU launches differ: " + c["name"]
        c["ref"] = ref
        c["outs"] = [torch.empty_like(ref) for _ in range(args.burst)]
        c["bad_launches"] = 0
        c["bad_elems"] = 0
        c["zeroed_elems"] = 0
        c["launches"] = 0
    big_a = torch.randn(64 << 20, device=dev)            # 256 MB
    big_b = torch.empty_like(big_a)
    t0 = time.time()
    for it in range(args.iters):
        c = cases[it % len(cases)]
        for o in c["outs"]:
            o.fill_(7.0)
        torch.cuda.synchronize()
        side.wait_stream(main_st)
        for k in range(6):                                # ~6 x 80 us of pure HBM streaming on the first stream
            big_b.copy_(big_a)
        with torch.cuda.stream(side):
            for o in c["outs"]:
                launch(c, o)
        torch.cuda.synchronize()
        for o in c["outs"]:
            c["launches"] += 1
            if not torch.equal(o, c["ref"]):
                d = o != c["ref"]
                c["bad_launches"] += 1
                c["bad_elems"] += int(d.sum())
                base = c.get("Cin")
                z = (o == (base if base is not None else 0)) & d
                c["zeroed_elems"] += int(z.sum())
    res = dict(lib=hip.LIB_PATH, iters=args.iters, burst=args.burst, seconds=round(time.time() - t0, 1),
               cases=[{k: c[k] for k in ("name", "launches", "bad_launches", "bad_elems", "zeroed_elems")} for c in cases])
    res["bad_launches_total"] = sum(c["bad_launches"] for c in cases)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--burst", type=int, default=12)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress_gemm_tail.json"))
    args = ap.parse_args()
    res = run(args.iters, args.burst)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res, indent=1))
    return 0 if res["bad_launches_total"] == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
