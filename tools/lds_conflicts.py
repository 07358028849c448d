#!/usr/bin/env python3
"""Bank-conflict check of the GEMM's row-major LDS images for ds_read_b128 fragment reads, by the lane-group model of
/opt/skills/guides/MI355X_MICROARCH.md (LDS table: a wave64 ds_read_b128 is served in four groups of 16 lanes, 64 banks of 4 B;
lanes of one group conflict when they touch the same bank at different addresses).  Prints the worst multiplicity per k16 step
(1 = conflict-free).  No GPU needed."""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def worst(addr_of_lane):
    w = 0
    for g in GROUPS:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            for d in range(4):
                banks.setdefault(((a >> 2) + d) % 64, set()).add(a + 4 * d)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def off64(row, kc):      # production image: 128-byte rows, chunk ^ ((row >> 1) & 7)     (gemm.hip lds_off_normal)
    return row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)


def off32(row, kc):      # 32-deep experiment image: 64-byte rows, chunk ^ ((row >> 2) & 3)  (gemm.hip lds_off_normal32)
    return row * 64 + ((kc ^ ((row >> 2) & 3)) << 4)


if __name__ == "__main__":
    for sub0 in (0, 32, 64, 96, 128, 224):
        print(f"rows {sub0:3d}..{sub0 + 31:3d}:  64-deep image {[worst(lambda l: off64(sub0 + (l & 31), 2 * s + (l >> 5))) for s in range(4)]}"
              f"   32-deep image {[worst(lambda l: off32(sub0 + (l & 31), 2 * s + (l >> 5))) for s in range(2)]}")
