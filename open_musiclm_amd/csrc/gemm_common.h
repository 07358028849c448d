// Shared device code of the tile GEMM kernels (gemm.hip, gemm_mx.hip): argument block, LDS images, the LDS-DMA stager, fragment reads and the
// LDS-transposed epilogue.  Included inside each translation unit's copy namespace (OMLM_NS), like everything of common.h.
#pragma once
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace OMLM_NS {

#ifndef OMLM_GEMM_TAIL_WAIT
#define OMLM_GEMM_TAIL_WAIT 1
#endif
#ifndef OMLM_GEMM_T8_DEFAULT
#define OMLM_GEMM_T8_DEFAULT 2     // round 5: gemm_tile8_body for the 256 x 256 tiles (see gemm_t8_mode): 2 = where it measured faster
#endif
#ifndef OMLM_EPI_CIN_AHEAD
#define OMLM_EPI_CIN_AHEAD 1       // epilogue: the residual pieces of strip i + 1 are requested before the stores of strip i (tile_epilogue)
#endif
#define BM 128
#define BN 128
#define BK 64
#define NTHREADS 256

struct GemmArgs {
    const void* A;  const void* B;  void* C;  const float* Cin;
    const int* a_map;   // normal A: physical row of logical row m;  k-major A: physical row of k
    const int* b_map;   // normal B: physical row of logical row n;  k-major B: physical row of k
    const int* c_map;   // physical row of logical row m in C / Cin
    long long a_rows, b_rows;   // physical row counts (for the buffer descriptors)
    int M, N, K;
    int lda, ldb, ldc, ldcin;
    float alpha;
    int kt_per_split;   // k-tiles handled by one blockIdx.y slice (split-K); gridDim.y == 1 -> all
    int bal_ck;         // > 0: balanced split-K ("chunked stream-K"): gridDim.x workgroups share tiles x k-tiles evenly; k-tiles per K chunk
    int bal_chunks;     //      number of K chunks
    int debug;          // profiling ablations only (OMLM_GEMM_DEBUG): bit 0 = skip the per-tile DMA, bit 1 = skip the MFMAs
    // hi/lo operand planes ("bf16x3" through the tile kernels; round 5: the ConvFeedForward forward of "fp16ff" on IEEE-half planes): A and B
    // point at a 16-bit hi plane, A_lo / B_lo at the matching lo plane (same layout, its own buffer descriptor: the planes may be separate
    // allocations), and the k-loop runs 3 x the k-tiles: (A_hi, B_hi), (A_hi, B_lo), (A_lo, B_hi).  C_lo (TOUT = h16pl_t instantiations):
    // the result leaves as planes too -- C = rne16(v), C_lo = rne16(v - C) at the same pitch.  c_lo8: the lo plane as bf8 (e5m2: the upper
    // byte of a half, same exponent range, no scale) BYTES at the same element pitch -- omlm_gemm_mx16's h1: half the lo plane's bytes on
    // both sides of it, and v - C keeps 3 significant bits (C + C_lo ~ v to 2^-14 instead of 2^-11 for C alone).
    int split3;
    const void* A_lo; const void* B_lo; void* C_lo;
    int c_lo8;
    // split-K into SLICES instead of atomics (the peeled tail, gemm_impl): split s stores its fp32 partial tile to C + s * c_split_stride
    // (elements); a reduction kernel adds the slices in a fixed order -- deterministic, and no pre-filled C
    long long c_split_stride;
    // l2-norm epilogue (EPI == 1 instantiations, omlm_gemm_qknorm): the first epi_groups 64-column groups of a row leave the kernel as
    // v / max(|v|, 1e-12) * epi_scale[col & 63] with the norm written to epi_norm[row * epi_ldnorm + group]; columns >= c2_col0 (if C2) go
    // to C2 + row * ldc2 + (col - c2_col0)
    const float* epi_scale; float* epi_norm; int epi_groups, epi_ldnorm;
    void* C2; int c2_col0, ldc2;
};

// ---- LDS images ---------------------------------------------------------------------------
// normal tile: [128 rows][64 k] bf16, 128 B per row, 16-B chunk index XOR ((row >> 1) & 7)
__device__ __forceinline__ int lds_off_normal(int row, int kchunk) {
    return row * 128 + ((kchunk ^ ((row >> 1) & 7)) << 4);
}
// k-major tile [64 k][W cols] bf16 (W = 128 or 256), stored as W/128 panels of [64 k][128 cols], 256 B per k-row, with
// the 64-byte piece index of a row XORed with (k & 3).  Why this image:
//   * global side: one LDS-DMA wave-instruction (1 KiB, lane-linear destination) is 4 k-rows x 256 contiguous bytes, i.e.
//     full 128-B lines read by adjacent lanes.  (The first version used a [4 k][16 col]-blocked image whose instruction
//     touched 32 separate 32-byte runs: the DMA phase of the weight-gradient GEMM measured 690 us vs 199 us row-major.)
//   * LDS side: ds_read_b64_tr_b16 is served in two 32-lane groups; a group reads 4 consecutive k-rows x 32 cols (64 B
//     each).  With 256-B rows those would all sit on the same quarter of the 64-bank row; the XOR puts row k on quarter
//     piece ^ (k & 3): four distinct quarters, conflict-free.
__device__ __forceinline__ int lds_off_kmaj(int k, int colbyte) {
    const int col = colbyte >> 1;
    return (col >> 7) * 16384 + k * 256 + ((((col >> 5) & 3) ^ (k & 3)) << 6) + (col & 31) * 2;
}

template <typename T, bool KMAJ>
struct Stager {
    static constexpr bool PRECISE = elt_traits<T>::precise;
    // per thread: 4 pieces of 8 elements
    u32x4 r[PRECISE ? 8 : 4];

    // tile origin: (r0 along the tile's 128-wide dim, k0 along the contraction)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, const int* map, int ld, int nvalid128,
                                         int r0, int k0, int K) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + NTHREADS * i;
            unsigned off;
            if (!KMAJ) {
                const int row = c >> 3, kc = c & 7;               // 8 chunks (of 8 k) per row
                const int gr = r0 + row, gk = k0 + kc * 8;
                const bool ok = (gr < nvalid128) && (gk < K);
                long long pr = gr;
                if (ok && map) pr = map[gr];
                off = ok ? (unsigned)((pr * ld + gk) * (long long)sizeof(T)) : OOB_OFF;
            } else {
                const int kr = c >> 4, cc = c & 15;               // 16 chunks (of 8 cols) per k row
                const int gk = k0 + kr, gc = r0 + cc * 8;
                const bool ok = (gk < K) && (gc < nvalid128);
                long long pr = gk;
                if (ok && map) pr = map[gk];
                off = ok ? (unsigned)((pr * ld + gc) * (long long)sizeof(T)) : OOB_OFF;
            }
            if (PRECISE) {
                r[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                r[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off == OOB_OFF ? OOB_OFF : off + 16, 0, 0);
            } else {
                r[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            }
        }
    }

    __device__ __forceinline__ void store(char* lds_hi, char* lds_lo) {
        const int t = threadIdx.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = t + NTHREADS * i;
            int o;
            if (!KMAJ) o = lds_off_normal(c >> 3, c & 7);
            else       o = lds_off_kmaj(c >> 4, (c & 15) * 16);
            if (PRECISE) {
                u32x4 hi, lo;
                const u32x4 x0 = r[2 * i], x1 = r[2 * i + 1];
                unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                split_pair(u2f(x0[0]), u2f(x0[1]), h0, l0);
                split_pair(u2f(x0[2]), u2f(x0[3]), h1, l1);
                split_pair(u2f(x1[0]), u2f(x1[1]), h2, l2);
                split_pair(u2f(x1[2]), u2f(x1[3]), h3, l3);
                hi[0] = h0; hi[1] = h1; hi[2] = h2; hi[3] = h3;
                lo[0] = l0; lo[1] = l1; lo[2] = l2; lo[3] = l3;
                *(u32x4*)(lds_hi + o) = hi;
                *(u32x4*)(lds_lo + o) = lo;
            } else {
                *(u32x4*)(lds_hi + o) = r[i];
            }
        }
    }
};

// fragment of a 32-wide sub-tile (rows/cols sub0..sub0+31 of the 128-wide tile), k16 step s
template <bool KMAJ>
__device__ __forceinline__ h16x8 read_frag(const char* lds, int sub0, int s, int lane) {
    if (!KMAJ) {
        const int row = sub0 + (lane & 31);
        const int kc = 2 * s + (lane >> 5);
        return *(const h16x8*)(lds + lds_off_normal(row, kc));
    } else {
        // two transpose reads of 4 k each: the lane ends up with column sub0 + (lane & 31), k = 16s + 8*(lane>>5) + 0..7.
        // Lane i of a 16-lane group supplies the address of the 8-byte piece (k-row i>>2, cols 4*(i&3)..+3) of the group's
        // [4 k][16 col] block; groups 0/1 are the two 16-col halves of the sub-tile, groups 2/3 the same for k + 8.
        const int i16 = lane & 15, grp = lane >> 4, r = i16 >> 2;
        const int k = 16 * s + 8 * (grp >> 1) + r;
        const char* base = lds + (sub0 >> 7) * 16384 + k * 256 + ((((sub0 >> 5) & 3) ^ r) << 6) + (16 * (grp & 1) + 4 * (i16 & 3)) * 2;
        s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
        s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 1024));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(h16x8, v);
    }
}

// epilogue: C-layout of v_mfma_f32_32x32x16: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
template <typename TOUT>
__device__ __forceinline__ void epilogue(const GemmArgs& g, f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int lane) {
    TOUT* C = (TOUT*)g.C;
    const bool split = gridDim.y > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (row >= g.M) continue;
            const long long prow = g.c_map ? (long long)g.c_map[row] : (long long)row;
            if (prow < 0) continue;          // row dropped by the scatter map (padding rows of a repacked weight)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = n0 + wn + 32 * j + (lane & 31);
                if (col >= g.N) continue;
                float v = g.alpha * acc[i][j][e];
                if (split) {                 // split-K slices accumulate into C (host guarantees fp32 C and Cin == C)
                    unsafeAtomicAdd((float*)g.C + prow * g.ldc + col, v);
                } else {
                    if (g.Cin) v += g.Cin[prow * g.ldcin + col];
                    store_from_float(C + prow * g.ldc + col, v);
                }
            }
        }
    }
}

template <typename T, bool A_KMAJ, bool B_KMAJ, typename TOUT>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs g) {
    constexpr bool PRECISE = elt_traits<T>::precise;
    constexpr int PLANE = BM * BK * 2;                       // 16 KiB per bf16 plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As_hi = smem;
    char* Bs_hi = smem + PLANE;
    char* As_lo = smem + 2 * PLANE;
    char* Bs_lo = smem + 3 * PLANE;

    // XCD-aware tile order: consecutive tiles of one XCD walk along N for a fixed M tile, so the
    // A panel of a tile row stays in that XCD's L2.
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // super-tile order: 8 tile-rows x all tile-columns per group, column-major inside the group, so the ~64 workgroups
    // co-resident on one XCD cover an ~8 x 8 patch of C (8 A panels + 8 B panels ~ 4 MiB: the XCD's L2)
    const int gsz = 8 * tiles_n;
    const int grp = bid / gsz, first_m = grp * 8;
    const int rows_in = min(8, tiles_m - first_m);
    const int tm = first_m + (bid - grp * gsz) % rows_in, tn = (bid - grp * gsz) / rows_in;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(g.A, (unsigned long long)g.a_rows * g.lda * sizeof(T));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(g.B, (unsigned long long)g.b_rows * g.ldb * sizeof(T));

    Stager<T, A_KMAJ> sa;
    Stager<T, B_KMAJ> sb;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk_all = (g.K + BK - 1) / BK;
    const int kt0 = blockIdx.y * g.kt_per_split;
    const int nk = min(nk_all, kt0 + g.kt_per_split);
    sa.load(rsA, g.a_map, g.lda, g.M, m0, kt0 * BK, g.K);
    sb.load(rsB, g.b_map, g.ldb, g.N, n0, kt0 * BK, g.K);

    for (int kt = kt0; kt < nk; ++kt) {
        sa.store(As_hi, As_lo);
        sb.store(Bs_hi, Bs_lo);
        __syncthreads();
        if (kt + 1 < nk) {
            sa.load(rsA, g.a_map, g.lda, g.M, m0, (kt + 1) * BK, g.K);
            sb.load(rsB, g.b_map, g.ldb, g.N, n0, (kt + 1) * BK, g.K);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h16x8 ah[2], bh[2], al[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = read_frag<A_KMAJ>(As_hi, wm + 32 * i, s, lane);
                bh[i] = read_frag<B_KMAJ>(Bs_hi, wn + 32 * i, s, lane);
                if (PRECISE) {
                    al[i] = read_frag<A_KMAJ>(As_lo, wm + 32 * i, s, lane);
                    bl[i] = read_frag<B_KMAJ>(Bs_lo, wn + 32 * i, s, lane);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (PRECISE) {
                        acc[i][j] = OMLM_MFMA_32x32x16(al[i], bh[j], acc[i][j]);
                        acc[i][j] = OMLM_MFMA_32x32x16(ah[i], bl[j], acc[i][j]);
                    }
                    acc[i][j] = OMLM_MFMA_32x32x16(ah[i], bh[j], acc[i][j]);
                }
        }
        __syncthreads();
    }

    epilogue<TOUT>(g, acc, m0, n0, wm, wn, lane);
}


// ---- bf16 fast path, generalised tile: BM_ x BN_ x 64 per workgroup, waves of WM_ x WN_ -------------------------------
// Measured on MI355X: the 128x128 kernel saturates the L2 -> LDS DMA path (~10-12 TB/s chip-wide) at ~600 TFLOP/s because
// a 128x128x64 tile moves 32 KiB per 2.1 MFLOP (64 FLOP/B).  256x256 (8 waves of 128x64) doubles that to 128 FLOP/B,
// 256x128 (8 waves of 64x64) gives 85 FLOP/B for the narrow-N GEMMs where 256-wide tiles would leave CUs idle.
// LDS-DMA issued as inline asm.  With the builtin (raw_ptr_buffer_load_lds) hipcc tracks "a pending write to LDS" and protects
// every LDS read it cannot prove disjoint: plain ds_read_b128 fragment reads pass, but the transpose-read builtin
// (ds_read_b64_tr_b16: every fragment of a k-major operand) gets an s_waitcnt vmcnt(0) in front of it -- seen in the ISA of the
// k-major kernels right after the DMA issue of each k16 step, i.e. the weight-gradient GEMMs waited for the tile they had just
// requested four times per k-tile (~620 TFLOP/s against ~1000 for the same contraction with k-contiguous operands).  The asm
// form is invisible to that bookkeeping; the k-loop's barrier is preceded by an explicit s_waitcnt vmcnt(0).
typedef u32x4 dma_rsrc;
__device__ __forceinline__ dma_rsrc make_dma_rsrc(const void* p, unsigned long long bytes) {      // wave-uniform inputs only
    const unsigned long long a = (unsigned long long)p;
    dma_rsrc r;
    r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32) & 0xFFFFu; r[2] = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes; r[3] = 0x00020000u;
    return r;
}
// (measured and dropped, round 3 first call, profiles/r03a_lib_ab.md: one wait state instead of five after the M0 write and no M0
// save / restore -- +-2 %, inside the run-to-run band of the probe)
__device__ __forceinline__ void dma_issue(dma_rsrc rs, unsigned lds_dst, unsigned off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(lds_dst), "s"(rs) : "memory");
}

// the same with a wave-uniform byte offset in the instruction's SGPR offset field: the per-lane offset register is loop-invariant
__device__ __forceinline__ void dma_issue_s(dma_rsrc rs, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(rs), "s"(soff) : "memory");
}

template <bool KMAJ, int ROWS, int NWAVES>
struct DmaStagerT {
    static constexpr int UPW = (ROWS / 8) / NWAVES;       // 1-KiB units per wave per tile
    unsigned base[UPW];
    int kidx[UPW];
    unsigned vfast[UPW];      // whole-k-tile form (K % 64 == 0, no k-row map): per-lane offset of k-tile 0; the tile's k offset travels as an SGPR
    __device__ __forceinline__ void init(const int* map, int ld, int nvalid, int r0, int wave, int lane) {
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int b = wave + NWAVES * i;
            if (!KMAJ) {
                const int row = 8 * b + (lane >> 3), slot = lane & 7;
                const int kc = slot ^ ((row >> 1) & 7);
                const int gr = r0 + row;
                const bool ok = gr < nvalid;
                const long long pr = (ok && map) ? (long long)map[gr] : (long long)gr;
                base[i] = ok ? (unsigned)((pr * ld + kc * 8) * 2) : OOB_OFF;
                kidx[i] = kc * 8;
                vfast[i] = base[i];
            } else {
                // unit b = (panel b >> 4, k-group b & 15): 4 k-rows x 256 B; lane = (row lane >> 4, 16-B chunk lane & 15)
                const int krow = 4 * (b & 15) + (lane >> 4);
                const int piece = ((lane & 15) >> 2) ^ (lane >> 4);                 // image piece -> logical piece (k & 3 == lane >> 4)
                const int gc = r0 + 128 * (b >> 4) + 32 * piece + 8 * (lane & 3);
                base[i] = gc < nvalid ? (unsigned)(gc * 2) : OOB_OFF;
                kidx[i] = krow;
                vfast[i] = gc < nvalid ? base[i] + (unsigned)krow * (unsigned)(ld * 2) : OOB_OFF;
            }
        }
    }
    // one 1-KiB unit (one wave-instruction).  `live` false -> out-of-bounds offset: the DMA writes zeros, no memory traffic
    // KMAP: a k-row map is honoured (an ORDINARY global load inside the k-loop).  It is a template switch because its mere
    // presence -- even behind a null-pointer test -- makes hipcc wait vmcnt(0) before every LDS-DMA issue and every
    // fragment read, which serialised the whole pipeline of the k-major GEMMs (2x slower; found in the ISA).
    // poff: byte offset of the operand plane this k-tile reads (0, or the lo plane of a split3 GEMM)
    template <bool KMAP, bool FAST = false>
    __device__ __forceinline__ void issue_one(int i, dma_rsrc rs, const int* map, int ld, int k0, int K,
                                              char* lds_tile, int wave, bool live, int aux = 0, unsigned poff = 0u) {
        const int b = wave + NWAVES * i;
        if constexpr (!KMAP && FAST) {
            // every k of the tile is inside K (host: K % 64 == 0): no per-piece compare / select / add chain (~6 VALU instructions per
            // piece, 50 per k-tile per wave next to 32 MFMAs) -- one select for a dead tile (`live` false: past the last one)
            dma_issue_s(rs, (unsigned)(size_t)LDS_PTR(char, lds_tile) + (unsigned)(b * 1024), live ? vfast[i] : OOB_OFF,
                        (KMAJ ? (unsigned)k0 * (unsigned)(ld * 2) : (unsigned)(k0 * 2)) + poff);
            return;
        }
        unsigned off;
        if (!KMAJ) {
            off = (live && base[i] != OOB_OFF && k0 + kidx[i] < K) ? base[i] + (unsigned)(k0 * 2) + poff : OOB_OFF;
        } else {
            const int gk = k0 + kidx[i];
            const bool ok = live && base[i] != OOB_OFF && gk < K;
            unsigned pr = (unsigned)gk;
            if (KMAP) { if (ok && map) pr = (unsigned)map[gk]; }
            off = ok ? base[i] + pr * (unsigned)(ld * 2) + poff : OOB_OFF;     // 32-bit: the host checks that the operand is < 4 GiB
        }
        (void)aux;
        dma_issue(rs, (unsigned)(size_t)LDS_PTR(char, lds_tile) + (unsigned)(b * 1024), off);
    }
    template <bool KMAP, bool FAST = false>
    __device__ __forceinline__ void issue(dma_rsrc rs, const int* map, int ld, int k0, int K,
                                          char* lds_tile, int wave, unsigned poff = 0u) {
#pragma unroll
        for (int i = 0; i < UPW; ++i) issue_one<KMAP, FAST>(i, rs, map, ld, k0, K, lds_tile, wave, true, 0, poff);
    }
};

// Output type tag of the plane-output instantiations: 16-bit elements, C receives rne16(v) and GemmArgs::C_lo receives rne16(v - C).
struct h16pl_t { h16_t v; };

// Epilogue shared by the tile kernels.  In the MFMA C-layout a lane owns ONE column and 16 rows, so direct stores are 2- or 4-byte
// scatters (measured: ~480 of 650 us of the FF-in GEMM).  Each 32-row strip of the wave's tile is therefore transposed
// through a per-wave LDS patch (the k-loop stages are dead: the caller has passed a barrier) and written as 16-byte
// row-contiguous stores: 8 bf16 / 4 fp32 per lane, full 128-byte lines per row.
// SLICE: the instantiation may be launched as the slice-storing split-K of a peeled tail (GemmArgs::c_split_stride; 128 x 128 fp32-output kernels
// only -- every other kernel keeps the code and registers it was measured with)
template <int MI, int NJ, int WN_, typename TOUT, int EPI = 0, bool AHEAD = (OMLM_EPI_CIN_AHEAD != 0) && (MI * NJ <= 4), bool SLICE = false>
__device__ __forceinline__ void tile_epilogue(const GemmArgs& g, f32x16 (&acc)[MI][NJ], char* smem, int m0, int n0,
                                              int wm, int wn, int wave, int lane, int dbg, bool split, float* patch = nullptr, int ksplit = 0) {
    constexpr int SROW = WN_ + 4;                                  // padded row (floats), keeps 16-B alignment
    float* stg = patch ? patch : (float*)smem + (size_t)wave * 32 * SROW;      // (persistent kernel: the patch sits where no DMA lands)
    constexpr int VEC = sizeof(TOUT) == 2 ? 8 : 4;                 // elements per 16-byte store
    constexpr int LPR = WN_ / VEC;                                 // lanes per row
    constexpr int RPP = 64 / LPR;                                  // rows per pass
    constexpr bool PLANES = std::is_same<TOUT, h16pl_t>::value;
    using TST = std::conditional_t<PLANES, h16_t, TOUT>;           // element type of the stores
    TST* C = (TST*)g.C;
    const bool vec_ok = (g.ldc % VEC == 0) && (((uintptr_t)g.C & 15) == 0) &&
                        (!g.Cin || ((g.ldcin % 4 == 0) && (((uintptr_t)g.Cin & 15) == 0)));
    const int hi = lane >> 5;
    if (dbg & 4) return;
    if (split) {      // split-K slices accumulate into fp32 C (Cin == C): the C-layout already gives 128-B coalesced atomics
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * hi;
                if (row >= g.M) continue;
                const long long prow = g.c_map ? (long long)g.c_map[row] : (long long)row;
                if (prow < 0) continue;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int col = n0 + wn + 32 * j + (lane & 31);
                    if (col >= g.N) continue;
                    if (SLICE && g.c_split_stride) ((float*)g.C + (long long)ksplit * g.c_split_stride)[prow * g.ldc + col] = g.alpha * acc[i][j][e];
                    else unsafeAtomicAdd((float*)g.C + prow * g.ldc + col, g.alpha * acc[i][j][e]);
                }
            }
        return;
    }
    constexpr int NP = 32 / RPP;                                   // passes per 32-row strip
    const bool cin_pref = g.Cin != nullptr && vec_ok;
    // Residual (Cin) pieces of a whole strip are requested BEFORE its LDS transposition: one memory round trip per strip, hidden behind
    // the ds_write / ds_read pass.  (Round 4: with the loads inside the pass loop -- behind its row / column early-outs -- every pass
    // waited for its own piece: 32 dependent round trips per 256x256 tile, to_out (K = 512) 130 us against a 60 us HBM floor at every
    // tile size, FF-out ~80 us of epilogue.)  AHEAD (64x64 wave tiles only: the 128x64 ones have no registers for it -- 32 to 64 spills): strip
    // i + 1's pieces leave before strip i's stores (second register set).  Requested after them, their first use is a `vmcnt` that also covers the older stores: every strip drained the previous
    // strip's stores (a write round trip per strip) before it could add its residual.
    float4 cinv[2][NP][VEC / 4];
    int prows[2][NP];                                              // physical C rows of a strip's passes (scatter map), -1: dropped
    auto request = [&](const int i, const int slot) {
        if (g.c_map) {                                             // the map entries of all passes in flight together, then the pieces
#pragma unroll
            for (int pass = 0; pass < NP; ++pass) {
                const int row = m0 + wm + 32 * i + pass * RPP + lane / LPR;
                prows[slot][pass] = g.c_map[row < g.M ? row : 0];
            }
        }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int r = pass * RPP + lane / LPR, c = (lane % LPR) * VEC;
            const int row = m0 + wm + 32 * i + r, col = n0 + wn + c;
            if (!g.c_map) prows[slot][pass] = row;
            const bool ok = row < g.M && col + VEC <= g.N && prows[slot][pass] >= 0;
            const float* src = g.Cin + (long long)(ok ? prows[slot][pass] : 0) * g.ldcin + (ok ? col : 0);
#pragma unroll
            for (int x = 0; x < VEC / 4; ++x) cinv[slot][pass][x] = *(const float4*)(src + 4 * x);
        }
    };
    if (AHEAD && cin_pref) request(0, 0);
    // l2-norm epilogue: a lane's eight scale values are the same in every pass and strip (its column inside the head) -- loaded once.  Inside
    // the pass loop each load's wait also covered the stores of the pass before: one write round trip per pass.
    float4 es0 = make_float4(0.f, 0.f, 0.f, 0.f), es1 = es0;
    if constexpr (EPI == 1) {
        const int c = (lane % LPR) * VEC;
        es0 = *(const float4*)(g.epi_scale + c);
        es1 = *(const float4*)(g.epi_scale + c + 4);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (cin_pref) {
            if (!AHEAD) request(i, i & 1);
            else if (i + 1 < MI) request(i + 1, (i + 1) & 1);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stg[((e & 3) + 8 * (e >> 2) + 4 * hi) * SROW + 32 * j + (lane & 31)] = g.alpha * acc[i][j][e];
        if constexpr (EPI == 1) {
            // the scale values are consumed HERE, on every path, behind the first strip's LDS writes: one wait, before any store exists.
            // (Left to their first use inside the pass loop -- a conditional region -- hipcc repeats the wait in every pass, and there
            // it covers the stores of the pass before.)
            if (i == 0) asm volatile("" : "+v"(es0.x), "+v"(es0.y), "+v"(es0.z), "+v"(es0.w), "+v"(es1.x), "+v"(es1.y), "+v"(es1.z), "+v"(es1.w));
        }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int r = pass * RPP + lane / LPR, c = (lane % LPR) * VEC;
            const int row = m0 + wm + 32 * i + r, col = n0 + wn + c;
            float v[VEC];
#pragma unroll
            for (int x = 0; x < VEC; x += 4) {
                const float4 t = *(const float4*)(stg + r * SROW + c + x);
                v[x] = t.x; v[x + 1] = t.y; v[x + 2] = t.z; v[x + 3] = t.w;
            }
            if (row >= g.M || col >= g.N) continue;
            const long long prow = cin_pref ? (long long)prows[i & 1][pass] : (g.c_map ? (long long)g.c_map[row] : (long long)row);
            if (prow < 0) continue;
            if constexpr (EPI == 1) {
                // q / k of the attention (transformer.py:265-271): l2-normalise each 64-wide head and apply the learned per-dim scale
                // here, in fp32 on the accumulator values -- the wave's 64 columns ARE one head (WN_ == 64, host: N % 64 == 0), a row's
                // eight lanes hold it whole.  Rows >= M left above as whole 8-lane groups, so the shuffles below see complete rows.
                static_assert(WN_ == 64 && VEC == 8, "the l2-norm epilogue needs one head per wave row and 16-bit output");
                const int grp = (n0 + wn) >> 6;
                if (grp < g.epi_groups) {
                    float ss = 0.f;
#pragma unroll
                    for (int x = 0; x < VEC; ++x) ss += v[x] * v[x];
                    ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
                    const float nrm = fmaxf(sqrtf(ss), 1e-12f), inv = 1.0f / nrm;
                    const float4 s0 = es0, s1 = es1;
                    v[0] = v[0] * inv * s0.x; v[1] = v[1] * inv * s0.y; v[2] = v[2] * inv * s0.z; v[3] = v[3] * inv * s0.w;
                    v[4] = v[4] * inv * s1.x; v[5] = v[5] * inv * s1.y; v[6] = v[6] * inv * s1.z; v[7] = v[7] * inv * s1.w;
                    if ((lane % LPR) == 0) g.epi_norm[prow * g.epi_ldnorm + grp] = nrm;
                }
                TST* dst = (g.C2 && col >= g.c2_col0) ? (TST*)g.C2 + prow * g.ldc2 + (col - g.c2_col0) : C + prow * g.ldc + col;
                u32x4 o;
                o[0] = pack_h16_rne(v[0], v[1]); o[1] = pack_h16_rne(v[2], v[3]);
                o[2] = pack_h16_rne(v[4 % VEC], v[5 % VEC]); o[3] = pack_h16_rne(v[6 % VEC], v[7 % VEC]);
                *(u32x4*)dst = o;
                continue;
            }
            if (vec_ok && col + VEC <= g.N) {
                if (cin_pref) {
#pragma unroll
                    for (int x = 0; x < VEC; x += 4) {
                        const float4 t = cinv[i & 1][pass][x / 4];
                        v[x] += t.x; v[x + 1] += t.y; v[x + 2] += t.z; v[x + 3] += t.w;
                    }
                } else if (g.Cin) {
#pragma unroll
                    for (int x = 0; x < VEC; x += 4) {
                        const float4 t = *(const float4*)(g.Cin + prow * g.ldcin + col + x);
                        v[x] += t.x; v[x + 1] += t.y; v[x + 2] += t.z; v[x + 3] += t.w;
                    }
                }
                if (sizeof(TOUT) == 2) {
                    u32x4 o;
                    o[0] = pack_h16_rne(v[0], v[1]); o[1] = pack_h16_rne(v[2], v[3]);
                    o[2] = pack_h16_rne(v[4 % VEC], v[5 % VEC]); o[3] = pack_h16_rne(v[6 % VEC], v[7 % VEC]);
                    *(u32x4*)(C + prow * g.ldc + col) = o;
                    if constexpr (PLANES) {
                        if (g.c_lo8) {
                            u32x2 l8;
                            l8[0] = pack4_bf8(v[0] - h16_lo_to_f(o[0]), v[1] - h16_hi_to_f(o[0]), v[2] - h16_lo_to_f(o[1]), v[3] - h16_hi_to_f(o[1]));
                            l8[1] = pack4_bf8(v[4 % VEC] - h16_lo_to_f(o[2]), v[5 % VEC] - h16_hi_to_f(o[2]), v[6 % VEC] - h16_lo_to_f(o[3]), v[7 % VEC] - h16_hi_to_f(o[3]));
                            *(u32x2*)((unsigned char*)g.C_lo + prow * g.ldc + col) = l8;
                        } else {
                            u32x4 l;
#pragma unroll
                            for (int x = 0; x < 4; ++x)
                                l[x] = pack_h16_rne(v[(2 * x) % VEC] - h16_lo_to_f(o[x]), v[(2 * x + 1) % VEC] - h16_hi_to_f(o[x]));
                            *(u32x4*)((h16_t*)g.C_lo + prow * g.ldc + col) = l;
                        }
                    }
                } else {
                    *(float4*)((float*)g.C + prow * g.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int x = 0; x < VEC; ++x)
                    if (col + x < g.N) {
                        float o = v[x];
                        if (g.Cin) o += g.Cin[prow * g.ldcin + col + x];
                        store_from_float(C + prow * g.ldc + col + x, o);
                        if constexpr (PLANES) {
                            const h16_t hi = (h16_t)o;
                            if (g.c_lo8) ((unsigned char*)g.C_lo)[prow * g.ldc + col + x] = (unsigned char)(pack4_bf8(o - (float)hi, 0.f, 0.f, 0.f) & 0xFFu);
                            else store_from_float((h16_t*)g.C_lo + prow * g.ldc + col + x, o - (float)hi);
                        }
                    }
            }
        }
    }
}

// XCD-aware order over a WHOLE grid of `total` workgroups: they are dealt round-robin to the 8 XCDs in linear dispatch order, so
// XCD c is given one contiguous chunk of the logical sequence.
__device__ __forceinline__ int xcd_logical_id(int lin, int total) {
    const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Sum of the S fp32 slices a slice-storing split-K left in a workspace (GemmArgs::c_split_stride), in a fixed order, plus the residual.
// MODE 0: fp32 out; 1: 16-bit out; 2: 16-bit hi/lo planes out; 3: 16-bit hi plane + bf8 lo plane (bytes).  One thread = 4 consecutive columns of a row.
template <int MODE>
__global__ __launch_bounds__(256) void gemm_tail_reduce_kernel(const float* __restrict__ ws, int S, long long stride, int M, int N, int ldw,
                                                               void* __restrict__ C, void* __restrict__ C_lo, int ldc,
                                                               const float* __restrict__ Cin, int ldcin) {
    const int nq = (N + 3) >> 2;
    const long long total = (long long)M * nq;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / nq), c = (int)(i - (long long)r * nq) * 4;
        float4 v = *(const float4*)(ws + (size_t)r * ldw + c);                  // ldw % 4 == 0: whole pieces (columns >= N hold what the tile kernel left: never stored)
        for (int s = 1; s < S; ++s) {
            const float4 t = *(const float4*)(ws + (size_t)s * stride + (size_t)r * ldw + c);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (c + x >= N) break;
            float u = o[x];
            if (Cin) u += Cin[(size_t)r * ldcin + c + x];
            if constexpr (MODE == 0) ((float*)C)[(size_t)r * ldc + c + x] = u;
            else {
                const h16_t hi = (h16_t)u;
                ((h16_t*)C)[(size_t)r * ldc + c + x] = hi;
                if constexpr (MODE == 2) ((h16_t*)C_lo)[(size_t)r * ldc + c + x] = (h16_t)(u - (float)hi);
                if constexpr (MODE == 3) ((unsigned char*)C_lo)[(size_t)r * ldc + c + x] = (unsigned char)(pack4_bf8(u - (float)hi, 0.f, 0.f, 0.f) & 0xFFu);
            }
        }
    }
}

}   // namespace OMLM_NS
