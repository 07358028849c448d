// Causal multi-query attention, second-generation kernels for bf16 operands (transformer.py:254-331 of the reference).
//
// What changed against attention.hip (whose kernels stay for fp32 / "bf16x3" operands), and why:
//   * one workgroup = 8 waves = 8 HEADS of the same 32*QB queries: the single shared K/V head (MQA) is staged once per tile
//     for all of them (the old geometry staged a tile for 4 heads through registers + ds_write behind two barriers per tile,
//     and paid a global round trip for the key mask inside the loop: ~3.9 k cycles per 32x32 block for 8 MFMAs);
//   * K / V tiles and the rel-pos bias window of the tile go HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds) into a 3-stage
//     ring: no VGPR staging, no ds_write, ONE barrier per 64-key tile, two tiles in flight behind counted vmcnt waits;
//   * the key mask costs nothing per score: masked keys get ZERO V rows (the DMA reads them through an out-of-bounds
//     offset) and the softmax denominator is produced by the matrix cores from a 1/0 "live" vector (2 extra MFMAs per block),
//     so masked keys contribute to neither numerator nor denominator.  The running max may include masked keys' scores
//     (they are bounded like the others: l2-normalised q, k); that only moves the reference point of the exponentials;
//   * the causal compare exists only in the blocks that touch the diagonal;
//   * the rel-pos bias is read from a per-tile window (LDS, 128 floats per head) with compile-time offsets from one lane base.
// The arithmetic per score is: fma (scale * log2 e, bias), max, subtract, exp2, pack -- everything else is on the matrix cores.
#include "common.h"

namespace OMLM_NS {

#define A2_THREADS 512
#define A2_TKV 64
#define A2_PAD 64            /* zero entries in front of each row of the transposed bias table (rel >= -64) */
#define A2_BWIN 128          /* floats per head in a tile's bias window */
#define A2_NEG (-1.0e30f)
#define A2_LOG2E 1.4426950408889634f
#define A2_STAGE (8192 + 8192 + 8 * A2_BWIN * 4)     /* K rows | V blocked | bias window = 20 KiB */
#define A2_NST 3
#ifndef A2_DQ_PK
#define A2_DQ_PK 1           /* backward dQ kernel: element arithmetic on register pairs (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32), -lse and -delta carried
                                in by the accumulators' initial values, the softmax scale applied once to dQ instead of to every dS */
#endif
#ifndef A2_DQ_BATCH
#define A2_DQ_BATCH 1        /* backward dQ kernel: fragment / bias reads issued in batches (scheduling only, same arithmetic) */
#endif
#ifndef A2_SUBFENCE
#define A2_SUBFENCE 0
#endif
#ifndef A2_ABLATE
#define A2_ABLATE 0          /* profiling builds only: 1 = skip the tile arithmetic, 2 = skip the steady-state DMA, 4 = no exp2; dQ kernel: 8 = no diagonal sums, 16 = no per-block table update, 32 = no global d(bias) flush */
#endif

#define MFMA16(a, b, c) OMLM_MFMA_32x32x16(a, b, c)

// [rows][64 dims] bf16 tile, 128 B per row; 16-B chunk index XOR ((row >> 1) & 7)   (same image as attention.hip)
__device__ __forceinline__ int a2_tile_off(int row, int colbyte) {
    return row * 128 + ((((colbyte >> 4) ^ ((row >> 1) & 7)) << 4) | (colbyte & 15));
}
__device__ __forceinline__ int a2_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ h16x8 a2_frag_rows(const char* lds, int row0, int s, int lane) {
    return *(const h16x8*)(lds + a2_tile_off(row0 + (lane & 31), (2 * s + (lane >> 5)) * 16));
}
// transposed operand from the BLOCKED image (see attention.hip tile_off_blk): both reads are linear in the lane id
__device__ __forceinline__ h16x8 a2_frag_cols_tr(const char* lds, int row0, int s, int col0, int lane) {
    const char* base = lds + ((((row0 >> 4) + s) << 1) + (col0 >> 5)) * 1024 + lane * 8;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ h16x8 a2_pack(const f32x16& p, int s) {
    u32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = pack_h16_rne(p[8 * s + 2 * e], p[8 * s + 2 * e + 1]);
    return __builtin_bit_cast(h16x8, h);
}

// ---- bias table [N, ld] (row = i - j, column = head) -> transposed, padded, pre-multiplied by log2 e: [H8][ldT] ----------
// One workgroup per head.  With a bound on |q.k| (qk_max, from the learned scales or given) the fixed reference point
// m_h = c qk_max + max_r table_h[r] - p_max_log2 is subtracted from every entry (pads included) and written to the row's tail
// [ldT - 2] = 1.0 (flag), [ldT - 1] = m_h; without a bound, or if the exponent range could leave the operand type's normal range, the
// flag is 0.  The decision is the SAME for every head (it looks at the widest head's range): the forward picks its kernel by it.
//   p_max_log2: the largest probability numerator is 2^p_max_log2.  0 for bf16 / fp32 operands (numerators <= 1, fp32's exponent range
//   below); 15 for IEEE half, whose normal range is 2^-14 .. 2^15.99: numerators in (2^-13, 2^15] while 2 c qk_max + range < 28.
// (round 6) grid.y = layers: every layer's table in ONE launch -- the tables of a forward differ only through their layer's learned scales, and
// six launches of 8 workgroups were 6 x 14 us of latency per step (omlm_attn_bias_prepare_group; the single-layer entry passes one layer).
#define A2_PREP_MAX 32
struct A2PrepGroup { float* out[A2_PREP_MAX]; const float* qs[A2_PREP_MAX]; const float* ks[A2_PREP_MAX]; };
__global__ void attn2_bias_prep_kernel(const float* __restrict__ bias, A2PrepGroup grp, int N, int H, int ld, int ldT,
                                       float qk_bound, float c, int p_max_log2) {
    __shared__ float red[4][2][8];
    __shared__ float redq[4];
    const int h = blockIdx.x, t = threadIdx.x;
    float* biasT = grp.out[blockIdx.y];
    const float* q_scale = grp.qs[blockIdx.y];
    const float* k_scale = grp.ks[blockIdx.y];
    float* row = biasT + (size_t)h * ldT;
    const bool has = bias && h < H;
    // every head's extremes (8 heads per pass): the widest range decides for all, this head's maximum sets its reference point
    float wide = 0.f, own_max = 0.f;
    if (bias) {
        for (int hb = 0; hb < H; hb += 8) {
            float mx[8], mn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { mx[j] = -3.0e38f; mn[j] = 3.0e38f; }
            for (int r = t; r < N; r += 256) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = hb + j < H ? bias[(size_t)r * ld + hb + j] * A2_LOG2E : 0.f;
                    mx[j] = fmaxf(mx[j], x); mn[j] = fminf(mn[j], x);
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = wave_max(mx[j]), b = -wave_max(-mn[j]);
                if ((t & 63) == 0) { red[t >> 6][0][j] = a; red[t >> 6][1][j] = b; }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (hb + j < H) {
                    const float a = fmaxf(fmaxf(red[0][0][j], red[1][0][j]), fmaxf(red[2][0][j], red[3][0][j]));
                    const float b = fminf(fminf(red[0][1][j], red[1][1][j]), fminf(red[2][1][j], red[3][1][j]));
                    wide = fmaxf(wide, a - b);
                    if (hb + j == h) own_max = a;
                }
            }
        }
    }
    float qk = qk_bound;
    if (q_scale && k_scale) {
        qk = t < 64 ? fabsf(q_scale[t] * k_scale[t]) : 0.f;
        qk = wave_max(qk);
        if ((t & 63) == 0) redq[t >> 6] = qk;
        __syncthreads();
        qk = redq[0];
    }
    const float B = c * qk;
    // all exponents lie in [p_max_log2 - (2 B + range), p_max_log2]
    const bool fixed = qk > 0.f && (2.f * B + wide) < (p_max_log2 > 0 ? 13.f + (float)p_max_log2 : 80.f);
    const float m = fixed ? B + own_max - (float)p_max_log2 : 0.f;
    for (int x = t; x < ldT - 2; x += 256) {
        const int r = x - A2_PAD;
        const float v = (has && r >= 0 && r < N) ? bias[(size_t)r * ld + h] * A2_LOG2E : 0.f;
        row[x] = v - m;
    }
    if (t == 0) { row[ldT - 2] = fixed ? 1.0f : 0.f; row[ldT - 1] = m; }
}

// One LDS-DMA wave-instruction = 1 KiB (64 lanes x 16 B), destination lane-linear (M0 = wave-uniform LDS byte address).
// Issued as inline asm on purpose: with the builtin, hipcc tracks "a pending LDS write" and protects LDS reads it cannot
// prove disjoint with s_waitcnt vmcnt(N) -- one build of this kernel waited for the K tile it had JUST requested in front
// of every first MFMA of a tile (seen in the ISA), i.e. the 3-stage ring ran with zero tiles in flight.  The asm form is
// invisible to that bookkeeping; ordering is by the counted s_waitcnt vmcnt + s_barrier at the top of the tile loop.
typedef u32x4 a2_rsrc;
__device__ __forceinline__ a2_rsrc a2_make_rsrc(const void* p, unsigned bytes) {     // wave-uniform inputs only
    const unsigned long long a = (unsigned long long)p;
    a2_rsrc r;
    r[0] = (unsigned)a; r[1] = (unsigned)(a >> 32) & 0xFFFFu; r[2] = bytes; r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void a2_dma(a2_rsrc rs, unsigned lds_dst, unsigned off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(lds_dst), "s"(rs) : "memory");
}

struct A2Stager {
    unsigned koff, voff, boff;       // per-lane source byte offsets relative to the tile (key j0) / window start
    int vrow;                        // this lane's key row (0..63) in the V unit it issues
    bool bias_wave;
    __device__ __forceinline__ void init(int wave, int lane, int ldT) {
        {   // K unit `wave`: rows 8 wave + (lane >> 3), slot lane & 7 holds chunk slot ^ ((row >> 1) & 7)
            const int row = 8 * wave + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            koff = (unsigned)(row * 128 + ch * 16);
        }
        {   // V unit `wave` of the blocked image: unit = (row >> 4) * 2 + (col >> 5); lane = p * 8 + (row & 3) * 2 + ((col >> 3) & 1)
            const int p = lane >> 3, rq = ((p >> 2) << 1) | ((p >> 1) & 1);
            vrow = (wave >> 1) * 16 + rq * 4 + ((lane >> 1) & 3);
            const int col = (wave & 1) * 32 + (p & 1) * 16 + (lane & 1) * 8;
            voff = (unsigned)(vrow * 128 + col * 2);
        }
        {   // bias unit (wave & 3): heads 2 u, 2 u + 1; lane = (head & 1) * 32 + float4 index
            const int u = wave & 3, hh = 2 * u + (lane >> 5);
            boff = (unsigned)(((size_t)hh * ldT + 4 * (lane & 31)) * 4);
            bias_wave = wave < 4;
        }
    }
};

// Work item of workgroup `lin`: (sample b, query tile qt, head group hy).  XCD-aware: the workgroups of one sample (they share its K / V
// through the XCD's L2) are dealt to one XCD (workgroup lin runs on XCD lin % 8).  When an XCD's share is whole samples (B a multiple of
// 8), its items run heaviest (latest) query tile first ACROSS its samples -- dealt sample by sample, the last sample's longest items
// started two thirds into the launch and the SIMDs averaged 1.3 of 2 resident waves (forward 94 -> 86 us per layer at B = 32, N = 1116).
__device__ __forceinline__ void a2_item_order(int lin, int nqt, int ny, int B, int& b, int& qt, int& hy) {
    const int per = nqt * ny, total = per * B;
    const int qq = total >> 3, rr = total & 7, xcd = lin & 7, idx = lin >> 3;
    const int start = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq, cnt = qq + (xcd < rr ? 1 : 0);
    if (start % per == 0 && cnt % per == 0) {
        const int ns = cnt / per, w = idx % (ns * ny);
        qt = nqt - 1 - idx / (ns * ny);
        b = start / per + w / ny;
        hy = w % ny;
    } else {                                                  // sample-major, heavy tiles first inside a sample
        const int lg = start + idx;
        b = lg / per;
        const int rem = lg - b * per;
        qt = nqt - 1 - rem / ny;
        hy = rem % ny;
    }
}

struct A4Acc {
    f32x16 acc[2][2];       // O^T: [head][d tile]
    f32x16 accl;            // denominators; row parity (i & 1) == hb holds those of head hb
    float m[2];             // running max (online form only)
};

// One 64-key tile against the wave's two heads x 32 queries.
//   FIXED: the exponentials are taken against a FIXED reference point instead of a running maximum.  q and k are l2-normalised
//          vectors times learned scales (transformer.py:269-271), so |q.k| <= max_d |q_scale_d k_scale_d| =: qk_max and every
//          score is <= m_h = scale log2(e) qk_max + max_rel bias_h.  omlm_attn_bias_prepare subtracts m_h from the table, so the
//          work per score is ONE fma (score * c + table) and ONE exp2: no row maximum, no cross-lane step, no rescaling of
//          the accumulators, no dependency between blocks other than the MFMA accumulation itself.  The result is the same
//          softmax (numerator and denominator share the factor 2^-m_h); it is only selected while 2 c qk_max + the table's
//          range stays inside the operand type's exponent range (flag in the table's tail), else the online form runs.
//   FULL:  every block of the tile lies strictly below the diagonal: straight-line code, no compares.
//   QSEL >= 0: only head QSEL is processed (the online form walks the heads one at a time: fewer live registers).
//   Addressing: every LDS read of the tile is (one lane-dependent base register) + (compile-time offset) -- the bias window from ONE
//   base below its lowest entry (the 64 window reads of a tile had an address add each), the live vectors from a per-head base that
//   points at the sample's liveness array for the lanes whose accumulator rows belong to that head and at a block of zeros for the
//   others (instead of 32 per-value selects): ~70 of ~260 VALU instructions per tile less in a loop that is VALU-issue bound.
template <bool FIXED, bool FULL, int QSEL = -1>
__device__ __forceinline__ void a4_tile(A4Acc& A, const h16x8 (&qf)[2][4], const char* Ks, const h16_t* live0, const h16_t* live1,
                                        float c, int i0, int j0, int wave, int lane) {
    const char* Vs = Ks + 8192;
    const int hi = lane >> 5, ql = lane & 31;
    // window index of (head hb, query ql, key 32 sub + 4 hi + cr): hb 128 + 64 + ql - 32 sub - 4 hi - cr, cr = crow(r, 0) <= 27
    // (the base as ONE opaque LDS address: hipcc otherwise folds the stage's 16 KB offset into each read's constant, past the offset field)
    unsigned bbo = (unsigned)(size_t)LDS_PTR(const float, (const float*)(Ks + 16384) + 2 * wave * A2_BWIN + ql - 4 * hi);
    asm volatile("" : "+v"(bbo));
    const __attribute__((address_space(3))) float* bb = (const __attribute__((address_space(3))) float*)(size_t)bbo;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int jb = j0 + 32 * sub;
        if (!FULL && jb > i0 + 31) break;                       // above the diagonal for every query of the workgroup
        h16x8 kf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = a2_frag_rows(Ks, 32 * sub, s, lane);
        h16x8 pb[2][2];
        const bool diag = !FULL && !(jb + 31 <= i0);
        const int d0 = (i0 + ql) - (jb + 4 * hi);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            if (QSEL >= 0 && hb != QSEL) continue;
            f32x16 st;
#pragma unroll
            for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st = MFMA16(kf[s], qf[hb][s], st);
            const __attribute__((address_space(3))) float* bp = bb + (hb * A2_BWIN + 64 - 32 * sub - 27);       // entries [27 - cr]: offsets >= 0
            if (FIXED) {
                if (!diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = (A2_ABLATE & 4) ? st[r] * c + bp[27 - ((r & 3) + 8 * (r >> 2))] : __builtin_amdgcn_exp2f(st[r] * c + bp[27 - ((r & 3) + 8 * (r >> 2))]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = (r & 3) + 8 * (r >> 2);
                        const float e2 = __builtin_amdgcn_exp2f(st[r] * c + bp[27 - cr]);
                        st[r] = (d0 - cr >= 0) ? e2 : 0.f;
                    }
                }
            } else {
                float mloc = A2_NEG;
                if (!diag) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        st[r] = st[r] * c + bp[27 - ((r & 3) + 8 * (r >> 2))];
                        mloc = fmaxf(mloc, st[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cr = (r & 3) + 8 * (r >> 2);
                        const float val = st[r] * c + bp[27 - cr];
                        st[r] = (d0 - cr >= 0) ? val : A2_NEG;
                        mloc = fmaxf(mloc, st[r]);
                    }
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mnew = fmaxf(A.m[hb], mloc);
                const float alpha = __builtin_amdgcn_exp2f(A.m[hb] - mnew);
                A.m[hb] = mnew;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = __builtin_amdgcn_exp2f(st[r] - mnew);
                if (!__all(alpha == 1.0f)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { A.acc[hb][0][e] *= alpha; A.acc[hb][1][e] *= alpha; }
                    A.accl[hb] *= alpha;                        // rows 0/4 (head 0) and 1/5 (head 1) are the ones read
                }
            }
            pb[hb][0] = a2_pack(st, 0);
            pb[hb][1] = a2_pack(st, 1);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h16x8 va0 = a2_frag_cols_tr(Vs, 32 * sub, s, 0, lane);
            const h16x8 va1 = a2_frag_cols_tr(Vs, 32 * sub, s, 32, lane);
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                if (QSEL >= 0 && hb != QSEL) continue;
                // live vector of the 16 keys of this k-step in MFMA k order: keys key0 + 4 hi + {0..3}, key0 + 8 + 4 hi + {0..3}; the
                // A rows of parity hb carry it for head hb, the others zeros: ONE denominator accumulator for both heads
                const h16_t* lp = (hb ? live1 : live0) + 32 * sub + 16 * s;
                const u32x2 l0 = *(const u32x2*)lp, l1 = *(const u32x2*)(lp + 8);
                u32x4 lv4;
                lv4[0] = l0[0]; lv4[1] = l0[1]; lv4[2] = l1[0]; lv4[3] = l1[1];
                A.acc[hb][0] = MFMA16(va0, pb[hb][s], A.acc[hb][0]);
                A.acc[hb][1] = MFMA16(va1, pb[hb][s], A.acc[hb][1]);
                A.accl = MFMA16(__builtin_bit_cast(h16x8, lv4), pb[hb][s], A.accl);      // sum over live keys of P
            }
        }
    }
}

// Forward: one workgroup = 4 waves = 8 heads x 32 queries of one sample, each wave TWO heads; two workgroups per CU (63 KB of LDS,
// <= 256 registers).  Against the 8-wave form it replaces (one head per wave, one workgroup per CU: 122 us -> 95 us per layer at B = 32,
// N = 1116 with half operands):
//   * the K fragments and the V (transposed) fragments of a 32-key block are read from LDS once for two heads, and a wave has four
//     independent (head, key block) chains for the scheduler to interleave exponentials with MFMAs;
//   * the two waves of a SIMD belong to different workgroups, so they do not meet at the same barrier -- with 8 waves in one workgroup
//     both waves of a SIMD ran the same phase of the same tile at the same time (both want the VALU, then both want the matrix pipe), and
//     a workgroup's prologue / epilogue (~3 dependent round trips) had nothing to hide behind.
#define A4_THREADS 256
struct A4Stager {
    unsigned koff[2], voff[2], boff;     // per-lane source byte offsets of the wave's two K units, two V units, one bias unit
    int vrow[2];
    // Recomputed from the lane id at every use (a dozen integer ops per tile): kept across the tile loop these seven registers were
    // spilled (the tile body wants all 256), and hipcc puts s_waitcnt vmcnt(0) behind a scratch reload -- which drains the DMA ring.
    __device__ __forceinline__ void init(int wave, int lane, int ldT) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int unit = 2 * wave + u;
            {   // K unit: rows 8 unit + (lane >> 3), slot lane & 7 holds chunk slot ^ ((row >> 1) & 7)
                const int row = 8 * unit + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
                koff[u] = (unsigned)(row * 128 + ch * 16);
            }
            {   // V unit of the blocked image (see A2Stager)
                const int p = lane >> 3, rq = ((p >> 2) << 1) | ((p >> 1) & 1);
                vrow[u] = (unit >> 1) * 16 + rq * 4 + ((lane >> 1) & 3);
                const int col = (unit & 1) * 32 + (p & 1) * 16 + (lane & 1) * 8;
                voff[u] = (unsigned)(vrow[u] * 128 + col * 2);
            }
        }
        const int hh = 2 * wave + (lane >> 5);                   // bias unit `wave`: heads 2 wave, 2 wave + 1
        boff = (unsigned)(((size_t)hh * ldT + 4 * (lane & 31)) * 4);
    }
};

// FIXED is a template parameter of the KERNEL, and both instances are launched: which softmax form applies is known on the device only
// (the flag in the table's tail, omlm_attn_bias_prepare), and the instance the flag does not name returns at its first instruction
// (~3 us per layer).  One kernel holding both forms was tried: at 256 registers it either spilled ~20 loop-invariant registers -- hipcc
// waits vmcnt(0) behind every scratch reload, which drains the DMA ring -- or, with the online form walking its heads one at a time, kept
// the fixed form 10 % slower than this split (103 vs 92 us).
//   FIXED: both heads in one straight line.  Online (more live state: running maxima, rescale factors): the two heads one after the
//   other (QSEL), re-reading the K / V fragments per head.
template <bool FIXED>
__global__ __launch_bounds__(A4_THREADS, 2) void attn4_fwd_kernel(const h16_t* __restrict__ q, const h16_t* __restrict__ k,
                                                                  const h16_t* __restrict__ v, const float* __restrict__ biasT, int ldT,
                                                                  const unsigned char* __restrict__ keymask, h16_t* __restrict__ out,
                                                                  float* __restrict__ lse, int B, int N, int H, float scale) {
    // the flag of head 0: omlm_attn_bias_prepare decides once for all heads
    if ((biasT && __builtin_amdgcn_readfirstlane(__float_as_int(biasT[ldT - 2])) != 0) != FIXED) return;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                        // A2_NST stages
    h16_t* livef = (h16_t*)(smem + A2_NST * A2_STAGE);      // [nkt_all * 64] 1.0 / 0.0 per key of this sample
    const int nqt = (N + 31) / 32, ny = (H + 7) / 8;
    int b, qt, hy;
    a2_item_order(blockIdx.x, nqt, ny, B, b, qt, hy);
    const int lane = threadIdx.x & 63, hi = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h0 = hy * 8 + 2 * wave;                        // heads h0, h0 + 1
    const int i0 = qt * 32;
    const size_t rowbase = (size_t)b * N;
    const int nkt = min((i0 + 32 + A2_TKV - 1) / A2_TKV, (N + A2_TKV - 1) / A2_TKV);   // key tiles this query tile needs

    // ---- prologue: liveness of this sample's keys, as a 1/0 array of the operand type (denominator operand) and one ballot word per key
    // tile (V rows of masked keys are DMA'd as zeros).  All byte loads are issued before the first wait.
    unsigned long long* livebits = (unsigned long long*)(livef + (size_t)((N + 63) / 64) * 64);     // [64 tiles]
    unsigned* zeros = (unsigned*)(livebits + 64);                                                 // 128 B of zeros (see a4_tile)
    {
        unsigned char mk[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) mk[it] = 1;
        if (keymask) {
#pragma unroll
            for (int it = 0; it < 16; ++it)
                if (it * A4_THREADS < nkt * A2_TKV) mk[it] = keymask[rowbase + min(it * A4_THREADS + (int)threadIdx.x, N - 1)];   // uniform condition, clamped index
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int j = it * A4_THREADS + threadIdx.x;
            if (it * A4_THREADS < nkt * A2_TKV) {               // uniform
                const bool lv = j < N && mk[it] != 0;
                const unsigned long long w = __ballot(lv);
                if (j < nkt * A2_TKV) livef[j] = lv ? (h16_t)1.0f : (h16_t)0.0f;
                if (lane == 0 && it * 4 + wave < nkt) livebits[it * 4 + wave] = w;
            }
        }
        if (threadIdx.x < 32) zeros[threadIdx.x] = 0u;
    }
    __syncthreads();                                          // livef / livebits visible; no LDS-DMA in flight yet
    const a2_rsrc rsK = a2_make_rsrc(k + rowbase * 64, (unsigned)N * 128u);
    const a2_rsrc rsV = a2_make_rsrc(v + rowbase * 64, (unsigned)N * 128u);
    const a2_rsrc rsB = a2_make_rsrc(biasT ? (const void*)(biasT + (size_t)hy * 8 * ldT) : (const void*)k, biasT ? (unsigned)(8 * ldT * 4) : 0u);
    const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(char, ring);

    auto issue = [&](int t, int lane_) {                       // 5 DMA wave-instructions per wave per tile
        A4Stager stg;
        stg.init(wave, lane_, ldT);
        const unsigned st = ring_lds + (unsigned)((t % A2_NST) * A2_STAGE);
        const int j0 = t * A2_TKV;
        const unsigned long long lb = livebits[t];             // one broadcast LDS read per tile
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            a2_dma(rsK, st + (2 * wave + u) * 1024, (unsigned)(j0 * 128) + stg.koff[u]);           // rows >= N: beyond the descriptor -> zeros
            a2_dma(rsV, st + 8192 + (2 * wave + u) * 1024, ((lb >> stg.vrow[u]) & 1ull) ? (unsigned)(j0 * 128) + stg.voff[u] : OOB_OFF);
        }
        // bias window of this tile: table index PAD + rel, rel from i0 - j0 - 64 (no table: empty descriptor -> zeros)
        a2_dma(rsB, st + 16384 + wave * 1024, (unsigned)((A2_PAD + i0 - j0 - 64) * 4) + stg.boff);
    };

    issue(0, lane);
    if (nkt > 1) issue(1, lane);
    // Q fragments (B operand of S^T = K Q^T): query i0 + ql, dims 16 s + 8 hi .. +7, of the wave's two heads
    h16x8 qf[2][4];
    const int qi = i0 + ql;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const bool act = h0 + hb < H;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 z = {0u, 0u, 0u, 0u};
            const h16_t* p = q + (rowbase + min(qi, N - 1)) * (size_t)(H * 64) + (size_t)(act ? h0 + hb : 0) * 64 + 16 * s + 8 * hi;
            u32x4 val = (act && qi < N) ? *(const u32x4*)p : z;
            qf[hb][s] = __builtin_bit_cast(h16x8, val);
        }
    }
    // Consume the Q loads HERE: hipcc then waits for them before the loop.  Left to their first use inside the loop, its
    // s_waitcnt vmcnt(0) sat in front of the first MFMA of every tile and drained the DMA ring each iteration (seen in the ISA).
#pragma unroll
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(qf[hb][s]));

    A4Acc A;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        A.m[hb] = A2_NEG;
#pragma unroll
        for (int e = 0; e < 16; ++e) { A.acc[hb][0][e] = 0.f; A.acc[hb][1][e] = 0.f; }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) A.accl[e] = 0.f;
    const float c = scale * A2_LOG2E;
    float mfix[2] = {0.f, 0.f};                              // fixed reference points of the two heads (table tails)
    if (FIXED) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
            mfix[hb] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(biasT[(size_t)(h0 + hb) * ldT + (ldT - 1)])));
    }
    asm volatile("" : "+s"(mfix[0]), "+s"(mfix[1]));          // loaded (and waited for) before the tile loop

    for (int t = 0; t < nkt; ++t) {
        // own DMA of tile t retired (tile t+1's five may stay in flight), then everybody's; the barrier also says that all
        // waves are done with tile t-1, whose stage tile t+2 is about to overwrite
        if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int lane_ = lane;                                      // opaque per tile: every lane-derived address is rebuilt, none carried (see A4Stager)
        asm volatile("" : "+v"(lane_));
        if (t + 2 < nkt && !(A2_ABLATE & 2)) issue(t + 2, lane_);
        if (A2_ABLATE & 1) continue;
        const char* Ks = ring + (t % A2_NST) * A2_STAGE;
        const int j0 = t * A2_TKV;
        const bool full = j0 + A2_TKV - 1 <= i0;               // every block of the tile lies below the diagonal
        const h16_t* lv = livef + j0 + 4 * (lane_ >> 5);
        const h16_t* live0 = (lane_ & 1) == 0 ? lv : (const h16_t*)zeros;
        const h16_t* live1 = (lane_ & 1) == 1 ? lv : (const h16_t*)zeros;
        if (FIXED) {
            if (full) a4_tile<true, true>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
            else      a4_tile<true, false>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
        } else if (full) {
            a4_tile<false, true, 0>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
            __builtin_amdgcn_sched_barrier(0);
            a4_tile<false, true, 1>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
        } else {
            a4_tile<false, false, 0>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
            __builtin_amdgcn_sched_barrier(0);
            a4_tile<false, false, 1>(A, qf, Ks, live0, live1, c, i0, j0, wave, lane_);
        }
    }
    if (qi >= N) return;
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        const int h = h0 + hb;
        if (h >= H) continue;
        const float lsum = A.accl[hb];                  // element e = hb: row crow(hb, hi) has parity hb
        const float mref = FIXED ? mfix[hb] : A.m[hb];
        // a query without any live causal key has no defined softmax: emit zeros and an lse that zeroes its backward
        const float inv = lsum > 0.f ? 1.0f / lsum : 0.f;
        h16_t* orow = out + (rowbase + qi) * (size_t)(H * 64) + (size_t)h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hi;
                store4_from_float(orow + d, A.acc[hb][dt][4 * g4] * inv, A.acc[hb][dt][4 * g4 + 1] * inv,
                                  A.acc[hb][dt][4 * g4 + 2] * inv, A.acc[hb][dt][4 * g4 + 3] * inv);
            }
        if (hi == 0 && lse) lse[((size_t)b * H + h) * N + qi] = lsum > 0.f ? mref + log2f(lsum) : 1.0e30f;   // log2 domain
    }
}

// =========================================================================================================================
// backward, dQ / d(bias) / delta kernel on the same skeleton: 8 heads x 32 queries per workgroup, 64-key tiles through the
// 3-stage LDS-DMA ring -- per stage K as rows (S^T = K Q^T), K blocked (dQ^T += K^T dS^T), V as rows (dP^T = V dO^T) and the
// bias window.  The key mask is an additive 0 / -1e30 vector in LDS (one aligned 16-byte read per 4 scores); the probabilities
// come straight from the stored log-sum-exp (no maximum to track), so the blocks of a tile are independent.
#define A2B_STAGE (3 * 8192 + 8 * A2_BWIN * 4)       /* 28 KiB */
__global__ __launch_bounds__(A2_THREADS) void attn2_bwd_dq_kernel(const h16_t* __restrict__ q, const h16_t* __restrict__ k,
                                                                  const h16_t* __restrict__ v, const float* __restrict__ biasT, int ldT,
                                                                  const unsigned char* __restrict__ keymask, const h16_t* __restrict__ out,
                                                                  const h16_t* __restrict__ dout, const float* __restrict__ lse,
                                                                  float* __restrict__ delta, float* __restrict__ dq, float* __restrict__ dbias,
                                                                  int bias_ld, float* __restrict__ dpart, int B, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    char* scratch = smem + A2_NST * A2B_STAGE;
    const int npad = (N + 63) / 64 * 64;
    float* mb = (float*)(scratch + 4096);                     // [npad] 0 / -1e30 per key of this sample
    float* dbl = mb + npad;                                   // [8 waves][nbp]
    const int nqt = (N + 31) / 32, ny = (H + 7) / 8;
    int b, qt, hy;
    a2_item_order(blockIdx.x, nqt, ny, B, b, qt, hy);
    const int lane = threadIdx.x & 63, hi = lane >> 5, ql = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = hy * 8 + wave;
    const bool active = h < H;
    const int i0 = qt * 32, qi = i0 + ql;
    const int nb = i0 + 32;                                   // rel in [0, i0 + 31]
    const size_t rowbase = (size_t)b * N;
    const int nkt = min((i0 + 32 + A2_TKV - 1) / A2_TKV, (N + A2_TKV - 1) / A2_TKV);
    float* dbw = dbl + (size_t)wave * nb;

    {   // additive key mask, all byte loads in flight at once; this wave's d(bias) bins zeroed
        unsigned char mk[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) mk[it] = 1;
        if (keymask) {
#pragma unroll
            for (int it = 0; it < 8; ++it) mk[it] = keymask[rowbase + min(it * A2_THREADS + (int)threadIdx.x, N - 1)];
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int j = it * A2_THREADS + threadIdx.x;
            if (j < nkt * A2_TKV) mb[j] = (j < N && mk[it] != 0) ? 0.f : A2_NEG;
        }
        if (dbias) for (int r = lane; r < nb; r += 64) dbw[r] = 0.f;
    }
    __syncthreads();
    // per-lane DMA source offsets: K rows and V rows use the row image, K blocked the blocked image (see A2Stager)
    A2Stager stg;
    stg.init(wave, lane, ldT);
    const a2_rsrc rsK = a2_make_rsrc(k + rowbase * 64, (unsigned)N * 128u);
    const a2_rsrc rsV = a2_make_rsrc(v + rowbase * 64, (unsigned)N * 128u);
    const a2_rsrc rsB = a2_make_rsrc(biasT ? (const void*)(biasT + (size_t)hy * 8 * ldT) : (const void*)k, biasT ? (unsigned)(8 * ldT * 4) : 0u);
    const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(char, ring), scratch_lds = (unsigned)(size_t)LDS_PTR(char, scratch);
    auto issue = [&](int t) {                                  // 4 DMA wave-instructions per wave per tile
        const unsigned st = ring_lds + (unsigned)((t % A2_NST) * A2B_STAGE);
        const int j0 = t * A2_TKV;
        a2_dma(rsK, st + wave * 1024, (unsigned)(j0 * 128) + stg.koff);
        a2_dma(rsK, st + 8192 + wave * 1024, (unsigned)(j0 * 128) + stg.voff);
        a2_dma(rsV, st + 16384 + wave * 1024, (unsigned)(j0 * 128) + stg.koff);
        const unsigned w0 = (unsigned)((A2_PAD + i0 - j0 - 64) * 4);
        if (stg.bias_wave) a2_dma(rsB, st + 24576 + (wave & 3) * 1024, w0 + stg.boff);
        else               a2_dma(rsB, scratch_lds + (wave & 3) * 1024, OOB_OFF);
    };
    issue(0);
    if (nkt > 1) issue(1);

    // Q and dO fragments (B operands), delta_i = sum_d dO O, the row's log-sum-exp relative to the table's reference point
    h16x8 qf[4], dof[4];
    float dl = 0.f;
    const size_t qrow = (rowbase + min(qi, N - 1)) * (size_t)(H * 64) + (size_t)(active ? h : 0) * 64;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 z = {0u, 0u, 0u, 0u};
        const bool ok = active && qi < N;
        const u32x4 qv = ok ? *(const u32x4*)(q + qrow + 16 * s + 8 * hi) : z;
        const u32x4 dv = ok ? *(const u32x4*)(dout + qrow + 16 * s + 8 * hi) : z;
        const u32x4 ov = ok ? *(const u32x4*)(out + qrow + 16 * s + 8 * hi) : z;
        qf[s] = __builtin_bit_cast(h16x8, qv);
        dof[s] = __builtin_bit_cast(h16x8, dv);
#pragma unroll
        for (int e = 0; e < 4; ++e) dl += h16_lo_to_f(dv[e]) * h16_lo_to_f(ov[e]) + h16_hi_to_f(dv[e]) * h16_hi_to_f(ov[e]);
    }
    dl += __shfl_xor(dl, 32, 64);
    float Lp = 0.f;
    if (active) {
        Lp = lse[((size_t)b * H + h) * N + min(qi, N - 1)];
        if (biasT) Lp -= biasT[(size_t)h * ldT + (ldT - 1)];     // the table is stored relative to its reference point m_h
        if (hi == 0 && qi < N) delta[((size_t)b * H + h) * N + qi] = dl;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { asm volatile("" : "+v"(qf[s])); asm volatile("" : "+v"(dof[s])); }
    asm volatile("" : "+v"(Lp), "+v"(dl));                    // every prologue load is consumed before the tile loop

    f32x16 acc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    const float c = scale * A2_LOG2E;
    const float qscale = (A2_DQ_PK && A2_DQ_BATCH) ? scale : 1.f;      // dQ = scale dS K: taken out of the element loop

    for (int t = 0; t < nkt; ++t) {
        if (t + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else             asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 2 < nkt) issue(t + 2);
        if (!active) continue;
        const char* Kr = ring + (t % A2_NST) * A2B_STAGE;
        const char* Kb = Kr + 8192;
        const char* Vr = Kr + 16384;
        const float* bw = (const float*)(Kr + 24576) + wave * A2_BWIN;
        const int j0 = t * A2_TKV;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int jb = j0 + 32 * sub;
            if (jb > i0 + 31) break;
            f32x16 st, dp;
#pragma unroll
            for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
#if A2_DQ_BATCH
            {   // all eight fragment reads in flight before the first MFMA, retired in two groups (hipcc issued them one at a time
                // through the same four registers: read -> wait -> MFMA, seen in the ISA)
                h16x8 kfr[4], vfr[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) { kfr[s] = a2_frag_rows(Kr, 32 * sub, s, lane); vfr[s] = a2_frag_rows(Vr, 32 * sub, s, lane); }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if ((s & 1) == 0) asm volatile("" : "+v"(kfr[s]), "+v"(vfr[s]), "+v"(kfr[s + 1]), "+v"(vfr[s + 1]));
                    st = MFMA16(kfr[s], qf[s], st);                               // S^T  = K Q^T
                    dp = MFMA16(vfr[s], dof[s], dp);                              // dP^T = V dO^T
                }
            }
#else
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = MFMA16(a2_frag_rows(Kr, 32 * sub, s, lane), qf[s], st);      // S^T  = K Q^T
                dp = MFMA16(a2_frag_rows(Vr, 32 * sub, s, lane), dof[s], dp);     // dP^T = V dO^T
            }
#endif
            const float* bp = bw + (64 - 32 * sub) + ql - 4 * hi;
            const float* mp = mb + jb + 4 * hi;
            float bv[16];
            const bool diag = jb + 31 > i0;
            const int d0 = qi - (jb + 4 * hi);
            float4 m4s[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) m4s[g] = *(const float4*)(mp + 8 * g);
#if A2_DQ_BATCH
            float bpv[16];                     // bias window gathered in one pass (see m4s: nothing waits element by element)
#pragma unroll
            for (int r = 0; r < 16; ++r) bpv[r] = bp[-((r & 3) + 8 * (r >> 2))];
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bpv[r]));
#if A2_DQ_PK
            if (diag) {                        // the block on the diagonal: keys above it leave through the bias term (a real branch: one block in nkt)
                int d0v = d0;
                asm volatile("" : "+v"(d0v));             // the selects depend on a value defined inside the branch: hipcc otherwise hoists all 16 of them in front of it
#pragma unroll
                for (int r = 0; r < 16; ++r) bpv[r] = (d0v - ((r & 3) + 8 * (r >> 2)) >= 0) ? bpv[r] : A2_NEG;
            }
#endif
#endif
#if A2_DQ_PK && A2_DQ_BATCH
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 m4 = m4s[g];
                const f32x2 mm2[2] = {{m4.x, m4.y}, {m4.z, m4.w}};
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) {
                    const int r = 4 * g + 2 * pq, cr = 2 * pq + 8 * g;
                    const f32x2 t2 = (f32x2{bpv[r], bpv[r + 1]} + mm2[pq]) - f32x2{Lp, Lp};
                    const f32x2 x2 = __builtin_elementwise_fma(f32x2{st[r], st[r + 1]}, f32x2{c, c}, t2);
                    const f32x2 pr2 = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
                    const f32x2 ds2 = pr2 * (f32x2{dp[r], dp[r + 1]} - f32x2{dl, dl});   // dS = P (dP - delta), 0 where masked; scale: see the dQ store
                    bv[r] = ds2[0]; bv[r + 1] = ds2[1];
                    st[r] = ds2[0]; st[r + 1] = ds2[1];
                }
            }
#else
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 m4 = m4s[g];
                const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e, cr = e + 8 * g;
#if A2_DQ_BATCH
                    float x = st[r] * c + bpv[r] + (mm[e] - Lp);
#else
                    float x = st[r] * c + bp[-cr] + (mm[e] - Lp);
#endif
                    if (diag) x = (d0 - cr >= 0) ? x : A2_NEG;
                    const float pr = __builtin_amdgcn_exp2f(x);
                    bv[r] = pr * (dp[r] - dl);                                    // dS (0 where masked)
                    st[r] = bv[r] * scale;
                }
            }
#endif
            if (dbias) {
                // d(bias)[rel] = sum of dS over the diagonal rel = i - j: output lane L stands for t = q - kr = L - 31 and pulls row
                // kr's element from query column q = t + kr through the cross-lane permute; then one read-add-write of this
                // wave's private table (every lane owns a distinct bin)
                const float dsum = (A2_ABLATE & 8) ? bv[0] + bv[15] : diag_sum_32x32(bv, lane);
                const int rel = (i0 - jb) + (lane - 31);
                if (!(A2_ABLATE & 16) && rel >= 0 && rel < nb) dbw[rel] += dsum;
                if (A2_ABLATE & 16) acc[0][0] += dsum * 1e-30f;      // (ds_add_f32 instead of this read-add-write: measured 20 us per layer SLOWER)
            }
#if A2_DQ_BATCH
            {
                h16x8 ktf[2][2], dsb[2];
#pragma unroll
                for (int s = 0; s < 2; ++s) { ktf[s][0] = a2_frag_cols_tr(Kb, 32 * sub, s, 0, lane); ktf[s][1] = a2_frag_cols_tr(Kb, 32 * sub, s, 32, lane); }
#pragma unroll
                for (int s = 0; s < 2; ++s) dsb[s] = a2_pack(st, s);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    asm volatile("" : "+v"(ktf[s][0]), "+v"(ktf[s][1]));
                    acc[0] = MFMA16(ktf[s][0], dsb[s], acc[0]);                                // dQ^T += K^T dS^T
                    acc[1] = MFMA16(ktf[s][1], dsb[s], acc[1]);
                }
            }
#else
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const h16x8 dsb = a2_pack(st, s);
                acc[0] = MFMA16(a2_frag_cols_tr(Kb, 32 * sub, s, 0, lane), dsb, acc[0]);       // dQ^T += K^T dS^T
                acc[1] = MFMA16(a2_frag_cols_tr(Kb, 32 * sub, s, 32, lane), dsb, acc[1]);
            }
#endif
        }
    }
    if (!active) return;
    if (qi < N) {
        float* drow = dq + (rowbase + qi) * (size_t)(H * 64) + (size_t)h * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = 32 * dt + 8 * g4 + 4 * hi;
                *(float4*)(drow + d) = make_float4(qscale * acc[dt][4 * g4], qscale * acc[dt][4 * g4 + 1], qscale * acc[dt][4 * g4 + 2], qscale * acc[dt][4 * g4 + 3]);
            }
    }
    if (dbias && !(A2_ABLATE & 32)) {
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // this wave's LDS updates are complete for its own reads
        if (dpart) {                                          // one row of the partial buffer (see attention.hip's dQ kernel): plain stores
            float* prow = dpart + (((size_t)b * H + h) * nqt + qt) * (size_t)(nqt * 32);
            for (int r = lane; r < nb; r += 64) prow[r] = dbw[r];
        } else
        for (int r = lane; r < min(nb, N); r += 64) {
            const float vv = dbw[r];
            if (vv != 0.f) unsafeAtomicAdd(dbias + (size_t)r * bias_ld + h, vv);
        }
    }
}

int attn2_bwd_dq_launch(const void* q, const void* k, const void* v, const float* biasT, const unsigned char* keymask,
                        const void* out, const void* dout, const float* lse, float* delta, float* dq, float* dbias, int bias_ld,
                        float* dpart, int B, int N, int H, float scale, hipStream_t st) {
    const int ldT = (A2_PAD + N + 2 * A2_BWIN + 3) / 4 * 4;
    const int nqt = (N + 31) / 32, ny = (H + 7) / 8, npad = (N + 63) / 64 * 64;
    const size_t lds = (size_t)A2_NST * A2B_STAGE + 4096 + (size_t)npad * 4 + (size_t)8 * (nqt * 32) * 4;
    if (lds > 160 * 1024 || N > 4096) return 1;                // caller falls back to the first-generation kernel
    static bool a1 = false;
    if (!a1) { (void)hipFuncSetAttribute((const void*)attn2_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a1 = true; }
    hipLaunchKernelGGL(attn2_bwd_dq_kernel, dim3(nqt * ny * B), dim3(A2_THREADS), lds, st, (const h16_t*)q, (const h16_t*)k, (const h16_t*)v,
                       biasT, ldT, keymask, (const h16_t*)out, (const h16_t*)dout, lse, delta, dq, dbias, bias_ld, dpart, B, N, H, scale);
    return omlm_post_launch("omlm_mqa_attn_bwd");
}

#if !OMLM_FP16      /* the bias table is fp32 in every precision: prepared by the bf16 copy of this file */
// -------------------------------------------------------------------------------------------------------------------------
extern "C" long long omlm_attn_bias_table_floats(int N, int H) {
    const int ldT = (A2_PAD + N + 2 * A2_BWIN + 3) / 4 * 4, H8 = (H + 7) / 8 * 8;
    return (long long)H8 * ldT;
}

// biasT: omlm_attn_bias_table_floats(N, H) floats.  bias may be null (no rel-pos bias).  q_scale / k_scale (64 floats each, optional):
// the learned per-dim scales applied after the l2 normalisation -- they give the bound max_d |q_scale_d k_scale_d| on |q.k| that
// selects the fixed-reference softmax; alternatively qk_bound > 0 states the bound directly (callers with unit q, k: 1.0);
// neither: online softmax.  scale: the attention scale (8).  p_max_log2: 0 for bf16 / fp32 attention operands, 15 for half operands (the
// fixed reference point is lowered by 15 so that the probability numerators use half's normal range; see attn2_bias_prep_kernel).
extern "C" int omlm_attn_bias_prepare(const float* bias, float* biasT, int N, int H, int bias_ld, const float* q_scale,
                                      const float* k_scale, float qk_bound, float scale, int p_max_log2, void* stream) {
    OMLM_CHECK_ARG(biasT && N > 0 && H > 0, "null table / sizes");
    OMLM_CHECK_ARG(p_max_log2 == 0 || p_max_log2 == 15, "p_max_log2: 0 (bf16 / fp32 operands) or 15 (half operands)");
    const int ldT = (A2_PAD + N + 2 * A2_BWIN + 3) / 4 * 4, H8 = (H + 7) / 8 * 8;
    A2PrepGroup grp;
    memset(&grp, 0, sizeof(grp));
    grp.out[0] = biasT; grp.qs[0] = q_scale; grp.ks[0] = k_scale;
    hipLaunchKernelGGL(attn2_bias_prep_kernel, dim3(H8, 1), dim3(256), 0, as_stream(stream), bias, grp, N, H, bias_ld, ldT,
                       qk_bound, scale * A2_LOG2E, p_max_log2);
    return omlm_post_launch("omlm_attn_bias_prepare");
}
// The tables of `layers` attention layers over ONE rel-pos table in one launch: biasT[l] (omlm_attn_bias_table_floats(N, H) floats each) from
// the layer's learned scales q_scale[l] / k_scale[l] (64 floats each; all given, or all NULL with qk_bound as in omlm_attn_bias_prepare).
// biasT / q_scale / k_scale: HOST arrays of device pointers.
extern "C" int omlm_attn_bias_prepare_group(const float* bias, float* const* biasT, int layers, int N, int H, int bias_ld,
                                            const float* const* q_scale, const float* const* k_scale, float qk_bound, float scale,
                                            int p_max_log2, void* stream) {
    if (layers <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(biasT && N > 0 && H > 0, "null table / sizes");
    OMLM_CHECK_ARG(p_max_log2 == 0 || p_max_log2 == 15, "p_max_log2: 0 (bf16 / fp32 operands) or 15 (half operands)");
    OMLM_CHECK_ARG((q_scale == nullptr) == (k_scale == nullptr), "q_scale and k_scale: both or neither");
    const int ldT = (A2_PAD + N + 2 * A2_BWIN + 3) / 4 * 4, H8 = (H + 7) / 8 * 8;
    for (int base = 0; base < layers; base += A2_PREP_MAX) {
        const int n = layers - base < A2_PREP_MAX ? layers - base : A2_PREP_MAX;
        A2PrepGroup grp;
        memset(&grp, 0, sizeof(grp));
        for (int l = 0; l < n; ++l) {
            OMLM_CHECK_ARG(biasT[base + l], "null table");
            grp.out[l] = biasT[base + l];
            grp.qs[l] = q_scale ? q_scale[base + l] : nullptr;
            grp.ks[l] = k_scale ? k_scale[base + l] : nullptr;
        }
        hipLaunchKernelGGL(attn2_bias_prep_kernel, dim3(H8, n), dim3(256), 0, as_stream(stream), bias, grp, N, H, bias_ld, ldT,
                           qk_bound, scale * A2_LOG2E, p_max_log2);
    }
    return omlm_post_launch("omlm_attn_bias_prepare_group");
}

// d(bias) partial rows -> the [N, bias_ld] table.  The dQ kernels leave one fp32 row of nqt*32 bins per (sample, head, query tile) in
// the workspace (bins [0, 32 (qt + 1)) of row qt are written; the rest is never read); a thread here owns one (bin, head) and adds
// the rows of every S-th sample down its column -- coalesced along the bins -- then one atomic per thread folds the S sample groups.
#define A2_DBR_S 8
__global__ __launch_bounds__(256) void attn_dbias_reduce_kernel(const float* __restrict__ dpart, float* __restrict__ dbias, int bias_ld,
                                                                int B, int N, int H, int nqt) {
    const int r = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y, s = blockIdx.z;
    if (r >= N) return;
    const size_t NB = (size_t)nqt * 32;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int b = s; b < B; b += A2_DBR_S) {
        const float* col = dpart + ((size_t)b * H + h) * nqt * NB + r;
        int qt = r >> 5;
        for (; qt + 4 <= nqt; qt += 4) {
            a0 += col[(size_t)qt * NB];
            a1 += col[(size_t)(qt + 1) * NB];
            a2 += col[(size_t)(qt + 2) * NB];
            a3 += col[(size_t)(qt + 3) * NB];
        }
        for (; qt < nqt; ++qt) a0 += col[(size_t)qt * NB];
    }
    const float t = (a0 + a1) + (a2 + a3);
    if (t != 0.f) unsafeAtomicAdd(dbias + (size_t)r * bias_ld + h, t);
}

extern "C" long long omlm_mqa_attn_bwd_workspace_bytes(int B, int N, int H) {
    const long long nqt = (N + 31) / 32;
    return (long long)B * H * nqt * nqt * 32 * (long long)sizeof(float);
}

extern "C" __attribute__((visibility("hidden"))) int omlm_attn_dbias_reduce_launch(const float* dpart, float* dbias, int bias_ld, int B, int N, int H, void* stream) {
    const int nqt = (N + 31) / 32;
    hipLaunchKernelGGL(attn_dbias_reduce_kernel, dim3((N + 255) / 256, H, min(B, A2_DBR_S)), dim3(256), 0, as_stream(stream), dpart, dbias,
                       bias_ld, B, N, H, nqt);
    return omlm_post_launch("omlm_mqa_attn_bwd");
}

#endif

// bf16 forward.  biasT from omlm_attn_bias_prepare (or null: no bias).
int attn2_fwd_launch(const void* q, const void* k, const void* v, const float* biasT, const unsigned char* keymask,
                     void* out, float* lse, int B, int N, int H, float scale, hipStream_t st) {
    const int ldT = (A2_PAD + N + 2 * A2_BWIN + 3) / 4 * 4;
    if (N > 64 * 64) { omlm_set_error("attention: N > 4096 keys per sample is not supported (liveness prologue covers 4096 keys)"); return OMLM_ERR_UNSUPPORTED; }
    const int nqt = (N + 31) / 32, ny = (H + 7) / 8;
    const size_t lds = (size_t)A2_NST * A2_STAGE + (size_t)((N + 63) / 64 * 64) * 2 + 64 * 8 + 128;
    static bool a4 = false;
    if (!a4) {
        (void)hipFuncSetAttribute((const void*)attn4_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)attn4_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        a4 = true;
    }
    // both softmax forms: the one the table's flag does not name returns at its first instruction (no table: online only)
    if (biasT) hipLaunchKernelGGL(attn4_fwd_kernel<true>, dim3(nqt * ny * B), dim3(A4_THREADS), lds, st, (const h16_t*)q, (const h16_t*)k,
                                  (const h16_t*)v, biasT, ldT, keymask, (h16_t*)out, lse, B, N, H, scale);
    hipLaunchKernelGGL(attn4_fwd_kernel<false>, dim3(nqt * ny * B), dim3(A4_THREADS), lds, st, (const h16_t*)q, (const h16_t*)k, (const h16_t*)v,
                       biasT, ldT, keymask, (h16_t*)out, lse, B, N, H, scale);
    return omlm_post_launch("omlm_mqa_attn_fwd");
}

}   // namespace OMLM_NS
