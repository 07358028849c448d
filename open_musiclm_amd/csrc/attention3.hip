// Attention backward, dK / dV (transformer.py:303-331 autograd), third generation for 16-bit operands.
//
// What the counters said about the second-generation kernel (attention.hip: attn_bwd_dkv_kernel, profiles/r03a_attn_bwd_pmc.md:
// 323 us per layer at B = 32, N = 1116, H = 8 -- 55 % of the whole attention backward, 44 % of its wave-cycles parked in waits,
// 1.73 M vector-memory instructions): a workgroup owned ONE 32-key tile and each of its waves fetched the Q / dO tile of its own
// (query tile, head) item straight into MFMA fragments -- 32 rows x 32 bytes per load instruction (every instruction touches 32
// cache lines and uses a quarter of each), and every Q / dO tile of a sample travelled L2 -> CU once per 32-key tile: 1.3 GB per
// layer.  Here
//   * a workgroup owns 128 keys = 4 waves x 32 keys, and ALL its waves work on the SAME item: the item's Q and dO tiles
//     ([32 queries][64 dims]) are staged ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4, 16 rows x 64 bytes per wave
//     instruction, straight into the blocked image the transpose reads want) into a 3-stage ring, two items ahead of the matrix
//     cores, one barrier per item -- a quarter of the L2 -> CU bytes, no VGPR staging, no per-item ds_write pass;
//   * lse / delta / the 63-value bias window of an item arrive the same way (two small DMA pieces per wave);
//   * a key range's items are cut into chunks of CH query tiles so that ~3 workgroups per CU exist and the longest one is short;
//     the chunks of a range add their dK / dV with fp32 atomics (64 KiB per workgroup, ~45 MB per layer) into zero-filled
//     outputs -- each wave owns its 32 keys, so there is no cross-wave reduction at all.
// Per element the arithmetic is the second generation's (same exponent form against the prepared table, same roundings).
#include "common.h"
#include <stdlib.h>

namespace OMLM_NS {

#define A3_T 256
#define A3_KR 128                      /* keys per workgroup */
#define A3_NST 3
#define A3_AUX 2048                    /* per-wave aux piece: [0,1024) window DMA (window 256 B | table tail 16 B | filler), [1024,1280) lse | delta */
#define A3_IMG (4096 + 256)            /* one [32][64] blocked image: the units of rows 16..31 sit 128 bytes further (see a3_blk_off) */
#define A3_STAGE (2 * A3_IMG + 4 * A3_AUX)
#define A3_NEG (-1.0e30f)
#define A3_LOG2E 1.4426950408889634f
#define A3_PAD 64                      /* zero entries in front of each row of the prepared bias table (attention2.hip: A2_PAD) */

__device__ __forceinline__ int a3_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// blocked image of a [32 rows][64 cols] 16-bit tile (attention.hip: tile_off_blk): 1-KiB units (row / 16, col / 32), 128-byte blocks of
// [4 rows][16 cols] ordered for ds_read_b64_tr_b16
__device__ __forceinline__ int a3_blk_off(int row, int colbyte) {
    const int col = colbyte >> 1;
    const int rq = (row >> 2) & 3;
    const int p = ((rq >> 1) << 2) | ((rq & 1) << 1) | ((col >> 4) & 1);
    // + 128 bytes for rows 16..31: a 16-lane group of a row-fragment ds_read_b128 holds two row quads of each half, and with all units on
    // 1-KiB boundaries the four quads met on the same 32 banks (4-way conflict, SQ_LDS_BANK_CONFLICT 8 % of the kernel's wave-cycles);
    // shifted, the two halves use disjoint bank halves (2-way, the best this image allows: tools/lds_conflicts.py model)
    return (((row >> 4) << 1) + (col >> 5)) * 1024 + (row >> 4) * 128 + p * 128 + (row & 3) * 32 + (col & 15) * 2;
}
// A-operand row fragment (row = row0 + (lane & 31), dims 16 s + 8 (lane >> 5) .. +7) out of the blocked image
__device__ __forceinline__ h16x8 a3_frag_rows(const char* lds, int s, int lane) {
    return *(const h16x8*)(lds + a3_blk_off(lane & 31, (2 * s + (lane >> 5)) * 16));
}
// transposed operand: lane gets column col0 + (lane & 31) and the 8 rows MFMA k-index 8 (lane >> 5) + e maps to (accumulator row order)
__device__ __forceinline__ h16x8 a3_frag_cols_tr(const char* lds, int s, int col0, int lane) {
    const char* base = lds + ((s << 1) + (col0 >> 5)) * 1024 + s * 128 + lane * 8;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, base + 512));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h16x8, v);
}
__device__ __forceinline__ h16x8 a3_pack(const f32x16& p, int s) {
    u32x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = pack_h16_rne(p[8 * s + 2 * e], p[8 * s + 2 * e + 1]);
    return __builtin_bit_cast(h16x8, h);
}

// LDS-DMA with per-lane 64-bit source addresses (any mix of buffers in one instruction); destination = M0 + lane * size.
// Inline asm: invisible to hipcc's vmcnt bookkeeping (attention2.hip explains why that is wanted); ordering is by the counted waits below.
// (M0 is written and read inside the one statement and declared clobbered: nothing else in this kernel uses it)
__device__ __forceinline__ void a3_dma16(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
// the same with a wave-uniform base and a per-lane 32-bit byte offset (no 64-bit address arithmetic per item)
__device__ __forceinline__ void a3_dma16s(const void* sbase, unsigned voff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void a3_dma4(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(gsrc), "s"(lds_dst) : "memory", "m0");
}

// grid: B x (workgroups per sample); a sample's workgroups walk its key ranges r = 0, 1, ... (128 keys each), range r being cut into
// ceil((nqt - 4 r) / CH) chunks of CH query tiles.
#ifndef A3_WAVES
#define A3_WAVES 2                     /* waves per SIMD the register allocation aims at */
#endif
__global__ __launch_bounds__(A3_T) __attribute__((amdgpu_waves_per_eu(A3_WAVES)))
void attn3_bwd_dkv_kernel(const h16_t* __restrict__ q, const h16_t* __restrict__ k, const h16_t* __restrict__ v,
                          const unsigned char* __restrict__ keymask, const h16_t* __restrict__ dout,
                          const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dk, float* __restrict__ dv,
                          const float* __restrict__ biasT, int ldT, int B, int N, int H, float scale, int CH, int wg_per_sample) {
    extern __shared__ __attribute__((aligned(16))) char smem3[];
    const int lane = threadIdx.x & 63, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nqt = (N + 31) / 32;
    // XCD-aware: the workgroups of one sample share its Q / dO tiles through an XCD's L2 (workgroup i runs on XCD i % 8); dealt in launch
    // order every XCD fetched every sample's Q and dO
    int lg = blockIdx.x;
#ifndef A3_NOXCD
    {
        const int total = B * wg_per_sample, lin = blockIdx.x;
        const int qq = total >> 3, rr = total & 7, xcd = lin & 7, idx = lin >> 3;
        lg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
#endif
    const int b = lg / wg_per_sample;
    int rem = lg - b * wg_per_sample, r = 0, chunk = 0;
    for (;; ++r) {                                            // uniform scalar scan: which key range this workgroup belongs to
        const int nch = (nqt - 4 * r + CH - 1) / CH;
        if (rem < nch) { chunk = rem; break; }
        rem -= nch;
    }
    const int it0 = 4 * r + chunk * CH, it1 = min(nqt, it0 + CH);
    const int nitems = (it1 - it0) * H;
    const int j0w = r * A3_KR + 32 * wave;                    // this wave's 32 keys
    const int jtw = 4 * r + wave;                             // ... as a 32-key tile index
    const int kj = j0w + (lane & 31);
    const size_t rowbase = (size_t)b * N;
    const float c = scale * A3_LOG2E;
    const bool has_bias = biasT != nullptr;

    // K^T, V^T B-operands: lane n = key kj, dims 16 s + 8 hi .. +7 -- resident for the whole kernel
    h16x8 kf[4], vf[4];
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const size_t off = (rowbase + min(kj, N - 1)) * 64 + 16 * s + 8 * hi;
            kf[s] = __builtin_bit_cast(h16x8, kj < N ? *(const u32x4*)(k + off) : z);
            vf[s] = __builtin_bit_cast(h16x8, kj < N ? *(const u32x4*)(v + off) : z);
        }
    }
    const bool keylive = kj < N && (keymask ? keymask[rowbase + min(kj, N - 1)] != 0 : true);
    // consumed here, so that hipcc's own waits for these loads sit in front of the loop and not inside it (they would drain the DMA ring)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(kf[s]), "+v"(vf[s]));

    // per-lane source coordinates of the unit this wave stages (unit = wave): row rowu, first column colu of the blocked image
    const int pp_ = lane >> 3, rq_ = ((pp_ >> 2) << 1) | ((pp_ >> 1) & 1);
    const int rowu = (wave >> 1) * 16 + rq_ * 4 + ((lane >> 1) & 3);
    const int colu = (wave & 1) * 32 + (pp_ & 1) * 16 + (lane & 1) * 8;
    const unsigned ring_lds = (unsigned)(size_t)LDS_PTR(char, smem3);

    // (it, h) of an item advance as counters, and so do the per-lane source offsets of its four DMA pieces: byte offsets from the tensor
    // bases (32 bits: the launch checks the extents), bumped by a constant per head and rebuilt once per query tile -- item / H, item % H
    // and the 64-bit address products were ~130 scalar + ~25 vector instructions per item (SQ_INSTS_SALU 2.1e7 in the counters).
    unsigned qoff = 0, boff = 0;                              // Q / dO piece; bias window piece (lanes 0-15) | table tail (the others)
    const float* ldp = lse;                                   // lse (lanes 0-31) | delta (lanes 32-63) element of the item
    const unsigned bstep = (unsigned)ldT * 4u;
    auto tile_offsets = [&](int it) {                         // head 0 of query tile `it`
        const int qi = min(32 * it + rowu, N - 1);            // rows past N: clamped (their scores are masked below)
        qoff = (unsigned)((((int)rowbase + qi) * H) * 64 + colu) * 2u;
        // lanes 0-15: the bias window for this wave's keys, table index A3_PAD + rel - 1 from rel = 32 (it - jtw) - 31 (one entry early:
        // 16-byte aligned); lane 16: the row's tail [.., flag, m_h]; the other lanes repeat lane 16's address
        const int w0 = max(A3_PAD + 32 * (it - jtw) - 32, 0);
        boff = has_bias ? (unsigned)(lane < 16 ? w0 + 4 * lane : ldT - 4) * 4u : 0u;
        ldp = (hi ? delta : lse) + ((size_t)b * H * N + min(32 * it + (lane & 31), N - 1));
    };
    auto issue = [&](int stage) {                             // 4 DMA wave-instructions per wave per item, then on to the next head
        const unsigned st = ring_lds + (unsigned)(stage * A3_STAGE);
        a3_dma16s(q, qoff, st + wave * 1024 + (wave >> 1) * 128);
        a3_dma16s(dout, qoff, st + A3_IMG + wave * 1024 + (wave >> 1) * 128);
        const unsigned ax = st + 2 * A3_IMG + wave * A3_AUX;
        a3_dma16s(has_bias ? (const void*)biasT : (const void*)lse, boff, ax);
        a3_dma4(ldp, ax + 1024);
        qoff += 128u; boff += has_bias ? bstep : 0u; ldp += N;
    };

    f32x16 dkacc[2], dvacc[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[0][e] = 0.f; dkacc[1][e] = 0.f; dvacc[0][e] = 0.f; dvacc[1][e] = 0.f; }

    int it_i = it0, h_i = 0, st_i = 0;                        // the next item to be issued
    tile_offsets(it0);
    auto issue_next = [&]() {
        issue(st_i);
        if (++h_i == H) { h_i = 0; ++it_i; tile_offsets(it_i); }
        if (++st_i == A3_NST) st_i = 0;
    };
    if (nitems > 0) issue_next();
    if (nitems > 1) issue_next();
    int it = it0, hcur = 0, stage = 0;                        // the item being multiplied
    for (int item = 0; item < nitems; ++item, (++hcur == H ? (hcur = 0, ++it) : 0), (++stage == A3_NST ? (stage = 0) : 0)) {
        // own pieces of `item` landed (the next item's four may stay in flight), then everybody's; the barrier also says that all
        // waves are done with item - 1, whose stage item + 2 is about to overwrite
        if (item + 1 < nitems) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else                   asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (item + 2 < nitems) issue_next();
        const int i0 = 32 * it;
        if (i0 + 31 < j0w) continue;                          // every query of the tile precedes every key of this wave (causal): nothing to do
        const char* Qs = smem3 + stage * A3_STAGE;
        const char* dOs = Qs + A3_IMG;
        const float* axa = (const float*)(Qs + 2 * A3_IMG + wave * A3_AUX);
        const float* axb = axa + 256;                         // lse[32] | delta[32]
        const float mh = has_bias ? axa[64 + 3] : 0.f;        // the head's reference point (table tail), 0 without a fixed one
        // window index of (query row crow(r, hi), this lane's key): rel - (32 (it - jtw) - 31) + 1 = cr + 4 hi - (lane & 31) + 32
        const float* bwp = axa + 32 + 4 * hi - (lane & 31);
        const float* lp = axb + 4 * hi;
        f32x16 st, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { st[e] = 0.f; dp[e] = 0.f; }
        {
            h16x8 qa[4], doa[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { qa[s] = a3_frag_rows(Qs, s, lane); doa[s] = a3_frag_rows(dOs, s, lane); }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                st = OMLM_MFMA_32x32x16(qa[s], kf[s], st);    // S  = Q K^T   (rows i, column = this lane's key)
                dp = OMLM_MFMA_32x32x16(doa[s], vf[s], dp);   // dP = dO V^T
            }
        }
        f32x16 pr;
        float bvv[16], lvv[16], dvv[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int cr = (rr & 3) + 8 * (rr >> 2);          // crow(rr, hi) - 4 hi
            bvv[rr] = bwp[cr];                                // (no bias: overwritten below -- one branch, not one per element)
            lvv[rr] = lp[cr];
            dvv[rr] = lp[32 + cr];
        }
        if (!has_bias) {
            float z = 0.f;
            asm volatile("" : "+v"(z));                       // (defined inside the branch: otherwise 16 selects on every item)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) bvv[rr] = z;
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) asm volatile("" : "+v"(bvv[rr]), "+v"(lvv[rr]), "+v"(dvv[rr]));
        // Element arithmetic on register pairs (v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32).  A dead key (this lane's column) leaves through
        // the reference point: mhk = m_h - 1e30 there, so x = c S + bias - (lse - mhk) = -1e30 and P = 0 without a select; dS carries no
        // softmax scale here -- dK = scale dS^T Q takes it once, at the final store.
        const float mhk = keylive ? mh : mh + A3_NEG;
        if (!(i0 >= j0w + 31 && i0 + 31 < N)) {               // the tile touches the diagonal or runs past N: those rows leave through the bias term
            int kjv = kj;
            asm volatile("" : "+v"(kjv));                     // (defined inside the branch: hipcc otherwise hoists the 16 selects in front of it)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int i = i0 + a3_crow(rr, hi);
                bvv[rr] = (i >= kjv && i < N) ? bvv[rr] : A3_NEG;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 16; rr += 2) {
            const f32x2 t2 = f32x2{bvv[rr], bvv[rr + 1]} - (f32x2{lvv[rr], lvv[rr + 1]} - f32x2{mhk, mhk});
            const f32x2 x2 = __builtin_elementwise_fma(f32x2{st[rr], st[rr + 1]}, f32x2{c, c}, t2);
            const f32x2 p2 = {__builtin_amdgcn_exp2f(x2[0]), __builtin_amdgcn_exp2f(x2[1])};
            const f32x2 ds2 = p2 * (f32x2{dp[rr], dp[rr + 1]} - f32x2{dvv[rr], dvv[rr + 1]});
            pr[rr] = p2[0]; pr[rr + 1] = p2[1];
            st[rr] = ds2[0]; st[rr + 1] = ds2[1];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const h16x8 pb = a3_pack(pr, s), dsb = a3_pack(st, s);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dvacc[dt] = OMLM_MFMA_32x32x16(a3_frag_cols_tr(dOs, s, 32 * dt, lane), pb, dvacc[dt]);    // dV^T += dO^T P
                dkacc[dt] = OMLM_MFMA_32x32x16(a3_frag_cols_tr(Qs, s, 32 * dt, lane), dsb, dkacc[dt]);    // dK^T += Q^T dS
            }
        }
    }
    // ---- this wave's 32 keys x 64 dims of dK and dV: transposed through LDS (pitch 33: conflict-free both ways) and added row by row ----
    __syncthreads();                                          // every wave is past its last reads of the ring
    float* red = (float*)smem3 + (size_t)wave * (64 * 33);
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int d = 32 * dt + a3_crow(rr, hi);
                red[d * 33 + (lane & 31)] = which == 0 ? scale * dkacc[dt][rr] : dvacc[dt][rr];
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // this wave's own LDS writes, then its own reads below
        float* dst = which == 0 ? dk : dv;
        for (int e = lane; e < 32 * 64; e += 64) {
            const int j = e >> 6, d = e & 63;                 // consecutive lanes -> consecutive d (coalesced 256-byte rows)
            const float val = red[d * 33 + j];
            if (j0w + j < N && val != 0.f) unsafeAtomicAdd(dst + (rowbase + j0w + j) * 64 + d, val);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

__global__ __launch_bounds__(256) void a3_zero_kernel(float4* __restrict__ p, size_t n4) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = z;
}

static int a3_chunk(int B, int N) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("OMLM_ATTN3_CH"); forced = e ? atoi(e) : 0; }
    if (forced > 0) return forced;
    const int nqt = (N + 31) / 32, nr = (N + A3_KR - 1) / A3_KR;
    long long units = 0;
    for (int r = 0; r < nr; ++r) units += nqt - 4 * r;
    long long ch = ((long long)B * units + 1399) / 1400;      // ~5 workgroups per CU (measured: B = 32, N = 1116: CH 4 -> 505 us per layer, 8 -> 522, 2 -> 528)
    if (ch < 2) ch = 2;
    if (ch > 16) ch = 16;
    return (int)ch;
}

// dk, dv [B * N, 64] fp32 are ZERO-FILLED here (memset node on the stream) and accumulated with atomics.  Returns 1 when the shape
// is not served (caller falls back to the second-generation kernel).
int attn3_bwd_dkv_launch(const void* q, const void* k, const void* v, const float* biasT, const unsigned char* keymask,
                         const void* dout, const float* lse, const float* delta, float* dk, float* dv,
                         int B, int N, int H, float scale, hipStream_t st) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("OMLM_ATTN_DKV3"); off = (e && e[0] == '0') ? 1 : 0; }
    if (off || N < 32 || (long long)B * N * H * 128 >= (1ll << 32)) return 1;      // (32-bit byte offsets into q / dout)
    const int ldT = ((A3_PAD + N + 2 * 128 + 3) / 4) * 4;    // layout of omlm_attn_bias_prepare (attention2.hip)
    const int CH = a3_chunk(B, N);
    const int nqt = (N + 31) / 32, nr = (N + A3_KR - 1) / A3_KR;
    int wps = 0;
    for (int r = 0; r < nr; ++r) wps += (nqt - 4 * r + CH - 1) / CH;
    const size_t lds = (size_t)A3_NST * A3_STAGE;             // 48 KiB (the final transposes reuse it: 4 x 8448 B)
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)attn3_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
    // Zero fill by a KERNEL of this library, not hipMemsetAsync: as a memset node of a captured micro-step the fill detached everything
    // behind it from the graph's completion -- the rest of the backward (this layer's dK / dV onwards) was still running when the launch
    // had "finished" and the optimizer's kernels started (round 4: after one fp16 overflow the skipped step's gradient clear raced with
    // those late writes and every later step stayed non-finite; OMLM_ATTN_DKV3=0, i.e. no memset node, or a host sync after the replay,
    // cured it).  A kernel node is ordered like every other launch.
    const size_t gfloats = (size_t)B * N * 64;
    auto fill = [&](float* p, size_t n) {
        const size_t n4 = n / 4;                               // n = B N 64: a multiple of 4; rows are 256-byte aligned
        unsigned blocks = (unsigned)((n4 + 255) / 256); if (blocks > 8192u) blocks = 8192u;
        hipLaunchKernelGGL(a3_zero_kernel, dim3(blocks), dim3(256), 0, st, (float4*)p, n4);
    };
    if (dv == dk + gfloats) fill(dk, 2 * gfloats);            // one allocation (the host's usual case): one fill launch instead of two
    else { fill(dk, gfloats); fill(dv, gfloats); }
    hipLaunchKernelGGL(attn3_bwd_dkv_kernel, dim3(B * wps), dim3(A3_T), lds, st, (const h16_t*)q, (const h16_t*)k, (const h16_t*)v, keymask,
                       (const h16_t*)dout, lse, delta, dk, dv, biasT, ldT, B, N, H, scale, CH, wps);
    return omlm_post_launch("omlm_mqa_attn_bwd");
}

}   // namespace OMLM_NS
