// Second-generation middle of ConvFeedForward (reference transformer.py:122-150), templated on the operand type (bf16 / fp32):
//   h1 [M, 2*Fp] --causal depthwise conv k=3--> u --GEGLU--> g --LayerNorm(F)--> --Dropout(p)--> h2 [M, Fp]
// Same contract, layouts and outputs as the wave-per-row kernels of ffmid.hip (which stay as the fallback: shapes beyond Fp = 4096,
// a backward without the saved normalised output, regenerated Philox masks); what changes is who owns what.
//
// The first generation gave a wave one row: every 8-channel chunk loaded 3 rows x 2 halves of h1 plus 6 tap vectors (12
// loads) and converted all of them (12 bf16 -> fp32 conversions per element) before the first useful FMA, ~70 VALU
// instructions per element.  At ~16 fp32 lanes per SIMD-cycle that is ~200 us of pure issue for the 97.5 M elements of a
// coarse-small layer against an HBM time of ~160 us: these kernels were issue-bound (profiles/r02b_kernel_stats.md:
// forward 309 us, backward 390 + 260 us).
//
// Here a THREAD owns a few channel pairs (value + gate) and walks DOWN the rows of a strip of one sample:
//   * the conv window is two rows of fp32 registers (x[t-1], x[t-2]); each h1 element is loaded and converted once;
//   * taps and gamma are converted once per strip and stay in registers as fp32;
//   * the arithmetic is written on float2 so that hipcc emits v_pk_fma_f32 / v_pk_mul_f32 (two lanes' worth per issue);
//   * forward: a workgroup covers ALL channels of its rows, the LayerNorm sums cross the waves through 1 KiB of LDS with
//     one barrier per batch of 4 rows, g never leaves registers;
//   * backward: the two row sums of LayerNorm^T come from a light prepass (reads dh2, the saved normalised output gh and
//     the keep bits; also d(gamma)); the main kernel is then column-local -- dropout^T, LayerNorm^T, GEGLU^T, conv^T and
//     d(conv taps) in ONE pass, du never exists in memory (first generation: bwd1 wrote it, bwd2 re-read it: 0.8 GB per
//     layer).  conv^T runs as two pending output rows in registers; a strip recomputes du for the 2 rows past its end.
#include "common.h"
#include <type_traits>

namespace OMLM_NS {

typedef f32x2 v2;

__device__ __forceinline__ v2 mk2(float a, float b) { v2 r; r[0] = a; r[1] = b; return r; }
__device__ __forceinline__ v2 splat2(float a) { return mk2(a, a); }
__device__ __forceinline__ v2 bf2_to_f2(unsigned w) { return mk2(h16_lo_to_f(w), h16_hi_to_f(w)); }      // the build's 16-bit type (common.h)
__device__ __forceinline__ unsigned f2_to_bf2(v2 v) { return pack_h16_rne(v[0], v[1]); }
__device__ __forceinline__ v2 fma2(v2 a, v2 b, v2 c) { return __builtin_elementwise_fma(a, b, c); }

// 8 / 4 consecutive channels as they sit in HBM (bf16: 16 / 8 bytes, fp32: 32 / 16 bytes), handed out as float2 pairs
template <typename T> struct Ch8;
template <> struct Ch8<h16_t> {
    u32x4 r;
    __device__ __forceinline__ void load(const h16_t* p) { r = *(const u32x4*)p; }
    __device__ __forceinline__ void zero() { r[0] = 0u; r[1] = 0u; r[2] = 0u; r[3] = 0u; }
    __device__ __forceinline__ v2 get(int i) const { return bf2_to_f2(r[i]); }
    static __device__ __forceinline__ void store(h16_t* p, const v2 (&y)[4]) {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = f2_to_bf2(y[i]);
        *(u32x4*)p = o;
    }
};
template <> struct Ch8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ v2 get(int i) const {
        return i == 0 ? mk2(a.x, a.y) : (i == 1 ? mk2(a.z, a.w) : (i == 2 ? mk2(b.x, b.y) : mk2(b.z, b.w)));
    }
    static __device__ __forceinline__ void store(float* p, const v2 (&y)[4]) {
        ((float4*)p)[0] = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]);
        ((float4*)p)[1] = make_float4(y[2][0], y[2][1], y[3][0], y[3][1]);
    }
};
template <typename T> struct Ch4;
template <> struct Ch4<h16_t> {
    u32x2 r;
    __device__ __forceinline__ void load(const h16_t* p) { r = *(const u32x2*)p; }
    __device__ __forceinline__ void zero() { r[0] = 0u; r[1] = 0u; }
    __device__ __forceinline__ v2 get(int i) const { return bf2_to_f2(r[i]); }
    static __device__ __forceinline__ void store(h16_t* p, const v2 (&y)[2]) {
        u32x2 o; o[0] = f2_to_bf2(y[0]); o[1] = f2_to_bf2(y[1]);
        *(u32x2*)p = o;
    }
};
template <> struct Ch4<float> {
    float4 a;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ v2 get(int i) const { return i == 0 ? mk2(a.x, a.y) : mk2(a.z, a.w); }
    static __device__ __forceinline__ void store(float* p, const v2 (&y)[2]) { *(float4*)p = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]); }
};

// 8 channels of an operand that may arrive as hi/lo planes of the 16-bit type (precision "fp16ff": the forward reads the un-rounded h1,
// taps and gamma as hi + lo and leaves h2 as planes; `lo` = element distance from the hi plane to the lo plane).  PL false: Ch8<T> itself.
template <typename T, bool PL> struct Row8 {
    Ch8<T> h;
    __device__ __forceinline__ void load(const T* p, long long) { h.load(p); }
    __device__ __forceinline__ void zero() { h.zero(); }
    __device__ __forceinline__ v2 get(int i) const { return h.get(i); }
};
template <typename T> struct Row8<T, true> {
    Ch8<T> h, l;
    __device__ __forceinline__ void load(const T* p, long long lo) { h.load(p); l.load(p + lo); }
    __device__ __forceinline__ void zero() { h.zero(); l.zero(); }
    __device__ __forceinline__ v2 get(int i) const { return h.get(i) + l.get(i); }
};
// y as planes: hi = rne16(y), lo = rne16(y - hi)
__device__ __forceinline__ void store_planes8(h16_t* hi, h16_t* lo, const v2 (&y)[4]) {
    u32x4 o, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[i] = f2_to_bf2(y[i]); l[i] = f2_to_bf2(y[i] - bf2_to_f2(o[i])); }
    *(u32x4*)hi = o;
    *(u32x4*)lo = l;
}
__device__ __forceinline__ void store_planes8(float*, float*, const v2 (&)[4]) {}          // (fp32 operands have no planes)
struct FfPlanes { long long h1_lo, convw_lo, gamma_lo; void* h2_lo;        // element distances hi -> lo of the inputs; lo plane of h2
                  unsigned char* h2_8; long long h2_8_stride; unsigned char* scale8;        // MX instantiations: h2's fp8 planes [hi8 | lo8] (pitch 2 Fp bytes) + row scales instead of h2_lo
                  const unsigned char* h1_8; };                                             // MX instantiations: h1's lo plane as bf8 bytes at h1's element pitch (instead of h1_lo)
// a pair of bf8 (e5m2) bytes of a word -> two floats (the conversion's word select is an immediate)
__device__ __forceinline__ v2 bf8_pair(u32x2 l, int i) {
    switch (i) {
        case 0: return __builtin_amdgcn_cvt_pk_f32_bf8((int)l[0], false);
        case 1: return __builtin_amdgcn_cvt_pk_f32_bf8((int)l[0], true);
        case 2: return __builtin_amdgcn_cvt_pk_f32_bf8((int)l[1], false);
        default: return __builtin_amdgcn_cvt_pk_f32_bf8((int)l[1], true);
    }
}
// 8 channels of h1 at element offset `off`: the operand itself (PL false), hi + half lo planes (PL), hi plane + bf8 lo bytes (MX:
// omlm_gemm_mx16 leaves h1's lo plane as e5m2 -- 1 byte per element, hi + lo ~ h1 to 2^-14)
template <typename T, bool PL, bool MX> struct H1Row {
    Row8<T, PL> r;
    __device__ __forceinline__ void load(const T* h1, size_t off, const FfPlanes& pl) { r.load(h1 + off, pl.h1_lo); }
    __device__ __forceinline__ void zero() { r.zero(); }
    __device__ __forceinline__ v2 get(int i) const { return r.get(i); }
};
template <typename T> struct H1Row<T, true, true> {
    Ch8<T> h; u32x2 l;
    __device__ __forceinline__ void load(const T* h1, size_t off, const FfPlanes& pl) { h.load(h1 + off); l = *(const u32x2*)(pl.h1_8 + off); }
    __device__ __forceinline__ void zero() { h.zero(); l[0] = 0u; l[1] = 0u; }
    __device__ __forceinline__ v2 get(int i) const { return h.get(i) + bf8_pair(l, i); }
};
// y as the half hi plane + fp8 planes (omlm_gemm_mx16's operand form): hi8 = e4m3(hi sh), lo8 = e4m3((y - hi) sl)
__device__ __forceinline__ void store_mx8(h16_t* hi, unsigned char* p8h, unsigned char* p8l, const v2 (&y)[4], float sh, float sl) {
    u32x4 o;
    u32x2 h8, l8;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f2_to_bf2(y[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const v2 ha = bf2_to_f2(o[2 * i]), hb = bf2_to_f2(o[2 * i + 1]);
        h8[i] = pack4_fp8(ha[0] * sh, ha[1] * sh, hb[0] * sh, hb[1] * sh);
        l8[i] = pack4_fp8((y[2 * i][0] - ha[0]) * sl, (y[2 * i][1] - ha[1]) * sl, (y[2 * i + 1][0] - hb[0]) * sl, (y[2 * i + 1][1] - hb[1]) * sl);
    }
    *(u32x4*)hi = o;
    *(u32x2*)p8h = h8;
    *(u32x2*)p8l = l8;
}
__device__ __forceinline__ void store_mx8(float*, unsigned char*, unsigned char*, const v2 (&)[4], float, float) {}

__device__ __forceinline__ void pin_regs(Ch8<h16_t>& a, Ch8<h16_t>& b) { asm volatile("" : "+v"(a.r), "+v"(b.r)); }
__device__ __forceinline__ void pin_regs(Ch8<float>&, Ch8<float>&) {}

// GELU pieces of a gate pair u:  h = Phi(u) = 0.5 (1 + erf(u / sqrt 2)),  ex = exp(-u^2 / 2).
// erf by Abramowitz-Stegun 7.1.26 like ffmid.hip (|abs err| <= 1.5e-7), with the 1/sqrt2 and the 0.5 folded into constants.
__device__ __forceinline__ void gelu_parts(v2 u, v2& h, v2& ex) {
    const v2 au = mk2(fabsf(u[0]), fabsf(u[1]));
    const v2 d = fma2(au, splat2(0.23164189f), splat2(1.0f));                  // 1 + 0.3275911 |u| / sqrt 2
    const v2 t = mk2(__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]));
    v2 poly = fma2(t, splat2(0.5f * 1.061405429f), splat2(0.5f * -1.453152027f));
    poly = fma2(poly, t, splat2(0.5f * 1.421413741f));
    poly = fma2(poly, t, splat2(0.5f * -0.284496736f));
    poly = fma2(poly, t, splat2(0.5f * 0.254829592f));
    poly = poly * t;
    const v2 q = u * u;
    ex = mk2(__builtin_amdgcn_exp2f(q[0] * -0.72134752f), __builtin_amdgcn_exp2f(q[1] * -0.72134752f));   // exp(-u^2/2)
    const v2 hp = fma2(-poly, ex, splat2(0.5f));                                // 0.5 erf(|u| / sqrt 2), in [0, 0.5]
    h = mk2(0.5f + copysignf(hp[0], u[0]), 0.5f + copysignf(hp[1], u[1]));
}

// 32-bit integer hash (lowbias32, Wellons) -- the dropout keep-mask of these kernels: 128 bits per 8 elements from one
// full hash of the (row, chunk) counter and three xorshift-multiply steps; each element draws 16 bits, keep iff >= p * 65536.
// (The first-generation kernels of ffmid.hip keep Philox: 4 quarter-rate 32-bit multiplies per round were ~12 issue slots per element.)
__device__ __forceinline__ unsigned lowbias32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ void keep_words(unsigned long long seed, unsigned long long blk, unsigned w[4]) {
    const unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
    unsigned x = lowbias32(((unsigned)blk ^ k0) + 0x9E3779B9u * ((unsigned)(blk >> 32) ^ k1));
    x = lowbias32(x ^ k1);
    w[0] = x;
    x = (x ^ (x >> 15)) * 0x2c1b3c6du; w[1] = x ^ (x >> 13);
    x = (x ^ (x >> 12)) * 0x297a2d39u; w[2] = x ^ (x >> 15);
    x = (x ^ (x >> 14)) * 0x85ebca6bu; w[3] = x ^ (x >> 16);
}

#ifndef FS_R
#define FS_R 4            // rows per batch of the forward (one barrier per batch)
#endif
#ifndef FS_OCC
#define FS_OCC 1          // workgroups per CU the forward's register budget is sized for
#endif
#ifndef FS_SPLIT
#define FS_SPLIT 1        // 1: full batches run in a steady-state loop without row conditions (counted waits), the general body takes the strip's tail
#endif
#ifndef FS_DIAG
#define FS_DIAG 0         // diagnostic builds only (tools/ab_variant.sh): forward -- 1 / 2 / 4 = the keep-bit / h2 / normalised-output stores are skipped at run
#endif                    // time (values still computed), 8 = no GELU and no hash arithmetic
#ifndef FS_PACKW
#define FS_PACKW 0        // 1: the forward keeps the conv taps of 16-bit operands packed (see the kernel)
#endif
#ifndef FS_PIN
#define FS_PIN 1          // 1: the early re-requests of the steady-state loop are pinned behind their row's arithmetic
#endif
#ifndef FS_EARLY
#define FS_EARLY 1        // 1: a row's registers are re-requested (next batch) the moment sweep 1 has consumed them -- the loads are in flight
#endif                    //    through the rest of the sweep too, not only through the reduction / barrier / store phase (0: the round-2 order)

// ---------------------------------------------------------------------------------------------------------------------
// forward.  grid: B * strips workgroups of NT threads (NT = chunks of 8 channels rounded up to waves); strip s of sample b
// covers rows [s * RB, min(nseq, (s + 1) * RB)).
// ---------------------------------------------------------------------------------------------------------------------
// MX (with PL): h2 leaves as the half hi plane + fp8 planes + one scale byte per row (csrc/gemm_mx.hip) instead of hi / lo half planes.  The scale
// must be known before a row is written; a LayerNorm output is bounded whatever its input: |y_i| <= sqrt(F - 1) max|gamma / keep|.  Every row takes
// that bound (3-4 binades above a typical row's largest entry, of the 17 e4m3 spans: profiles/r06_error_budget_fp8corr.md, "one scale per tensor")
// -- no per-element max / min in a kernel that is bound by instruction issue (a data-derived row bound measured +54 us per launch).
template <typename T, int NT, bool TRAIN, bool PL = false, bool MX = false>
__global__ __launch_bounds__(NT, (FS_OCC * NT + 255) / 256) void ffmid2_fwd_kernel(const T* __restrict__ h1, const T* __restrict__ convw,
                                                        const T* __restrict__ gamma, T* __restrict__ h2,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        int nseq, int F, int Fp, int RB, int strips, float eps, float p,
                                                        unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                        unsigned char* __restrict__ drop_bits, T* __restrict__ gh_out, const FfPlanes pl) {
    constexpr int NW = NT / 64;
    constexpr int RB_ = (sizeof(T) == 2 && !PL) ? FS_R : (FS_R > 2 ? 2 : FS_R);        // rows per batch: fp32 rows (and hi + lo rows) cost twice the registers
    __shared__ float st[2][FS_R][NW][2];
    __shared__ float gmx[MX ? NW : 1];
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;
    const int b = blockIdx.x / strips, s = blockIdx.x - b * strips;
    const int t0 = s * RB, t1 = min(nseq, t0 + RB);
    if (t0 >= t1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // The launch has EXACTLY Fp / 8 threads (one per 8-channel chunk; the last wave may be partial): there is no thread without a chunk,
    // so no memory instruction of the row loop sits under a divergent branch and hipcc's s_waitcnt bookkeeping can count the stores that
    // follow a load instead of assuming none (see the steady-state loop below).  Lanes the hardware never started read as 0 in the
    // ds_bpermute steps of wave_sum.
#if OMLM_WAVE_DPP
    // wave_sum_dpp finishes with v_readlane of lanes 16 / 32 / 48, which ignores EXEC: in a PARTIAL last wave (Fp = 2752: 24 lanes) it would
    // read the never-written registers of absent lanes -- wrong LayerNorm statistics.  This forward needs the bpermute ladder.
#error "ffmid2_fwd_kernel launches a partial last wave: build it with the ds_bpermute wave_sum (OMLM_WAVE_DPP=0)"
#endif
    const int col = threadIdx.x * 8;
    const int ld = 2 * Fp;
    const size_t row0 = (size_t)b * nseq;
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);

    // taps / gamma: registers for the whole strip.  FS_PACKW (16-bit operands): the taps stay PACKED (24 registers instead of 48) and are
    // unpacked where they are used -- 2 VALU instructions per pair and use, ~12 % more issue, for a register budget that admits a
    // second workgroup per CU (6 + 6 waves = 3 per SIMD instead of the 2 / 2 / 1 / 1 of a lone 6-wave workgroup).
    constexpr bool PACKW = FS_PACKW && sizeof(T) == 2 && !PL;
    v2 wv[PACKW ? 1 : 3][4], wg[PACKW ? 1 : 3][4], gm[4];
    Row8<T, PL> tv[3], tg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        tv[k].load(convw + (size_t)k * ld + col, pl.convw_lo);
        tg[k].load(convw + (size_t)k * ld + Fp + col, pl.convw_lo);
        if constexpr (!PACKW) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { wv[k][i] = tv[k].get(i); wg[k][i] = tg[k].get(i); }
        }
    }
    auto WV = [&](int k, int i) -> v2 { if constexpr (PACKW) return tv[k].get(i); else return wv[k][i]; };
    auto WG = [&](int k, int i) -> v2 { if constexpr (PACKW) return tg[k].get(i); else return wg[k][i]; };
    {
        Row8<T, PL> a;
        a.load(gamma + col, pl.gamma_lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) gm[i] = a.get(i) * inv;                  // dropout scale folded into gamma
    }
    float gmmax = 0.f;                                   // MX: max |gamma / keep| over the row (pad channels carry 0)
    if constexpr (MX) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gmmax = fmaxf(gmmax, fmaxf(fabsf(gm[i][0]), fabsf(gm[i][1])));
        gmmax = wave_max(gmmax);                         // (absent lanes of a partial last wave read as 0: harmless for a max of magnitudes)
        if (lane == 0) gmx[wave] = gmmax;
        __syncthreads();
        gmmax = gmx[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) gmmax = fmaxf(gmmax, gmx[w]);
    }
    const int e_mx = MX ? mx_row_exp(sqrtf((float)F) * gmmax) : 0;
    const float sh_mx = ldexpf(1.0f, -e_mx), sl_mx = ldexpf(1.0f, 11 - e_mx);
    // conv window: rows t0 - 1 and t0 - 2 of the same sample (zero before the sample starts, transformer.py:129)
    v2 x1v[4], x1g[4], x2v[4], x2g[4];
    {
        H1Row<T, PL, MX> a, c, d, e;
        a.zero(); c.zero(); d.zero(); e.zero();
        if (t0 >= 1) { a.load(h1, (row0 + t0 - 1) * ld + col, pl); c.load(h1, (row0 + t0 - 1) * ld + Fp + col, pl); }
        if (t0 >= 2) { d.load(h1, (row0 + t0 - 2) * ld + col, pl); e.load(h1, (row0 + t0 - 2) * ld + Fp + col, pl); }
#pragma unroll
        for (int i = 0; i < 4; ++i) { x1v[i] = a.get(i); x1g[i] = c.get(i); x2v[i] = d.get(i); x2g[i] = e.get(i); }
    }
    H1Row<T, PL, MX> rv[RB_], rg[RB_];
#pragma unroll
    for (int r = 0; r < RB_; ++r)
        if (t0 + r < t1) {
            rv[r].load(h1, (row0 + t0 + r) * ld + col, pl);
            rg[r].load(h1, (row0 + t0 + r) * ld + Fp + col, pl);
        }
    const float invF = 1.0f / (float)F;

    // One batch of RB_ rows.  FULL: every row of this batch and of the next one exists -- no row condition is left, every load and store of
    // the body is issued on every path, and the waits hipcc places in front of a row's first use become counted (`vmcnt(n)` with the batch's
    // stores and the later rows' loads still in flight).  The general body (strip tails, short strips) keeps the conditions; with them the
    // fewest-operations path has nothing behind a load, so every batch ended in `vmcnt(0)`: a full drain of the 12 stores just issued, with
    // one workgroup per CU and nothing else to run (round 4: 46 % of the wave cycles parked, 3.3 TB/s).
    auto batch = [&](auto full_tag, const int tb, const int it) {
        constexpr bool FULL = decltype(full_tag)::value;
        v2 g[RB_][4];
        float ls[RB_], lq[RB_];
        // sweep 1: conv + GEGLU of the batch, per-thread sums for LayerNorm
#pragma unroll
        for (int r = 0; r < RB_; ++r) {
            ls[r] = 0.f; lq[r] = 0.f;
            if (FULL || tb + r < t1) {
                v2 s2 = splat2(0.f), q2 = splat2(0.f);
                if constexpr (PACKW) {                   // opaque per row: the unpacked taps must not be hoisted back into loop-invariant registers
#pragma unroll
                    for (int k = 0; k < 3; ++k) pin_regs(tv[k].h, tg[k].h);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const v2 xv = rv[r].get(i), xg = rg[r].get(i);
                    const v2 uv = fma2(WV(2, i), xv, fma2(WV(1, i), x1v[i], WV(0, i) * x2v[i]));
                    const v2 ug = fma2(WG(2, i), xg, fma2(WG(1, i), x1g[i], WG(0, i) * x2g[i]));
                    x2v[i] = x1v[i]; x1v[i] = xv; x2g[i] = x1g[i]; x1g[i] = xg;
                    v2 h, ex;
#if FS_DIAG & 8
                    h = ug; ex = ug;                     // diagnostic build: no GELU arithmetic
#else
                    gelu_parts(ug, h, ex);
#endif
                    const v2 gv = (ug * h) * uv;
                    g[r][i] = gv;
                    s2 += gv;
                    q2 = fma2(gv, gv, q2);
                }
                ls[r] = s2[0] + s2[1];
                lq[r] = q2[0] + q2[1];
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) g[r][i] = splat2(0.f);
            }
#if FS_EARLY
            if (FULL || tb + RB_ + r < t1) {             // this row's registers are free: the next batch's row r leaves now
#if FS_PIN
                if (FULL) __builtin_amdgcn_sched_barrier(0);     // (hipcc's scheduler otherwise sinks the four requests below the whole sweep)
#endif
                rv[r].load(h1, (row0 + tb + RB_ + r) * ld + col, pl);
                rg[r].load(h1, (row0 + tb + RB_ + r) * ld + Fp + col, pl);
#if FS_PIN
                if (FULL) __builtin_amdgcn_sched_barrier(0);
#endif
            }
#endif
        }
#if !FS_EARLY
        // next batch's rows: in flight during the reduction and the second sweep
#pragma unroll
        for (int r = 0; r < RB_; ++r)
            if (FULL || tb + RB_ + r < t1) {
                rv[r].load(h1, (row0 + tb + RB_ + r) * ld + col, pl);
                rg[r].load(h1, (row0 + tb + RB_ + r) * ld + Fp + col, pl);
            }
#endif
#pragma unroll
        for (int r = 0; r < RB_; ++r) { ls[r] = wave_sum(ls[r]); lq[r] = wave_sum(lq[r]); }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < RB_; ++r) { st[it & 1][r][wave][0] = ls[r]; st[it & 1][r][wave][1] = lq[r]; }
        }
        __syncthreads();          // one barrier per batch: the other parity's slots are rewritten only after the next one
        // sweep 2: normalise, gamma, dropout, store
        float mu_r[RB_], rs_r[RB_];
#pragma unroll
        for (int r = 0; r < RB_; ++r) {
            if (!FULL && tb + r >= t1) continue;
            float S = 0.f, Q = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { S += st[it & 1][r][w][0]; Q += st[it & 1][r][w][1]; }
            const float mu = S * invF;
            const float var = fmaxf(Q * invF - mu * mu, 0.f);
            const float rs = rsqrtf(var + eps);
            const size_t row = row0 + tb + r;
            const float sh = sh_mx, sl = sl_mx;
            if (FULL) { mu_r[r] = mu; rs_r[r] = rs; }
            else if (threadIdx.x == 0) { mean[row] = mu; rstd[row] = rs; if constexpr (MX) pl.scale8[row] = (unsigned char)(e_mx + 127); }
            const v2 nmr = splat2(-mu * rs), rs2 = splat2(rs);
            v2 gh[4], y[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) gh[i] = fma2(g[r][i], rs2, nmr);
            if (col + 8 > F) {                                   // the chunk holding the F boundary: pad channels carry 0
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (col + 2 * i >= F) gh[i][0] = 0.f;
                    if (col + 2 * i + 1 >= F) gh[i][1] = 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) y[i] = gh[i] * gm[i];
            if (TRAIN || p > 0.f) {
                unsigned w[4], bits = 0;
#if FS_DIAG & 8
                w[0] = w[1] = w[2] = w[3] = (unsigned)row * 0x9E3779B9u + (unsigned)col;     // diagnostic build: no hash
#else
                keep_words(seed, row * (unsigned long long)(Fp >> 3) + (unsigned)(col >> 3), w);
#endif
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool k0 = (w[i] & 0xFFFFu) >= thr, k1 = (w[i] >> 16) >= thr;
                    y[i][0] = k0 ? y[i][0] : 0.f;
                    y[i][1] = k1 ? y[i][1] : 0.f;
                    bits |= (k0 ? 1u : 0u) << (2 * i);
                    bits |= (k1 ? 1u : 0u) << (2 * i + 1);
                }
#if FS_DIAG
                asm volatile("" :: "v"(bits));           // diagnostic builds: the value stays computed although its store may be skipped
                if (!(FS_DIAG & 1) || eps < 0.f)
#endif
                if (TRAIN || drop_bits) drop_bits[row * (size_t)(Fp >> 3) + (col >> 3)] = (unsigned char)bits;
            }
#if FS_DIAG
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(y[i]), "v"(gh[i]));
            if (!(FS_DIAG & 2) || eps < 0.f) Ch8<T>::store(h2 + row * Fp + col, y);
            if (!(FS_DIAG & 4) || eps < 0.f) { if (TRAIN || gh_out) Ch8<T>::store(gh_out + row * Fp + col, gh); }
#else
            if constexpr (MX) {
                store_mx8(h2 + row * Fp + col, pl.h2_8 + row * (2 * (size_t)Fp) + col, pl.h2_8 + pl.h2_8_stride + row * (2 * (size_t)Fp) + col, y, sh, sl);
                // the row's tail up to a whole 128-byte fp8 k-tile is zero (omlm_gemm_mx16 reads it): written here, the planes need no zero fill
                if ((int)threadIdx.x < ((((Fp + 127) & ~127) - Fp) >> 3)) {
                    unsigned char* z8 = pl.h2_8 + row * (2 * (size_t)Fp) + Fp + 8 * threadIdx.x;
                    u32x2 z; z[0] = 0u; z[1] = 0u;
                    *(u32x2*)z8 = z;
                    *(u32x2*)(z8 + pl.h2_8_stride) = z;
                }
            }
            else if constexpr (PL) store_planes8(h2 + row * Fp + col, (T*)pl.h2_lo + row * Fp + col, y);
            else Ch8<T>::store(h2 + row * Fp + col, y);
            if (TRAIN || gh_out) Ch8<T>::store(gh_out + row * Fp + col, gh);
#endif
        }
        if (FULL) {
            // the batch's statistics: lane l of EVERY wave stores row (l mod RB_)'s pair -- the same values to the same addresses from
            // every wave, two store instructions per batch on every path (a `threadIdx.x == 0` region is a second, shorter path)
            const int q = lane % RB_;
            float m = mu_r[0], s_ = rs_r[0];
#pragma unroll
            for (int r = 1; r < RB_; ++r) { m = q == r ? mu_r[r] : m; s_ = q == r ? rs_r[r] : s_; }
            mean[row0 + tb + q] = m;
            rstd[row0 + tb + q] = s_;
            if constexpr (MX) pl.scale8[row0 + tb + q] = (unsigned char)(e_mx + 127);
        }
    };
    int it = 0, tb = t0;
#if FS_SPLIT
    // The loop header joins the entry edge with the back edge: with the prologue's (conditional) loads still pending on entry, the join
    // would again see "nothing behind this load" for every iteration.  Draining them once per strip leaves the back edge's counts.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                        // vmcnt(0)
#pragma unroll 1
    for (; tb + 2 * RB_ <= t1; tb += RB_, ++it) batch(std::true_type{}, tb, it);
#endif
#pragma unroll 1
    for (; tb < t1; tb += RB_, ++it) batch(std::false_type{}, tb, it);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward prepass: per row  bc[row] = (rstd * sum(gy) / F, rstd * sum(gy * gh) / F)  with gy = dropout^T(dh2) * gamma,
// and per-workgroup partial rows of d(gamma) = sum_rows dropout^T(dh2) * gh.  Wave per row, 8 channels per lane and step.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ffmid2_rowsum_kernel(const T* __restrict__ dh2, const T* __restrict__ gamma,
                                                            const T* __restrict__ ghs, const unsigned char* __restrict__ drop_bits,
                                                            const float* __restrict__ rstd, float* __restrict__ bc,
                                                            float* __restrict__ part_dgamma, int M, int F, int Fp, float p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const int nwaves = gridDim.x * 4;
    v2 dg[MAXC][4], gmv[MAXC][4];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int ch = (lane + 64 * k) * 8;
        Ch8<T> a;
        a.zero();
        if (ch < Fp) a.load(gamma + ch);
#pragma unroll
        for (int i = 0; i < 4; ++i) { dg[k][i] = splat2(0.f); gmv[k][i] = a.get(i); }
    }
    // Software-pipelined over the wave's rows without a second register set: chunk k of the NEXT row is requested the moment chunk k of
    // this row has been consumed, and the row's rstd travels with its data -- the reductions and the store of a row overlap the next row's
    // loads (round 3: every row paid two serial round trips, its data and then, behind the reductions, rstd[row]: 3.8 TB/s).
    Ch8<T> d[MAXC], gq[MAXC];
    unsigned bits[MAXC];
    float rs_next = 0.f;
    auto request = [&](int row, int k) {
        const int ch = (lane + 64 * k) * 8;
        if (ch < Fp) {
            d[k].load(dh2 + (size_t)row * Fp + ch);
            gq[k].load(ghs + (size_t)row * Fp + ch);
            bits[k] = (p > 0.f) ? drop_bits[(size_t)row * (Fp >> 3) + (ch >> 3)] : 0xFFu;
        }
    };
    {
        const int row0 = blockIdx.x * 4 + wave;
        if (row0 < M) {
#pragma unroll
            for (int k = 0; k < MAXC; ++k) request(row0, k);
            rs_next = rstd[row0];
        }
    }
    for (int row = blockIdx.x * 4 + wave; row < M; row += nwaves) {
        v2 s1 = splat2(0.f), s2 = splat2(0.f);
        const int nxt = row + nwaves;
        const float rs_cur = rs_next;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) {
            const int ch = (lane + 64 * k) * 8;
            if (ch < Fp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v2 dy = d[k].get(i) * inv;
                    if (!((bits[k] >> (2 * i)) & 1u)) dy[0] = 0.f;
                    if (!((bits[k] >> (2 * i + 1)) & 1u)) dy[1] = 0.f;
                    const v2 gh = gq[k].get(i);
                    dg[k][i] = fma2(dy, gh, dg[k][i]);                // pad columns: gh == 0 and gamma == 0
                    const v2 gy = dy * gmv[k][i];
                    s1 += gy;
                    s2 = fma2(gy, gh, s2);
                }
            }
            if (nxt < M) request(nxt, k);
        }
        if (nxt < M) rs_next = rstd[nxt];
        const float t1 = wave_sum(s1[0] + s1[1]), t2 = wave_sum(s2[0] + s2[1]);
        if (lane == 0) {
            const float rs = rs_cur / (float)F;
            bc[2 * (size_t)row] = rs * t1;
            bc[2 * (size_t)row + 1] = rs * t2;
        }
    }
    // the four waves' d(gamma) partials -> one partial row per workgroup
    extern __shared__ __attribute__((aligned(16))) float red[];     // [4][Fp]
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {
        const int ch = (lane + 64 * k) * 8;
        if (ch < Fp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { red[(size_t)wave * Fp + ch + 2 * i] = dg[k][i][0]; red[(size_t)wave * Fp + ch + 2 * i + 1] = dg[k][i][1]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Fp; c += 256)
        part_dgamma[(size_t)blockIdx.x * Fp + c] = red[c] + red[Fp + c] + red[2 * Fp + c] + red[3 * Fp + c];
}

// ---------------------------------------------------------------------------------------------------------------------
// backward main.  grid (column blocks of 256 threads x 4 channels, NY); workgroup y walks strips y, y + NY, ... keeping its
// d(conv taps) sums in registers, and leaves ONE partial row per y.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct Bwd2Row { Ch4<T> dy, gh, xv, xg; unsigned bits; float a, b, c; };

template <typename T>
__global__ __launch_bounds__(256) void ffmid2_bwd_kernel(const T* __restrict__ dh2, const T* __restrict__ h1,
                                                         const T* __restrict__ convw, const T* __restrict__ gamma,
                                                         const float* __restrict__ rstd, const float* __restrict__ bc,
                                                         const T* __restrict__ ghs, const unsigned char* __restrict__ drop_bits,
                                                         T* __restrict__ dh1, float* __restrict__ part_dconv,
                                                         int nseq, int F, int Fp, int RB, int strips, int total_strips, float p) {
    const int c4 = blockIdx.x * 256 + threadIdx.x;
    const int col = c4 * 4;
    const bool act = col < Fp;
    const int colc = act ? col : 0;
    const int ld = 2 * Fp;
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const int nib = (c4 & 1) * 4;

    v2 wv[3][2], wg[3][2], gm[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Ch4<T> a, c;
        a.load(convw + (size_t)k * ld + colc);
        c.load(convw + (size_t)k * ld + Fp + colc);
#pragma unroll
        for (int i = 0; i < 2; ++i) { wv[k][i] = act ? a.get(i) : splat2(0.f); wg[k][i] = act ? c.get(i) : splat2(0.f); }
    }
    {
        Ch4<T> a;
        a.load(gamma + colc);
#pragma unroll
        for (int i = 0; i < 2; ++i) gm[i] = act ? a.get(i) * inv : splat2(0.f);
    }
    v2 dcv[3][2], dcg[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) { dcv[k][i] = splat2(0.f); dcg[k][i] = splat2(0.f); }

    auto load_row = [&](Bwd2Row<T>& R, size_t row) {
        R.dy.load(dh2 + row * Fp + colc);
        R.gh.load(ghs + row * Fp + colc);
        R.xv.load(h1 + row * ld + colc);
        R.xg.load(h1 + row * ld + Fp + colc);
        R.bits = (p > 0.f) ? (unsigned)drop_bits[row * (size_t)(Fp >> 3) + (colc >> 3)] : 0xFFu;
        R.a = rstd[row]; R.b = bc[2 * row]; R.c = bc[2 * row + 1];
    };

#pragma unroll 1
    for (int strip = blockIdx.y; strip < total_strips; strip += gridDim.y) {
        const int b = strip / strips, s = strip - b * strips;
        const int t0 = s * RB, t1 = min(nseq, t0 + RB);
        if (t0 >= t1) continue;
        const size_t row0 = (size_t)b * nseq;
        v2 x1v[2], x1g[2], x2v[2], x2g[2], p1v[2], p1g[2], p2v[2], p2g[2];
        {
            Ch4<T> a, c, d, e;
            a.zero(); c.zero(); d.zero(); e.zero();
            if (t0 >= 1) { a.load(h1 + (row0 + t0 - 1) * ld + colc); c.load(h1 + (row0 + t0 - 1) * ld + Fp + colc); }
            if (t0 >= 2) { d.load(h1 + (row0 + t0 - 2) * ld + colc); e.load(h1 + (row0 + t0 - 2) * ld + Fp + colc); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                x1v[i] = a.get(i); x1g[i] = c.get(i); x2v[i] = d.get(i); x2g[i] = e.get(i);
                p1v[i] = splat2(0.f); p1g[i] = splat2(0.f); p2v[i] = splat2(0.f); p2g[i] = splat2(0.f);
            }
        }
        const int tend = t1 + 2;                      // rows t1, t1 + 1: du recomputed for the conv^T of this strip's last two rows
        Bwd2Row<T> cur, nxt;
        load_row(cur, row0 + t0);
#pragma unroll 1
        for (int t = t0; t < tend; ++t) {
            if (t + 1 < min(tend, nseq)) load_row(nxt, row0 + t + 1);
            v2 duv[2], dug[2];
            if (t < nseq) {
                const float own = t < t1 ? 1.f : 0.f;
                const unsigned kb = cur.bits >> nib;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    v2 gy = cur.dy.get(i) * gm[i];
                    if (!((kb >> (2 * i)) & 1u)) gy[0] = 0.f;
                    if (!((kb >> (2 * i + 1)) & 1u)) gy[1] = 0.f;
                    const v2 gh = cur.gh.get(i);
                    v2 dg = fma2(gy, splat2(cur.a), splat2(-cur.b));
                    dg = fma2(gh, splat2(-cur.c), dg);
                    const v2 xv = cur.xv.get(i), xg = cur.xg.get(i);
                    const v2 uv = fma2(wv[2][i], xv, fma2(wv[1][i], x1v[i], wv[0][i] * x2v[i]));
                    const v2 ug = fma2(wg[2][i], xg, fma2(wg[1][i], x1g[i], wg[0][i] * x2g[i]));
                    v2 h, ex;
                    gelu_parts(ug, h, ex);
                    const v2 gp = fma2(ug * 0.3989422804f, ex, h);          // GELU'(u) = Phi(u) + u phi(u)
                    const v2 a_ = dg * (ug * h);                             // d(value conv output)
                    const v2 g_ = (dg * uv) * gp;                            // d(gate conv output)
                    duv[i] = a_; dug[i] = g_;
                    const v2 ao = a_ * own, go = g_ * own;                   // d(taps): owned rows only
                    dcv[0][i] = fma2(ao, x2v[i], dcv[0][i]); dcv[1][i] = fma2(ao, x1v[i], dcv[1][i]); dcv[2][i] = fma2(ao, xv, dcv[2][i]);
                    dcg[0][i] = fma2(go, x2g[i], dcg[0][i]); dcg[1][i] = fma2(go, x1g[i], dcg[1][i]); dcg[2][i] = fma2(go, xg, dcg[2][i]);
                    x2v[i] = x1v[i]; x1v[i] = xv; x2g[i] = x1g[i]; x1g[i] = xg;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) { duv[i] = splat2(0.f); dug[i] = splat2(0.f); }
            }
            // conv^T: dh1[t-2] = w2 du[t-2] + w1 du[t-1] + w0 du[t] is complete now
            if (t - 2 >= t0 && act) {
                v2 ov[2], og[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ov[i] = fma2(wv[0][i], duv[i], p2v[i]);
                    og[i] = fma2(wg[0][i], dug[i], p2g[i]);
                }
                Ch4<T>::store(dh1 + (row0 + t - 2) * ld + col, ov);
                Ch4<T>::store(dh1 + (row0 + t - 2) * ld + Fp + col, og);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                p2v[i] = fma2(wv[1][i], duv[i], p1v[i]); p1v[i] = wv[2][i] * duv[i];
                p2g[i] = fma2(wg[1][i], dug[i], p1g[i]); p1g[i] = wg[2][i] * dug[i];
            }
            cur = nxt;
        }
    }
    // partial row of d(conv taps): layout [2F real channels][3] like the reference weight [2F, 1, 3]
    if (act) {
        float* pr = part_dconv + (size_t)blockIdx.y * 2 * F * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = col + 2 * i + e;
                if (ch < F) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { pr[(size_t)ch * 3 + k] = dcv[k][i][e]; pr[(size_t)(F + ch) * 3 + k] = dcg[k][i][e]; }
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward in ONE pass (Fp <= 4 NT channels).  The prepass above exists because LayerNorm^T needs two sums over a WHOLE row before the
// column-local kernel can touch it -- and it reads dh2, gh and the keep bits (400 MB per coarse-small layer, ~90 us) that the main kernel
// reads again.  Here a workgroup of NT threads covers ALL channels of its rows (4 per thread), so the row sums cross the waves through
// LDS: a row's partial sums are taken one row AHEAD of its main pass (its data is in registers by then: rows are requested two ahead),
// published behind the one barrier of the iteration, and read at the top of the next one -- the reduction's latency sits under the
// main pass of the previous row.  d(gamma) = sum_rows dropout^T(dh2) gh joins d(conv taps) in registers: one partial row per workgroup
// of each.  Same element arithmetic and the same summation order inside a row as prepass + main (wave sums, then the waves in order).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef FF3_DPP
#define FF3_DPP 1
#endif
template <typename T> struct Bwd3Row { Ch4<T> dy, gh, xv, xg; unsigned bits; float a; };
// "defined here" for the optimiser: what is derived from the registers after this point is not hoisted above it
__device__ __forceinline__ void opaque(Ch4<h16_t>& c) { asm volatile("" : "+v"(c.r[0]), "+v"(c.r[1])); }
__device__ __forceinline__ void opaque(Ch4<float>&) {}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void ffmid3_bwd_kernel(const T* __restrict__ dh2, const T* __restrict__ h1,
                                                        const T* __restrict__ convw, const T* __restrict__ gamma,
                                                        const float* __restrict__ rstd, const T* __restrict__ ghs,
                                                        const unsigned char* __restrict__ drop_bits, T* __restrict__ dh1,
                                                        float* __restrict__ part_dconv, float* __restrict__ part_dgamma,
                                                        int nseq, int F, int Fp, int RB, int strips, int total_strips, float p) {
    constexpr int NW = NT / 64;
    __shared__ __attribute__((aligned(16))) float red[2][NW][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = threadIdx.x * 4;
    const bool act = col < Fp;
    const int colc = act ? col : 0;
    const int ld = 2 * Fp;
    const float inv = p > 0.f ? 1.0f / (1.0f - p) : 1.0f, invF = 1.0f / (float)F;
    const int nib = (threadIdx.x & 1) * 4;

    // taps stay in the operand type (converted where used): as fp32 pairs they were 24 of the 168 registers three waves per SIMD allow
    Ch4<T> wvr[3], wgr[3];
    v2 gm[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        wvr[k].zero(); wgr[k].zero();
        if (act) { wvr[k].load(convw + (size_t)k * ld + colc); wgr[k].load(convw + (size_t)k * ld + Fp + colc); }
    }
    {
        Ch4<T> a;
        a.load(gamma + colc);
#pragma unroll
        for (int i = 0; i < 2; ++i) gm[i] = act ? a.get(i) : splat2(0.f);
    }
    v2 dcv[3][2], dcg[3][2], dgam[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dgam[i] = splat2(0.f);
#pragma unroll
        for (int k = 0; k < 3; ++k) { dcv[k][i] = splat2(0.f); dcg[k][i] = splat2(0.f); }
    }

    auto load_row = [&](Bwd3Row<T>& R, size_t row) {
        R.dy.load(dh2 + row * Fp + colc);
        R.gh.load(ghs + row * Fp + colc);
        R.xv.load(h1 + row * ld + colc);
        R.xg.load(h1 + row * ld + Fp + colc);
        R.bits = (p > 0.f) ? (unsigned)drop_bits[row * (size_t)(Fp >> 3) + (colc >> 3)] : 0xFFu;
        R.a = rstd[row];
    };
    // dropout^T of the row's dh2 (inactive threads: gamma == 0 keeps them out of every sum)
    auto dy_kept = [&](const Bwd3Row<T>& R, v2 (&dy)[2]) {
        const unsigned kb = R.bits >> nib;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dy[i] = R.dy.get(i) * inv;
            if (!((kb >> (2 * i)) & 1u)) dy[i][0] = 0.f;
            if (!((kb >> (2 * i + 1)) & 1u)) dy[i][1] = 0.f;
        }
    };
    // this thread's part of the two LayerNorm^T sums of a row -> the wave's -> LDS slot `buf`
    auto publish_sums = [&](const Bwd3Row<T>& R, int buf) {
        v2 dy[2], s1 = splat2(0.f), s2 = splat2(0.f);
        dy_kept(R, dy);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const v2 gy = dy[i] * gm[i];
            s1 += gy;
            s2 = fma2(gy, R.gh.get(i), s2);
        }
#if FF3_DPP
        // (every lane of the NT-thread workgroup exists -- inactive channels carry zeros --, so the VALU form is safe here: four DPP adds + four
        // readlanes per sum instead of six dependent LDS-crossbar round trips on the row's critical path)
        const float t1 = wave_sum_dpp(s1[0] + s1[1]), t2 = wave_sum_dpp(s2[0] + s2[1]);
#else
        const float t1 = wave_sum(s1[0] + s1[1]), t2 = wave_sum(s2[0] + s2[1]);
#endif
        if (lane == 0) { red[buf][wave][0] = t1; red[buf][wave][1] = t2; }
    };

#pragma unroll 1
    for (int strip = blockIdx.y; strip < total_strips; strip += gridDim.y) {
        const int b = strip / strips, s = strip - b * strips;
        const int t0 = s * RB, t1 = min(nseq, t0 + RB);
        if (t0 >= t1) continue;                                  // (uniform)
        const size_t row0 = (size_t)b * nseq;
        const int tlast = min(t1 + 2, nseq);                     // rows [t0, tlast) are read; du of rows t1, t1 + 1 is recomputed for the conv^T of the strip's last two rows
        v2 x1v[2], x1g[2], x2v[2], x2g[2], p1v[2], p1g[2], p2v[2], p2g[2];
        {
            Ch4<T> a, c, d, e;
            a.zero(); c.zero(); d.zero(); e.zero();
            if (t0 >= 1) { a.load(h1 + (row0 + t0 - 1) * ld + colc); c.load(h1 + (row0 + t0 - 1) * ld + Fp + colc); }
            if (t0 >= 2) { d.load(h1 + (row0 + t0 - 2) * ld + colc); e.load(h1 + (row0 + t0 - 2) * ld + Fp + colc); }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                x1v[i] = a.get(i); x1g[i] = c.get(i); x2v[i] = d.get(i); x2g[i] = e.get(i);
                p1v[i] = splat2(0.f); p1g[i] = splat2(0.f); p2v[i] = splat2(0.f); p2g[i] = splat2(0.f);
            }
        }
        Bwd3Row<T> cur, nx1, nx2;
        load_row(cur, row0 + t0);
        nx1 = cur;
        if (t0 + 1 < tlast) load_row(nx1, row0 + t0 + 1);
        __syncthreads();                                         // the previous strip's last reads of `red` are done
        publish_sums(cur, t0 & 1);
        __syncthreads();
        const int tend = t1 + 2;
        // (measured and dropped, round 6: the loop unrolled by three with rotating register roles instead of `cur = nx1; nx1 = nx2` -- 20 fewer
        // moves per row, 162 registers, no spills, and 375 us against 324: hipcc's schedule of the tripled body loses more than the moves cost)
#pragma unroll 1
        for (int t = t0; t < tend; ++t) {
            nx2 = nx1;
            if (t + 2 < tlast) load_row(nx2, row0 + t + 2);

            if (t + 1 < tlast) publish_sums(nx1, (t + 1) & 1);   // next row's sums: visible behind this iteration's barrier
            v2 duv[2], dug[2];
            if (t < nseq) {
                v2 S12 = splat2(0.f);                                    // (S1, S2) as a register pair: one packed add per wave's partials, same order
#pragma unroll
                for (int w = 0; w < NW; ++w) S12 += *(const v2*)&red[t & 1][w][0];
                const float S1 = S12[0], S2 = S12[1];
                const float rs = cur.a * invF;
                const float bsum = rs * S1, csum = rs * S2;
                const float own = t < t1 ? 1.f : 0.f;
                v2 dy[2];
                dy_kept(cur, dy);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v2 gh = cur.gh.get(i);
                    const v2 gy = dy[i] * gm[i];
                    dgam[i] = fma2(dy[i] * own, gh, dgam[i]);                 // pad columns: gh == 0
                    v2 dg = fma2(gy, splat2(cur.a), splat2(-bsum));
                    dg = fma2(gh, splat2(-csum), dg);
                    const v2 xv = cur.xv.get(i), xg = cur.xg.get(i);
                    const v2 uv = fma2(wvr[2].get(i), xv, fma2(wvr[1].get(i), x1v[i], wvr[0].get(i) * x2v[i]));
                    const v2 ug = fma2(wgr[2].get(i), xg, fma2(wgr[1].get(i), x1g[i], wgr[0].get(i) * x2g[i]));
                    v2 h, ex;
                    gelu_parts(ug, h, ex);
                    const v2 gp = fma2(ug * 0.3989422804f, ex, h);          // GELU'(u) = Phi(u) + u phi(u)
                    const v2 a_ = dg * (ug * h);                             // d(value conv output)
                    const v2 g_ = (dg * uv) * gp;                            // d(gate conv output)
                    duv[i] = a_; dug[i] = g_;
                    const v2 ao = a_ * own, go = g_ * own;                   // d(taps): owned rows only
                    dcv[0][i] = fma2(ao, x2v[i], dcv[0][i]); dcv[1][i] = fma2(ao, x1v[i], dcv[1][i]); dcv[2][i] = fma2(ao, xv, dcv[2][i]);
                    dcg[0][i] = fma2(go, x2g[i], dcg[0][i]); dcg[1][i] = fma2(go, x1g[i], dcg[1][i]); dcg[2][i] = fma2(go, xg, dcg[2][i]);
                    x2v[i] = x1v[i]; x1v[i] = xv; x2g[i] = x1g[i]; x1g[i] = xg;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) { duv[i] = splat2(0.f); dug[i] = splat2(0.f); }
            }
            // conv^T: dh1[t-2] = w2 du[t-2] + w1 du[t-1] + w0 du[t] is complete now
            if (t - 2 >= t0 && act) {
                v2 ov[2], og[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ov[i] = fma2(wvr[0].get(i), duv[i], p2v[i]);
                    og[i] = fma2(wgr[0].get(i), dug[i], p2g[i]);
                }
                Ch4<T>::store(dh1 + (row0 + t - 2) * ld + col, ov);
                Ch4<T>::store(dh1 + (row0 + t - 2) * ld + Fp + col, og);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                p2v[i] = fma2(wvr[1].get(i), duv[i], p1v[i]); p1v[i] = wvr[2].get(i) * duv[i];
                p2g[i] = fma2(wgr[1].get(i), dug[i], p1g[i]); p1g[i] = wgr[2].get(i) * dug[i];
            }
            cur = nx1; nx1 = nx2;
            __syncthreads();
        }
    }
    // partial rows: d(conv taps) in the layout [2F real channels][3] of the reference weight [2F, 1, 3]; d(gamma) [Fp]
    if (act) {
        float* pr = part_dconv + (size_t)blockIdx.y * 2 * F * 3;
        float* pg = part_dgamma + (size_t)blockIdx.y * Fp;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = col + 2 * i + e;
                pg[ch] = dgam[i][e];
                if (ch < F) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { pr[(size_t)ch * 3 + k] = dcv[k][i][e]; pr[(size_t)(F + ch) * 3 + k] = dcg[k][i][e]; }
                }
            }
    }
}

// ---- host side (called by the C-ABI entry points of ffmid.hip) ----------------------------------------------------------
static int strip_rows(int nseq, int target) {
    const int n = (nseq + target - 1) / target;
    return (nseq + n - 1) / n;
}

bool ffmid2_supported(int Fp) { return Fp % 8 == 0 && Fp / 8 <= 512; }

template <typename T, bool PL = false, bool MX = false>
static int fwd_launch_t(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd, int M, int nseq,
                        int F, int Fp, float eps, float p, unsigned long long seed, const unsigned long long* seed_dev,
                        unsigned char* drop_bits, void* gh, hipStream_t st, const FfPlanes pl = FfPlanes{0, 0, 0, nullptr, nullptr, 0, nullptr, nullptr}) {
    const int B = M / nseq;
    const int RB = strip_rows(nseq, 36);
    const int strips = (nseq + RB - 1) / RB;
    const int nt = ((Fp / 8 + 63) / 64) * 64;                                  // kernel instantiation: whole waves
    const int nthr = Fp / 8;                                                   // threads launched: one per chunk
    dim3 grid(B * strips);
    const bool train = p > 0.f && drop_bits != nullptr && gh != nullptr;       // the training call: all three row stores are unconditional
#define FF2_FWD(NT_) do { if (train) hipLaunchKernelGGL((ffmid2_fwd_kernel<T, NT_, true, PL, MX>), grid, dim3(nthr), 0, st, (const T*)h1, (const T*)convw, \
        (const T*)gamma, (T*)h2, mean, rstd, nseq, F, Fp, RB, strips, eps, p, seed, seed_dev, drop_bits, (T*)gh, pl); \
    else hipLaunchKernelGGL((ffmid2_fwd_kernel<T, NT_, false, PL, MX>), grid, dim3(nthr), 0, st, (const T*)h1, (const T*)convw, \
        (const T*)gamma, (T*)h2, mean, rstd, nseq, F, Fp, RB, strips, eps, p, seed, seed_dev, drop_bits, (T*)gh, pl); } while (0)
    switch (nt) {
        case 64: FF2_FWD(64); break;
        case 128: FF2_FWD(128); break;
        case 192: FF2_FWD(192); break;
        case 256: FF2_FWD(256); break;
        case 320: FF2_FWD(320); break;
        case 384: FF2_FWD(384); break;
        case 448: FF2_FWD(448); break;
        default: FF2_FWD(512); break;
    }
#undef FF2_FWD
    return omlm_post_launch("omlm_ffmid_fwd (strip)");
}

// dtype: 0 = fp32 operands ("bf16x3" mode), 1 = bf16
int ffmid2_fwd_launch(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd, int M, int nseq,
                      int F, int Fp, float eps, float p, unsigned long long seed, const unsigned long long* seed_dev,
                      unsigned char* drop_bits, void* gh, int dtype, hipStream_t st) {
#if !OMLM_FP16
    if (dtype == 0) return fwd_launch_t<float>(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, st);
#endif
    return fwd_launch_t<h16_t>(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, st);
}

// the same forward on hi/lo planes of the 16-bit type (omlm_ffmid_fwd_planes): h1, taps and gamma are read as hi + lo, h2 leaves as planes
int ffmid2_fwd_planes_launch(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                             void* h2, void* h2_lo, float* mean, float* rstd, int M, int nseq, int F, int Fp, float eps, float p,
                             unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, hipStream_t st) {
    FfPlanes pl;
    pl.h1_lo = (const h16_t*)h1_lo - (const h16_t*)h1;
    pl.convw_lo = (const h16_t*)convw_lo - (const h16_t*)convw;
    pl.gamma_lo = (const h16_t*)gamma_lo - (const h16_t*)gamma;
    pl.h2_lo = h2_lo;
    pl.h2_8 = nullptr; pl.h2_8_stride = 0; pl.scale8 = nullptr; pl.h1_8 = nullptr;
    return fwd_launch_t<h16_t, true>(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, st, pl);
}

#if OMLM_FP16
// the plane forward with h2 leaving in omlm_gemm_mx16's operand form: half hi plane + fp8 planes [hi8 | lo8] (row pitch 2 Fp bytes) + row scales;
// h1's lo plane arrives as bf8 bytes (h1_lo8, at h1's element pitch: what omlm_gemm_mx16 writes with c_lo_bf8)
int ffmid2_fwd_mx_launch(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                         void* h2, void* h2_8, long long h2_8_stride, unsigned char* scale8, float* mean, float* rstd, int M, int nseq, int F, int Fp,
                         float eps, float p, unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, hipStream_t st) {
    FfPlanes pl;
    pl.h1_lo = 0;
    pl.h1_8 = (const unsigned char*)h1_lo;
    pl.convw_lo = (const h16_t*)convw_lo - (const h16_t*)convw;
    pl.gamma_lo = (const h16_t*)gamma_lo - (const h16_t*)gamma;
    pl.h2_lo = nullptr;
    pl.h2_8 = (unsigned char*)h2_8; pl.h2_8_stride = h2_8_stride; pl.scale8 = scale8;
    return fwd_launch_t<h16_t, true, true>(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, st, pl);
}
#endif

// bc: [M][2] floats of scratch; part_g: [>= rowsum blocks][Fp]; part_c: [>= NY][2F*3]
template <typename T>
static int bwd_launch_t(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* rstd, float* bc,
                        void* dh1, float* part_g, int max_g_rows, float* part_c, int max_c_rows, int* g_rows, int* c_rows,
                        int M, int nseq, int F, int Fp, float p, const unsigned char* drop_bits, const void* gh, hipStream_t st) {
    const int B = M / nseq;
    static int fused = -1;
    if (fused < 0) { const char* e = getenv("OMLM_FFMID_BWD_FUSED"); fused = (e && e[0] == '0') ? 0 : 1; }
    if (fused && Fp / 4 <= 768) {
        // one pass: a workgroup covers all channels of its rows (ffmid3_bwd_kernel); `bc` is not used
        const int RB = strip_rows(nseq, 35);
        const int strips = (nseq + RB - 1) / RB;
        const int total = B * strips;
        const int per = (total + 255) / 256;                 // strips per workgroup: one workgroup of 8-12 waves per CU
        int ny = (total + per - 1) / per;
        if (ny > max_c_rows) ny = max_c_rows;
        if (ny > max_g_rows) ny = max_g_rows;
        const int nthr = (Fp / 4 + 63) / 64 * 64;
#define FF3_BWD(NT_) hipLaunchKernelGGL((ffmid3_bwd_kernel<T, NT_>), dim3(1, ny), dim3(NT_), 0, st, (const T*)dh2, (const T*)h1, (const T*)convw, \
                       (const T*)gamma, rstd, (const T*)gh, drop_bits, (T*)dh1, part_c, part_g, nseq, F, Fp, RB, strips, total, p)
        if (nthr <= 128) FF3_BWD(128); else if (nthr <= 256) FF3_BWD(256); else if (nthr <= 512) FF3_BWD(512); else FF3_BWD(768);
#undef FF3_BWD
        *g_rows = ny;
        *c_rows = ny;
        return omlm_post_launch("omlm_ffmid_bwd (fused strip)");
    }
    // prepass
    const int rows4 = (M + 3) / 4;
    const int b1 = rows4 < max_g_rows ? rows4 : max_g_rows;
    const size_t lds = (size_t)4 * Fp * sizeof(float);
    const int mc = (Fp / 8 + 63) / 64;
#define FF2_RS(MC_) do { if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)ffmid2_rowsum_kernel<T, MC_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((ffmid2_rowsum_kernel<T, MC_>), dim3(b1), dim3(256), lds, st, (const T*)dh2, (const T*)gamma, (const T*)gh, \
                           drop_bits, rstd, bc, part_g, M, F, Fp, p); } while (0)
    if (mc <= 1) FF2_RS(1); else if (mc <= 2) FF2_RS(2); else if (mc <= 4) FF2_RS(4); else if (mc <= 6) FF2_RS(6); else FF2_RS(8);
#undef FF2_RS
    int rc = omlm_post_launch("omlm_ffmid_bwd (row sums)");
    if (rc) return rc;
    // main
    const int RB = strip_rows(nseq, 35);
    const int strips = (nseq + RB - 1) / RB;
    const int total = B * strips;
    int per = (total + 255) / 256;                       // strips per workgroup-y: ~3 workgroups of 4 waves per CU
    int ny = (total + per - 1) / per;
    if (ny > max_c_rows) ny = max_c_rows;
    dim3 grid((Fp / 4 + 255) / 256, ny);
    hipLaunchKernelGGL(ffmid2_bwd_kernel<T>, grid, dim3(256), 0, st, (const T*)dh2, (const T*)h1, (const T*)convw,
                       (const T*)gamma, rstd, (const float*)bc, (const T*)gh, drop_bits, (T*)dh1, part_c,
                       nseq, F, Fp, RB, strips, total, p);
    *g_rows = b1;
    *c_rows = ny;
    return omlm_post_launch("omlm_ffmid_bwd (strip)");
}

int ffmid2_bwd_launch(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* rstd, float* bc,
                      void* dh1, float* part_g, int max_g_rows, float* part_c, int max_c_rows, int* g_rows, int* c_rows,
                      int M, int nseq, int F, int Fp, float p, const unsigned char* drop_bits, const void* gh, int dtype, hipStream_t st) {
#if !OMLM_FP16
    if (dtype == 0)
        return bwd_launch_t<float>(dh2, h1, convw, gamma, rstd, bc, dh1, part_g, max_g_rows, part_c, max_c_rows, g_rows, c_rows,
                                   M, nseq, F, Fp, p, drop_bits, gh, st);
#endif
    return bwd_launch_t<h16_t>(dh2, h1, convw, gamma, rstd, bc, dh1, part_g, max_g_rows, part_c, max_c_rows, g_rows, c_rows,
                                M, nseq, F, Fp, p, drop_bits, gh, st);
}

}   // namespace OMLM_NS
