// Fused middle of ConvFeedForward (reference transformer.py:122-150):
//   h1 [M, 2*Fp]  --causal depthwise conv k=3-->  u  --GEGLU (erf GELU on the gate half)-->  g
//      --LayerNorm over the F real channels-->  --Dropout(p)-->  h2 [M, Fp]
// h1 is the output of the FF-in GEMM in a padded layout: value half in columns [0, F), gate half in
// columns [Fp, Fp + F), Fp = F rounded up to 8 so both halves are 16-byte aligned; pad columns are 0.
// Rows are tokens (b * n_seq + t); the conv is causal *within a sample* (left pad 2, :129).
// With use_conv_ff=False (plain FeedForward :152-161) the host passes conv weights (0, 0, 1).
// Conv taps and gamma arrive re-packed by the host (ops.pack_conv_taps / padded gamma): taps TAP-MAJOR and padded,
// convT[3][2*Fp] with the same column layout as h1 (pad columns 0), gamma_p[Fp] (pad 0), so that a lane's 8 channels
// are two aligned float4 loads per tap instead of 24 scattered dwords.
//
// HBM-bound: forward reads h1 once (the two halo rows come from L2) and writes h2; nothing but the
// per-row LN statistics is saved -- the backward recomputes u and g from h1.
#include "common.h"
#include <stdlib.h>

namespace OMLM_NS {

#define FF_THREADS 256
#define FF_MAXC_LIMIT 16   // 8-channel chunks per lane (wave-per-row kernels) -> Fp <= 8192

template <typename T> struct vec8;
template <> struct vec8<float> {
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void store(float* p) const {
        ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        ((float4*)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct vec8<h16_t> {
    float v[8];
    __device__ __forceinline__ void load(const h16_t* p) {
        const u32x4 a = *(const u32x4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo_to_f(a[i]); v[2 * i + 1] = h16_hi_to_f(a[i]); }
    }
    __device__ __forceinline__ void store(h16_t* p) const {
        u32x4 a;
        a[0] = pack_h16_rne(v[0], v[1]); a[1] = pack_h16_rne(v[2], v[3]);
        a[2] = pack_h16_rne(v[4], v[5]); a[3] = pack_h16_rne(v[6], v[7]);
        *(u32x4*)p = a;
    }
};
__device__ __forceinline__ void zero8(float* v) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 0.f;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. fp32 rounding level for GELU) -- ~12 VALU + one v_exp
// instead of libm erff's ~60: these kernels were VALU-bound on it.
// v_rcp_f32 (1 ulp) instead of __frcp_rn / '/', which hipcc expands to the full IEEE division sequence (v_div_scale x2, v_rcp,
// 4 fma, v_div_fmas, v_div_fixup: 164 of the 958 static VALU instructions of the forward, found in the ISA).  The erf argument
// is >= 1 and the change is below the 1.5e-7 approximation error: the kernel parity tests report the same errors to all
// printed digits with either build; forward 328 -> 304 us at the bench shape.  FF_FAST_RCP=0 restores the exact quotients.
#ifndef FF_FAST_RCP
#define FF_FAST_RCP 1
#endif
__device__ __forceinline__ float ff_rcp(float x) {
#if FF_FAST_RCP
    return __builtin_amdgcn_rcpf(x);
#else
    return __frcp_rn(x);
#endif
}
__device__ __forceinline__ float ff_div(float a, float b) {       // same switch for the Welford quotients
#if FF_FAST_RCP
    return a * __builtin_amdgcn_rcpf(b);
#else
    return a / b;
#endif
}
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = ff_rcp(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    return copysignf(1.0f - poly * __expf(-ax * ax), x);
}
// same, also handing back exp(-x^2) of its argument: for x = u / sqrt(2) that is the exp(-u^2 / 2) of the GELU derivative
__device__ __forceinline__ float fast_erf_exp(float x, float& ex) {
    const float ax = fabsf(x);
    const float t = ff_rcp(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    ex = __expf(-ax * ax);
    return copysignf(1.0f - poly * ex, x);
}
__device__ __forceinline__ float gelu_from_erf(float x, float e) { return 0.5f * x * (1.0f + e); }
__device__ __forceinline__ float gelu_grad_from_erf_exp(float x, float e, float ex) {
    return 0.5f * (1.0f + e) + x * 0.3989422804014327f * ex;
}
__device__ __forceinline__ float gelu_grad_from_erf(float x, float e) {
    return 0.5f * (1.0f + e) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float gelu_f(float x) { return gelu_from_erf(x, fast_erf(x * 0.70710678118654752f)); }

// y[t] = w0 x[t-2] + w1 x[t-1] + w2 x[t]; taps from convT[3][ld] at the same column as the data
template <typename T>
__device__ __forceinline__ void conv_row(const T* __restrict__ h1, const T* __restrict__ convT, size_t row, int t, int ld,
                                         int col, float* u) {
    vec8<T> c0, c1, c2;
    vec8<T> w0, w1, w2;      // taps / gamma travel in the operand dtype: they are row-invariant, and as fp32 they were 70 % of
                             // the L2->L1 bytes of these kernels (which run at the L2 bandwidth, not at HBM's)
    c2.load(h1 + row * ld + col);
    if (t >= 1) c1.load(h1 + (row - 1) * ld + col); else zero8(c1.v);
    if (t >= 2) c0.load(h1 + (row - 2) * ld + col); else zero8(c0.v);
    w0.load(convT + col); w1.load(convT + ld + col); w2.load(convT + 2 * (size_t)ld + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) u[i] = w0.v[i] * c0.v[i] + w1.v[i] * c1.v[i] + w2.v[i] * c2.v[i];
}

// Packed 8-element loads kept un-converted, so that the NEXT chunk's operands can sit in registers (16 B each in bf16)
// while the current chunk is computed: each wave walks its row chunk by chunk, and with only 2-3 waves per SIMD a chunk's
// load latency (~1.7 us under load) was serialised with its ~1.7 us of arithmetic.
template <typename T> struct raw8;
template <> struct raw8<h16_t> {
    u32x4 r;
    __device__ __forceinline__ void load(const h16_t* p) { r = *(const u32x4*)p; }
    __device__ __forceinline__ void zero() { r[0] = 0u; r[1] = 0u; r[2] = 0u; r[3] = 0u; }
    __device__ __forceinline__ void unpack(float* v) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = h16_lo_to_f(r[i]); v[2 * i + 1] = h16_hi_to_f(r[i]); }
    }
};
template <> struct raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void unpack(float* v) const {
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
// operands of y[t] = w0 x[t-2] + w1 x[t-1] + w2 x[t] for 8 channels
template <typename T> struct ConvIn {
    raw8<T> c0, c1, c2, w0, w1, w2;
    __device__ __forceinline__ void load(const T* __restrict__ h1, const T* __restrict__ convT, size_t row, int t, int ld, int col) {
        c2.load(h1 + row * ld + col);
        if (t >= 1) c1.load(h1 + (row - 1) * ld + col); else c1.zero();
        if (t >= 2) c0.load(h1 + (row - 2) * ld + col); else c0.zero();
        w0.load(convT + col); w1.load(convT + ld + col); w2.load(convT + 2 * (size_t)ld + col);
    }
    __device__ __forceinline__ void eval(float* u) const {
        float a0[8], a1[8], a2[8], b0[8], b1[8], b2[8];
        c0.unpack(a0); c1.unpack(a1); c2.unpack(a2); w0.unpack(b0); w1.unpack(b1); w2.unpack(b2);
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = b0[i] * a0[i] + b1[i] * a1[i] + b2[i] * a2[i];
    }
};

// everything one chunk of the backward row sweep reads: conv operands of both halves, dh2, gamma, the keep-mask byte
template <typename T> struct Bwd1In {
    ConvIn<T> x, g;
    raw8<T> d, gm;
    unsigned bits;
    __device__ __forceinline__ void load(const T* __restrict__ h1, const T* __restrict__ convw, const T* __restrict__ dh2,
                                         const T* __restrict__ gamma, const unsigned char* __restrict__ drop_bits,
                                         size_t row, int t, int ld, int Fp, int ch) {
        x.load(h1, convw, row, t, ld, ch);
        g.load(h1, convw, row, t, ld, Fp + ch);
        d.load(dh2 + row * Fp + ch);
        gm.load(gamma + ch);
        bits = drop_bits ? drop_bits[row * (Fp >> 3) + (ch >> 3)] : 0xFFu;
    }
};

// what the first backward sweep reads when the forward saved the normalised GEGLU output: dh2, gamma, gh, the keep-mask byte
template <typename T> struct Bwd1Lite {
    raw8<T> d, gm, gh;
    unsigned bits;
    __device__ __forceinline__ void load(const T* __restrict__ dh2, const T* __restrict__ gamma, const T* __restrict__ ghs,
                                         const unsigned char* __restrict__ drop_bits, size_t row, int Fp, int ch) {
        d.load(dh2 + row * Fp + ch);
        gh.load(ghs + row * Fp + ch);
        gm.load(gamma + ch);
        bits = drop_bits ? drop_bits[row * (Fp >> 3) + (ch >> 3)] : 0xFFu;
    }
};

// keep-mask * 1/(1-p) for 8 consecutive elements starting at element index e0 (multiple of 8).  ONE Philox-4x32-10 call per
// 8 elements: each element gets a 16-bit draw (keep iff draw >= p * 65536).  The first version spent a 24-bit draw per
// element = two calls per chunk, ~140 of the ~560 VALU instructions a row-chunk of the forward costs (350 -> 328 us).
__device__ __forceinline__ void dropout8(unsigned long long seed, unsigned long long e0, float p, float* m) {
    const float inv = 1.0f / (1.0f - p);
    const unsigned thr = (unsigned)(p * 65536.0f + 0.5f);
    unsigned o[4];
    const unsigned long long blk = e0 >> 3;
    // 5 of Philox's 10 rounds: a dropout keep-mask needs decorrelated, well-mixed bits per element, not a Crush-resistant stream
    // (7 rounds already pass BigCrush; each round is ~14 of the forward's VALU instructions per 8 elements)
    philox4x32<5>((unsigned)blk, (unsigned)(blk >> 32), 0u, 0u, (unsigned)seed, (unsigned)(seed >> 32), o);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m[2 * i] = (o[i] & 0xFFFFu) >= thr ? inv : 0.f;
        m[2 * i + 1] = (o[i] >> 16) >= thr ? inv : 0.f;
    }
}

// keep-mask of 8 consecutive elements from the byte the forward kernel stored (bit i = element i kept)
__device__ __forceinline__ void dropout8_from_bits(unsigned bits, float p, float* m) {
    const float inv = 1.0f / (1.0f - p);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = ((bits >> i) & 1u) ? inv : 0.f;
}

// (A strip formulation -- a thread owns 8 channels for a strip of rows, taps and the conv halo in registers, h1 read once,
// R rows requested together, LayerNorm statistics exchanged through LDS -- was built and measured at 400 us against this
// kernel's 328 us, with or without a prefetched next batch: with the memory side out of the way the kernel is no faster,
// i.e. what is left is instruction issue (~560 VALU per 8-channel chunk), and the strip form pays 4x the shuffles per row.
// profiles/r01_ffmid_strip_ab.md; the code is in history at 36014b0.)
// One WAVE per row: the row's F channels are spread over the 64 lanes in 8-channel chunks, the two LayerNorm
// reductions are wave shuffles (no LDS, no barrier), and a 256-thread workgroup keeps 4 independent rows in
// flight -- at ~4 waves/SIMD that is ~16 rows per CU hiding HBM latency, instead of one row per workgroup
// serialised behind two block-wide reductions.
// (n, mean, M2) merge of two partial statistics (Chan et al.); n == 0 partials are neutral
__device__ __forceinline__ void welford_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
    const float nt = n + nb;
    if (nt > 0.f) {
        const float d = meanb - mean, r = ff_div(nb, nt);   // (FF_FAST_RCP: a 1-ulp error in r enters mean / M2 at the 1e-7 relative level)
        mean += d * r;
        m2 += m2b + d * d * n * r;
    }
    n = nt;
}

template <typename T, int MAXC>
__global__ __launch_bounds__(FF_THREADS, 2) void ffmid_fwd_kernel(const T* __restrict__ h1, const T* __restrict__ convw,
                                                                  const T* __restrict__ gamma, T* __restrict__ h2,
                                                                  float* __restrict__ mean, float* __restrict__ rstd,
                                                                  int M, int nseq, int F, int Fp, float eps, float p,
                                                                  unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                                  unsigned char* __restrict__ drop_bits, T* __restrict__ gh_out) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;      // per-step salt from device memory (graph replays differ)
    extern __shared__ __attribute__((aligned(16))) float ff_lds[];     // [4 waves][Fp]: this wave's g row between the sweeps
    const int lane = threadIdx.x & 63;
    float* gl = ff_lds + (size_t)(threadIdx.x >> 6) * Fp;
    const int ld = 2 * Fp;
    const int nwaves = gridDim.x * (FF_THREADS / 64);
    for (int row = blockIdx.x * (FF_THREADS / 64) + (threadIdx.x >> 6); row < M; row += nwaves) {
        const int t = row % nseq;
        // sweep 1: LayerNorm statistics of g = gelu(gate) * value over the F real channels (single pass, Welford)
        float wn = 0.f, wmean = 0.f, wm2 = 0.f;
        constexpr bool PF = sizeof(T) == 2;                 // operand prefetch one chunk ahead (register budget: bf16 only)
        ConvIn<T> xin, gin, xnx, gnx;
        if (PF && lane * 8 < F) { xin.load(h1, convw, row, t, ld, lane * 8); gin.load(h1, convw, row, t, ld, Fp + lane * 8); }
#pragma unroll 1
        for (int k = 0; k < MAXC; ++k) {
            const int ch = (lane + 64 * k) * 8;
            if (PF) {
                const int chn = ch + 512;
                if (k + 1 < MAXC && chn < F) { xnx.load(h1, convw, row, t, ld, chn); gnx.load(h1, convw, row, t, ld, Fp + chn); }
            } else if (ch < F) {
                xin.load(h1, convw, row, t, ld, ch); gin.load(h1, convw, row, t, ld, Fp + ch);
            }
            if (ch < F) {
                float ux[8], ug[8], gv[8];
                xin.eval(ux);
                gin.eval(ug);
                const int nv = min(8, F - ch);
                float cs = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) { gv[i] = gelu_f(ug[i]) * ux[i]; if (i < nv) cs += gv[i]; }
                ((float4*)(gl + ch))[0] = make_float4(gv[0], gv[1], gv[2], gv[3]);
                ((float4*)(gl + ch))[1] = make_float4(gv[4], gv[5], gv[6], gv[7]);
                const float cm = ff_div(cs, (float)nv);
                float c2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (i < nv) { const float d = gv[i] - cm; c2 += d * d; }
                welford_merge(wn, wmean, wm2, (float)nv, cm, c2);
            }
            if (PF) { xin = xnx; gin = gnx; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float nb = __shfl_xor(wn, o, 64), mb = __shfl_xor(wmean, o, 64), qb = __shfl_xor(wm2, o, 64);
            welford_merge(wn, wmean, wm2, nb, mb, qb);
        }
        const float mu = wmean;
        const float rs = rsqrtf(wm2 / (float)F + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        // sweep 2: g comes back from this wave's LDS row (same-wave LDS ops are ordered); normalise, dropout, store
#pragma unroll 1
        for (int k = 0; k < MAXC; ++k) {
            const int ch = (lane + 64 * k) * 8;
            if (ch < Fp) {
                float gv[8], m[8];
                if (ch < F) {
                    const float4 a = ((const float4*)(gl + ch))[0], b = ((const float4*)(gl + ch))[1];
                    gv[0] = a.x; gv[1] = a.y; gv[2] = a.z; gv[3] = a.w; gv[4] = b.x; gv[5] = b.y; gv[6] = b.z; gv[7] = b.w;
                } else {
                    zero8(gv);
                }
                if (p > 0.f) {
                    dropout8(seed, (unsigned long long)row * Fp + ch, p, m);
                    if (drop_bits) {                               // 1 bit per element for the backward (Philox there was ~40 % of its VALU work)
                        unsigned bits = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) bits |= (m[i] != 0.f ? 1u : 0u) << i;
                        drop_bits[(size_t)row * (Fp >> 3) + (ch >> 3)] = (unsigned char)bits;
                    }
                }
                vec8<T> o, gh;
                vec8<T> gm;
                gm.load(gamma + ch);                               // padded gamma: 0 beyond F
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    gh.v[i] = ch + i < F ? (gv[i] - mu) * rs : 0.f;
                    float y = gh.v[i] * gm.v[i];
                    if (p > 0.f) y *= m[i];
                    o.v[i] = y;
                }
                o.store(h2 + (size_t)row * Fp + ch);
                // the normalised GEGLU output, for the backward: its first sweep (the two LayerNorm^T sums and d(gamma)) then needs
                // neither the conv nor the erf again -- these kernels are VALU-bound with HBM to spare
                if (gh_out) gh.store(gh_out + (size_t)row * Fp + ch);
            }
        }
    }
}

// Backward, stage 1 (row-local): dh2 -> dropout^T -> LayerNorm^T -> GEGLU^T  => du [M, 2*Fp].
// Wave per row, two sweeps over the row's chunks (sums first, outputs second; the second sweep re-reads the three
// h1 rows and dh2 from L1/L2) so that nothing but the two reduction scalars lives across the sweep boundary.
// dgamma is accumulated in an LDS array per workgroup (ds_add_f32) and written once as a partial row.
template <typename T, int MAXC, bool GH>
__global__ __launch_bounds__(FF_THREADS, 2) void ffmid_bwd1_kernel(const T* __restrict__ dh2, const T* __restrict__ h1,
                                                                const T* __restrict__ convw, const T* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                T* __restrict__ du, float* __restrict__ part_dgamma,
                                                                int M, int nseq, int F, int Fp, float p, unsigned long long seed,
                                                                const unsigned long long* __restrict__ seed_dev,
                                                                const unsigned char* __restrict__ drop_bits, const T* __restrict__ ghs) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B97F4A7C15ull;
    // [4 waves][Fp]: each wave's PRIVATE d(gamma) partial, updated with plain vector read-add-write (a lane always owns
    // the same channels, so there is nothing to arbitrate).  The first version used one shared array with ds_add_f32 per
    // element: LDS float atomics turned out to run at ~1 lane per clock (they cost 930 us per layer in the attention
    // backward), i.e. ~0.5 ms of this kernel.  erf(gate) is recomputed in the second sweep instead of being cached, which
    // keeps the LDS footprint at 4*Fp floats (3 workgroups per CU).
    extern __shared__ __attribute__((aligned(16))) float dg_lds[];
    for (int c = threadIdx.x; c < 4 * Fp; c += FF_THREADS) dg_lds[c] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float* dgw = dg_lds + (size_t)(threadIdx.x >> 6) * Fp;
    const int ld = 2 * Fp;
    const int nwaves = gridDim.x * (FF_THREADS / 64);
    for (int row = blockIdx.x * (FF_THREADS / 64) + (threadIdx.x >> 6); row < M; row += nwaves) {
        const int t = row % nseq;
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        constexpr bool PF = sizeof(T) == 2;                 // operand prefetch one chunk ahead (register budget: bf16 only)
        Bwd1In<T> cur, nxt;
        if (GH) {
            // sweep 1 from the saved normalised output gh: s1 = sum dy gamma, s2 = sum dy gamma gh, d(gamma) += dy gh
            Bwd1Lite<T> lc, ln;
            if (lane * 8 < Fp) lc.load(dh2, gamma, ghs, drop_bits, (size_t)row, Fp, lane * 8);
#pragma unroll 1
            for (int k = 0; k < MAXC; ++k) {
                const int ch = (lane + 64 * k) * 8;
                if (k + 1 < MAXC && ch + 512 < Fp) ln.load(dh2, gamma, ghs, drop_bits, (size_t)row, Fp, ch + 512);
                if (ch < Fp) {
                    float dv[8], gmv[8], ghv[8], m[8];
                    lc.d.unpack(dv);
                    lc.gm.unpack(gmv);
                    lc.gh.unpack(ghv);
                    if (p > 0.f) {
                        if (drop_bits) dropout8_from_bits(lc.bits, p, m);
                        else dropout8(seed, (unsigned long long)row * Fp + ch, p, m);
                    }
                    float dgv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {               // pad columns: gamma == 0 and gh == 0 there
                        float dyv = dv[i];
                        if (p > 0.f) dyv *= m[i];
                        dgv[i] = dyv * ghv[i];
                        const float gy = dyv * gmv[i];
                        s1 += gy;
                        s2 += gy * ghv[i];
                    }
                    float4 a = ((float4*)(dgw + ch))[0], b = ((float4*)(dgw + ch))[1];
                    a.x += dgv[0]; a.y += dgv[1]; a.z += dgv[2]; a.w += dgv[3];
                    b.x += dgv[4]; b.y += dgv[5]; b.z += dgv[6]; b.w += dgv[7];
                    ((float4*)(dgw + ch))[0] = a;
                    ((float4*)(dgw + ch))[1] = b;
                }
                lc = ln;
            }
        } else {
        if (PF && lane * 8 < Fp) cur.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, lane * 8);
#pragma unroll 1
        for (int k = 0; k < MAXC; ++k) {
            const int ch = (lane + 64 * k) * 8;
            if (PF) { if (k + 1 < MAXC && ch + 512 < Fp) nxt.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, ch + 512); }
            else if (ch < Fp) cur.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, ch);
            if (ch < Fp) {
                float ux[8], ug[8], m[8];
                cur.x.eval(ux);
                cur.g.eval(ug);
                vec8<T> d, gm;
                cur.d.unpack(d.v);
                cur.gm.unpack(gm.v);
                if (p > 0.f) {
                    if (drop_bits) dropout8_from_bits(cur.bits, p, m);
                    else dropout8(seed, (unsigned long long)row * Fp + ch, p, m);
                }
                float dgv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    dgv[i] = 0.f;
                    if (ch + i < F) {
                        const float gh = (gelu_f(ug[i]) * ux[i] - mu) * rs;
                        float dyv = d.v[i];
                        if (p > 0.f) dyv *= m[i];
                        dgv[i] = dyv * gh;
                        const float gy = dyv * gm.v[i];
                        s1 += gy;
                        s2 += gy * gh;
                    }
                }
                float4 a = ((float4*)(dgw + ch))[0], b = ((float4*)(dgw + ch))[1];
                a.x += dgv[0]; a.y += dgv[1]; a.z += dgv[2]; a.w += dgv[3];
                b.x += dgv[4]; b.y += dgv[5]; b.z += dgv[6]; b.w += dgv[7];
                ((float4*)(dgw + ch))[0] = a;
                ((float4*)(dgw + ch))[1] = b;
            }
            if (PF) cur = nxt;
        }
        }
        const float m1 = wave_sum(s1) / (float)F;
        const float m2 = wave_sum(s2) / (float)F;
        if (PF && lane * 8 < Fp) cur.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, lane * 8);
#pragma unroll 1
        for (int k = 0; k < MAXC; ++k) {
            const int ch = (lane + 64 * k) * 8;
            if (PF) { if (k + 1 < MAXC && ch + 512 < Fp) nxt.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, ch + 512); }
            else if (ch < Fp) cur.load(h1, convw, dh2, gamma, drop_bits, (size_t)row, t, ld, Fp, ch);
            if (ch < Fp) {
                float ux[8], ug[8], m[8];
                cur.x.eval(ux);
                cur.g.eval(ug);
                vec8<T> d, ox, og, gm;
                cur.d.unpack(d.v);
                cur.gm.unpack(gm.v);
                if (p > 0.f) {
                    if (drop_bits) dropout8_from_bits(cur.bits, p, m);
                    else dropout8(seed, (unsigned long long)row * Fp + ch, p, m);
                }
                float ev[8], ex[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) ev[i] = fast_erf_exp(ug[i] * 0.70710678118654752f, ex[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float dx = 0.f, dgt = 0.f;
                    if (ch + i < F) {
                        const float ge = gelu_from_erf(ug[i], ev[i]);
                        const float gh = (ge * ux[i] - mu) * rs;
                        float dyv = d.v[i];
                        if (p > 0.f) dyv *= m[i];
                        const float dg = rs * (dyv * gm.v[i] - m1 - gh * m2);
                        dx = dg * ge;
                        dgt = dg * ux[i] * gelu_grad_from_erf_exp(ug[i], ev[i], ex[i]);
                    }
                    ox.v[i] = dx;
                    og.v[i] = dgt;
                }
                ox.store(du + (size_t)row * ld + ch);
                og.store(du + (size_t)row * ld + Fp + ch);
            }
            if (PF) cur = nxt;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < Fp; c += FF_THREADS)
        part_dgamma[(size_t)blockIdx.x * Fp + c] = dg_lds[c] + dg_lds[Fp + c] + dg_lds[2 * Fp + c] + dg_lds[3 * Fp + c];
}

// Backward, stage 2 (conv^T): dh1[t] = w2 du[t] + w1 du[t+1] + w0 du[t+2] (within the sample), and
// per-block partials of dconv_w[ch][k] = sum_t du[t, ch] * h1[t-2+k, ch].
// One thread owns 8 channels of the padded 2*Fp layout for a strip of rows.
template <typename T>
__global__ __launch_bounds__(FF_THREADS) void ffmid_bwd2_kernel(const T* __restrict__ du, const T* __restrict__ h1,
                                                                const T* __restrict__ convw, T* __restrict__ dh1,
                                                                float* __restrict__ part_dconv, int M, int nseq, int F, int Fp) {
    const int ld = 2 * Fp, nchunk = ld / 8;
    const int strips = gridDim.y;
    const int chunk = blockIdx.x * FF_THREADS + threadIdx.x;
    if (chunk >= nchunk) return;
    const int col = chunk * 8;
    const bool gate = col >= Fp;
    const int chreal = gate ? col - Fp : col;          // channel index inside its half
    const int wch = gate ? F + chreal : chreal;        // row of the reference conv weight [2F, 1, 3]
    float w[8][3], dw[8][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vec8<T> wv;
        wv.load(convw + (size_t)k * ld + col);         // tap-major padded taps (0 in pad columns)
#pragma unroll
        for (int i = 0; i < 8; ++i) { w[i][k] = wv.v[i]; dw[i][k] = 0.f; }
    }
    const int rows_per = (M + strips - 1) / strips;
    const int r_begin = blockIdx.y * rows_per, r_end = min(M, r_begin + rows_per);
    // rolling window over the strip: du rows (row, row+1, row+2) and h1 rows (row-2, row-1, row) are carried in registers, so
    // each step loads ONE new du row and ONE new h1 row instead of three of each (sequence boundaries are applied as masks)
    vec8<T> da, db, dc, xa, xb, xc;
    zero8(da.v); zero8(db.v); zero8(dc.v); zero8(xa.v); zero8(xb.v); zero8(xc.v);
    if (r_begin < r_end) {
        db.load(du + (size_t)r_begin * ld + col);
        if (r_begin + 1 < M) dc.load(du + (size_t)(r_begin + 1) * ld + col);
        if (r_begin >= 1) xc.load(h1 + (size_t)(r_begin - 1) * ld + col);
        if (r_begin >= 2) xb.load(h1 + (size_t)(r_begin - 2) * ld + col);
    }
    for (int row = r_begin; row < r_end; ++row) {
        const int t = row % nseq;
        da = db; db = dc;                                            // du[row], du[row + 1]
        if (row + 2 < M) dc.load(du + (size_t)(row + 2) * ld + col); else zero8(dc.v);
        xa = xb; xb = xc;                                            // h1[row - 2], h1[row - 1]
        xc.load(h1 + (size_t)row * ld + col);
        const float m1 = t + 1 < nseq ? 1.f : 0.f, m2 = t + 2 < nseq ? 1.f : 0.f;
        const float p1 = t >= 1 ? 1.f : 0.f, p2 = t >= 2 ? 1.f : 0.f;
        vec8<T> o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            o.v[i] = w[i][2] * da.v[i] + w[i][1] * (m1 * db.v[i]) + w[i][0] * (m2 * dc.v[i]);
            dw[i][0] += da.v[i] * (p2 * xa.v[i]);
            dw[i][1] += da.v[i] * (p1 * xb.v[i]);
            dw[i][2] += da.v[i] * xc.v[i];
        }
        o.store(dh1 + (size_t)row * ld + col);
    }
    // partials layout: [strip][2F real channels][3]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (chreal + i < F) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                part_dconv[((size_t)blockIdx.y * 2 * F + wch + i) * 3 + k] = dw[i][k];
        }
    }
}

// out[c] += sum_p part[p * ldp + c],  c < C
__global__ void colsum_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int C, int ldp) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int per = (P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(P, p0 + per);
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += part[(size_t)p * ldp + c];
    if (p1 > p0) unsafeAtomicAdd(out + c, s);
}

// several column sums as ONE launch (round 5): the LayerNorm backward's d(gamma) partial rows of a whole backward pass (13 launches of
// ~8 us each in the coarse-small step) and the two of every ConvFeedForward backward.  blockIdx.z = problem.
struct omlm_colsum_desc { const float* part; float* out; int P, C, ldp; };      // include/omlm.h
#define OMLM_COLSUM_MAX 32
struct ColsumGroupArgs { int n; omlm_colsum_desc p[OMLM_COLSUM_MAX]; };
__global__ void colsum_group_kernel(ColsumGroupArgs ga) {
    const omlm_colsum_desc& q = ga.p[blockIdx.z];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= q.C) return;
    const int per = (q.P + gridDim.y - 1) / gridDim.y;
    const int p0 = blockIdx.y * per, p1 = min(q.P, p0 + per);
    float s = 0.f;
    for (int p = p0; p < p1; ++p) s += q.part[(size_t)p * q.ldp + c];
    if (p1 > p0) unsafeAtomicAdd(q.out + c, s);
}

#if OMLM_FP16
extern "C" int omlm_colsum_accumulate(const float* part, float* out, int P, int C, int ldp, void* stream);     // bf16 copy
extern "C" int omlm_colsum_group(const omlm_colsum_desc* d, int count, void* stream);
#else
extern "C" int omlm_colsum_group(const omlm_colsum_desc* d, int count, void* stream) {
    if (count <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(d, "colsum_group: null descriptor array");
    for (int base = 0; base < count; base += OMLM_COLSUM_MAX) {
        ColsumGroupArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.n = count - base < OMLM_COLSUM_MAX ? count - base : OMLM_COLSUM_MAX;
        int cmax = 0, pmax = 0;
        for (int i = 0; i < ga.n; ++i) {
            const omlm_colsum_desc& q = d[base + i];
            OMLM_CHECK_ARG(q.part && q.out && q.P > 0 && q.C > 0 && q.ldp >= q.C, "colsum_group: bad problem");
            ga.p[i] = q;
            if (q.C > cmax) cmax = q.C;
            if (q.P > pmax) pmax = q.P;
        }
        const int ysplit = pmax >= 1024 ? 128 : (pmax >= 64 ? 16 : 1);
        hipLaunchKernelGGL(colsum_group_kernel, dim3((cmax + 255) / 256, ysplit, ga.n), dim3(256), 0, as_stream(stream), ga);
    }
    return omlm_post_launch("omlm_colsum_group");
}

extern "C" int omlm_colsum_accumulate(const float* part, float* out, int P, int C, int ldp, void* stream) {
    if (P <= 0 || C <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(part && out && ldp >= C, "colsum arguments");
    const int ysplit = P >= 1024 ? 128 : (P >= 64 ? 16 : 1);      // enough workgroups to stream a [2048, D] partial buffer
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, ysplit), dim3(256), 0, as_stream(stream), part, out, P, C, ldp);
    return omlm_post_launch("omlm_colsum_accumulate");
}

#endif

#define FF_BWD1_BLOCKS 1024
#define FF_BWD2_STRIPS 512

// second-generation kernels for bf16 operands (ffmid2.hip: column strips, fused backward)
bool ffmid2_supported(int Fp);
int ffmid2_fwd_launch(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd, int M, int nseq,
                      int F, int Fp, float eps, float p, unsigned long long seed, const unsigned long long* seed_dev,
                      unsigned char* drop_bits, void* gh, int dtype, hipStream_t st);
int ffmid2_bwd_launch(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* rstd, float* bc,
                      void* dh1, float* part_g, int max_g_rows, float* part_c, int max_c_rows, int* g_rows, int* c_rows,
                      int M, int nseq, int F, int Fp, float p, const unsigned char* drop_bits, const void* gh, int dtype, hipStream_t st);
// 1 (default): the column-strip kernels (both operand dtypes) where their preconditions hold; 0: the wave-per-row kernels.  $OMLM_FFMID_IMPL or
// omlm_ffmid_set_impl (A/B runs, tests).  The two generations share every buffer layout; their dropout streams differ
// (the keep-mask travels from forward to backward as drop_bits, so a step may not mix them only when drop_bits is null).
static int g_ffmid_impl = -1;
#if !OMLM_FP16
extern "C" int omlm_ffmid_set_impl_h(int impl);
#endif
extern "C" int OMLM_API(omlm_ffmid_set_impl)(int impl) {
    OMLM_CHECK_ARG(impl == 0 || impl == 1, "impl: 0 = wave-per-row, 1 = column strips");
    g_ffmid_impl = impl;
#if !OMLM_FP16
    return omlm_ffmid_set_impl_h(impl);         // the fp16 copy of this file keeps its own switch
#else
    return OMLM_OK;
#endif
}
static int ffmid_impl() {
    if (g_ffmid_impl < 0) { const char* e = getenv("OMLM_FFMID_IMPL"); g_ffmid_impl = (e && e[0] == '0') ? 0 : 1; }
    return g_ffmid_impl;
}

#if !OMLM_FP16
extern "C" long long omlm_ffmid_bwd_workspace_bytes(int F, int Fp) {
    return (long long)sizeof(float) * ((long long)FF_BWD1_BLOCKS * Fp + (long long)FF_BWD2_STRIPS * 2 * F * 3);
}
#endif

// dtype: 0 = fp32 operands ("bf16x3"), 1 = bf16, 2 = fp16 (forwarded to the fp16 copy of this file)
#if !OMLM_FP16
extern "C" int omlm_ffmid_fwd_h(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd,
                                int M, int nseq, int F, int Fp, float eps, float p, unsigned long long seed,
                                const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, int dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_ffmid_fwd)(const void* h1, const void* convw, const void* gamma, void* h2, float* mean, float* rstd,
                              int M, int nseq, int F, int Fp, float eps, float p, unsigned long long seed,
                              const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, int dtype, void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16) return omlm_ffmid_fwd_h(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, 1, stream);
#else
    OMLM_CHECK_ARG(dtype == 1, "the fp16 copy serves fp16 operands only");
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(h1 && convw && gamma && h2 && mean && rstd, "null pointer");
    OMLM_CHECK_ARG(Fp % 8 == 0 && Fp >= F && Fp <= 8192 && (size_t)5 * Fp * sizeof(float) <= 160 * 1024, "Fp must be F rounded up to 8 and <= 8192");
    OMLM_CHECK_ARG(nseq > 0 && M % nseq == 0, "M must be batch * nseq");
    OMLM_CHECK_ARG(p >= 0.f && p < 1.f, "dropout p");
    hipStream_t st = as_stream(stream);
    if (ffmid_impl() == 1 && ffmid2_supported(Fp) && (p == 0.f || drop_bits))
        return ffmid2_fwd_launch(h1, convw, gamma, h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, dtype, st);
    const int rows4 = (M + 3) / 4;
    dim3 grid(rows4 < 4096 ? rows4 : 4096), block(FF_THREADS);
    const size_t lds_fwd = (size_t)4 * Fp * sizeof(float);
    if (lds_fwd > 48 * 1024) {
#if !OMLM_FP16
        (void)hipFuncSetAttribute((const void*)ffmid_fwd_kernel<float, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)ffmid_fwd_kernel<float, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        (void)hipFuncSetAttribute((const void*)ffmid_fwd_kernel<h16_t, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)ffmid_fwd_kernel<h16_t, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
#define FF_FWD(T_, MC_) hipLaunchKernelGGL((ffmid_fwd_kernel<T_, MC_>), grid, block, lds_fwd, st, (const T_*)h1, (const T_*)convw, (const T_*)gamma, (T_*)h2, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, (T_*)gh)
#define FF_FWD_DISPATCH(T_) do { const int mc = (Fp / 8 + 63) / 64; \
        if (mc <= 2) FF_FWD(T_, 2); else if (mc <= 6) FF_FWD(T_, 6); else if (mc <= 8) FF_FWD(T_, 8); else FF_FWD(T_, 16); } while (0)
#if !OMLM_FP16
    if (dtype == 0) FF_FWD_DISPATCH(float); else
#endif
    FF_FWD_DISPATCH(h16_t);
    return omlm_post_launch("omlm_ffmid_fwd");
}

// The same forward on hi/lo planes of the 16-bit type `dtype` (1 = bf16, 2 = fp16; precision "fp16ff"): h1, the conv taps and gamma are read
// as hi + lo (each *_lo plane has its hi plane's layout), h2 leaves as planes h2 = rne16(y), h2_lo = rne16(y - h2); gh, the statistics and
// the keep bits as in omlm_ffmid_fwd.  The strip kernels only (Fp <= 4096, drop_bits present when p > 0).
int ffmid2_fwd_planes_launch(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                             void* h2, void* h2_lo, float* mean, float* rstd, int M, int nseq, int F, int Fp, float eps, float p,
                             unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, hipStream_t st);
#if !OMLM_FP16
extern "C" int omlm_ffmid_fwd_planes_h(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                                       void* h2, void* h2_lo, float* mean, float* rstd, int M, int nseq, int F, int Fp, float eps, float p,
                                       unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, int dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_ffmid_fwd_planes)(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma,
                                               const void* gamma_lo, void* h2, void* h2_lo, float* mean, float* rstd, int M, int nseq, int F, int Fp,
                                               float eps, float p, unsigned long long seed, const unsigned long long* seed_dev,
                                               unsigned char* drop_bits, void* gh, int dtype, void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16)
        return omlm_ffmid_fwd_planes_h(h1, h1_lo, convw, convw_lo, gamma, gamma_lo, h2, h2_lo, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev, drop_bits, gh, 1, stream);
#endif
    OMLM_CHECK_ARG(dtype == 1, "ffmid_fwd_planes: dtype 1 (bf16) or 2 (fp16)");
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(h1 && h1_lo && convw && convw_lo && gamma && gamma_lo && h2 && h2_lo && mean && rstd, "null pointer");
    OMLM_CHECK_ARG(Fp % 8 == 0 && Fp >= F && ffmid2_supported(Fp), "ffmid_fwd_planes: Fp must be F rounded up to 8 and <= 4096");
    OMLM_CHECK_ARG(nseq > 0 && M % nseq == 0, "M must be batch * nseq");
    OMLM_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || drop_bits), "dropout p (keep bits required when p > 0)");
    OMLM_CHECK_ARG((((uintptr_t)h1 | (uintptr_t)h1_lo | (uintptr_t)convw | (uintptr_t)convw_lo | (uintptr_t)gamma | (uintptr_t)gamma_lo |
                     (uintptr_t)h2 | (uintptr_t)h2_lo) % 16) == 0, "ffmid_fwd_planes: 16-byte aligned planes");
    return ffmid2_fwd_planes_launch(h1, h1_lo, convw, convw_lo, gamma, gamma_lo, h2, h2_lo, mean, rstd, M, nseq, F, Fp, eps, p, seed, seed_dev,
                                    drop_bits, gh, as_stream(stream));
}

#if OMLM_FP16
// The plane forward with h2 leaving in omlm_gemm_mx16's operand form (round 6): the half hi plane h2 [M, Fp], fp8 planes [hi8 | lo8] at row pitch
// 2 Fp bytes (the lo8 plane h2_8_stride bytes behind the hi8 plane) and one E8M0 scale byte per row; everything else as omlm_ffmid_fwd_planes.
int ffmid2_fwd_mx_launch(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                         void* h2, void* h2_8, long long h2_8_stride, unsigned char* scale8, float* mean, float* rstd, int M, int nseq, int F, int Fp,
                         float eps, float p, unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh, hipStream_t st);
}   // namespace OMLM_NS
extern "C" int omlm_ffmid_fwd_mx(const void* h1, const void* h1_lo, const void* convw, const void* convw_lo, const void* gamma, const void* gamma_lo,
                                 void* h2, void* h2_8, long long h2_8_stride, unsigned char* scale8, float* mean, float* rstd, int M, int nseq, int F, int Fp,
                                 float eps, float p, unsigned long long seed, const unsigned long long* seed_dev, unsigned char* drop_bits, void* gh,
                                 void* stream) {
    using namespace OMLM_NS;
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(h1 && h1_lo && convw && convw_lo && gamma && gamma_lo && h2 && h2_8 && scale8 && mean && rstd, "null pointer");
    OMLM_CHECK_ARG(Fp % 8 == 0 && Fp >= F && ffmid2_supported(Fp), "ffmid_fwd_mx: Fp must be F rounded up to 8 and <= 4096");
    OMLM_CHECK_ARG(nseq > 0 && M % nseq == 0, "M must be batch * nseq");
    OMLM_CHECK_ARG(p >= 0.f && p < 1.f && (p == 0.f || drop_bits), "dropout p (keep bits required when p > 0)");
    OMLM_CHECK_ARG(h2_8_stride >= (long long)M * 2 * Fp && h2_8_stride % 16 == 0, "ffmid_fwd_mx: fp8 plane stride");
    OMLM_CHECK_ARG((((uintptr_t)h1 | (uintptr_t)h1_lo | (uintptr_t)convw | (uintptr_t)convw_lo | (uintptr_t)gamma | (uintptr_t)gamma_lo |
                     (uintptr_t)h2 | (uintptr_t)h2_8) % 16) == 0, "ffmid_fwd_mx: 16-byte aligned planes");
    return ffmid2_fwd_mx_launch(h1, h1_lo, convw, convw_lo, gamma, gamma_lo, h2, h2_8, h2_8_stride, scale8, mean, rstd, M, nseq, F, Fp, eps, p, seed,
                                seed_dev, drop_bits, gh, as_stream(stream));
}
namespace OMLM_NS {
#endif

// du_tmp: [M, 2*Fp] scratch of the operand dtype; dh1: [M, 2*Fp] output; workspace: omlm_ffmid_bwd_workspace_bytes.
// dgamma [F], dconv [2F*3] are accumulated into (+=).
#if !OMLM_FP16
extern "C" int omlm_ffmid_bwd_h(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* mean,
                                const float* rstd, void* du_tmp, void* dh1, float* dgamma, float* dconv, float* workspace,
                                int M, int nseq, int F, int Fp, float p, unsigned long long seed,
                                const unsigned long long* seed_dev, const unsigned char* drop_bits, const void* gh, int dtype,
                                void* stream);
#endif
extern "C" int OMLM_API(omlm_ffmid_bwd)(const void* dh2, const void* h1, const void* convw, const void* gamma, const float* mean,
                              const float* rstd, void* du_tmp, void* dh1, float* dgamma, float* dconv, float* workspace,
                              int M, int nseq, int F, int Fp, float p, unsigned long long seed,
                              const unsigned long long* seed_dev, const unsigned char* drop_bits, const void* gh, int dtype,
                              void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16)
        return omlm_ffmid_bwd_h(dh2, h1, convw, gamma, mean, rstd, du_tmp, dh1, dgamma, dconv, workspace, M, nseq, F, Fp, p, seed, seed_dev,
                                drop_bits, gh, 1, stream);
#else
    OMLM_CHECK_ARG(dtype == 1, "the fp16 copy serves fp16 operands only");
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dh2 && h1 && convw && gamma && mean && rstd && du_tmp && dh1 && workspace, "null pointer");
    OMLM_CHECK_ARG(Fp % 8 == 0 && Fp >= F && Fp <= 8192 && (size_t)5 * Fp * sizeof(float) <= 160 * 1024, "Fp must be F rounded up to 8 and <= 8192");
    OMLM_CHECK_ARG(nseq > 0 && M % nseq == 0, "M must be batch * nseq");
    hipStream_t st = as_stream(stream);
    float* part_g = workspace;
    float* part_c = workspace + (size_t)FF_BWD1_BLOCKS * Fp;
    if (ffmid_impl() == 1 && ffmid2_supported(Fp) && gh && (p == 0.f || drop_bits)) {
        // du_tmp is not needed by the fused kernel: its first M * 2 floats carry the per-row LayerNorm^T sums
        OMLM_CHECK_ARG(((uintptr_t)du_tmp % 8) == 0, "du_tmp must be 8-byte aligned");
        int g_rows = 0, c_rows = 0;
        int rc2 = ffmid2_bwd_launch(dh2, h1, convw, gamma, rstd, (float*)du_tmp, dh1, part_g, FF_BWD1_BLOCKS, part_c, FF_BWD2_STRIPS,
                                    &g_rows, &c_rows, M, nseq, F, Fp, p, drop_bits, gh, dtype, st);
        if (rc2) return rc2;
        if (dgamma && dconv) {
            const omlm_colsum_desc two[2] = {{part_g, dgamma, g_rows, F, Fp}, {part_c, dconv, c_rows, 2 * F * 3, 2 * F * 3}};
            return omlm_colsum_group(two, 2, stream);
        }
        if (dgamma) { rc2 = omlm_colsum_accumulate(part_g, dgamma, g_rows, F, Fp, stream); if (rc2) return rc2; }
        if (dconv)  { rc2 = omlm_colsum_accumulate(part_c, dconv, c_rows, 2 * F * 3, 2 * F * 3, stream); if (rc2) return rc2; }
        return OMLM_OK;
    }
    const int rows4 = (M + 3) / 4;
    const int b1 = rows4 < FF_BWD1_BLOCKS ? rows4 : FF_BWD1_BLOCKS;
    const int strips = M < FF_BWD2_STRIPS ? M : FF_BWD2_STRIPS;
    dim3 g2((2 * Fp / 8 + FF_THREADS - 1) / FF_THREADS, strips);
    const size_t lds1 = (size_t)4 * Fp * sizeof(float);
#define FF_B1_ATTR(T_, MC_) do { (void)hipFuncSetAttribute((const void*)ffmid_bwd1_kernel<T_, MC_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        (void)hipFuncSetAttribute((const void*)ffmid_bwd1_kernel<T_, MC_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } while (0)
    if (lds1 > 48 * 1024) {
#if !OMLM_FP16
        FF_B1_ATTR(float, 6); FF_B1_ATTR(float, 8); FF_B1_ATTR(float, 16);
#endif
        FF_B1_ATTR(h16_t, 6); FF_B1_ATTR(h16_t, 8); FF_B1_ATTR(h16_t, 16);
    }
#define FF_B1G(T_, MC_, GH_) hipLaunchKernelGGL((ffmid_bwd1_kernel<T_, MC_, GH_>), dim3(b1), dim3(FF_THREADS), lds1, st, (const T_*)dh2, (const T_*)h1, (const T_*)convw, (const T_*)gamma, mean, rstd, (T_*)du_tmp, part_g, M, nseq, F, Fp, p, seed, seed_dev, drop_bits, (const T_*)gh)
#define FF_B1(T_, MC_) do { if (gh) FF_B1G(T_, MC_, true); else FF_B1G(T_, MC_, false); } while (0)
#define FF_B1_DISPATCH(T_) do { const int mc = (Fp / 8 + 63) / 64; \
        if (mc <= 2) FF_B1(T_, 2); else if (mc <= 6) FF_B1(T_, 6); else if (mc <= 8) FF_B1(T_, 8); else FF_B1(T_, 16); } while (0)
#if !OMLM_FP16
    if (dtype == 0) {
        FF_B1_DISPATCH(float);
        hipLaunchKernelGGL(ffmid_bwd2_kernel<float>, g2, dim3(FF_THREADS), 0, st, (const float*)du_tmp, (const float*)h1, (const float*)convw, (float*)dh1, part_c, M, nseq, F, Fp);
    } else
#endif
    {
        FF_B1_DISPATCH(h16_t);
        hipLaunchKernelGGL(ffmid_bwd2_kernel<h16_t>, g2, dim3(FF_THREADS), 0, st, (const h16_t*)du_tmp, (const h16_t*)h1, (const h16_t*)convw, (h16_t*)dh1, part_c, M, nseq, F, Fp);
    }
    int rc = omlm_post_launch("omlm_ffmid_bwd");
    if (rc) return rc;
    if (dgamma && dconv) {
        const omlm_colsum_desc two[2] = {{part_g, dgamma, b1, F, Fp}, {part_c, dconv, strips, 2 * F * 3, 2 * F * 3}};
        return omlm_colsum_group(two, 2, stream);
    }
    if (dgamma) { rc = omlm_colsum_accumulate(part_g, dgamma, b1, F, Fp, stream); if (rc) return rc; }
    if (dconv)  { rc = omlm_colsum_accumulate(part_c, dconv, strips, 2 * F * 3, 2 * F * 3, stream); if (rc) return rc; }
    return OMLM_OK;
}

}   // namespace OMLM_NS
