// HBM-bound row kernels of the transformer trunk:
//   * bias-less LayerNorm forward / backward   (reference transformer.py:24-31)
//   * q/k l2-normalise * learned scale, v cast  (reference transformer.py:265-271, utils.py:68-69)
// Residual stream and statistics are fp32; the operand copies handed to the MFMA GEMMs are
// written in the same pass in the GEMM operand type T (bf16, or fp32 for the bf16x3 mode).
#include "common.h"

namespace OMLM_NS {

#define LN_THREADS 256
#define LN_MAXV 4   // float4 pieces per thread  -> D <= 4096

// y = (x - mean) * rstd * gamma ; optionally also emit cast(x) (the K/V projection reads the
// un-normalised residual: reference transformer.py:228 binds kv_input before the pre-norm :250).
// four outputs as hi/lo planes of the 16-bit type: y = rne16(v), ylo = rne16(v - y)  (precision "fp16ff": the FF-in GEMM reads both)
__device__ __forceinline__ void store4_planes(h16_t* y, h16_t* ylo, float a, float b, float c, float d) {
    u32x2 o, l;
    o[0] = pack_h16_rne(a, b);
    o[1] = pack_h16_rne(c, d);
    l[0] = pack_h16_rne(a - h16_lo_to_f(o[0]), b - h16_hi_to_f(o[0]));
    l[1] = pack_h16_rne(c - h16_lo_to_f(o[1]), d - h16_hi_to_f(o[1]));
    *(u32x2*)y = o;
    *(u32x2*)ylo = l;
}
__device__ __forceinline__ void store4_planes(float*, float*, float, float, float, float) {}      // (fp32 output has no planes)

// LO: `xcast` is the LO PLANE of y (pitch ldy) instead of a cast copy of x
template <typename T, bool LO = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            T* __restrict__ y, T* __restrict__ xcast,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int M, int D, int ldy, float eps) {
    __shared__ float red[LN_THREADS / 64];
    const int nv = D / 4;
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        const float4* xr = (const float4*)(x + (size_t)row * D);
        float4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        }
        const float mu = block_sum<LN_THREADS>(s, red) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float var = block_sum<LN_THREADS>(q, red) / (float)D;
        const float rs = rsqrtf(var + eps);
        if (threadIdx.x == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                const float4 g = ((const float4*)gamma)[c];
                T* yo = y + (size_t)row * ldy + 4 * c;
                if constexpr (LO) {
                    store4_planes(yo, xcast + (size_t)row * ldy + 4 * c, (v[i].x - mu) * rs * g.x, (v[i].y - mu) * rs * g.y, (v[i].z - mu) * rs * g.z, (v[i].w - mu) * rs * g.w);
                    continue;
                }
                store4_from_float(yo, (v[i].x - mu) * rs * g.x, (v[i].y - mu) * rs * g.y, (v[i].z - mu) * rs * g.z, (v[i].w - mu) * rs * g.w);
                if (xcast) {
                    T* xo = xcast + (size_t)row * D + 4 * c;
                    store4_from_float(xo, v[i].x, v[i].y, v[i].z, v[i].w);
                }
            }
        }
    }
}

// The same forward for D == 4 * LN_THREADS (d = 1024), round 4: gamma stays in registers (the general kernel re-reads it per row, behind the
// reductions: a dependent round trip per row), the NEXT row is requested before this one is reduced (two register sets, rows alternate),
// the two reductions use parity-double-buffered LDS slots (two barriers per row instead of four), and no memory instruction sits under a
// column test or a thread test (the row's statistics are stored by every thread: same value, same address).  XC: cast copy of x wanted.
// Same summation order as the general kernel: identical bits.
template <typename T, bool XC, bool LO = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_row1_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                 T* __restrict__ y, T* __restrict__ xcast,
                                                                 float* __restrict__ mean, float* __restrict__ rstd, int M, int ldy, float eps) {
    constexpr int D = 4 * LN_THREADS, NW = LN_THREADS / 64;
    __shared__ float red[2][2][NW];
    const int c = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = gridDim.x;
    const float4 g = ((const float4*)gamma)[c];
    auto process = [&](const float4 v, const int row, const int par) {
        const float s = wave_sum((v.x + v.y) + (v.z + v.w));
        if (lane == 0) red[par][0][wave] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[par][0][w];
        const float mu = t / (float)D;
        const float a = v.x - mu, b = v.y - mu, cc = v.z - mu, d = v.w - mu;
        const float q = wave_sum((a * a + b * b) + (cc * cc + d * d));
        if (lane == 0) red[par][1][wave] = q;
        __syncthreads();                                           // parity `par` is rewritten two rows later, two barriers past its last read
        float u = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) u += red[par][1][w];
        const float rs = rsqrtf(u / (float)D + eps);
        mean[row] = mu; rstd[row] = rs;
        if constexpr (LO) store4_planes(y + (size_t)row * ldy + 4 * c, xcast + (size_t)row * ldy + 4 * c, a * rs * g.x, b * rs * g.y, cc * rs * g.z, d * rs * g.w);
        else store4_from_float(y + (size_t)row * ldy + 4 * c, a * rs * g.x, b * rs * g.y, cc * rs * g.z, d * rs * g.w);
        if constexpr (XC) store4_from_float(xcast + (size_t)row * D + 4 * c, v.x, v.y, v.z, v.w);
    };
    int row = blockIdx.x;
    if (row >= M) return;
    float4 A = ((const float4*)(x + (size_t)row * D))[c], B;
#pragma unroll 1
    for (;;) {
        const int r1 = row + G;
        B = ((const float4*)(x + (size_t)(r1 < M ? r1 : row) * D))[c];         // past the end: a redundant re-read instead of a conditional request
        __builtin_amdgcn_sched_barrier(0);
        process(A, row, 0);
        if (r1 >= M) break;
        const int r2 = r1 + G;
        A = ((const float4*)(x + (size_t)(r2 < M ? r2 : r1) * D))[c];
        __builtin_amdgcn_sched_barrier(0);
        process(B, r1, 1);
        if (r2 >= M) break;
        row = r2;
    }
}

#if OMLM_FP16
// ---- LayerNorm forward for omlm_gemm_mx16 (round 6): y as the half hi plane PLUS the two fp8 planes and the row scale ----------------------------
// The FF-in GEMM of precision "fp16ff" multiplies y_hi on the half matrix cores and corrects with fp8 products (csrc/gemm_mx.hip): per row one
// power-of-two scale 2^e with |y| <= 2^(e + 8), hi8 = e4m3(y_hi 2^-e), lo8 = e4m3((y - y_hi) 2^-(e - 11)).  The scale must be known before the
// row is written: a LayerNorm output is bounded whatever its input, |y_i| <= sqrt(D - 1) max|gamma|, and every row takes that bound (3 binades above
// a typical row's largest entry, of the 17 e4m3 spans; profiles/r06_error_budget_fp8corr.md "one scale per tensor": the same logits error as
// data-derived row scales).  fp8 plane rows have the pitch of the half plane in bytes (2 ldy), the lo8 plane y8_stride bytes behind the hi8
// plane; scale8[row] = e + 127 (E8M0).
// hi plane (half), hi8 / lo8 bytes of four consecutive outputs; sh = 2^-e, sl = 2^-(e - 11)
__device__ __forceinline__ void store4_mx(h16_t* y, unsigned char* y8h, unsigned char* y8l, float a, float b, float c, float d, float sh, float sl) {
    u32x2 o;
    o[0] = pack_h16_rne(a, b);
    o[1] = pack_h16_rne(c, d);
    const float ha = h16_lo_to_f(o[0]), hb = h16_hi_to_f(o[0]), hc = h16_lo_to_f(o[1]), hd = h16_hi_to_f(o[1]);
    *(u32x2*)y = o;
    *(unsigned*)y8h = pack4_fp8(ha * sh, hb * sh, hc * sh, hd * sh);
    *(unsigned*)y8l = pack4_fp8((a - ha) * sl, (b - hb) * sl, (c - hc) * sl, (d - hd) * sl);
}

template <int MAXV>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_mx_kernel(const float* __restrict__ x, const float* __restrict__ gamma, h16_t* __restrict__ y,
                                                               unsigned char* __restrict__ y8, long long y8_stride, unsigned char* __restrict__ scale8,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int M, int D, int ldy, float eps) {
    constexpr int NW = LN_THREADS / 64;
    __shared__ float red[2][2][NW];                     // [parity][sum | sum of squares][wave]
    __shared__ float gred[NW];
    const int nv = D / 4, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 g[MAXV];
    float gm = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = threadIdx.x + i * LN_THREADS;
        g[i] = c < nv ? ((const float4*)gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        gm = fmaxf(fmaxf(gm, fmaxf(fabsf(g[i].x), fabsf(g[i].y))), fmaxf(fabsf(g[i].z), fabsf(g[i].w)));
    }
    gm = wave_max(gm);
    if (lane == 0) gred[wave] = gm;
    __syncthreads();
    gm = gred[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) gm = fmaxf(gm, gred[w]);
    const int e = mx_row_exp(sqrtf((float)D) * gm);
    const float sh = ldexpf(1.0f, -e), sl = ldexpf(1.0f, 11 - e);
    int par = 0;
    for (int row = blockIdx.x; row < M; row += gridDim.x, par ^= 1) {
        const float4* xr = (const float4*)(x + (size_t)row * D);
        float4 v[MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) { v[i] = xr[c]; s += (v[i].x + v[i].y) + (v[i].z + v[i].w); }
        }
        s = wave_sum(s);
        if (lane == 0) red[par][0][wave] = s;
        __syncthreads();
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[par][0][w];
        const float mu = t / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        q = wave_sum(q);
        if (lane == 0) red[par][1][wave] = q;
        __syncthreads();                                           // parity `par` is rewritten two rows later, two barriers past its last read
        float u = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) u += red[par][1][w];
        const float rs = rsqrtf(u / (float)D + eps);
        if (threadIdx.x == 0) { mean[row] = mu; rstd[row] = rs; scale8[row] = (unsigned char)(e + 127); }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                unsigned char* p8 = y8 + (size_t)row * (2 * (size_t)ldy) + 4 * c;
                store4_mx(y + (size_t)row * ldy + 4 * c, p8, p8 + y8_stride, (v[i].x - mu) * rs * g[i].x, (v[i].y - mu) * rs * g[i].y,
                          (v[i].z - mu) * rs * g[i].z, (v[i].w - mu) * rs * g[i].w, sh, sl);
            }
        }
        // the row's tail up to a whole 128-byte fp8 k-tile is zero (omlm_gemm_mx16 reads it): written here, so the planes need no zero fill
        const int tail4 = (((D + 127) & ~127) - D) >> 2;
        if ((int)threadIdx.x < tail4) {
            unsigned char* p8 = y8 + (size_t)row * (2 * (size_t)ldy) + D + 4 * threadIdx.x;
            *(unsigned*)p8 = 0u;
            *(unsigned*)(p8 + y8_stride) = 0u;
        }
    }
}
}   // namespace OMLM_NS
using namespace OMLM_NS;
// y: half hi plane [M, ldy]; y8: fp8 planes [hi8 | lo8] at row pitch 2 ldy bytes, the lo8 plane y8_stride bytes behind the hi8 plane; scale8 [M]
// E8M0 (include/omlm.h).  The hi plane and the statistics are omlm_layernorm_fwd's (bit for bit at D = 1024).
extern "C" int omlm_layernorm_fwd_mx(const float* x, const float* gamma, void* y, void* y8, long long y8_stride, unsigned char* scale8,
                                     float* mean, float* rstd, int M, int D, int ldy, float eps, void* stream) {
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && gamma && y && y8 && scale8 && mean && rstd, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    OMLM_CHECK_ARG(ldy >= D && ldy % 4 == 0 && y8_stride >= (long long)M * 2 * ldy, "ldy / plane stride");
    dim3 grid(M < 2048 ? M : 2048), block(LN_THREADS);
    if (D <= 4 * LN_THREADS)
        hipLaunchKernelGGL((ln_fwd_mx_kernel<1>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (unsigned char*)y8, y8_stride, scale8, mean, rstd, M, D, ldy, eps);
    else
        hipLaunchKernelGGL((ln_fwd_mx_kernel<LN_MAXV>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (unsigned char*)y8, y8_stride, scale8, mean, rstd, M, D, ldy, eps);
    return omlm_post_launch("omlm_layernorm_fwd_mx");
}
namespace OMLM_NS {
#endif

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ; dgamma += dy * xhat
// A thread owns fixed columns, so dgamma is accumulated in registers over the block's rows and
// flushed with one atomic per column per block.
// dy arrives as fp32 or -- in bf16 mode, straight from the input-gradient GEMM's epilogue -- as bf16: the GEMM accumulates in fp32
// and rounds once (the same rounding every other GEMM operand of that mode gets); 73 MB less to write and to re-read per call
__device__ __forceinline__ float4 load4f(const float* p, int c) { return ((const float4*)p)[c]; }
__device__ __forceinline__ float4 load4f(const h16_t* p, int c) {
    const u32x2 w = ((const u32x2*)p)[c];
    return make_float4(h16_lo_to_f(w[0]), h16_hi_to_f(w[0]), h16_lo_to_f(w[1]), h16_hi_to_f(w[1]));
}
template <typename T, typename TDY>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            const T* __restrict__ dres2,
                                                            float* __restrict__ dx, T* __restrict__ dxcast,
                                                            float* __restrict__ dgamma, float* __restrict__ part, int M, int D,
                                                            float dx_scale) {
    __shared__ float red[LN_THREADS / 64];
    const int nv = D / 4;
    float4 dg[LN_MAXV];
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        const float mu = mean[row], rs = rstd[row];
        float4 xh[LN_MAXV], g[LN_MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                const float4 xv = ((const float4*)(x + (size_t)row * D))[c];
                const float4 dv = load4f(dy + (size_t)row * D, c);
                const float4 gm = ((const float4*)gamma)[c];
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                g[i] = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
                dg[i].x += dv.x * xh[i].x; dg[i].y += dv.y * xh[i].y; dg[i].z += dv.z * xh[i].z; dg[i].w += dv.w * xh[i].w;
                s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
                s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            }
        }
        const float m1 = block_sum<LN_THREADS>(s1, red) / (float)D;
        const float m2 = block_sum<LN_THREADS>(s2, red) / (float)D;
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                float4 r = make_float4(rs * (g[i].x - m1 - xh[i].x * m2), rs * (g[i].y - m1 - xh[i].y * m2),
                                       rs * (g[i].z - m1 - xh[i].z * m2), rs * (g[i].w - m1 - xh[i].w * m2));
                if (dres) {
                    const float4 d0 = ((const float4*)(dres + (size_t)row * D))[c];
                    r.x += d0.x; r.y += d0.y; r.z += d0.z; r.w += d0.w;
                }
                if (dres2) {                // a second residual-gradient term in the cast type (the K/V input gradient: see omlm_layernorm_bwd2)
                    const float4 d1 = load4f(dres2 + (size_t)row * D, c);
                    r.x += d1.x; r.y += d1.y; r.z += d1.z; r.w += d1.w;
                }
                r.x *= dx_scale; r.y *= dx_scale; r.z *= dx_scale; r.w *= dx_scale;
                ((float4*)(dx + (size_t)row * D))[c] = r;
                if (dxcast) {
                    T* o = dxcast + (size_t)row * D + 4 * c;
                    store4_from_float(o, r.x, r.y, r.z, r.w);
                }
            }
        }
    }
    if (part) {                         // one partial row per workgroup, reduced by colsum afterwards (no contended atomics)
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) ((float4*)(part + (size_t)blockIdx.x * D))[c] = dg[i];
        }
    } else if (dgamma) {
#pragma unroll
        for (int i = 0; i < LN_MAXV; ++i) {
            const int c = threadIdx.x + i * LN_THREADS;
            if (c < nv) {
                unsafeAtomicAdd(dgamma + 4 * c + 0, dg[i].x); unsafeAtomicAdd(dgamma + 4 * c + 1, dg[i].y);
                unsafeAtomicAdd(dgamma + 4 * c + 2, dg[i].z); unsafeAtomicAdd(dgamma + 4 * c + 3, dg[i].w);
            }
        }
    }
}

// four consecutive elements as they sit in memory (fp32: 16 bytes, 16-bit: 8 bytes), converted on demand
template <typename TT> struct Raw4;
template <> struct Raw4<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p, int c) { v = ((const float4*)p)[c]; }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ float4 get() const { return v; }
};
template <> struct Raw4<h16_t> {
    u32x2 w;
    __device__ __forceinline__ void load(const h16_t* p, int c) { w = ((const u32x2*)p)[c]; }
    __device__ __forceinline__ void zero() { w[0] = 0u; w[1] = 0u; }
    __device__ __forceinline__ float4 get() const { return make_float4(h16_lo_to_f(w[0]), h16_hi_to_f(w[0]), h16_lo_to_f(w[1]), h16_hi_to_f(w[1])); }
};

// The same backward for D == 4 * LN_THREADS (one float4 per thread: d = 1024, every trunk LayerNorm of both model sizes), round 4.
// The general kernel above makes each row a chain of dependent round trips: x / dy, then two block reductions of two barriers each,
// then the residual-gradient row, then the store; its rate comes from 20 waves per CU hiding that (5.45 TB/s).  Here a row is one
// thread's registers, so: the NEXT row's x / dy / residual pieces / statistics are requested before this row is reduced (a clamped row
// index keeps the request unconditional), the two sums cross the waves through one parity-double-buffered LDS slot pair with ONE
// barrier per row, and no memory instruction sits under a column test.
// FL: bit 0 = dres, bit 1 = dres2, bit 2 = dxcast present.  Template switches, not pointer tests: a request or store behind a run-time test
// is a second, shorter path, and hipcc's s_waitcnt count for "this row's pieces have landed" is taken from the shortest one -- with
// the optional pieces really in flight that count stalls every row on the requests it has just issued (seen in the ISA: vmcnt(2) behind
// four loads).
template <typename T, typename TDY, int FL>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_row1_kernel(const TDY* __restrict__ dy, const float* __restrict__ x,
                                                                 const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, const float* __restrict__ dres,
                                                                 const T* __restrict__ dres2, float* __restrict__ dx, T* __restrict__ dxcast,
                                                                 float* __restrict__ dgamma, float* __restrict__ part, int M, float dx_scale) {
    constexpr int D = 4 * LN_THREADS, NW = LN_THREADS / 64;
    __shared__ float red[2][NW][2];
    const int c = threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = gridDim.x;
    const float4 gm = ((const float4*)gamma)[c];
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
    // the 16-bit pieces stay RAW until they are used: a conversion next to the request is a wait next to the request
    struct Row { float4 xv, d0; Raw4<TDY> dv; Raw4<T> d1; float mu, rs; };
    auto request = [&](Row& R, int row) {
        R.xv = ((const float4*)(x + (size_t)row * D))[c];
        R.dv.load(dy + (size_t)row * D, c);
        if constexpr (FL & 1) R.d0 = ((const float4*)(dres + (size_t)row * D))[c]; else R.d0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (FL & 2) R.d1.load(dres2 + (size_t)row * D, c); else R.d1.zero();
        R.mu = mean[row]; R.rs = rstd[row];
    };
    auto process = [&](const Row& cur, const int row, const int it) {
        const float mu = cur.mu, rs = cur.rs;
        const float4 dv = cur.dv.get();
        const float4 xh = make_float4((cur.xv.x - mu) * rs, (cur.xv.y - mu) * rs, (cur.xv.z - mu) * rs, (cur.xv.w - mu) * rs);
        const float4 g = make_float4(dv.x * gm.x, dv.y * gm.y, dv.z * gm.z, dv.w * gm.w);
        dg.x += dv.x * xh.x; dg.y += dv.y * xh.y; dg.z += dv.z * xh.z; dg.w += dv.w * xh.w;
        float s1 = wave_sum((g.x + g.y) + (g.z + g.w));
        float s2 = wave_sum((g.x * xh.x + g.y * xh.y) + (g.z * xh.z + g.w * xh.w));
        if (lane == 0) { red[it & 1][wave][0] = s1; red[it & 1][wave][1] = s2; }
        __syncthreads();                                           // the other parity's slots are rewritten only after the next barrier
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t1 += red[it & 1][w][0]; t2 += red[it & 1][w][1]; }
        const float m1 = t1 / (float)D, m2 = t2 / (float)D;
        float4 r = make_float4(rs * (g.x - m1 - xh.x * m2), rs * (g.y - m1 - xh.y * m2), rs * (g.z - m1 - xh.z * m2), rs * (g.w - m1 - xh.w * m2));
        r.x += cur.d0.x; r.y += cur.d0.y; r.z += cur.d0.z; r.w += cur.d0.w;
        const float4 d1 = cur.d1.get();
        r.x += d1.x; r.y += d1.y; r.z += d1.z; r.w += d1.w;
        r.x *= dx_scale; r.y *= dx_scale; r.z *= dx_scale; r.w *= dx_scale;
        ((float4*)(dx + (size_t)row * D))[c] = r;
        if constexpr (FL & 4) store4_from_float(dxcast + (size_t)row * D + 4 * c, r.x, r.y, r.z, r.w);
    };
    int row = blockIdx.x;
    if (row < M) {
        // two register sets, rows alternate between them: a set is requested one whole row (reduction, barrier, stores) before it is used,
        // and never copied -- a `cur = nxt` at the end of the trip is a wait for the requests of that same trip
        Row A, B;
        request(A, row);
#pragma unroll 1
        for (;;) {
            const int r1 = row + G;
            request(B, r1 < M ? r1 : row);                         // past the end: a redundant re-read instead of a conditional request
            __builtin_amdgcn_sched_barrier(0);
            process(A, row, 0);
            if (r1 >= M) break;
            const int r2 = r1 + G;
            request(A, r2 < M ? r2 : r1);
            __builtin_amdgcn_sched_barrier(0);
            process(B, r1, 1);
            if (r2 >= M) break;
            row = r2;
        }
    }
    if (part) ((float4*)(part + (size_t)blockIdx.x * D))[c] = dg;
    else if (dgamma) {
        unsafeAtomicAdd(dgamma + 4 * c + 0, dg.x); unsafeAtomicAdd(dgamma + 4 * c + 1, dg.y);
        unsafeAtomicAdd(dgamma + 4 * c + 2, dg.z); unsafeAtomicAdd(dgamma + 4 * c + 3, dg.w);
    }
}

static int ln_grid(int M) { return M < 2048 ? M : 2048; }

#if !OMLM_FP16
extern "C" int omlm_layernorm_fwd_h(const float* x, const float* gamma, void* y, void* xcast, float* mean, float* rstd, int M, int D, int ldy, float eps, int out_dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_layernorm_fwd)(const float* x, const float* gamma, void* y, void* xcast, float* mean, float* rstd,
                                  int M, int D, int ldy, float eps, int out_dtype, void* stream) {
#if !OMLM_FP16
    if (out_dtype == OMLM_DT_F16) return omlm_layernorm_fwd_h(x, gamma, y, xcast, mean, rstd, M, D, ldy, eps, 1, stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && gamma && y, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    OMLM_CHECK_ARG(ldy >= D, "ldy < D");
    dim3 grid(ln_grid(M)), block(LN_THREADS);
#ifndef OMLM_LN_ROW1
#define OMLM_LN_ROW1 1          /* 1: D == 1024 rows take the row-pair kernels (next-row prefetch, fewer barriers); 0: the general kernels */
#endif
    if (OMLM_LN_ROW1 && D == 4 * LN_THREADS && mean && rstd) {
        if (out_dtype == 0) {
            if (xcast) hipLaunchKernelGGL((ln_fwd_row1_kernel<float, true>), grid, block, 0, as_stream(stream), x, gamma, (float*)y, (float*)xcast, mean, rstd, M, ldy, eps);
            else       hipLaunchKernelGGL((ln_fwd_row1_kernel<float, false>), grid, block, 0, as_stream(stream), x, gamma, (float*)y, (float*)xcast, mean, rstd, M, ldy, eps);
        } else {
            if (xcast) hipLaunchKernelGGL((ln_fwd_row1_kernel<h16_t, true>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (h16_t*)xcast, mean, rstd, M, ldy, eps);
            else       hipLaunchKernelGGL((ln_fwd_row1_kernel<h16_t, false>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (h16_t*)xcast, mean, rstd, M, ldy, eps);
        }
        return omlm_post_launch("omlm_layernorm_fwd");
    }
    if (out_dtype == 0)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, as_stream(stream), x, gamma, (float*)y, (float*)xcast, mean, rstd, M, D, ldy, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<h16_t>, grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (h16_t*)xcast, mean, rstd, M, D, ldy, eps);
    return omlm_post_launch("omlm_layernorm_fwd");
}

// The same forward with the result as hi/lo planes of the 16-bit type (1 = bf16, 2 = fp16): y = rne16(v), y_lo = rne16(v - y), both at
// pitch ldy (precision "fp16ff": the FF-in GEMM of the forward reads both planes, the backward reads y alone).
#if !OMLM_FP16
extern "C" int omlm_layernorm_fwd_planes_h(const float* x, const float* gamma, void* y, void* y_lo, float* mean, float* rstd, int M, int D, int ldy, float eps, int out_dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_layernorm_fwd_planes)(const float* x, const float* gamma, void* y, void* y_lo, float* mean, float* rstd,
                                                   int M, int D, int ldy, float eps, int out_dtype, void* stream) {
#if !OMLM_FP16
    if (out_dtype == OMLM_DT_F16) return omlm_layernorm_fwd_planes_h(x, gamma, y, y_lo, mean, rstd, M, D, ldy, eps, 1, stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(out_dtype == 1, "layernorm_fwd_planes: out_dtype 1 (bf16) or 2 (fp16)");
    OMLM_CHECK_ARG(x && gamma && y && y_lo && mean && rstd, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    OMLM_CHECK_ARG(ldy >= D && ldy % 4 == 0, "ldy < D");
    dim3 grid(ln_grid(M)), block(LN_THREADS);
    if (D == 4 * LN_THREADS)
        hipLaunchKernelGGL((ln_fwd_row1_kernel<h16_t, false, true>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (h16_t*)y_lo, mean, rstd, M, ldy, eps);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<h16_t, true>), grid, block, 0, as_stream(stream), x, gamma, (h16_t*)y, (h16_t*)y_lo, mean, rstd, M, D, ldy, eps);
    return omlm_post_launch("omlm_layernorm_fwd_planes");
}

extern "C" int omlm_colsum_accumulate(const float* part, float* out, int P, int C, int ldp, void* stream);
#if !OMLM_FP16
extern "C" long long omlm_layernorm_bwd_workspace_bytes(int D) { return (long long)2048 * D * sizeof(float); }
#endif

#if !OMLM_FP16
extern "C" int omlm_layernorm_bwd2_h(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd, const float* dres, const void* dres2, float* dx, void* dxcast, float* dgamma, float* workspace, int M, int D, float dx_scale, int cast_dtype, int dy_dtype, void* stream);
#endif
// dres2 (optional): a second residual-gradient term [M, D] in the type named by cast_dtype (dxcast itself may be null)
extern "C" int OMLM_API(omlm_layernorm_bwd2)(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* dres, const void* dres2, float* dx, void* dxcast, float* dgamma, float* workspace, int M, int D,
                                  float dx_scale, int cast_dtype, int dy_dtype, void* stream) {
#if !OMLM_FP16
    if (cast_dtype == OMLM_DT_F16 || dy_dtype == OMLM_DT_F16) return omlm_layernorm_bwd2_h(dy, x, gamma, mean, rstd, dres, dres2, dx, dxcast, dgamma, workspace, M, D, dx_scale, OMLM_H_CODE(cast_dtype), OMLM_H_CODE(dy_dtype), stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    // Each row is a dependent chain (load -> two block reductions -> load dres -> store), so the rate is set by the rows
    // in flight.  With a workspace (omlm_layernorm_bwd_workspace_bytes) 2048 workgroups each leave one partial dgamma row
    // for colsum; without it dgamma is accumulated with atomics, which only scale to 512 workgroups (2048: 194 -> 273 us).
    // dgamma null WITH a workspace (round 5): the partial rows are left in the workspace ([min(M, 2048), D]) and the caller sums them later --
    // engine.trunk_backward folds the column sums of all LayerNorms of a backward pass into one omlm_colsum_group launch.
    const bool two_level = workspace != nullptr;
    const int blocks = two_level ? (M < 2048 ? M : 2048) : (M < 512 ? M : 512);
    dim3 grid(blocks), block(LN_THREADS);
    float* part = two_level ? workspace : nullptr;
    OMLM_CHECK_ARG(dy_dtype == 0 || dy_dtype == 1, "dy_dtype: 0 = fp32, 1 = bf16, 2 = fp16 (same 16-bit type as the cast output)");
#ifndef OMLM_LN_ROW1
#define OMLM_LN_ROW1 1          /* 1: D == 1024 rows take ln_bwd_row1_kernel (next-row prefetch, one barrier per row) */
#endif
    if (OMLM_LN_ROW1 && D == 4 * LN_THREADS) {
        const int fl = (dres ? 1 : 0) | (dres2 ? 2 : 0) | (dxcast ? 4 : 0);
#define LN_ROW1(TT, TD, F) hipLaunchKernelGGL((ln_bwd_row1_kernel<TT, TD, F>), grid, block, 0, as_stream(stream), (const TD*)dy, x, gamma, mean, rstd, dres, \
                                              (const TT*)dres2, dx, (TT*)dxcast, dgamma, part, M, dx_scale)
#define LN_ROW1_FL(TT, TD) do { switch (fl) { case 0: LN_ROW1(TT, TD, 0); break; case 1: LN_ROW1(TT, TD, 1); break; case 2: LN_ROW1(TT, TD, 2); break; \
        case 3: LN_ROW1(TT, TD, 3); break; case 4: LN_ROW1(TT, TD, 4); break; case 5: LN_ROW1(TT, TD, 5); break; case 6: LN_ROW1(TT, TD, 6); break; \
        default: LN_ROW1(TT, TD, 7); break; } } while (0)
        if (cast_dtype == 0 && dy_dtype == 0) LN_ROW1_FL(float, float);
        else if (cast_dtype == 0)              LN_ROW1_FL(float, h16_t);
        else if (dy_dtype == 0)                LN_ROW1_FL(h16_t, float);
        else                                   LN_ROW1_FL(h16_t, h16_t);
#undef LN_ROW1_FL
#undef LN_ROW1
    } else
    if (cast_dtype == 0 && dy_dtype == 0)
        hipLaunchKernelGGL((ln_bwd_kernel<float, float>), grid, block, 0, as_stream(stream), (const float*)dy, x, gamma, mean, rstd, dres, (const float*)dres2, dx, (float*)dxcast, dgamma, part, M, D, dx_scale);
    else if (cast_dtype == 0)
        hipLaunchKernelGGL((ln_bwd_kernel<float, h16_t>), grid, block, 0, as_stream(stream), (const h16_t*)dy, x, gamma, mean, rstd, dres, (const float*)dres2, dx, (float*)dxcast, dgamma, part, M, D, dx_scale);
    else if (dy_dtype == 0)
        hipLaunchKernelGGL((ln_bwd_kernel<h16_t, float>), grid, block, 0, as_stream(stream), (const float*)dy, x, gamma, mean, rstd, dres, (const h16_t*)dres2, dx, (h16_t*)dxcast, dgamma, part, M, D, dx_scale);
    else
        hipLaunchKernelGGL((ln_bwd_kernel<h16_t, h16_t>), grid, block, 0, as_stream(stream), (const h16_t*)dy, x, gamma, mean, rstd, dres, (const h16_t*)dres2, dx, (h16_t*)dxcast, dgamma, part, M, D, dx_scale);
    int rc = omlm_post_launch("omlm_layernorm_bwd");
    if (rc) return rc;
    if (two_level && dgamma) return omlm_colsum_accumulate(part, dgamma, blocks, D, D, stream);
    return OMLM_OK;
}
#if !OMLM_FP16
extern "C" int omlm_layernorm_bwd(const void* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* dres, float* dx, void* dxcast, float* dgamma, float* workspace, int M, int D,
                                  float dx_scale, int cast_dtype, int dy_dtype, void* stream) {
    return omlm_layernorm_bwd2(dy, x, gamma, mean, rstd, dres, nullptr, dx, dxcast, dgamma, workspace, M, D, dx_scale, cast_dtype, dy_dtype, stream);
}
#endif

// ---------------------------------------------------------------------------------------------
// q/k l2norm * scale.  One wave per 64-wide vector, one element per lane (dim_head == 64).
//   q_raw [M, H*64] fp32, kv_raw [M, 128] fp32  ->  q [M, H*64] T, k [M, 64] T, v [M, 64] T
// 16 lanes per 64-dim vector (a lane owns 4 consecutive dims: 16-byte loads, 8-byte bf16 stores), 4 vectors per
// wave-iteration, norms reduced over the 16-lane group.  (One lane per dim -- 4-byte loads, 2-byte stores, 64-lane
// reductions -- ran at a third of the HBM rate: 155 us for 220 MB in the backward.)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
__device__ __forceinline__ void store4(h16_t* p, float a, float b, float c, float d) { store4_from_float(p, a, b, c, d); }

template <typename T>
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const float* __restrict__ q_raw, const float* __restrict__ kv_raw,
                                                          const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                          T* __restrict__ q, T* __restrict__ k, T* __restrict__ v, int M, int H) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, d0 = 4 * (lane & 15);
    const long long nvec = (long long)M * (H + 2);
    const float4 qs = *(const float4*)(q_scale + d0), ks = *(const float4*)(k_scale + d0);
    for (long long base = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; base < nvec; base += (long long)gridDim.x * 16) {
        const long long vec = base + sub;
        const bool live = vec < nvec;
        const int row = live ? (int)(vec / (H + 2)) : 0, j = live ? (int)(vec % (H + 2)) : 0;
        const float* src = j < H ? q_raw + (size_t)row * H * 64 + j * 64 + d0 : kv_raw + (size_t)row * 128 + (j == H ? 0 : 64) + d0;
        const float4 x = live ? *(const float4*)src : make_float4(0.f, 0.f, 0.f, 0.f);
        const float n2 = group16_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);      // all lanes take part
        if (!live) continue;
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        if (j < H)       store4(q + (size_t)row * H * 64 + j * 64 + d0, x.x * inv * qs.x, x.y * inv * qs.y, x.z * inv * qs.z, x.w * inv * qs.w);
        else if (j == H) store4(k + (size_t)row * 64 + d0, x.x * inv * ks.x, x.y * inv * ks.y, x.z * inv * ks.z, x.w * inv * ks.w);
        else             store4(v + (size_t)row * 64 + d0, x.x, x.y, x.z, x.w);
    }
}

// y = s * x / n, n = max(|x|, eps):  dx = (s*dy - xh * sum(xh * s * dy)) / n   (xh = x / n; for n clamped the
// projection term is dropped exactly like autograd of clamp_min does) ; ds += dy * xh
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ dk,
                                                          const float* __restrict__ dv, const float* __restrict__ q_raw,
                                                          const float* __restrict__ kv_raw, const float* __restrict__ q_scale,
                                                          const float* __restrict__ k_scale, T* __restrict__ dq_raw,
                                                          T* __restrict__ dkv_raw, float* __restrict__ dq_scale,
                                                          float* __restrict__ dk_scale, int M, int H) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, d0 = 4 * (lane & 15);
    const unsigned nvec = (unsigned)M * (unsigned)(H + 2), hp2 = (unsigned)(H + 2);      // host: M * (H + 2) < 2^31
    const float4 qs = *(const float4*)(q_scale + d0), ks = *(const float4*)(k_scale + d0);
    float aq[4] = {0.f, 0.f, 0.f, 0.f}, ak[4] = {0.f, 0.f, 0.f, 0.f};
    // several vectors per lane per trip, all their loads requested before the first reduction (one vector per trip with a 64-bit
    // division in front of its loads ran at 2.7 TB/s -- 83 us at M = 35712, H = 8: one dependent round trip at a time per wave; now 51 us)
#ifndef QKB_U
#define QKB_U 4                 /* vectors per lane per trip */
#endif
#ifndef QKB_BLOCKS
#define QKB_BLOCKS 512          /* workgroups: each ends with 128 atomics into the two scale gradients (2048 workgroups: 69 us, 512: 51 us) */
#endif
    constexpr int U = QKB_U;
    const unsigned stride = gridDim.x * 16u;
    for (unsigned base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u; base < nvec; base += stride * U) {
        bool live[U], isq[U], isk[U];
        int row[U], j[U];
        float4 x[U], dy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned vec = base + u * stride + sub;
            live[u] = vec < nvec;
            row[u] = live[u] ? (int)(vec / hp2) : 0;
            j[u] = live[u] ? (int)(vec - (unsigned)row[u] * hp2) : H + 1;
            isq[u] = j[u] < H; isk[u] = j[u] == H;
            x[u] = make_float4(0.f, 0.f, 0.f, 0.f); dy[u] = x[u];
            if (live[u]) {
                if (isq[u])      { x[u] = *(const float4*)(q_raw + (size_t)row[u] * H * 64 + j[u] * 64 + d0); dy[u] = *(const float4*)(dq + (size_t)row[u] * H * 64 + j[u] * 64 + d0); }
                else if (isk[u]) { x[u] = *(const float4*)(kv_raw + (size_t)row[u] * 128 + d0);               dy[u] = *(const float4*)(dk + (size_t)row[u] * 64 + d0); }
                else             { dy[u] = *(const float4*)(dv + (size_t)row[u] * 64 + d0); }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 s = isq[u] ? qs : ks;
            const float n2 = group16_sum(x[u].x * x[u].x + x[u].y * x[u].y + x[u].z * x[u].z + x[u].w * x[u].w);
            const float nr = sqrtf(n2);
            const bool clamped = nr < 1e-12f;
            const float inv = 1.0f / (clamped ? 1e-12f : nr);
            const float xh[4] = {x[u].x * inv, x[u].y * inv, x[u].z * inv, x[u].w * inv};
            const float gy[4] = {s.x * dy[u].x, s.y * dy[u].y, s.z * dy[u].z, s.w * dy[u].w};
            float proj = group16_sum(xh[0] * gy[0] + xh[1] * gy[1] + xh[2] * gy[2] + xh[3] * gy[3]);
            if (clamped) proj = 0.f;
            if (!live[u]) continue;
            if (isq[u] || isk[u]) {
                const float o0 = (gy[0] - xh[0] * proj) * inv, o1 = (gy[1] - xh[1] * proj) * inv;
                const float o2 = (gy[2] - xh[2] * proj) * inv, o3 = (gy[3] - xh[3] * proj) * inv;
                if (isq[u]) {
                    store4(dq_raw + (size_t)row[u] * H * 64 + j[u] * 64 + d0, o0, o1, o2, o3);
                    aq[0] += dy[u].x * xh[0]; aq[1] += dy[u].y * xh[1]; aq[2] += dy[u].z * xh[2]; aq[3] += dy[u].w * xh[3];
                } else {
                    store4(dkv_raw + (size_t)row[u] * 128 + d0, o0, o1, o2, o3);
                    ak[0] += dy[u].x * xh[0]; ak[1] += dy[u].y * xh[1]; ak[2] += dy[u].z * xh[2]; ak[3] += dy[u].w * xh[3];
                }
            } else {
                store4(dkv_raw + (size_t)row[u] * 128 + 64 + d0, dy[u].x, dy[u].y, dy[u].z, dy[u].w);
            }
        }
    }
    // fold the 4 sub-groups of the wave, then the 4 waves of the block through LDS, then one atomic per dim per block
    __shared__ float sq[4][64], sk[4][64];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aq[i] += __shfl_xor(aq[i], 16, 64); aq[i] += __shfl_xor(aq[i], 32, 64);
        ak[i] += __shfl_xor(ak[i], 16, 64); ak[i] += __shfl_xor(ak[i], 32, 64);
        if (lane < 16) { sq[threadIdx.x >> 6][d0 + i] = aq[i]; sk[threadIdx.x >> 6][d0 + i] = ak[i]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        unsafeAtomicAdd(dq_scale + lane, sq[0][lane] + sq[1][lane] + sq[2][lane] + sq[3][lane]);
        unsafeAtomicAdd(dk_scale + lane, sk[0][lane] + sk[1][lane] + sk[2][lane] + sk[3][lane]);
    }
}

// The same backward from what the fused projection epilogue (omlm_gemm_qknorm) leaves behind: y = s * x / n in the 16-bit operand type
// and n = max(|x|, 1e-12) in fp32 -- xh = y / s (one operand rounding, like every other operand of the 16-bit modes; a zero scale
// gives xh = 0: its dx is 0 anyway, only its d(scale) is lost).  No fp32 pre-norm projections are read (146 -> 64 MB per layer).
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_bwd2_mixed_kernel(const float* __restrict__ dq, const float* __restrict__ dk,
                                                           const float* __restrict__ dv, const T* __restrict__ q, const T* __restrict__ k,
                                                           const float* __restrict__ qn, const float* __restrict__ kn,
                                                           const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                           T* __restrict__ dq_raw, T* __restrict__ dkv_raw, float* __restrict__ dq_scale,
                                                           float* __restrict__ dk_scale, int M, int H) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, d0 = 4 * (lane & 15);
    const unsigned nvec = (unsigned)M * (unsigned)(H + 2), hp2 = (unsigned)(H + 2);
    const float4 qs = *(const float4*)(q_scale + d0), ks = *(const float4*)(k_scale + d0);
    auto rcp0 = [](float v) { return v != 0.f ? 1.0f / v : 0.f; };
    const float4 qsi = make_float4(rcp0(qs.x), rcp0(qs.y), rcp0(qs.z), rcp0(qs.w)), ksi = make_float4(rcp0(ks.x), rcp0(ks.y), rcp0(ks.z), rcp0(ks.w));
    float aq[4] = {0.f, 0.f, 0.f, 0.f}, ak[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = QKB_U;
    const unsigned stride = gridDim.x * 16u;
    for (unsigned base = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u; base < nvec; base += stride * U) {
        bool live[U], isq[U], isk[U];
        int row[U], j[U];
        float4 y[U], dy[U];
        float nr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned vec = base + u * stride + sub;
            live[u] = vec < nvec;
            row[u] = live[u] ? (int)(vec / hp2) : 0;
            j[u] = live[u] ? (int)(vec - (unsigned)row[u] * hp2) : H + 1;
            isq[u] = j[u] < H; isk[u] = j[u] == H;
            y[u] = make_float4(0.f, 0.f, 0.f, 0.f); dy[u] = y[u]; nr[u] = 1.f;
            if (live[u]) {
                if (isq[u])      { y[u] = load4f(q + (size_t)row[u] * H * 64 + j[u] * 64, d0 >> 2); dy[u] = *(const float4*)(dq + (size_t)row[u] * H * 64 + j[u] * 64 + d0); nr[u] = qn[(size_t)row[u] * H + j[u]]; }
                else if (isk[u]) { y[u] = load4f(k + (size_t)row[u] * 64, d0 >> 2);               dy[u] = *(const float4*)(dk + (size_t)row[u] * 64 + d0); nr[u] = kn[row[u]]; }
                else             { dy[u] = *(const float4*)(dv + (size_t)row[u] * 64 + d0); }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 s = isq[u] ? qs : ks, si = isq[u] ? qsi : ksi;
            const bool clamped = nr[u] <= 1e-12f;
            const float inv = 1.0f / nr[u];
            const float xh[4] = {y[u].x * si.x, y[u].y * si.y, y[u].z * si.z, y[u].w * si.w};
            const float gy[4] = {s.x * dy[u].x, s.y * dy[u].y, s.z * dy[u].z, s.w * dy[u].w};
            float proj = group16_sum(xh[0] * gy[0] + xh[1] * gy[1] + xh[2] * gy[2] + xh[3] * gy[3]);      // all lanes take part
            if (clamped) proj = 0.f;
            if (!live[u]) continue;
            if (isq[u] || isk[u]) {
                const float o0 = (gy[0] - xh[0] * proj) * inv, o1 = (gy[1] - xh[1] * proj) * inv;
                const float o2 = (gy[2] - xh[2] * proj) * inv, o3 = (gy[3] - xh[3] * proj) * inv;
                if (isq[u]) {
                    store4(dq_raw + (size_t)row[u] * H * 64 + j[u] * 64 + d0, o0, o1, o2, o3);
                    aq[0] += dy[u].x * xh[0]; aq[1] += dy[u].y * xh[1]; aq[2] += dy[u].z * xh[2]; aq[3] += dy[u].w * xh[3];
                } else {
                    store4(dkv_raw + (size_t)row[u] * 128 + d0, o0, o1, o2, o3);
                    ak[0] += dy[u].x * xh[0]; ak[1] += dy[u].y * xh[1]; ak[2] += dy[u].z * xh[2]; ak[3] += dy[u].w * xh[3];
                }
            } else {
                store4(dkv_raw + (size_t)row[u] * 128 + 64 + d0, dy[u].x, dy[u].y, dy[u].z, dy[u].w);
            }
        }
    }
    __shared__ float sq[4][64], sk[4][64];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aq[i] += __shfl_xor(aq[i], 16, 64); aq[i] += __shfl_xor(aq[i], 32, 64);
        ak[i] += __shfl_xor(ak[i], 16, 64); ak[i] += __shfl_xor(ak[i], 32, 64);
        if (lane < 16) { sq[threadIdx.x >> 6][d0 + i] = aq[i]; sk[threadIdx.x >> 6][d0 + i] = ak[i]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        unsafeAtomicAdd(dq_scale + lane, sq[0][lane] + sq[1][lane] + sq[2][lane] + sq[3][lane]);
        unsafeAtomicAdd(dk_scale + lane, sk[0][lane] + sk[1][lane] + sk[2][lane] + sk[3][lane]);
    }
}


// Round 4, second form (QKB2_STREAM, default): the mixed kernel above walks one list of (row, head | k | v) vectors, so every load and
// store of its trip sits behind a per-vector kind test; hipcc's s_waitcnt bookkeeping then has to assume the shortest path ("nothing
// was issued behind this load") and the four vectors of a trip were in fact requested one after the other, each wait also draining
// the stores before it (seen in the ISA: vmcnt(0) between the units; 62 us per layer = 2.9 TB/s).  Here the q vectors are what they
// are in memory -- ONE contiguous stream of M * H 64-wide vectors with identical work -- and a unit is 8 consecutive vectors: lane l
// holds elements 8 (l & 7) .. + 7 of vector (l >> 3), a wave-instruction reads 1 KiB of q / 2 x 1 KiB of dq contiguously.  Whole units
// run two at a time without a single condition (counted waits); the ragged tail and the k | v rows (1 / (H + 1) of the bytes) take
// the predicated body.
#ifndef QKB2_THREADS
#define QKB2_THREADS 1024       /* 16 waves per workgroup: a quarter of the atomics of 256-thread workgroups at the same waves per CU (1024 workgroups x 128 atomics onto 128
                                   addresses were ~10 us of serialised tail) */
#endif
struct QkUnit { u32x4 y; float4 d0, d1; float nr; };

template <typename T>
__global__ __launch_bounds__(QKB2_THREADS) void qk_norm_bwd2_kernel(const float* __restrict__ dq, const float* __restrict__ dk,
                                                           const float* __restrict__ dv, const T* __restrict__ q, const T* __restrict__ k,
                                                           const float* __restrict__ qn, const float* __restrict__ kn,
                                                           const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                           T* __restrict__ dq_raw, T* __restrict__ dkv_raw, float* __restrict__ dq_scale,
                                                           float* __restrict__ dk_scale, int M, int H) {
    static_assert(sizeof(T) == 2, "16-bit operands");
    const int lane = threadIdx.x & 63, slot = lane >> 3, d0 = 8 * (lane & 7);
    constexpr int NW = QKB2_THREADS / 64;
    const int wave_g = blockIdx.x * NW + (threadIdx.x >> 6), nwaves = gridDim.x * NW;
    auto rcp0 = [](float v) { return v != 0.f ? 1.0f / v : 0.f; };
    float sc[8], sci[8], acc[8];
    auto take_scale = [&](const float* p) {
        const float4 a = *(const float4*)(p + d0), b = *(const float4*)(p + d0 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = v[e]; sci[e] = rcp0(v[e]); acc[e] = 0.f; }
    };
    auto load = [&](QkUnit& U, const T* y, const float* dy, const float* nrm, long long vec) {
        U.y = *(const u32x4*)(y + vec * 64 + d0);
        U.d0 = *(const float4*)(dy + vec * 64 + d0);
        U.d1 = *(const float4*)(dy + vec * 64 + d0 + 4);
        U.nr = nrm[vec];
    };
    // dx of one vector from its normalised output: xh = y / s, gy = s dy, dx = (gy - xh <xh, gy>) / n; d(scale) += dy xh
    auto finish = [&](const QkUnit& U, T* dst, const bool live) {
        const float dy[8] = {U.d0.x, U.d0.y, U.d0.z, U.d0.w, U.d1.x, U.d1.y, U.d1.z, U.d1.w};
        float xh[8], gy[8], part = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xh[2 * e] = h16_lo_to_f(U.y[e]) * sci[2 * e];
            xh[2 * e + 1] = h16_hi_to_f(U.y[e]) * sci[2 * e + 1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { gy[e] = sc[e] * dy[e]; part += xh[e] * gy[e]; }
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);      // the vector's 8 lanes
        const float proj = U.nr <= 1e-12f ? 0.f : part;
        const float inv = 1.0f / U.nr;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = pack_h16_rne((gy[2 * e] - xh[2 * e] * proj) * inv, (gy[2 * e + 1] - xh[2 * e + 1] * proj) * inv);
        if (live) {
            *(u32x4*)dst = o;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += dy[e] * xh[e];
        }
    };
    __shared__ float red[NW][64];
    auto flush = [&](float* dscale) {                       // this workgroup's share of d(scale): 8 slots -> NW waves -> 64 atomics
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = acc[e];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 8) red[threadIdx.x >> 6][d0 + e] = v;
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w][lane];
            unsafeAtomicAdd(dscale + lane, t);
        }
        __syncthreads();
    };

    // ---- q: M * H vectors, one contiguous stream ------------------------------------------------------------------------
    take_scale(q_scale);
    const long long nq = (long long)M * H;
    const int nfull = (int)(nq >> 3), nunits = (int)((nq + 7) >> 3);          // host: M * (H + 2) < 2^31
    int u = wave_g;
#pragma unroll 1
    for (; u + nwaves < nfull; u += 2 * nwaves) {          // two whole units per trip, nothing predicated
        QkUnit A, B;
        const long long va = (long long)u * 8 + slot, vb = (long long)(u + nwaves) * 8 + slot;
        load(A, q, dq, qn, va);
        load(B, q, dq, qn, vb);
        __builtin_amdgcn_sched_barrier(0);                 // all eight requests leave before the first use (hipcc otherwise sinks each load to its use)
        finish(A, dq_raw + va * 64 + d0, true);
        finish(B, dq_raw + vb * 64 + d0, true);
    }
#pragma unroll 1
    for (; u < nunits; u += nwaves) {                      // the last whole unit of a wave and the ragged one
        QkUnit A;
        const long long va = (long long)u * 8 + slot;
        const bool live = va < nq;
        load(A, q, dq, qn, live ? va : 0);
        __builtin_amdgcn_sched_barrier(0);
        finish(A, dq_raw + (live ? va : 0) * 64 + d0, live);
    }
    flush(dq_scale);
    // ---- k | v: one k vector (norm backward) and one v vector (cast) per row, into the [M, 128] kv gradient ----------------
    take_scale(k_scale);
    const int nurow = (M + 7) >> 3;
#pragma unroll 1
    for (int r = wave_g; r < nurow; r += nwaves) {
        const long long row = (long long)r * 8 + slot;
        const bool live = row < M;
        const long long rc = live ? row : 0;
        QkUnit A;
        load(A, k, dk, kn, rc);
        const float4 v0 = *(const float4*)(dv + rc * 64 + d0), v1 = *(const float4*)(dv + rc * 64 + d0 + 4);
        __builtin_amdgcn_sched_barrier(0);
        finish(A, dkv_raw + rc * 128 + d0, live);
        if (live) {
            u32x4 o;
            o[0] = pack_h16_rne(v0.x, v0.y); o[1] = pack_h16_rne(v0.z, v0.w); o[2] = pack_h16_rne(v1.x, v1.y); o[3] = pack_h16_rne(v1.z, v1.w);
            *(u32x4*)(dkv_raw + rc * 128 + 64 + d0) = o;
        }
    }
    flush(dk_scale);
}

#if !OMLM_FP16
extern "C" int omlm_qk_norm_bwd2_h(const float* dq, const float* dk, const float* dv, const void* q, const void* k, const float* qn, const float* kn, const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw, float* dq_scale, float* dk_scale, int M, int H, int dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_qk_norm_bwd2)(const float* dq, const float* dk, const float* dv, const void* q, const void* k, const float* qn,
                                           const float* kn, const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw,
                                           float* dq_scale, float* dk_scale, int M, int H, int dtype, void* stream) {
#if !OMLM_FP16
    if (dtype == OMLM_DT_F16) return omlm_qk_norm_bwd2_h(dq, dk, dv, q, k, qn, kn, q_scale, k_scale, dq_raw, dkv_raw, dq_scale, dk_scale, M, H, 1, stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dtype == 1, "qk_norm_bwd2: 16-bit operand types only (1 = bf16, 2 = fp16)");
    OMLM_CHECK_ARG(dq && dk && dv && q && k && qn && kn && dq_raw && dkv_raw && dq_scale && dk_scale, "null pointer");
    long long nvec = (long long)M * (H + 2);
    OMLM_CHECK_ARG(nvec < (1ll << 31), "qk_norm_bwd2: M * (H + 2) must stay below 2^31 (32-bit vector index)");
#ifndef QKB2_STREAM
#define QKB2_STREAM 1           /* 1: the stream form (qk_norm_bwd2_kernel), 0: the mixed-vector form of round 4's first half */
#endif
#ifndef QKB2_BLOCKS
#define QKB2_BLOCKS 256         /* workgroups of the stream form (QKB2_THREADS / 64 waves each; 128 atomics per workgroup at the end) */
#endif
#if QKB2_STREAM
    long long want = ((long long)M * H / 8 + 2 * (QKB2_THREADS / 64) - 1) / (2 * (QKB2_THREADS / 64));      // ~2 units per wave at least
    int blocks = want > QKB2_BLOCKS ? QKB2_BLOCKS : (want < 1 ? 1 : (int)want);
    hipLaunchKernelGGL(qk_norm_bwd2_kernel<h16_t>, dim3(blocks), dim3(QKB2_THREADS), 0, as_stream(stream), dq, dk, dv, (const h16_t*)q, (const h16_t*)k, qn, kn,
                       q_scale, k_scale, (h16_t*)dq_raw, (h16_t*)dkv_raw, dq_scale, dk_scale, M, H);
#else
    int blocks = (int)((nvec + 15) / 16); if (blocks > QKB_BLOCKS) blocks = QKB_BLOCKS;
    hipLaunchKernelGGL(qk_norm_bwd2_mixed_kernel<h16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), dq, dk, dv, (const h16_t*)q, (const h16_t*)k, qn, kn,
                       q_scale, k_scale, (h16_t*)dq_raw, (h16_t*)dkv_raw, dq_scale, dk_scale, M, H);
#endif
    return omlm_post_launch("omlm_qk_norm_bwd2");
}

#if !OMLM_FP16
extern "C" int omlm_qk_norm_fwd_h(const float* q_raw, const float* kv_raw, const float* q_scale, const float* k_scale, void* q, void* k, void* v, int M, int H, int out_dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_qk_norm_fwd)(const float* q_raw, const float* kv_raw, const float* q_scale, const float* k_scale,
                                void* q, void* k, void* v, int M, int H, int out_dtype, void* stream) {
#if !OMLM_FP16
    if (out_dtype == OMLM_DT_F16) return omlm_qk_norm_fwd_h(q_raw, kv_raw, q_scale, k_scale, q, k, v, M, H, 1, stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(q_raw && kv_raw && q_scale && k_scale && q && k && v, "null pointer");
    long long nvec = (long long)M * (H + 2);
    int blocks = (int)((nvec + 15) / 16); if (blocks > 4096) blocks = 4096;
    if (out_dtype == 0)
        hipLaunchKernelGGL(qk_norm_fwd_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), q_raw, kv_raw, q_scale, k_scale, (float*)q, (float*)k, (float*)v, M, H);
    else
        hipLaunchKernelGGL(qk_norm_fwd_kernel<h16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), q_raw, kv_raw, q_scale, k_scale, (h16_t*)q, (h16_t*)k, (h16_t*)v, M, H);
    return omlm_post_launch("omlm_qk_norm_fwd");
}

#if !OMLM_FP16
extern "C" int omlm_qk_norm_bwd_h(const float* dq, const float* dk, const float* dv, const float* q_raw, const float* kv_raw, const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw, float* dq_scale, float* dk_scale, int M, int H, int out_dtype, void* stream);
#endif
extern "C" int OMLM_API(omlm_qk_norm_bwd)(const float* dq, const float* dk, const float* dv, const float* q_raw, const float* kv_raw,
                                const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw,
                                float* dq_scale, float* dk_scale, int M, int H, int out_dtype, void* stream) {
#if !OMLM_FP16
    if (out_dtype == OMLM_DT_F16) return omlm_qk_norm_bwd_h(dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, dq_raw, dkv_raw, dq_scale, dk_scale, M, H, 1, stream);
#endif
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dq && dk && dv && q_raw && kv_raw && dq_raw && dkv_raw && dq_scale && dk_scale, "null pointer");
    long long nvec = (long long)M * (H + 2);
    OMLM_CHECK_ARG(nvec < (1ll << 31), "qk_norm_bwd: M * (H + 2) must stay below 2^31 (32-bit vector index)");
    int blocks = (int)((nvec + 15) / 16); if (blocks > QKB_BLOCKS) blocks = QKB_BLOCKS;
    if (out_dtype == 0)
        hipLaunchKernelGGL(qk_norm_bwd_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, (float*)dq_raw, (float*)dkv_raw, dq_scale, dk_scale, M, H);
    else
        hipLaunchKernelGGL(qk_norm_bwd_kernel<h16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, (h16_t*)dq_raw, (h16_t*)dkv_raw, dq_scale, dk_scale, M, H);
    return omlm_post_launch("omlm_qk_norm_bwd");
}

}   // namespace OMLM_NS
