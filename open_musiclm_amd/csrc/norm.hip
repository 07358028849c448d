// HBM-bound row kernels of the transformer trunk:
//   * bias-less LayerNorm forward / backward   (reference transformer.py:24-31)
//   * q/k l2-normalise * learned scale, v cast  (reference transformer.py:265-271, utils.py:68-69)
// Residual stream and statistics are fp32; the operand copies handed to the MFMA GEMMs are
// written in the same pass in the GEMM operand type T (bf16, or fp32 for the bf16x3 mode).
#include "common.h"

#define LN_THREADS 256
#define LN_MAXV 4   // float4 pieces per thread  -> D <= 4096

// y = (x - mean) * rstd * gamma ; optionally also emit cast(x) (the K/V projection reads the
// un-normalised residual: reference transformer.py:228 binds kv_input before the pre-norm :250).
template <typename T>
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
                store4_from_float(yo, (v[i].x - mu) * rs * g.x, (v[i].y - mu) * rs * g.y, (v[i].z - mu) * rs * g.z, (v[i].w - mu) * rs * g.w);
                if (xcast) {
                    T* xo = xcast + (size_t)row * D + 4 * c;
                    store4_from_float(xo, v[i].x, v[i].y, v[i].z, v[i].w);
                }
            }
        }
    }
}

// dx = dres + rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ; dgamma += dy * xhat
// A thread owns fixed columns, so dgamma is accumulated in registers over the block's rows and
// flushed with one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dx, T* __restrict__ dxcast,
                                                            float* __restrict__ dgamma, int M, int D, float dx_scale) {
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
                const float4 dv = ((const float4*)(dy + (size_t)row * D))[c];
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
                r.x *= dx_scale; r.y *= dx_scale; r.z *= dx_scale; r.w *= dx_scale;
                ((float4*)(dx + (size_t)row * D))[c] = r;
                if (dxcast) {
                    T* o = dxcast + (size_t)row * D + 4 * c;
                    store4_from_float(o, r.x, r.y, r.z, r.w);
                }
            }
        }
    }
    if (dgamma) {
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

static int ln_grid(int M) { return M < 2048 ? M : 2048; }

extern "C" int omlm_layernorm_fwd(const float* x, const float* gamma, void* y, void* xcast, float* mean, float* rstd,
                                  int M, int D, int ldy, float eps, int out_dtype, void* stream) {
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && gamma && y, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    OMLM_CHECK_ARG(ldy >= D, "ldy < D");
    dim3 grid(ln_grid(M)), block(LN_THREADS);
    if (out_dtype == 0)
        hipLaunchKernelGGL(ln_fwd_kernel<float>, grid, block, 0, as_stream(stream), x, gamma, (float*)y, (float*)xcast, mean, rstd, M, D, ldy, eps);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, grid, block, 0, as_stream(stream), x, gamma, (bf16_t*)y, (bf16_t*)xcast, mean, rstd, M, D, ldy, eps);
    return omlm_post_launch("omlm_layernorm_fwd");
}

extern "C" int omlm_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                                  const float* dres, float* dx, void* dxcast, float* dgamma, int M, int D,
                                  float dx_scale, int cast_dtype, void* stream) {
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dy && x && gamma && mean && rstd && dx, "null pointer");
    OMLM_CHECK_ARG(D % 4 == 0 && D <= 4 * LN_THREADS * LN_MAXV, "D must be a multiple of 4 and <= 4096");
    dim3 grid(M < 512 ? M : 512), block(LN_THREADS);
    if (cast_dtype == 0)
        hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, block, 0, as_stream(stream), dy, x, gamma, mean, rstd, dres, dx, (float*)dxcast, dgamma, M, D, dx_scale);
    else
        hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, block, 0, as_stream(stream), dy, x, gamma, mean, rstd, dres, dx, (bf16_t*)dxcast, dgamma, M, D, dx_scale);
    return omlm_post_launch("omlm_layernorm_bwd");
}

// ---------------------------------------------------------------------------------------------
// q/k l2norm * scale.  One wave per 64-wide vector, one element per lane (dim_head == 64).
//   q_raw [M, H*64] fp32, kv_raw [M, 128] fp32  ->  q [M, H*64] T, k [M, 64] T, v [M, 64] T
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_fwd_kernel(const float* __restrict__ q_raw, const float* __restrict__ kv_raw,
                                                          const float* __restrict__ q_scale, const float* __restrict__ k_scale,
                                                          T* __restrict__ q, T* __restrict__ k, T* __restrict__ v, int M, int H) {
    const int lane = threadIdx.x & 63;
    const long long nvec = (long long)M * (H + 2);
    const float qs = q_scale[lane], ks = k_scale[lane];
    for (long long vec = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); vec < nvec; vec += (long long)gridDim.x * 4) {
        const int row = (int)(vec / (H + 2)), j = (int)(vec % (H + 2));
        if (j < H) {
            const float xv = q_raw[(size_t)row * H * 64 + j * 64 + lane];
            const float nrm = fmaxf(sqrtf(wave_sum(xv * xv)), 1e-12f);
            store_from_float(q + (size_t)row * H * 64 + j * 64 + lane, xv / nrm * qs);
        } else if (j == H) {
            const float xv = kv_raw[(size_t)row * 128 + lane];
            const float nrm = fmaxf(sqrtf(wave_sum(xv * xv)), 1e-12f);
            store_from_float(k + (size_t)row * 64 + lane, xv / nrm * ks);
        } else {
            store_from_float(v + (size_t)row * 64 + lane, kv_raw[(size_t)row * 128 + 64 + lane]);
        }
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
    const int lane = threadIdx.x & 63;
    const long long nvec = (long long)M * (H + 2);
    const float qs = q_scale[lane], ks = k_scale[lane];
    float acc_qs = 0.f, acc_ks = 0.f;
    for (long long vec = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); vec < nvec; vec += (long long)gridDim.x * 4) {
        const int row = (int)(vec / (H + 2)), j = (int)(vec % (H + 2));
        if (j <= H) {
            const bool isq = j < H;
            const float xv = isq ? q_raw[(size_t)row * H * 64 + j * 64 + lane] : kv_raw[(size_t)row * 128 + lane];
            const float dyv = isq ? dq[(size_t)row * H * 64 + j * 64 + lane] : dk[(size_t)row * 64 + lane];
            const float s = isq ? qs : ks;
            const float n2 = wave_sum(xv * xv);
            const float nr = sqrtf(n2);
            const bool clamped = nr < 1e-12f;
            const float n = clamped ? 1e-12f : nr;
            const float xh = xv / n;
            const float gy = s * dyv;
            const float proj = clamped ? 0.f : wave_sum(xh * gy);
            const float dxv = (gy - xh * proj) / n;
            if (isq) { store_from_float(dq_raw + (size_t)row * H * 64 + j * 64 + lane, dxv); acc_qs += dyv * xh; }
            else     { store_from_float(dkv_raw + (size_t)row * 128 + lane, dxv);             acc_ks += dyv * xh; }
        } else {
            store_from_float(dkv_raw + (size_t)row * 128 + 64 + lane, dv[(size_t)row * 64 + lane]);
        }
    }
    // reduce the 4 waves of the block through LDS, then one atomic per lane per block
    __shared__ float sq[4][64], sk[4][64];
    sq[threadIdx.x >> 6][lane] = acc_qs;
    sk[threadIdx.x >> 6][lane] = acc_ks;
    __syncthreads();
    if (threadIdx.x < 64) {
        unsafeAtomicAdd(dq_scale + lane, sq[0][lane] + sq[1][lane] + sq[2][lane] + sq[3][lane]);
        unsafeAtomicAdd(dk_scale + lane, sk[0][lane] + sk[1][lane] + sk[2][lane] + sk[3][lane]);
    }
}

extern "C" int omlm_qk_norm_fwd(const float* q_raw, const float* kv_raw, const float* q_scale, const float* k_scale,
                                void* q, void* k, void* v, int M, int H, int out_dtype, void* stream) {
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(q_raw && kv_raw && q_scale && k_scale && q && k && v, "null pointer");
    long long nvec = (long long)M * (H + 2);
    int blocks = (int)((nvec + 3) / 4); if (blocks > 4096) blocks = 4096;
    if (out_dtype == 0)
        hipLaunchKernelGGL(qk_norm_fwd_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), q_raw, kv_raw, q_scale, k_scale, (float*)q, (float*)k, (float*)v, M, H);
    else
        hipLaunchKernelGGL(qk_norm_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), q_raw, kv_raw, q_scale, k_scale, (bf16_t*)q, (bf16_t*)k, (bf16_t*)v, M, H);
    return omlm_post_launch("omlm_qk_norm_fwd");
}

extern "C" int omlm_qk_norm_bwd(const float* dq, const float* dk, const float* dv, const float* q_raw, const float* kv_raw,
                                const float* q_scale, const float* k_scale, void* dq_raw, void* dkv_raw,
                                float* dq_scale, float* dk_scale, int M, int H, int out_dtype, void* stream) {
    if (M <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(dq && dk && dv && q_raw && kv_raw && dq_raw && dkv_raw && dq_scale && dk_scale, "null pointer");
    long long nvec = (long long)M * (H + 2);
    int blocks = (int)((nvec + 3) / 4); if (blocks > 1024) blocks = 1024;
    if (out_dtype == 0)
        hipLaunchKernelGGL(qk_norm_bwd_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, (float*)dq_raw, (float*)dkv_raw, dq_scale, dk_scale, M, H);
    else
        hipLaunchKernelGGL(qk_norm_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, as_stream(stream), dq, dk, dv, q_raw, kv_raw, q_scale, k_scale, (bf16_t*)dq_raw, (bf16_t*)dkv_raw, dq_scale, dk_scale, M, H);
    return omlm_post_launch("omlm_qk_norm_bwd");
}
