// Optimizer step, casts/repacks, small element-wise helpers, quantizer argmin and a hardware probe.
#include "common.h"
#include <string.h>
#include <math.h>

// ---------------------------------------------------------------------------------------------------------
// global grad-norm (sum of squares, one atomic per block) and fused clip + Adam/AdamW step on flat buffers
// (reference optimizer.py:10-34 -> torch.optim.Adam/AdamW defaults; trainer.py:444-447 clip then step).
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ out,
                                                    float* __restrict__ partials) {
    __shared__ float red[4];
    float s = 0.f;
    const long long n4 = n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = ((const float4*)g)[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[n4 * 4 + threadIdx.x]; s += v * v; }
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) { if (partials) partials[blockIdx.x] = s; else unsafeAtomicAdd(out, s); }
}
// second pass of the deterministic form: one workgroup adds the per-block partials in a fixed order
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nb, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) out[0] += s;
}

// out[0] += sum g^2.  partials (optional, >= 2048 floats): per-workgroup partial sums + a fixed-order final pass, so the result
// is bit-reproducible -- data-parallel replicas then clip with the SAME coefficient and stay bit-identical; without it the
// workgroups add to out[0] with float atomics (arrival order).
extern "C" int omlm_sumsq_accumulate(const float* g, long long n, float* out, float* partials, void* stream) {
    if (n <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(g && out && ((uintptr_t)g % 16) == 0, "sumsq arguments");
    long long blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, n, out, partials);
    if (partials) hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), partials, (int)blocks, out);
    return omlm_post_launch("omlm_sumsq_accumulate");
}

// p, g, m, v: flat fp32.  g is first scaled by gscale (1/world_size for the DP mean) and by the clip coefficient
// min(1, max_norm / (gscale * sqrt(*gnorm_sq) + 1e-6)) (torch.nn.utils.clip_grad_norm_), all on device: no host sync.
// decoupled != 0 -> AdamW (p *= 1 - lr*wd); else Adam with L2 folded into the gradient (wd is 0 on the reference's Adam path).
// p16 (optional) receives the 16-bit copy of the updated parameters (p16_dtype: 1 = bf16, 2 = fp16: the GEMM operand type of the
// model's precision mode); zero_grad != 0 clears g in the same pass.  gscale also carries 1 / loss-scale in fp16 mode.
template <typename T16>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, T16* __restrict__ p16, long long n,
                                                    float lr, float beta1, float beta2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float gscale, const float* __restrict__ gnorm_sq,
                                                    float max_norm, int decoupled, int zero_grad, const float* __restrict__ ls_state) {
    // ls_state (precision "fp16", optional): {loss scale, good steps since its last change, skipped steps, applied steps, the scale the
    // LAST step's gradients carried (written by omlm_loss_scale_update before it moves the scale)}, all on the device: the backward multiplied the loss gradient by ls_state[0] (engine.LossFunction), this kernel divides it out, and the bias
    // corrections count APPLIED steps (ls_state[3] + 1), so a skipped step leaves the Adam clock where it was.
    if (ls_state) {
        gscale /= ls_state[0];
        // the bias corrections of the APPLIED-step clock: formed once per workgroup (two double-precision pow per THREAD before round 5)
        __shared__ float sbc[2];
        if (threadIdx.x == 0) {
            const double t = (double)ls_state[3] + 1.0;
            sbc[0] = (float)(1.0 - pow((double)beta1, t));
            sbc[1] = (float)sqrt(1.0 - pow((double)beta2, t));
        }
        __syncthreads();
        bc1 = sbc[0];
        bc2_sqrt = sbc[1];
    }
    float clip = 1.0f;
    if (gnorm_sq && max_norm > 0.f) {
        const float nrm = sqrtf(gnorm_sq[0]) * gscale;
        clip = fminf(1.0f, max_norm / (nrm + 1e-6f));
    }
    // A non-finite gradient norm (an fp16 operand overflowed somewhere in the backward) must not reach the weights or the moments:
    // the step is skipped on the device -- gradients are still cleared, the 16-bit shadow stays current -- and omlm_loss_scale_update
    // (launched by the host after the last parameter group) halves the loss scale.
    const bool skip = gnorm_sq && !(gnorm_sq[0] < 3.0e38f);
    const float gs = gscale * clip;
    const float step = lr / bc1;
    // one element: the update of AdamW / Adam exactly as written above (the vector path below runs the same expression per lane element)
    auto upd = [&](float& pi, float gi, float& mi, float& vi) {
        gi *= gs;
        if (decoupled) pi *= (1.0f - lr * wd); else gi += wd * pi;
        mi = beta1 * mi + (1.0f - beta1) * gi;
        vi = beta2 * vi + (1.0f - beta2) * gi * gi;
        pi -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    };
    // 16-byte pieces (round 6: dword loads / stores moved 34 B per parameter at 4.9 TB/s; four parameters per lane per trip), scalar tail
    const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && (!p16 || ((uintptr_t)p16 & 7) == 0);
#ifdef OMLM_ADAMW_SCALAR
    const long long n4 = 0;                                // (A/B build: the element-wise form)
#else
    const long long n4 = vec ? n >> 2 : 0;
#endif
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        if (skip) { if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
        float4 p4 = ((const float4*)p)[i], m4 = ((const float4*)m)[i], v4 = ((const float4*)v)[i];
        const float4 g4 = ((const float4*)g)[i];
        upd(p4.x, g4.x, m4.x, v4.x); upd(p4.y, g4.y, m4.y, v4.y); upd(p4.z, g4.z, m4.z, v4.z); upd(p4.w, g4.w, m4.w, v4.w);
        ((float4*)p)[i] = p4; ((float4*)m)[i] = m4; ((float4*)v)[i] = v4;
        if (p16) {
            union { T16 h[4]; uint2 u; } pk;
            pk.h[0] = (T16)p4.x; pk.h[1] = (T16)p4.y; pk.h[2] = (T16)p4.z; pk.h[3] = (T16)p4.w;
            *(uint2*)(p16 + 4 * i) = pk.u;
        }
        if (zero_grad) ((float4*)g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (long long i = 4 * n4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        if (skip) { if (zero_grad) g[i] = 0.f; continue; }
        float pi = p[i], mi = m[i], vi = v[i];
        upd(pi, g[i], mi, vi);
        p[i] = pi; m[i] = mi; v[i] = vi;
        if (p16) p16[i] = (T16)pi;
        if (zero_grad) g[i] = 0.f;
    }
}

extern "C" int omlm_adamw_clip_step(float* p, float* g, float* m, float* v, void* p16, long long n,
                                    float lr, float beta1, float beta2, float eps, float wd, int step,
                                    float gscale, const float* gnorm_sq, float max_norm, int decoupled, int zero_grad,
                                    int p16_dtype, const float* ls_state, void* stream) {
    if (n <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(p && g && m && v && step >= 1, "adamw arguments");
    OMLM_CHECK_ARG(!p16 || p16_dtype == OMLM_DT_BF16 || p16_dtype == OMLM_DT_F16, "p16_dtype: 1 = bf16, 2 = fp16");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step)), bc2 = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long long blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    if (p16 && p16_dtype == OMLM_DT_F16)
        hipLaunchKernelGGL(adamw_kernel<f16_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, (f16_t*)p16, n,
                           lr, beta1, beta2, eps, wd, bc1, bc2, gscale, gnorm_sq, max_norm, decoupled, zero_grad, ls_state);
    else
        hipLaunchKernelGGL(adamw_kernel<h16_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v, (h16_t*)p16, n,
                           lr, beta1, beta2, eps, wd, bc1, bc2, gscale, gnorm_sq, max_norm, decoupled, zero_grad, ls_state);
    return omlm_post_launch("omlm_adamw_clip_step");
}

// Dynamic loss scale of precision "fp16", entirely on the device (no host read of the norm): after the parameter groups of a step,
// a non-finite norm halves the scale (the step was skipped by adamw_kernel), `interval` consecutive good steps double it.
__global__ void loss_scale_update_kernel(float* st, const float* __restrict__ gnorm_sq, float growth, float backoff, float interval,
                                         float smin, float smax) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st[4] = st[0];                 // the scale this step's gradients (and its norm) carried: readers of the norm divide by THIS, not by the updated scale
    if (!(gnorm_sq[0] < 3.0e38f)) { st[0] = fmaxf(st[0] * backoff, smin); st[1] = 0.f; st[2] += 1.f; }
    else {
        st[3] += 1.f; st[1] += 1.f;
        if (interval > 0.f && st[1] >= interval) { st[0] = fminf(st[0] * growth, smax); st[1] = 0.f; }
    }
}
extern "C" int omlm_loss_scale_update(float* ls_state, const float* gnorm_sq, float growth, float backoff, int interval,
                                      float scale_min, float scale_max, void* stream) {
    OMLM_CHECK_ARG(ls_state && gnorm_sq && growth >= 1.f && backoff > 0.f && backoff <= 1.f && scale_min > 0.f && scale_max >= scale_min,
                   "loss_scale_update arguments");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), ls_state, gnorm_sq, growth, backoff, (float)interval,
                       scale_min, scale_max);
    return omlm_post_launch("omlm_loss_scale_update");
}

// ---------------------------------------------------------------------------------------------------------
// dst[c, r] = (T) src[r, c]  (r < R, c < C): transposed operand copies of the weights, so that the input-gradient GEMMs
// (dX = dY W) read W^T k-contiguous instead of W k-major.  Measured on MI355X: the same contraction runs at 795 TFLOP/s
// with a k-contiguous B versus 517 with a k-major B; the weights are tiny next to the activations, so the copy is ~free.
// 64x64 tile through LDS (pitch 65 floats): 256-byte coalesced reads, 128-byte coalesced bf16 writes.
template <typename T>
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int R, int C,
                                                             int ld_src, int ld_dst) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? src[(long long)r * ld_src + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < C && r < R) store_from_float(dst + (long long)c * ld_dst + r, tile[tx][ty + 4 * i]);
    }
}
extern "C" int omlm_transpose_cast(const float* src, void* dst, int R, int C, int ld_src, int ld_dst, int out_dtype, void* stream) {
    if (R <= 0 || C <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(src && dst && ld_src >= C && ld_dst >= R, "transpose_cast arguments");
    dim3 grid((C + 63) / 64, (R + 63) / 64), block(256);
    OMLM_CHECK_ARG(out_dtype >= 0 && out_dtype <= 2, "out_dtype: 0 = fp32, 1 = bf16, 2 = fp16");
    if (out_dtype == 0) hipLaunchKernelGGL(transpose_cast_kernel<float>, grid, block, 0, as_stream(stream), src, (float*)dst, R, C, ld_src, ld_dst);
    else if (out_dtype == OMLM_DT_F16) hipLaunchKernelGGL(transpose_cast_kernel<f16_t>, grid, block, 0, as_stream(stream), src, (f16_t*)dst, R, C, ld_src, ld_dst);
    else hipLaunchKernelGGL(transpose_cast_kernel<h16_t>, grid, block, 0, as_stream(stream), src, (h16_t*)dst, R, C, ld_src, ld_dst);
    return omlm_post_launch("omlm_transpose_cast");
}

// ---------------------------------------------------------------------------------------------------------
// The per-step weight re-packs of a model (FF-in value / gate halves, FF-out rows at the padded pitch, conv taps transposed, padded
// LayerNorm gammas: ~6 small launches per layer, 36 per coarse-small step) as ONE launch: problem i copies / casts src [R, C] (pitch
// ld_src) to dst [R, ld_dst] with zero pad columns, or -- transpose set -- writes dst[c, r] = src[r, c] (pad untouched).
#define OMLM_CAST_GROUP_MAX 64
struct omlm_cast_pad_desc { const float* src; void* dst; int R, C, ld_src, ld_dst, transpose, lo; };             // include/omlm.h
struct CastGroupArgs { int n; int start[OMLM_CAST_GROUP_MAX + 1]; omlm_cast_pad_desc d[OMLM_CAST_GROUP_MAX]; };
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_group_kernel(CastGroupArgs ga) {
    int pi = 0;
    for (int i = 1; i < ga.n; ++i) if ((int)blockIdx.x >= ga.start[i]) pi = i;           // uniform scalar scan (starts ascend)
    const omlm_cast_pad_desc q = ga.d[pi];
    const int blk = blockIdx.x - ga.start[pi], nblk = ga.start[pi + 1] - ga.start[pi];
    T* dst = (T*)q.dst;
    // lo (round 5, precision "fp16ff"): dst receives the LO PLANE of the cast, rne(v - rne(v)) -- the un-rounded weight is hi + lo
    auto val = [&](float v) -> float { return q.lo ? v - (float)(T)v : v; };
    if (q.transpose) {
        for (long long e = (long long)blk * 256 + threadIdx.x; e < (long long)q.R * q.C; e += (long long)nblk * 256) {
            const int r = (int)(e / q.C), c = (int)(e - (long long)r * q.C);
            store_from_float(dst + (size_t)c * q.ld_dst + r, val(q.src[(size_t)r * q.ld_src + c]));
        }
    } else if (sizeof(T) == 2 && (q.ld_dst & 7) == 0 && (q.ld_src & 3) == 0 && (((uintptr_t)q.src | (uintptr_t)q.dst) & 15) == 0) {
        // 16-bit outputs in 16-byte pieces: eight elements per thread per trip (round 6: the element-wise form below moved the ~400 MB of a
        // coarse-small re-pack at 2.7 TB/s)
        // the block's rows blk, blk + nblk, ... flattened with their 8-element pieces: every thread has a piece on every trip (a 1024-wide row alone
        // is half a block)
        const int P = q.ld_dst >> 3, nrows = (q.R - blk + nblk - 1) / nblk;
        for (int idx = threadIdx.x; idx < nrows * P; idx += 256) {
            const int ri = idx / P, c = (idx - ri * P) * 8, r = blk + ri * nblk;
            const float* sr = q.src + (size_t)r * q.ld_src;
            {
                float v[8];
                if (c + 8 <= q.C) {
                    const float4 a = *(const float4*)(sr + c), b = *(const float4*)(sr + c + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = c + e < q.C ? sr[c + e] : 0.f;
                }
                union { T h[8]; uint4 u; } pk;
#pragma unroll
                for (int e = 0; e < 8; ++e) pk.h[e] = (T)(c + e < q.C ? val(v[e]) : 0.f);
                *(uint4*)(dst + (size_t)r * q.ld_dst + c) = pk.u;
            }
        }
    } else {
        for (int r = blk; r < q.R; r += nblk)
            for (int c = threadIdx.x; c < q.ld_dst; c += 256)
                store_from_float(dst + (size_t)r * q.ld_dst + c, c < q.C ? val(q.src[(size_t)r * q.ld_src + c]) : 0.f);
    }
}
extern "C" int omlm_cast_pad_group(const omlm_cast_pad_desc* d, int count, int out_dtype, void* stream) {
    if (count <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(d != nullptr, "null descriptor array");
    OMLM_CHECK_ARG(out_dtype >= 0 && out_dtype <= 2, "out_dtype: 0 = fp32, 1 = bf16, 2 = fp16");
    for (int base = 0; base < count; base += OMLM_CAST_GROUP_MAX) {
        const int n = count - base < OMLM_CAST_GROUP_MAX ? count - base : OMLM_CAST_GROUP_MAX;
        CastGroupArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.n = n;
        int start = 0;
        for (int i = 0; i < n; ++i) {
            const omlm_cast_pad_desc& q = d[base + i];
            OMLM_CHECK_ARG(q.src && q.dst && q.R > 0 && q.C > 0 && q.ld_src >= q.C && (q.transpose ? q.ld_dst >= q.R : q.ld_dst >= q.C), "cast_pad group: bad problem");
            ga.d[i] = q;
            ga.start[i] = start;
            long long work = q.transpose ? ((long long)q.R * q.C + 255) / 256 : (long long)q.R;
            int blocks = (int)(work < 1 ? 1 : (work > 1024 ? 1024 : work));
            start += blocks;
        }
        ga.start[n] = start;
        if (out_dtype == 0) hipLaunchKernelGGL(cast_pad_group_kernel<float>, dim3(start), dim3(256), 0, as_stream(stream), ga);
        else if (out_dtype == OMLM_DT_F16) hipLaunchKernelGGL(cast_pad_group_kernel<f16_t>, dim3(start), dim3(256), 0, as_stream(stream), ga);
        else hipLaunchKernelGGL(cast_pad_group_kernel<h16_t>, dim3(start), dim3(256), 0, as_stream(stream), ga);
    }
    return omlm_post_launch("omlm_cast_pad_group");
}

// ---------------------------------------------------------------------------------------------------------
// fp8 planes of fp32 weights for omlm_gemm_mx16 (round 6, csrc/gemm_mx.hip), all problems of a model in ONE launch: with hi = rne_half(w) (the
// half operand plane the re-pack above writes) and lo = w - hi, row r gets the scale 2^e with max_c |hi| <= 2^(e + 8) and
//   dst8[r, c] = e4m3(hi 2^-e),  (dst8 + lo_stride)[r, c] = e4m3(lo 2^-(e - 11)),  scale8[r] = e + 127      (row pitch ld8 bytes).
// Bytes behind column C of a row are left alone (the caller's buffer is zero-filled once: omlm_gemm_mx16 wants zeros up to ceil128(K)).
struct omlm_quant_rows_desc { const float* src; unsigned char* dst8; long long lo_stride; unsigned char* scale8; int R, C, ld_src, ld8; };     // include/omlm.h
struct QuantGroupArgs { int n; int start[OMLM_CAST_GROUP_MAX + 1]; omlm_quant_rows_desc d[OMLM_CAST_GROUP_MAX]; };
__global__ __launch_bounds__(256) void quant_rows_mx_kernel(QuantGroupArgs ga) {
    __shared__ float red[4];
    int pi = 0;
    for (int i = 1; i < ga.n; ++i) if ((int)blockIdx.x >= ga.start[i]) pi = i;           // uniform scalar scan (starts ascend)
    const omlm_quant_rows_desc q = ga.d[pi];
    const int blk = blockIdx.x - ga.start[pi], nblk = ga.start[pi + 1] - ga.start[pi];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blk; r < q.R; r += nblk) {
        const float* sr = q.src + (size_t)r * q.ld_src;
        float m = 0.f;
        const bool vec4 = ((q.ld_src & 3) == 0) && (((uintptr_t)q.src & 15) == 0);
        if (vec4) {                                        // 16-byte loads (round 6; the element-wise passes ran at a third of the HBM rate)
            for (int c = threadIdx.x * 4; c < q.C; c += 1024) {
                if (c + 4 <= q.C) {
                    const float4 a = *(const float4*)(sr + c);
                    m = fmaxf(fmaxf(m, fabsf((float)(f16_t)a.x)), fmaxf(fabsf((float)(f16_t)a.y), fmaxf(fabsf((float)(f16_t)a.z), fabsf((float)(f16_t)a.w))));
                } else {
                    for (int x = 0; c + x < q.C; ++x) m = fmaxf(m, fabsf((float)(f16_t)sr[c + x]));
                }
            }
        } else
        for (int c = threadIdx.x; c < q.C; c += 256) m = fmaxf(m, fabsf((float)(f16_t)sr[c]));
        m = wave_max(m);
        __syncthreads();                                   // (red is re-used row after row)
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const int e = mx_row_exp(m);
        const float sh = ldexpf(1.0f, -e), sl = ldexpf(1.0f, 11 - e);
        if (threadIdx.x == 0) q.scale8[r] = (unsigned char)(e + 127);
        unsigned char* d8 = q.dst8 + (size_t)r * q.ld8;
        for (int c = threadIdx.x * 4; c < q.C; c += 1024) {
            float w[4], h[4];
            if (vec4 && c + 4 <= q.C) { const float4 a = *(const float4*)(sr + c); w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; }
            else {
#pragma unroll
                for (int x = 0; x < 4; ++x) w[x] = c + x < q.C ? sr[c + x] : 0.f;
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) h[x] = (float)(f16_t)w[x];
            *(unsigned*)(d8 + c) = pack4_fp8(h[0] * sh, h[1] * sh, h[2] * sh, h[3] * sh);
            *(unsigned*)(d8 + q.lo_stride + c) = pack4_fp8((w[0] - h[0]) * sl, (w[1] - h[1]) * sl, (w[2] - h[2]) * sl, (w[3] - h[3]) * sl);
        }
    }
}
extern "C" int omlm_quant_rows_mx(const omlm_quant_rows_desc* d, int count, void* stream) {
    if (count <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(d != nullptr, "null descriptor array");
    for (int base = 0; base < count; base += OMLM_CAST_GROUP_MAX) {
        const int n = count - base < OMLM_CAST_GROUP_MAX ? count - base : OMLM_CAST_GROUP_MAX;
        QuantGroupArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.n = n;
        int start = 0;
        for (int i = 0; i < n; ++i) {
            const omlm_quant_rows_desc& q = d[base + i];
            OMLM_CHECK_ARG(q.src && q.dst8 && q.scale8 && q.R > 0 && q.C > 0 && q.ld_src >= q.C && q.ld8 >= (q.C + 3) / 4 * 4 && q.ld8 % 4 == 0 &&
                           q.lo_stride % 4 == 0 && ((uintptr_t)q.dst8 % 4) == 0, "quant_rows_mx: bad problem");
            ga.d[i] = q;
            ga.start[i] = start;
            start += q.R < 1024 ? q.R : 1024;
        }
        ga.start[n] = start;
        hipLaunchKernelGGL(quant_rows_mx_kernel, dim3(start), dim3(256), 0, as_stream(stream), ga);
    }
    return omlm_post_launch("omlm_quant_rows_mx");
}

// ---------------------------------------------------------------------------------------------------------
// dst[r, c] = (T) src[r, c] for c < C ; 0 for C <= c < ldd.   (weight repack / operand casts)
template <typename T>
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ src, T* __restrict__ dst, long long R, int C, int lds_, int ldd) {
    for (long long r = blockIdx.x; r < R; r += gridDim.x)
        for (int c = threadIdx.x; c < ldd; c += 256)
            store_from_float(dst + r * ldd + c, c < C ? src[r * lds_ + c] : 0.f);
}
extern "C" int omlm_cast_pad(const float* src, void* dst, long long R, int C, int ld_src, int ld_dst, int out_dtype, void* stream) {
    if (R <= 0 || C <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(src && dst && ld_src >= C && ld_dst >= C, "cast_pad arguments");
    dim3 grid((unsigned)(R < 8192 ? R : 8192)), block(256);
    OMLM_CHECK_ARG(out_dtype >= 0 && out_dtype <= 2, "out_dtype: 0 = fp32, 1 = bf16, 2 = fp16");
    if (out_dtype == 0) hipLaunchKernelGGL(cast_pad_kernel<float>, grid, block, 0, as_stream(stream), src, (float*)dst, R, C, ld_src, ld_dst);
    else if (out_dtype == OMLM_DT_F16) hipLaunchKernelGGL(cast_pad_kernel<f16_t>, grid, block, 0, as_stream(stream), src, (f16_t*)dst, R, C, ld_src, ld_dst);
    else hipLaunchKernelGGL(cast_pad_kernel<h16_t>, grid, block, 0, as_stream(stream), src, (h16_t*)dst, R, C, ld_src, ld_dst);
    return omlm_post_launch("omlm_cast_pad");
}

// ---------------------------------------------------------------------------------------------------------
// rel-pos MLP helpers (reference transformer.py:36-67): SiLU layers.  pre = a + bias is saved for the backward.
//   first layer (Linear(1, Hd)): a[r, c] = r * w0[c]
__global__ void relpos_first_kernel(const float* __restrict__ w0, const float* __restrict__ b0, float* __restrict__ pre,
                                    float* __restrict__ z, int n, int Hd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * Hd) return;
    const int r = (int)(i / Hd), c = (int)(i % Hd);
    const float s = (float)r * w0[c] + b0[c];
    pre[i] = s;
    z[i] = s / (1.0f + __expf(-s));
}
__global__ void bias_silu_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ pre,
                                     float* __restrict__ z, long long total, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float s = a[i] + b[i % C];
    pre[i] = s;
    z[i] = s / (1.0f + __expf(-s));
}
// ds = dz * silu'(pre)
__global__ void silu_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ pre, float* __restrict__ ds, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float s = pre[i], sg = 1.0f / (1.0f + __expf(-s));
    ds[i] = dz[i] * (sg * (1.0f + s * (1.0f - sg)));
}
// out[r, c] = a[r, c] + b[c] (no activation: last rel-pos layer), row pitch ld (pad columns -> 0)
__global__ void bias_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int R, int C, int ld) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)R * ld) return;
    const int c = (int)(i % ld);
    out[i] = c < C ? a[i] + b[c] : 0.f;
}
// dw0[c] += sum_r r * ds[r, c]   (first-layer weight gradient).  64 columns x 4 row lanes per workgroup, 16 row chunks in grid.y:
// the first version walked all n rows with ONE thread per column (2 workgroups, 1116 dependent loads: 201 us per step).
__global__ __launch_bounds__(256) void relpos_first_bwd_kernel(const float* __restrict__ ds, float* __restrict__ dw0, int n, int Hd) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < Hd) for (int r = blockIdx.y * 4 + rl; r < n; r += 4 * gridDim.y) s += (float)r * ds[(size_t)r * Hd + c];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && c < Hd) unsafeAtomicAdd(dw0 + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

extern "C" int omlm_relpos_first_fwd(const float* w0, const float* b0, float* pre, float* z, int n, int Hd, void* stream) {
    OMLM_CHECK_ARG(w0 && b0 && pre && z && n > 0 && Hd > 0, "relpos_first arguments");
    const long long tot = (long long)n * Hd;
    hipLaunchKernelGGL(relpos_first_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), w0, b0, pre, z, n, Hd);
    return omlm_post_launch("omlm_relpos_first_fwd");
}
extern "C" int omlm_bias_silu_fwd(const float* a, const float* b, float* pre, float* z, long long R, int C, void* stream) {
    OMLM_CHECK_ARG(a && b && pre && z && R > 0 && C > 0, "bias_silu arguments");
    const long long tot = R * C;
    hipLaunchKernelGGL(bias_silu_fwd_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, pre, z, tot, C);
    return omlm_post_launch("omlm_bias_silu_fwd");
}
extern "C" int omlm_silu_bwd(const float* dz, const float* pre, float* ds, long long total, void* stream) {
    OMLM_CHECK_ARG(dz && pre && ds && total > 0, "silu_bwd arguments");
    hipLaunchKernelGGL(silu_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), dz, pre, ds, total);
    return omlm_post_launch("omlm_silu_bwd");
}
extern "C" int omlm_bias_add(const float* a, const float* b, float* out, int R, int C, int ld, void* stream) {
    OMLM_CHECK_ARG(a && b && out && R > 0 && ld >= C, "bias_add arguments");
    const long long tot = (long long)R * ld;
    hipLaunchKernelGGL(bias_add_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), a, b, out, R, C, ld);
    return omlm_post_launch("omlm_bias_add");
}
extern "C" int omlm_relpos_first_bwd(const float* ds, float* dw0, int n, int Hd, void* stream) {
    OMLM_CHECK_ARG(ds && dw0 && n > 0 && Hd > 0, "relpos_first_bwd arguments");
    hipLaunchKernelGGL(relpos_first_bwd_kernel, dim3((Hd + 63) / 64, 16), dim3(256), 0, as_stream(stream), ds, dw0, n, Hd);
    return omlm_post_launch("omlm_relpos_first_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Fused rel-pos MLP (round 5).  The continuous relative-position bias (reference transformer.py:36-67) is Linear(1, Hd) + SiLU,
// 2 x [Linear(Hd, Hd) + SiLU], Linear(Hd, H) on the n causal distances 0 .. n - 1: 0.6 GMAC forward for coarse-small -- which ran as
// 7 forward and 14 backward launches (three latency-bound register-staged fp32 GEMMs each way, 25-50 us apiece at 12-19 TFLOP/s, plus
// element-wise and column-sum kernels: ~0.4 ms of the 23.5 ms step).  Rows are independent, so a workgroup carries 16 rows through ALL
// layers with its activations in LDS: ONE forward launch and, for the backward, one row-chain launch (d(pre) of every layer) plus one
// launch for every weight / bias gradient (each output element owned by one lane: plain += into the gradient buffers, fixed summation
// order -- deterministic, which the 16-bit modes downstream need: DESIGN.md section 4.1, round 4).
// The contractions run on the EXACT fp32 matrix instruction v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain, bitwise: guide section 3):
// the first version of these kernels did them as VALU FMAs against activations broadcast from LDS and was bound by that broadcast at
// the same ~12 TFLOP/s as the GEMM path (forward 96 us, backward 270 us: no gain).  Operand maps of the instruction (lane l, i = l & 15,
// g = l >> 4): A[i][k = g], B[k = g][j = i], D[row = 4 g + reg][col = i].  Which four k a step takes is free (any fixed order is a fixed
// sum): where an operand's k runs along a lane's own row the step m takes k = 8 g + m (two 16-byte reads give a lane its 8 steps), where k
// runs across rows it takes k = 4 m + g.
// Hd must be 256 or 512 (512 in every shipped config); other widths keep the launch-per-layer path (engine.relpos_forward).
struct relpos_mlp_params {
    const float *w0, *b0, *W1, *b1, *W2, *b2, *W3, *b3;      // net.0.0.weight [Hd] (as a vector), .bias; net.1.0 / net.2.0 [Hd, Hd]; net.3 [H, Hd], [H]
    float *pre0, *z0, *pre1, *z1, *pre2, *z2;                // [n, Hd] each: saved for the backward (forward: written when non-null)
    float* table;                                            // forward out [n, ldb] (pad columns zeroed)
    const float* dtable;                                     // backward in [n, ldb]
    float *ds0, *ds1, *ds2;                                  // backward scratch [n, Hd]: d(pre) of layers 0 .. 2
    float *gw0, *gb0, *gW1, *gb1, *gW2, *gb2, *gW3, *gb3;    // gradient buffers (accumulated into)
    int n, Hd, H, ldb;
};
__device__ __forceinline__ float silu_f(float s) { return s / (1.0f + __expf(-s)); }
__device__ __forceinline__ float silu_grad_f(float s) { const float sg = 1.0f / (1.0f + __expf(-s)); return sg * (1.0f + s * (1.0f - sg)); }
#define RP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define RP_RB 16                                             /* rows per workgroup = one MFMA tile row */

template <int HD>
__global__ __launch_bounds__(256) void relpos_mlp_fwd_kernel(relpos_mlp_params a) {
    constexpr int KC = 32, WP = KC + 4, ZP = HD + 4, NT = HD / 64;      // wt pitch 36 floats / z pitch HD + 4: 16-byte rows, conflict-free 16-lane b128 reads
    extern __shared__ __attribute__((aligned(16))) float rp_sm[];
    float* zA = rp_sm;                      // [16][ZP]
    float* zB = zA + RP_RB * ZP;            // [16][ZP]
    float* wt = zB + RP_RB * ZP;            // [HD][WP]: a 32-k chunk of the layer's weight, row = output column
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4, r0 = blockIdx.x * RP_RB;
    const int colbase = wave * (HD / 4);
    for (int e = t; e < RP_RB * HD; e += 256) {
        const int r = e / HD, c = e - r * HD, row = r0 + r;
        const float s = (float)row * a.w0[c] + a.b0[c], z = silu_f(s);
        zA[r * ZP + c] = z;
        if (a.pre0 && row < a.n) { a.pre0[(size_t)row * HD + c] = s; a.z0[(size_t)row * HD + c] = z; }
    }
    __syncthreads();
#pragma unroll 1
    for (int layer = 1; layer <= 2; ++layer) {
        const float* W = layer == 1 ? a.W1 : a.W2;
        const float* bias = layer == 1 ? a.b1 : a.b2;
        const float* zin = layer == 1 ? zA : zB;
        float* zout = layer == 1 ? zB : zA;
        float* pre = layer == 1 ? a.pre1 : a.pre2;
        float* zsv = layer == 1 ? a.z1 : a.z2;
        f32x4 acc[NT];
#pragma unroll
        for (int x = 0; x < NT; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the weight streams through LDS in chunks of 32 k; the NEXT chunk is requested into registers before this chunk's MFMAs
        constexpr int NV = (HD * KC / 4) / 256;
        f32x4 wreg[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; wreg[i] = *(const f32x4*)(W + (size_t)(idx >> 3) * HD + 4 * (idx & 7)); }
#pragma unroll 1
        for (int k0 = 0; k0 < HD; k0 += KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; *(f32x4*)(wt + (idx >> 3) * WP + 4 * (idx & 7)) = wreg[i]; }
            __syncthreads();
            if (k0 + KC < HD) {
#pragma unroll
                for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; wreg[i] = *(const f32x4*)(W + (size_t)(idx >> 3) * HD + k0 + KC + 4 * (idx & 7)); }
            }
            // step m of this chunk contracts k = k0 + 8 g + m: lane (i, g) holds A = z[i][k0 + 8 g + m] and B = W[col][k0 + 8 g + m]
            const float4 a0 = *(const float4*)(zin + li * ZP + k0 + 8 * lg), a1 = *(const float4*)(zin + li * ZP + k0 + 8 * lg + 4);
            const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int x = 0; x < NT; ++x) {
                const float* wr = wt + (colbase + 16 * x + li) * WP + 8 * lg;
                const float4 b0 = *(const float4*)wr, b1 = *(const float4*)(wr + 4);
                const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[x] = RP_MFMA(av[m], bv[m], acc[x]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 4 * lg + reg, c = colbase + 16 * x + li, row = r0 + r;
                const float s = acc[x][reg] + bias[c], z = silu_f(s);
                zout[r * ZP + c] = z;
                if (pre && row < a.n) { pre[(size_t)row * HD + c] = s; zsv[(size_t)row * HD + c] = z; }
            }
        __syncthreads();
    }
    // last layer: table[row][h] = z2[row] . W3[h] + b3[h]; (row, h) pairs over groups of 4 lanes (each a quarter of k), fixed order
    const float* zf = zA;                   // layer 2 wrote zA
    for (int o = t >> 2; o < RP_RB * a.ldb; o += 64) {
        const int r = o / a.ldb, h = o - r * a.ldb, part = t & 3, row = r0 + r;
        float v = 0.f;
        if (h < a.H) {
            const float* w = a.W3 + (size_t)h * HD + part * (HD / 4);
            const float* z = zf + r * ZP + part * (HD / 4);
            for (int k = 0; k < HD / 4; ++k) v = fmaf(z[k], w[k], v);
        }
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if (part == 0 && row < a.n) a.table[(size_t)row * a.ldb + h] = h < a.H ? v + a.b3[h] : 0.f;
    }
}

// backward, row chain: ds2 = (dtable W3) * silu'(pre2); ds1 = (ds2 W2) * silu'(pre1); ds0 = (ds1 W1) * silu'(pre0).  The weight rows W[c][:]
// (c = the contraction index here) stream through LDS 32 at a time; step m of a chunk contracts c = c0 + 4 m + g.
template <int HD>
__global__ __launch_bounds__(256) void relpos_mlp_bwd_rows_kernel(relpos_mlp_params a) {
    constexpr int KC = 32, ZP = HD + 4, WP2 = HD + 16, NT = HD / 64;       // wt2 pitch HD + 16: rows 4 m + g of the two lane groups of a half-wave sit 16 banks apart
    extern __shared__ __attribute__((aligned(16))) float rp_sm[];
    float* dA = rp_sm;                      // [16][ZP]
    float* dB = dA + RP_RB * ZP;            // [16][ZP]
    float* wt2 = dB + RP_RB * ZP;           // [32][WP2]: weight rows c0 .. c0 + 31
    float* dt = wt2 + KC * WP2;             // [16][16]: the block's dtable rows (H <= 16)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4, r0 = blockIdx.x * RP_RB;
    const int colbase = wave * (HD / 4);
    {
        const int r = t >> 4, h = t & 15, row = r0 + r;
        dt[t] = (h < a.H && row < a.n) ? a.dtable[(size_t)row * a.ldb + h] : 0.f;
    }
    __syncthreads();
    {   // dz2 = dtable W3: contraction over the (padded) 16 heads = 4 steps, h = 4 m + g; W3 rows straight from global (coalesced along k)
        f32x4 acc[NT];
#pragma unroll
        for (int x = 0; x < NT; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int h = 4 * m + lg;
            const float av = dt[li * 16 + h];
#pragma unroll
            for (int x = 0; x < NT; ++x) {
                const float bv = h < a.H ? a.W3[(size_t)h * HD + colbase + 16 * x + li] : 0.f;
                acc[x] = RP_MFMA(av, bv, acc[x]);
            }
        }
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 4 * lg + reg, k = colbase + 16 * x + li, row = r0 + r;
                const float ds = row < a.n ? acc[x][reg] * silu_grad_f(a.pre2[(size_t)row * HD + k]) : 0.f;
                dA[r * ZP + k] = ds;
                if (row < a.n) a.ds2[(size_t)row * HD + k] = ds;
            }
    }
    __syncthreads();
#pragma unroll 1
    for (int layer = 2; layer >= 1; --layer) {
        const float* W = layer == 2 ? a.W2 : a.W1;           // dz_{layer-1}[r][k] = sum_c ds_layer[r][c] W[c][k]
        const float* din = layer == 2 ? dA : dB;
        float* dout = layer == 2 ? dB : dA;
        const float* pre = layer == 2 ? a.pre1 : a.pre0;
        float* dsg = layer == 2 ? a.ds1 : a.ds0;
        f32x4 acc[NT];
#pragma unroll
        for (int x = 0; x < NT; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NV = (KC * HD / 4) / 256;              // float4 pieces of a 32-row chunk per thread
        f32x4 wreg[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; wreg[i] = *(const f32x4*)(W + (size_t)(idx / (HD / 4)) * HD + 4 * (idx % (HD / 4))); }
#pragma unroll 1
        for (int c0 = 0; c0 < HD; c0 += KC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; *(f32x4*)(wt2 + (idx / (HD / 4)) * WP2 + 4 * (idx % (HD / 4))) = wreg[i]; }
            __syncthreads();
            if (c0 + KC < HD) {
#pragma unroll
                for (int i = 0; i < NV; ++i) { const int idx = t + 256 * i; wreg[i] = *(const f32x4*)(W + (size_t)(c0 + KC + idx / (HD / 4)) * HD + 4 * (idx % (HD / 4))); }
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int cc = 4 * m + lg;                   // row of the chunk this lane group contributes in step m
                const float av = din[li * ZP + c0 + cc];
#pragma unroll
                for (int x = 0; x < NT; ++x) acc[x] = RP_MFMA(av, wt2[cc * WP2 + colbase + 16 * x + li], acc[x]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int x = 0; x < NT; ++x)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = 4 * lg + reg, k = colbase + 16 * x + li, row = r0 + r;
                const float ds = row < a.n ? acc[x][reg] * silu_grad_f(pre[(size_t)row * HD + k]) : 0.f;
                dout[r * ZP + k] = ds;
                if (row < a.n) dsg[(size_t)row * HD + k] = ds;
            }
        __syncthreads();
    }
}

// backward, parameter gradients.  blockIdx.x enumerates 64 x 64 output tiles: [0, T2) of gW2 (+ gb2 on its first tile column), [T2, 2 T2) of
// gW1 (+ gb1), then Hd / 64 workgroups for gW3 / gb3 and Hd / 64 for gw0 / gb0.  gW[c][k] += sum_r ds[r][c] z[r][k]: the rows r are the
// contraction (64 staged per trip, step m of a 16-row group takes r = 4 m + g); wave w owns output rows c0 + 16 w .. + 15 and four 16-column
// tiles.  Every output element belongs to one lane and is accumulated over the n rows in a fixed order.
__global__ __launch_bounds__(256) void relpos_mlp_bwd_params_kernel(relpos_mlp_params a) {
    constexpr int RC = 64, SP = 64 + 16;                     // rows per trip; pitch 80 floats: rows 4 m + g of a half-wave's two groups 16 banks apart
    __shared__ __attribute__((aligned(16))) float sd[RC * SP];
    __shared__ __attribute__((aligned(16))) float sz[RC * SP];
    const int Hd = a.Hd, TPD = Hd / 64, T2 = TPD * TPD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lg = lane >> 4;
    int b = blockIdx.x;
    if (b < 2 * T2) {
        const bool second = b >= T2;
        if (second) b -= T2;
        const float* ds = second ? a.ds1 : a.ds2;            // [n, Hd] rows = distances, cols = output feature c
        const float* z = second ? a.z0 : a.z1;               // [n, Hd] cols = input feature k
        float* gW = second ? a.gW1 : a.gW2;
        float* gb = second ? a.gb1 : a.gb2;
        const int c0 = (b / TPD) * 64, k0 = (b % TPD) * 64;
        f32x4 acc[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
        float bacc = 0.f;                                    // thread t < 64 of a k0 == 0 tile: column sum of ds (the bias gradient)
        float4 rd[4], rz[4];                                 // 64 rows x 64 cols of each operand = 1024 float4 / 256 threads, requested one trip ahead
        auto request = [&](int r0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, rr = idx >> 4, q = idx & 15, row = r0 + rr;
                rd[i] = row < a.n ? *(const float4*)(ds + (size_t)row * Hd + c0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
                rz[i] = row < a.n ? *(const float4*)(z + (size_t)row * Hd + k0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        request(0);
        for (int r0 = 0; r0 < a.n; r0 += RC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, rr = idx >> 4, q = idx & 15;
                *(float4*)(sd + rr * SP + 4 * q) = rd[i];
                *(float4*)(sz + rr * SP + 4 * q) = rz[i];
            }
            __syncthreads();
            if (r0 + RC < a.n) request(r0 + RC);
#pragma unroll
            for (int m = 0; m < RC / 4; ++m) {
                const int rr = 4 * m + lg;
                const float av = sd[rr * SP + 16 * wave + li];
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x] = RP_MFMA(av, sz[rr * SP + 16 * x + li], acc[x]);
            }
            if (k0 == 0 && t < 64) {
#pragma unroll 8
                for (int rr = 0; rr < RC; ++rr) bacc += sd[rr * SP + t];
            }
            __syncthreads();
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) gW[(size_t)(c0 + 16 * wave + 4 * lg + reg) * Hd + k0 + 16 * x + li] += acc[x][reg];
        if (k0 == 0 && t < 64) gb[c0 + t] += bacc;
        return;
    }
    b -= 2 * T2;
    if (b < TPD) {
        // gW3[h][k] += sum_r dtable[r][h] z2[r][k] for 64 columns k (H <= 16 rows h); gb3[h] += sum_r dtable[r][h] (first of these
        // workgroups).  Rows staged 16 at a time: thread (h = t >> 4, 4 columns).
        const int k0 = b * 64, h = t >> 4, kx = t & 15;
        float acc[4] = {0.f, 0.f, 0.f, 0.f}, bacc = 0.f;
        // rows staged 64 at a time, the next trip requested under this one's arithmetic (round 6: 16 rows per trip were 70 trips of two barriers
        // + one exposed L2 round trip each -- this branch, 8 workgroups, was the launch's long pole).  Same row order per element: same bits.
        float dq[4], zq[16];
        auto request3 = [&](int r0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = t + 256 * i, rr = idx >> 4, hh = idx & 15, row = r0 + rr;      // 64 rows x 16 (padded) heads
                dq[i] = (row < a.n && hh < a.H) ? a.dtable[(size_t)row * a.ldb + hh] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = t + 256 * i, rr = idx >> 6, cc = idx & 63, row = r0 + rr;
                zq[i] = row < a.n ? a.z2[(size_t)row * Hd + k0 + cc] : 0.f;
            }
        };
        request3(0);
        for (int r0 = 0; r0 < a.n; r0 += RC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int idx = t + 256 * i; sd[(idx >> 4) * SP + (idx & 15)] = dq[i]; }
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int idx = t + 256 * i; sz[(idx >> 6) * SP + (idx & 63)] = zq[i]; }
            __syncthreads();
            if (r0 + RC < a.n) request3(r0 + RC);
#pragma unroll 16
            for (int rr = 0; rr < RC; ++rr) {
                const float dv = sd[rr * SP + h];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = fmaf(dv, sz[rr * SP + kx + 16 * j], acc[j]);
                if (b == 0 && t < 16) bacc += sd[rr * SP + t];
            }
            __syncthreads();
        }
        if (h < a.H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a.gW3[(size_t)h * Hd + k0 + kx + 16 * j] += acc[j];
        }
        if (b == 0 && t < a.H) a.gb3[t] += bacc;
        return;
    }
    b -= TPD;
    {   // gw0[c] += sum_r r ds0[r][c]; gb0[c] += sum_r ds0[r][c]: 64 columns per workgroup, rows 64 at a time, four row lanes combined in a
        // fixed order
        float* red = sd;                                     // [2][4][64]
        const int c0 = b * 64, cx = t & 63, rl = t >> 6;
        float sw = 0.f, sb = 0.f;
        for (int r0 = 0; r0 < a.n; r0 += 64) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) { const int row = r0 + rl + 4 * i; v[i] = row < a.n ? a.ds0[(size_t)row * Hd + c0 + cx] : 0.f; }
#pragma unroll
            for (int i = 0; i < 16; ++i) { sw = fmaf((float)(r0 + rl + 4 * i), v[i], sw); sb += v[i]; }
        }
        red[rl * 64 + cx] = sw; red[256 + rl * 64 + cx] = sb;
        __syncthreads();
        if (rl == 0) {
            a.gw0[c0 + t] += (red[t] + red[64 + t]) + (red[128 + t] + red[192 + t]);
            a.gb0[c0 + t] += (red[256 + t] + red[320 + t]) + (red[384 + t] + red[448 + t]);
        }
    }
}

extern "C" int omlm_relpos_mlp_fwd(const float* w0, const float* b0, const float* W1, const float* b1, const float* W2, const float* b2,
                                   const float* W3, const float* b3, float* pre0, float* z0, float* pre1, float* z1, float* pre2, float* z2,
                                   float* table, int n, int Hd, int H, int ldb, void* stream) {
    OMLM_CHECK_ARG(w0 && b0 && W1 && b1 && W2 && b2 && W3 && b3 && table && n > 0, "relpos_mlp_fwd: null argument");
    OMLM_CHECK_ARG((Hd == 256 || Hd == 512) && H >= 1 && H <= 16 && ldb >= H && ldb <= 16, "relpos_mlp_fwd: Hd must be 256 or 512, H <= 16");
    OMLM_CHECK_ARG(!pre0 || (z0 && pre1 && z1 && pre2 && z2), "relpos_mlp_fwd: save buffers come all or none");
    relpos_mlp_params a;
    memset(&a, 0, sizeof(a));
    a.w0 = w0; a.b0 = b0; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.W3 = W3; a.b3 = b3;
    a.pre0 = pre0; a.z0 = z0; a.pre1 = pre1; a.z1 = z1; a.pre2 = pre2; a.z2 = z2; a.table = table; a.n = n; a.Hd = Hd; a.H = H; a.ldb = ldb;
    const size_t lds = (size_t)(2 * RP_RB * (Hd + 4) + Hd * 36) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)relpos_mlp_fwd_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * RP_RB * 516 + 512 * 36) * 4));
        (void)hipFuncSetAttribute((const void*)relpos_mlp_fwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * RP_RB * 260 + 256 * 36) * 4));
        attr = true;
    }
    const dim3 grid((n + RP_RB - 1) / RP_RB);
    if (Hd == 512) hipLaunchKernelGGL(relpos_mlp_fwd_kernel<512>, grid, dim3(256), lds, as_stream(stream), a);
    else           hipLaunchKernelGGL(relpos_mlp_fwd_kernel<256>, grid, dim3(256), lds, as_stream(stream), a);
    return omlm_post_launch("omlm_relpos_mlp_fwd");
}

// scratch: 3 * n * Hd floats (ds0 | ds1 | ds2).  Gradients are ACCUMULATED into g* (fp32, the optimizer's flat buffer views).
extern "C" int omlm_relpos_mlp_bwd(const float* dtable, const float* W1, const float* W2, const float* W3, const float* pre0, const float* z0,
                                   const float* pre1, const float* z1, const float* pre2, const float* z2, float* scratch, float* gw0, float* gb0,
                                   float* gW1, float* gb1, float* gW2, float* gb2, float* gW3, float* gb3, int n, int Hd, int H, int ldb, void* stream) {
    OMLM_CHECK_ARG(dtable && W1 && W2 && W3 && pre0 && z0 && pre1 && z1 && pre2 && z2 && scratch && gw0 && gb0 && gW1 && gb1 && gW2 && gb2 && gW3 && gb3 && n > 0,
                   "relpos_mlp_bwd: null argument");
    OMLM_CHECK_ARG((Hd == 256 || Hd == 512) && H >= 1 && H <= 16 && ldb >= H && ldb <= 16, "relpos_mlp_bwd: Hd must be 256 or 512, H <= 16");
    relpos_mlp_params a;
    memset(&a, 0, sizeof(a));
    a.W1 = W1; a.W2 = W2; a.W3 = W3; a.pre0 = (float*)pre0; a.z0 = (float*)z0; a.pre1 = (float*)pre1; a.z1 = (float*)z1; a.pre2 = (float*)pre2; a.z2 = (float*)z2;
    a.dtable = dtable; a.ds0 = scratch; a.ds1 = scratch + (size_t)n * Hd; a.ds2 = scratch + 2 * (size_t)n * Hd;
    a.gw0 = gw0; a.gb0 = gb0; a.gW1 = gW1; a.gb1 = gb1; a.gW2 = gW2; a.gb2 = gb2; a.gW3 = gW3; a.gb3 = gb3; a.n = n; a.Hd = Hd; a.H = H; a.ldb = ldb;
    const size_t lds = (size_t)(2 * RP_RB * (Hd + 4) + 32 * (Hd + 16) + 256) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)relpos_mlp_bwd_rows_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * RP_RB * 516 + 32 * 528 + 256) * 4));
        (void)hipFuncSetAttribute((const void*)relpos_mlp_bwd_rows_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((2 * RP_RB * 260 + 32 * 272 + 256) * 4));
        attr = true;
    }
    const dim3 grid((n + RP_RB - 1) / RP_RB);
    if (Hd == 512) hipLaunchKernelGGL(relpos_mlp_bwd_rows_kernel<512>, grid, dim3(256), lds, as_stream(stream), a);
    else           hipLaunchKernelGGL(relpos_mlp_bwd_rows_kernel<256>, grid, dim3(256), lds, as_stream(stream), a);
    int rc = omlm_post_launch("omlm_relpos_mlp_bwd (rows)");
    if (rc) return rc;
    const int tpd = Hd / 64;
    hipLaunchKernelGGL(relpos_mlp_bwd_params_kernel, dim3(2 * tpd * tpd + 2 * tpd), dim3(256), 0, as_stream(stream), a);
    return omlm_post_launch("omlm_relpos_mlp_bwd (parameters)");
}

// ---------------------------------------------------------------------------------------------------------
// Residual-VQ / k-means nearest-codeword (clap_quantized.py:75-87 -> ResidualVQ eval path; hf_hubert_kmeans.py:87).
// cbT is the codebook TRANSPOSED: [n_stage][D][C], so that consecutive threads (codes) read consecutive addresses.  Two stated
// distance forms, both in fp32 with every multiply and add rounded separately (no FMA contraction), d in index order:
//   FORM_SQ    dist(c) = sum_d (x_d - e_{c,d})^2                                  (oracle.nearest_code; the k-means assign step,
//              pinned bit for bit against sklearn.MiniBatchKMeans.predict at the shipped dimensions)
//   FORM_CDIST dist(c) = sqrt(max((x2 + e2_c) - 2 * xy_c, 0)),  x2 = sum x_d^2, e2_c = sum e_{c,d}^2, xy_c = sum x_d e_{c,d}
//              (oracle.nearest_code_cdist) -- the form of vector-quantize-pytorch's EuclideanCodebook (`-cdist(x, embed)`, then
//              argmax = first maximum): the expanded euclidean distance, square root INCLUDED (it merges distances that differ by
//              less than a rounding of the root into ties, which the lowest index then wins).  Pinned against torch.cdist itself:
//              on inputs whose products and sums are exact in fp32 every summation order gives the same bits, and the ids agree
//              bit for bit (tests); on general inputs a BLAS's summation order is its own, and so are the library's ids.
// argmin with ties -> lowest index; then r <- r - e_idx (one fp32 subtraction per element).
enum { FORM_SQ = 0, FORM_CDIST = 1 };
template <int FORM>
__global__ __launch_bounds__(256) void rvq_kernel(const float* __restrict__ x, const float* __restrict__ cbT,
                                                  int* __restrict__ idx_out, float* __restrict__ resid_out,
                                                  int n, int D, int C, int nstage, int idx_stride) {
    extern __shared__ float r[];                  // [D] running residual
    __shared__ float bd[4];
    __shared__ int bi[4];
    __shared__ int chosen;
    const int row = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += 256) r[d] = x[(size_t)row * D + d];
    __syncthreads();
    for (int s = 0; s < nstage; ++s) {
        const float* cb = cbT + (size_t)s * D * C;
        float best = INFINITY;
        int besti = 0x7fffffff;
        float x2 = 0.f;
        if (FORM == FORM_CDIST)                   // every thread forms the same sequential sum (LDS broadcast reads)
            for (int d = 0; d < D; ++d) x2 = __fadd_rn(x2, __fmul_rn(r[d], r[d]));
        for (int c = threadIdx.x; c < C; c += 256) {
            float dist;
            if (FORM == FORM_SQ) {
                dist = 0.f;
                for (int d = 0; d < D; ++d) {
                    const float diff = __fsub_rn(r[d], cb[(size_t)d * C + c]);
                    dist = __fadd_rn(dist, __fmul_rn(diff, diff));
                }
            } else {
                float xy = 0.f, e2 = 0.f;
                for (int d = 0; d < D; ++d) {
                    const float e = cb[(size_t)d * C + c];
                    xy = __fadd_rn(xy, __fmul_rn(r[d], e));
                    e2 = __fadd_rn(e2, __fmul_rn(e, e));
                }
                // IEEE root: through fp64 (a correctly rounded fp64 root rounded to fp32 IS the correctly rounded fp32 root: 53 >= 2 * 24 + 2);
                // HIP's __fsqrt_rn is the native 1-ulp v_sqrt_f32, which keeps root-merged near-ties apart
                dist = (float)sqrt((double)fmaxf(__fsub_rn(__fadd_rn(x2, e2), __fmul_rn(2.f, xy)), 0.f));
            }
            if (dist < best) { best = dist; besti = c; }      // ascending c per thread: strict < keeps the lowest index
        }
        // lexicographic (dist, index) min over the block
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float od = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(besti, o, 64);
            if (od < best || (od == best && oi < besti)) { best = od; besti = oi; }
        }
        if ((threadIdx.x & 63) == 0) { bd[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = besti; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float b = bd[0]; int i0 = bi[0];
            for (int w = 1; w < 4; ++w) if (bd[w] < b || (bd[w] == b && bi[w] < i0)) { b = bd[w]; i0 = bi[w]; }
            chosen = i0;
            idx_out[(size_t)row * idx_stride + s] = i0;
        }
        __syncthreads();
        const int ci = chosen;
        for (int d = threadIdx.x; d < D; d += 256) r[d] = __fsub_rn(r[d], cb[(size_t)d * C + ci]);
        __syncthreads();
    }
    if (resid_out) for (int d = threadIdx.x; d < D; d += 256) resid_out[(size_t)row * D + d] = r[d];
}

static int rvq_launch(int form, const float* x, const float* cbT, int* indices, float* residual_out, int n, int D, int C,
                      int nstage, int idx_stride, void* stream, const char* what) {
    if (n <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && cbT && indices && D > 0 && C > 0 && nstage > 0 && idx_stride >= 1, "rvq arguments");
    OMLM_CHECK_ARG((size_t)D * sizeof(float) <= 48 * 1024, "D too large");
    if (form == FORM_CDIST)
        hipLaunchKernelGGL(rvq_kernel<FORM_CDIST>, dim3(n), dim3(256), D * sizeof(float), as_stream(stream), x, cbT, indices, residual_out, n, D, C, nstage, idx_stride);
    else
        hipLaunchKernelGGL(rvq_kernel<FORM_SQ>, dim3(n), dim3(256), D * sizeof(float), as_stream(stream), x, cbT, indices, residual_out, n, D, C, nstage, idx_stride);
    return omlm_post_launch(what);
}
// residual-VQ chain of nstage codebooks in the library's distance form (FORM_CDIST)
extern "C" int omlm_rvq_encode(const float* x, const float* codebooks_T, int* indices, float* residual_out,
                               int n, int D, int C, int nstage, void* stream) {
    return rvq_launch(FORM_CDIST, x, codebooks_T, indices, residual_out, n, D, C, nstage, nstage, stream, "omlm_rvq_encode");
}
// one stage, index of row i written to indices[i * idx_stride] (a column of an [n, stages] table: the RVQ fit step walks the layers
// one launch at a time because every layer's codebook changes between its assignment and the next layer's); same form
extern "C" int omlm_rvq_encode_strided(const float* x, const float* codebook_T, int* indices, int idx_stride, float* residual_out,
                                       int n, int D, int C, void* stream) {
    return rvq_launch(FORM_CDIST, x, codebook_T, indices, residual_out, n, D, C, 1, idx_stride, stream, "omlm_rvq_encode_strided");
}
// k-means assign (hf_hubert_kmeans.py:87): squared-difference form (FORM_SQ)
extern "C" int omlm_nearest_centroid(const float* x, const float* centroids_T, int* indices, int n, int D, int C, void* stream) {
    return rvq_launch(FORM_SQ, x, centroids_T, indices, nullptr, n, D, C, 1, 1, stream, "omlm_nearest_centroid");
}

// ---------------------------------------------------------------------------------------------------------
// fused sampler of the AR loop (open_musiclm.py:309-316; utils.py:65-84): last-position logits [B, V] ->
//   eos logit -> -inf (unless allowed), keep the k = max(int((1-thres) V), 1) largest logits, argmax(l / T + Gumbel(u)).
// One workgroup per row; V <= 2048.  The k-th largest value is found by a bitwise radix descent on the
// order-preserving integer image of the floats (exact, no sort); ties at the threshold are kept in index order
// like torch.topk + scatter (which keeps exactly k entries: the lowest indices among equals).
__device__ __forceinline__ unsigned f_ord(float f) { unsigned u = f2u(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// One WAVE per row: the row's logits sit in registers (V <= 2048 -> <= 32 per lane, element c = lane + 64 j), every count of the
// radix descent is a ballot + popcount on the scalar unit -- no LDS, no barrier.  (The first version used 256 threads, an LDS
// image and two barriers per bit: ~25 us of the ~200 us a sampled id costs at B = 1.)  Optionally gathers the embedding row of the
// sampled id (open_musiclm.py:123-134: id + quantizer offset) into x, so the decode step needs no separate gather launch.
// SAMPLE_NV (template): register slots per lane, 17 for V <= 1088 (the 1025-entry heads of every shipped model), 32 up to 2048.  A
// single wave is a serial instruction stream (~5-8 cycles per dependent instruction): the round-4 trace showed 43.9 us per call with the
// slot loops unrolled to 32 behind `j < nv` branches (32 bits x 32 slots of compare / branch / count), so the slot count is a compile-time
// constant and the descent stops at the first threshold that cuts exactly k keys.
template <int SAMPLE_NV>
__global__ __launch_bounds__(64) void sample_kernel(const float* __restrict__ logits, const float* __restrict__ uniform,
                                                    long long* __restrict__ out, int V, int ld, int k, float temperature,
                                                    int forbid_last, const int* __restrict__ step_dev, long long* __restrict__ hist,
                                                    const float* __restrict__ emb_table, long long emb_row_offset, long long emb_rows,
                                                    float* __restrict__ x, int D) {
    if (step_dev) {          // graph-replayable form: this step's uniforms / history slot are selected by a DEVICE counter
        const long long sidx = step_dev[0];
        uniform += sidx * (long long)gridDim.x * V;
        if (hist) hist += sidx * gridDim.x;
    }
    const int row = blockIdx.x, lane = threadIdx.x;
    const float* lr = logits + (size_t)row * ld;
    const float* ur = uniform + (size_t)row * V;
    unsigned keys[SAMPLE_NV];
    float lv[SAMPLE_NV], uv[SAMPLE_NV];
    constexpr int nv = SAMPLE_NV;                      // every slot is live or clamped: no per-slot branches
    // every load of the row -- logits AND uniforms -- is requested before anything waits.  (Round 4 kernel trace: 43.9 us per call at
    // B = 1, a quarter of a decode step: the uniforms were loaded inside `if (keep)`, one dependent memory round trip per register
    // slot, 17 in a row behind the running arg-max.)
    // (hipcc sank each uniform's first log next to its predicated load, with a vmcnt(0) in between -- seen in the ISA: the loads are
    // therefore unconditional (clamped index, no branch) and all consumed by the empty asm below before any arithmetic.)
#pragma unroll
    for (int j = 0; j < SAMPLE_NV; ++j) {
        const int c = lane + 64 * j, cc = c < V ? c : V - 1;
        lv[j] = lr[cc];
        uv[j] = ur[cc];
    }
#pragma unroll
    for (int j = 0; j < SAMPLE_NV; ++j) asm volatile("" : "+v"(lv[j]), "+v"(uv[j]));
#pragma unroll
    for (int j = 0; j < SAMPLE_NV; ++j) {
        const int c = lane + 64 * j;
        float v = c < V ? lv[j] : -INFINITY;
        if (forbid_last && c == V - 1) v = -INFINITY;
        lv[j] = v;
        keys[j] = c < V ? f_ord(v) : 0u;               // 0 sorts below every real key (f_ord(-inf) = 0x007fffff)
    }
    // largest threshold t such that count(keys >= t) >= k
    unsigned t = 0;
    bool exact = false;                                 // count(keys >= t) == k: the kept set is exactly {keys >= t}
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = t | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < SAMPLE_NV; ++j) cnt += __popcll(__ballot(keys[j] >= cand));
        if (cnt >= k) t = cand;
        if (cnt == k) { exact = true; break; }
    }
    // strictly-greater entries are all kept; of the entries equal to t keep the first (k - n_greater) by index
    int ng = 0;
    if (!exact) {
#pragma unroll
        for (int j = 0; j < SAMPLE_NV; ++j) ng += __popcll(__ballot(keys[j] > t));
    }
    const int n_equal_keep = exact ? 0x7fffffff : k - ng;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    int seen_eq = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < SAMPLE_NV; ++j) {
        const int c = lane + 64 * j;
        const bool in = c < V;
        const bool eq = in && keys[j] == t;
        const unsigned long long eqmask = __ballot(eq);
        const int rank = seen_eq + __popcll(eqmask & below);
        const bool keep = in && (keys[j] > t || (eq && rank < n_equal_keep));
        seen_eq += __popcll(eqmask);
        // branch-free: the Gumbel term of every slot is formed (2 logs per slot), dead slots lose the comparison
        const float gum = -logf(-logf(uv[j] + 1e-20f) + 1e-20f);
        const float v = keep ? lv[j] / temperature + gum : -INFINITY;
        if (v > best) { best = v; besti = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { out[row] = besti; if (hist) hist[row] = besti; }
    if (emb_table) {
        long long r = (long long)besti + emb_row_offset;
        r = r < 0 ? 0 : (r >= emb_rows ? emb_rows - 1 : r);
        const float4* src = (const float4*)(emb_table + r * D);
        float4* dst = (float4*)(x + (size_t)row * D);
        for (int i = lane; i < D / 4; i += 64) dst[i] = src[i];
    }
}

extern "C" int omlm_sample_topk_gumbel(const float* logits, const float* uniform, long long* out, int B, int V, int ld,
                                       int k, float temperature, int forbid_last, void* stream) {
    if (B <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(logits && uniform && out && V > 0 && V <= 2048 && k >= 1 && k <= V && temperature > 0.f, "sampler arguments");
    if (V <= 64 * 17) hipLaunchKernelGGL(sample_kernel<17>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform, out, V, ld, k, temperature, forbid_last,
                                         (const int*)nullptr, (long long*)nullptr, (const float*)nullptr, 0ll, 0ll, (float*)nullptr, 0);
    else hipLaunchKernelGGL(sample_kernel<32>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform, out, V, ld, k, temperature, forbid_last,
                            (const int*)nullptr, (long long*)nullptr, (const float*)nullptr, 0ll, 0ll, (float*)nullptr, 0);
    return omlm_post_launch("omlm_sample_topk_gumbel");
}

// Same sampler for a captured decode step: uniforms [steps, B, V] and the id history [steps, B] are indexed by *step_dev.
extern "C" int omlm_sample_topk_gumbel_at(const float* logits, const float* uniform_base, const int* step_dev, long long* out,
                                          long long* hist, int B, int V, int ld, int k, float temperature, int forbid_last,
                                          void* stream) {
    if (B <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(logits && uniform_base && step_dev && out && V > 0 && V <= 2048 && k >= 1 && k <= V && temperature > 0.f, "sampler arguments");
    if (V <= 64 * 17) hipLaunchKernelGGL(sample_kernel<17>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform_base, out, V, ld, k, temperature,
                                         forbid_last, step_dev, hist, (const float*)nullptr, 0ll, 0ll, (float*)nullptr, 0);
    else hipLaunchKernelGGL(sample_kernel<32>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform_base, out, V, ld, k, temperature,
                            forbid_last, step_dev, hist, (const float*)nullptr, 0ll, 0ll, (float*)nullptr, 0);
    return omlm_post_launch("omlm_sample_topk_gumbel_at");
}

// Sampler + embedding gather of the sampled id (x[b, :] = emb_table[id_b + emb_row_offset], rows clamped to [0, emb_rows)):
// the first launch of the next decode step folded into the sampler (then omlm_decode_step runs with emb_table == NULL).
extern "C" int omlm_sample_embed_at(const float* logits, const float* uniform_base, const int* step_dev, long long* out,
                                    long long* hist, int B, int V, int ld, int k, float temperature, int forbid_last,
                                    const float* emb_table, long long emb_row_offset, long long emb_rows, float* x, int D,
                                    void* stream) {
    if (B <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(logits && uniform_base && step_dev && out && V > 0 && V <= 2048 && k >= 1 && k <= V && temperature > 0.f, "sampler arguments");
    OMLM_CHECK_ARG(emb_table && x && D > 0 && D % 4 == 0 && emb_rows > 0, "embedding arguments");
    if (V <= 64 * 17) hipLaunchKernelGGL(sample_kernel<17>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform_base, out, V, ld, k, temperature,
                                         forbid_last, step_dev, hist, emb_table, emb_row_offset, emb_rows, x, D);
    else hipLaunchKernelGGL(sample_kernel<32>, dim3(B), dim3(64), 0, as_stream(stream), logits, uniform_base, out, V, ld, k, temperature,
                            forbid_last, step_dev, hist, emb_table, emb_row_offset, emb_rows, x, D);
    return omlm_post_launch("omlm_sample_embed_at");
}

// ---------------------------------------------------------------------------------------------------------
// Hardware probe used by tests/test_gpu_probe.py: dumps what ds_read_b64_tr_b16 returns for a linear LDS image
// (lds[i] = i as 16-bit) when lane l supplies byte address 8*l, so the transpose-read assumption that the GEMM
// and attention fragment loaders are built on is checked on the real chip.
__global__ void probe_tr16_kernel(short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, (char*)lds + threadIdx.x * 8));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
extern "C" int omlm_probe_tr16(short* out, void* stream) {
    OMLM_CHECK_ARG(out, "null");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, as_stream(stream), out);
    return omlm_post_launch("omlm_probe_tr16");
}
