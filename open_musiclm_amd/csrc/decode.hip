// KV-cached autoregressive decode step of the TokenConditionedTransformer trunk.
//
// The reference samples by re-running the full causal forward over the grown sequence for every new id
// (open_musiclm.py:301-321: no KV cache).  Every op of the trunk is causal -- attention (transformer.py:303-331), the
// depthwise conv of ConvFeedForward (left pad 2, :122-137), LayerNorm per token -- so row p depends on rows <= p only,
// and the same logits are obtained by computing ONE new row per step against
//   * the l2-normalised keys / values of rows < p          (K/V cache, fp32, [B, Nmax, 64] per layer), and
//   * the FF-in outputs of rows p-2, p-1                    (conv state,  fp32, [B, 2, 2*Fp] per layer).
// One step is weight-bandwidth and latency bound (all 91.6 M parameters are read once for up to 8 samples), so:
//   * the projections are skinny GEMVs on the vector ALUs, 16 weight rows per workgroup (hundreds of workgroups keep
//     enough loads in flight); the B activation rows sit in LDS and LayerNorm is recomputed per workgroup;
//   * attention is split over 64-key ranges (the single K/V head is read once for all heads: MQA) into partial
//     (max, sum, o) triples that the out-projection kernel combines while staging its activation;
//   * the row index lives in DEVICE memory (*pos_dev), grids and LDS sizes do not depend on it, so a whole step
//     (31 launches + sampler + advance) can be captured once into a HIP graph and replayed per sampled id.
//
// TW = h16_t: weights are the bf16 operand copies ("bf16" mode; activations that the batched path rounds to bf16 before
// its GEMMs are rounded here too, so both paths see the same operands); TW = float: fp32 weights, fp32 FMA chains.
#include "common.h"
// The step kernels are single dependent chains (load -> LayerNorm statistics -> dot products -> reduction -> store): the wave
// reductions run on the VALU (DPP row steps + readlanes) instead of six dependent ds_bpermute round trips each.  Measured on the
// B = 1 step (MI355X, tools/decode_probe.py, same box): 154.6 -> 145.8 us per id with the three reductions of the GEMV kernels alone.
// -DDEC_DPP_SUM=0 restores the __shfl ladders.
#ifndef DEC_DPP_SUM
#define DEC_DPP_SUM 1
#endif
#if DEC_DPP_SUM
#define wave_sum(x) wave_sum_dpp(x)
#define wave_max(x) wave_max_dpp(x)
#endif
#include <stdlib.h>
#include <type_traits>

namespace OMLM_NS {

#define DEC_T 256
#define DEC_BMAX 8
#define DEC_ROWS 16            // weight rows per workgroup (4 per wave)
#define DEC_KS 64              // keys per attention split
#define DEC_PART 66            // floats per (split, head) partial: max, sum, o[64]

struct omlm_decode_args {
    int B, D, H, L, F, Fp, Nmax, w_dtype, round_bf16, nsplit;
    float eps, scale;
    const int* pos_dev;
    const void* const* Wq; const void* const* Wkv; const void* const* Wo; const void* const* W1p; const void* const* W2p;
    const float* const* attn_gamma; const float* const* q_scale; const float* const* k_scale;
    const float* const* ffin_gamma; const float* const* convw; const float* const* mid_gamma;
    float* const* Kc; float* const* Vc; float* const* hist;
    const float* bias_table; int bias_ld;
    const float* final_gamma; const void* head_W; int V1; int ldV;
    const float* emb_table; long long emb_row_offset; long long emb_rows;
    float* x; float* x1; float* q; float* parts; float* u; float* logits;
    int* advance_pos; int* advance_step;
    float* ln_parts;
    // precision "fp16ff": lo planes of the FF-in / FF-out / head weights (W = hi + lo, fp32-grade); with them the three launches keep their
    // activation rows and h1 un-rounded, like the three-product forward of the batched path (W1p_lo == NULL: plain 16-bit step)
    const void* const* W1p_lo; const void* const* W2p_lo; const void* head_W_lo;
    // split-K scratch of the batched FF-out launch (optional; NULL: one workgroup per 16 output rows walks the whole row):
    // splitk_ws 4 * ceil(D / 16) * 256 floats, contents irrelevant; splitk_cnt ceil(D / 16) ints, ZERO before the first step (every launch
    // leaves them zero)
    float* splitk_ws; int* splitk_cnt;
};

__device__ __forceinline__ float round_if(float v, int on) { return on ? (float)(h16_t)v : v; }

__device__ __forceinline__ void load_w8(const float* p, float* w) {
    const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void load_w8(const h16_t* p, float* w) {
    const u32x4 a = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[2 * i] = h16_lo_to_f(a[i]); w[2 * i + 1] = h16_hi_to_f(a[i]); }
}

// vals[r * DEC_BMAX + b] = sum_k W[r, k] * xs[b * Kp + k]  for r < nrows <= 16 (weight rows of pitch ldw), b < B.
// Each wave owns 4 rows (4 independent load streams per lane); a lane owns the 8-element chunks lane, lane+64, ...
// Wlo (optional, uniform): the weights' lo plane -- the row is W + Wlo, summed in fp32
template <typename TW>
__device__ void wg_gemv(const TW* __restrict__ W, long long ldw, int K, int nrows, const float* xs, int Kp, int B, float* vals,
                        const TW* __restrict__ Wlo = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = K >> 3;
    const int r0 = wave * 4;
    if (r0 >= nrows) return;
    float acc[4][DEC_BMAX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < DEC_BMAX; ++b) acc[i][b] = 0.f;
    const TW* wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = W + (long long)min(r0 + i, nrows - 1) * ldw;     // clamped rows are dropped below
    const TW* wrl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wrl[i] = (Wlo ? Wlo : W) + (long long)min(r0 + i, nrows - 1) * ldw;
#pragma unroll 2
    for (int c = lane; c < nch; c += 64) {
        float w[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) load_w8(wr[i] + c * 8, w[i]);
        if (Wlo) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wl[8];
                load_w8(wrl[i] + c * 8, wl);
#pragma unroll
                for (int e = 0; e < 8; ++e) w[i][e] += wl[e];
            }
        }
#pragma unroll
        for (int b = 0; b < DEC_BMAX; ++b) {
            if (b < B) {
                const float4 x0 = *(const float4*)(xs + (size_t)b * Kp + c * 8), x1 = *(const float4*)(xs + (size_t)b * Kp + c * 8 + 4);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i][b] += w[i][0] * x0.x + w[i][1] * x0.y + w[i][2] * x0.z + w[i][3] * x0.w +
                                 w[i][4] * x1.x + w[i][5] * x1.y + w[i][6] * x1.z + w[i][7] * x1.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < DEC_BMAX; ++b) {
            if (b < B) {
                const float s = wave_sum(acc[i][b]);
                if (lane == 0 && r0 + i < nrows) vals[(r0 + i) * DEC_BMAX + b] = s;
            }
        }
}

// xs[b][0..K) <- in[b][0..K); optional LayerNorm over the first Kstat entries (gamma has K entries, 0 in any padding)
__device__ void stage_activation(const float* __restrict__ in, int ldin, int K, int Kstat, const float* __restrict__ gamma, float eps,
                                 int B, int round_bf16, float* xs, float* red) {
    for (int b = 0; b < B; ++b) {
        float s = 0.f;
        for (int i = threadIdx.x; i < K; i += DEC_T) {
            const float v = in[(size_t)b * ldin + i];
            xs[(size_t)b * K + i] = v;
            if (i < Kstat) s += v;
        }
        if (gamma) {
            const float mean = block_sum<DEC_T>(s, red) / (float)Kstat;
            float q = 0.f;
            for (int i = threadIdx.x; i < Kstat; i += DEC_T) { const float d = xs[(size_t)b * K + i] - mean; q += d * d; }
            const float rstd = rsqrtf(block_sum<DEC_T>(q, red) / (float)Kstat + eps);
            for (int i = threadIdx.x; i < K; i += DEC_T)
                xs[(size_t)b * K + i] = round_if((xs[(size_t)b * K + i] - mean) * rstd * gamma[i], round_bf16);
        } else if (round_bf16) {
            for (int i = threadIdx.x; i < K; i += DEC_T) xs[(size_t)b * K + i] = round_if(xs[(size_t)b * K + i], 1);
        }
    }
    __syncthreads();
}

// xs[b][h*64 + d] <- softmax-combine of the attention partials of the splits covering keys 0..pos
__device__ void stage_attention(const float* __restrict__ parts, int nsplit, int H, int pos, int B, int round_bf16, float* xs) {
    const int ns = pos / DEC_KS + 1;
    for (int idx = threadIdx.x; idx < B * H * 64; idx += DEC_T) {
        const int b = idx / (H * 64), hd = idx - b * H * 64, h = hd >> 6, d = hd & 63;
        const float* pb = parts + ((size_t)b * nsplit * H + h) * DEC_PART;
        float m = -3.0e38f;
        for (int s = 0; s < ns; ++s) m = fmaxf(m, pb[(size_t)s * H * DEC_PART]);
        float l = 0.f, o = 0.f;
        for (int s = 0; s < ns; ++s) {
            const float* p = pb + (size_t)s * H * DEC_PART;
            const float w = __expf(p[0] - m);
            l += w * p[1];
            o += w * p[2 + d];
        }
        xs[idx] = round_if(o / l, round_bf16);
    }
    __syncthreads();
}

// ---- embedding gather of the ids sampled in the previous step (open_musiclm.py:123-134: id + quantizer offset) ----
// stat_out (optional): the row's (sum, sum of squares) as LayerNorm partial 0 of sample b (layout [partial][8 samples][2], see dec4_kernel)
__global__ __launch_bounds__(DEC_T) void dec_embed_kernel(const long long* __restrict__ ids, const float* __restrict__ table,
                                                         long long row_off, long long rows, float* __restrict__ x, int D,
                                                         float* __restrict__ stat_out) {
    __shared__ float red[2 * (DEC_T / 64)];
    const int b = blockIdx.x;
    long long r = ids[b] + row_off;
    r = r < 0 ? 0 : (r >= rows ? rows - 1 : r);
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < D; i += DEC_T) { const float v = table[r * D + i]; x[(size_t)b * D + i] = v; s += v; q += v * v; }
    if (!stat_out) return;
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s; red[2 * (threadIdx.x >> 6) + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tq = 0.f;
        for (int w = 0; w < DEC_T / 64; ++w) { ts += red[2 * w]; tq += red[2 * w + 1]; }
        stat_out[b * 2] = ts; stat_out[b * 2 + 1] = tq;
    }
}

// (sum, sum of squares) of each sample's row of x as LayerNorm partial 0 (rows embedded by the caller)
__global__ __launch_bounds__(DEC_T) void dec_rowstat_kernel(const float* __restrict__ x, int D, float* __restrict__ stat_out) {
    __shared__ float red[2 * (DEC_T / 64)];
    const int b = blockIdx.x;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < D; i += DEC_T) { const float v = x[(size_t)b * D + i]; s += v; q += v * v; }
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s; red[2 * (threadIdx.x >> 6) + 1] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tq = 0.f;
        for (int w = 0; w < DEC_T / 64; ++w) { ts += red[2 * w]; tq += red[2 * w + 1]; }
        stat_out[b * 2] = ts; stat_out[b * 2 + 1] = tq;
    }
}

// ---- A: raw projections of the new row: q_raw = LN(x) Wq^T -> q;  [k_raw | v] = x Wkv^T -> cache row pos ----
// (transformer.py:228,250-254: keys / values are projected from the UN-normalised residual.)  The l2 normalisations
// (:265-271) happen where the 64 dims of a head meet again: in the attention kernel.
template <typename TW>
__global__ __launch_bounds__(DEC_T) void dec_qkv_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const TW* __restrict__ Wq, const TW* __restrict__ Wkv,
                                                       float* __restrict__ q, float* __restrict__ Kc, float* __restrict__ Vc,
                                                       int B, int D, int H, int Nmax, const int* __restrict__ pos_dev, float eps,
                                                       int round_bf16) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* xs = dsm;                                   // [B][D]
    float* vals = xs + (size_t)B * D;                  // [16][DEC_BMAX]
    float* red = vals + DEC_ROWS * DEC_BMAX;           // [8]
    const int pos = *pos_dev;
    const int n0 = blockIdx.x * DEC_ROWS, HD = H * 64;
    const bool isq = n0 < HD;
    stage_activation(x, D, D, D, isq ? gamma : nullptr, eps, B, round_bf16, xs, red);
    const TW* W = isq ? Wq + (size_t)n0 * D : Wkv + (size_t)(n0 - HD) * D;
    wg_gemv<TW>(W, D, D, DEC_ROWS, xs, D, B, vals);
    __syncthreads();
    for (int idx = threadIdx.x; idx < B * DEC_ROWS; idx += DEC_T) {
        const int b = idx / DEC_ROWS, r = idx - b * DEC_ROWS, n = n0 + r;
        const float v = vals[r * DEC_BMAX + b];
        if (isq) q[(size_t)b * HD + n] = v;
        else if (n < HD + 64) Kc[((size_t)b * Nmax + pos) * 64 + (n - HD)] = v;                       // raw: normalised by dec_attn
        else Vc[((size_t)b * Nmax + pos) * 64 + (n - HD - 64)] = round_if(v, round_bf16);
    }
}

// ---- B1: attention partials of one 64-key range for ALL heads (the single K/V head is read once) ----
//   part[b][s][h] = { m = max_j s_j,  l = sum_j e^(s_j - m),  o[d] = sum_j e^(s_j - m) v_j[d] },  s_j = scale <q_h, k_j> + bias[pos - j, h]
// The split that holds row `pos` finds it un-normalised (written by dec_qkv this step): it l2-normalises it, uses it and
// stores it back, so the cache holds final keys from then on.  q is l2-normalised here as well.
__global__ __launch_bounds__(DEC_T) void dec_attn_kernel(const float* __restrict__ q, float* __restrict__ Kc,
                                                        const float* __restrict__ Vc, const float* __restrict__ q_scale,
                                                        const float* __restrict__ k_scale, const float* __restrict__ bias, int bias_ld,
                                                        float* __restrict__ parts, int H, int Nmax, int nsplit,
                                                        const int* __restrict__ pos_dev, float scale, int round_bf16) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int pos = *pos_dev;
    const int s = blockIdx.x, b = blockIdx.y;
    const int j0 = s * DEC_KS;
    if (j0 > pos) return;
    const int nk = min(DEC_KS, pos + 1 - j0);
    float* Ks = dsm;                        // [64][65]
    float* Vs = Ks + 64 * 65;               // [64][64]
    float* qn = Vs + 64 * 64;               // [H][64]
    float* sc = qn + H * 64;                // [H][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int h = wave; h < H; h += 4) {                                   // q: l2norm * q_scale per head (utils.py:68-69)
        const float v = q[(size_t)b * H * 64 + h * 64 + lane];
        const float nrm = fmaxf(sqrtf(wave_sum(v * v)), 1e-12f);
        qn[h * 64 + lane] = round_if(v / nrm * q_scale[lane], round_bf16);
    }
    float* Kb = Kc + ((size_t)b * Nmax + j0) * 64;
    const float* Vb = Vc + ((size_t)b * Nmax + j0) * 64;
    for (int idx = threadIdx.x; idx < nk * 16; idx += DEC_T) {            // coalesced float4 rows
        const int j = idx >> 4, c = idx & 15;
        const float4 kk = ((const float4*)(Kb + (size_t)j * 64))[c];
        const float4 vv = ((const float4*)(Vb + (size_t)j * 64))[c];
        Ks[j * 65 + 4 * c] = kk.x; Ks[j * 65 + 4 * c + 1] = kk.y; Ks[j * 65 + 4 * c + 2] = kk.z; Ks[j * 65 + 4 * c + 3] = kk.w;
        *(float4*)(Vs + j * 64 + 4 * c) = vv;
    }
    __syncthreads();
    if (pos - j0 < DEC_KS && wave == 0) {                                 // the new key: normalise once, keep it in the cache
        const int j = pos - j0;
        const float v = Ks[j * 65 + lane];
        const float nrm = fmaxf(sqrtf(wave_sum(v * v)), 1e-12f);
        const float kn = round_if(v / nrm * k_scale[lane], round_bf16);
        Ks[j * 65 + lane] = kn;
        Kb[(size_t)j * 64 + lane] = kn;
    }
    __syncthreads();
    for (int h = wave; h < H; h += 4) {                                   // one wave per head: lane = key
        float dot = 0.f;
        if (lane < nk) {
#pragma unroll 16
            for (int d = 0; d < 64; ++d) dot += qn[h * 64 + d] * Ks[lane * 65 + d];
        }
        const float sv = lane < nk ? dot * scale + (bias ? bias[(size_t)(pos - j0 - lane) * bias_ld + h] : 0.f) : -3.0e38f;
        const float m = wave_max(sv);
        const float p = lane < nk ? __expf(sv - m) : 0.f;
        const float l = wave_sum(p);
        sc[h * 64 + lane] = p;
        if (lane == 0) {
            float* pp = parts + (((size_t)b * nsplit + s) * H + h) * DEC_PART;
            pp[0] = m; pp[1] = l;
        }
    }
    __syncthreads();
    for (int h = wave; h < H; h += 4) {                                   // lane = dim
        float acc = 0.f;
        for (int j = 0; j < nk; ++j) acc += sc[h * 64 + j] * Vs[j * 64 + lane];
        parts[(((size_t)b * nsplit + s) * H + h) * DEC_PART + 2 + lane] = acc;
    }
}

// ---- B1, second generation: the same partials from 8 waves (one head per wave and pass) with ONE barrier: every global load
// (q, the 64-key K / V tiles, scales) is requested at the top; the new key is l2-normalised in registers by the 16 lanes that
// loaded it (sum of squares through 4 lane swaps) before it goes to LDS and back to the cache; the probabilities reach the
// P.V loop through the wave's own LDS row (same-wave LDS ordering, no barrier).
#define DEC_AT2 512
// comb_out / comb_cnt (optional; the batched matrix-core step): the workgroups of a sample count their arrivals in comb_cnt[b] (zero on
// entry and exit) and the last one combines the sample's partials into comb_out[b, H * 64] -- what dec_attn_combine_kernel did as a
// launch of its own (5 us of every layer's ~50).  comb_out may be q's buffer: every reader of q[b, :] has arrived by then.
__global__ __launch_bounds__(DEC_AT2) void dec_attn2_kernel(const float* q, float* __restrict__ Kc,
                                                           const float* __restrict__ Vc, const float* __restrict__ q_scale,
                                                           const float* __restrict__ k_scale, const float* __restrict__ bias, int bias_ld,
                                                           float* parts, int H, int Nmax, int nsplit,
                                                           const int* __restrict__ pos_dev, float scale, int round_bf16,
                                                           float* comb_out, int* comb_cnt) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int pos = *pos_dev;
    const int s = blockIdx.x, b = blockIdx.y;
    const int j0 = s * DEC_KS;
    if (j0 > pos) return;
    const int nk = min(DEC_KS, pos + 1 - j0);
    float* Ks = dsm;                        // [64][65]
    float* Vs = Ks + 64 * 65;               // [64][64]
    float* qn = Vs + 64 * 64;               // [H][64]
    float* sc = qn + H * 64;                // [8 waves][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* Kb = Kc + ((size_t)b * Nmax + j0) * 64;
    const float* Vb = Vc + ((size_t)b * Nmax + j0) * 64;
    // ---- all global loads first ----
    float4 kk[2], vv[2];
    const int c16 = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (threadIdx.x >> 4) + 32 * i;
        const int jj = min(j, nk - 1);
        kk[i] = ((const float4*)(Kb + (size_t)jj * 64))[c16];
        vv[i] = ((const float4*)(Vb + (size_t)jj * 64))[c16];
    }
    const float4 ks4 = ((const float4*)k_scale)[c16];
    float qv[2] = {0.f, 0.f};
    const float qs = q_scale[lane];
    for (int h = wave, i = 0; h < H; h += 8, ++i) qv[i] = q[(size_t)b * H * 64 + h * 64 + lane];
    // ---- q: l2norm * q_scale per head (utils.py:68-69), one wave per head ----
    for (int h = wave, i = 0; h < H; h += 8, ++i) {
        const float nrm = fmaxf(sqrtf(wave_sum(qv[i] * qv[i])), 1e-12f);
        qn[h * 64 + lane] = round_if(qv[i] / nrm * qs, round_bf16);
    }
    // ---- K / V tiles to LDS; the row written by dec_qkv this step is raw: normalise it here, keep it in the cache ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (threadIdx.x >> 4) + 32 * i;
        float4 kx = kk[i];
        if (j == pos - j0) {                                    // uniform over the 16 lanes that hold this row
            float ss = kx.x * kx.x + kx.y * kx.y + kx.z * kx.z + kx.w * kx.w;
            ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
            const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
            kx.x = round_if(kx.x * inv * ks4.x, round_bf16); kx.y = round_if(kx.y * inv * ks4.y, round_bf16);
            kx.z = round_if(kx.z * inv * ks4.z, round_bf16); kx.w = round_if(kx.w * inv * ks4.w, round_bf16);
            ((float4*)(Kb + (size_t)j * 64))[c16] = kx;
        }
        if (j < nk) {
            Ks[j * 65 + 4 * c16] = kx.x; Ks[j * 65 + 4 * c16 + 1] = kx.y; Ks[j * 65 + 4 * c16 + 2] = kx.z; Ks[j * 65 + 4 * c16 + 3] = kx.w;
            *(float4*)(Vs + j * 64 + 4 * c16) = vv[i];
        }
    }
    __syncthreads();
    for (int h = wave; h < H; h += 8) {                                   // lane = key, then lane = dim
        float dot = 0.f;
        if (lane < nk) {
#pragma unroll 16
            for (int d = 0; d < 64; ++d) dot += qn[h * 64 + d] * Ks[lane * 65 + d];
        }
        const float sv = lane < nk ? dot * scale + (bias ? bias[(size_t)(pos - j0 - lane) * bias_ld + h] : 0.f) : -3.0e38f;
        const float m = wave_max(sv);
        const float p = lane < nk ? __expf(sv - m) : 0.f;
        const float l = wave_sum(p);
        sc[wave * 64 + lane] = p;
        float* pp = parts + (((size_t)b * nsplit + s) * H + h) * DEC_PART;
        if (lane == 0) { pp[0] = m; pp[1] = l; }
        float acc = 0.f;
        for (int j = 0; j < nk; ++j) acc += sc[wave * 64 + j] * Vs[j * 64 + lane];
        pp[2 + lane] = acc;
    }
    if (!comb_cnt) return;
    // publish the partials, take a ticket (cdna_hip_programming.md: slab stores -> every wave vmcnt(0) -> barrier -> one lane: agent-scope
    // release, vmcnt(0), relaxed ticket; the last arriver: agent-scope acquire -> barrier -> plain loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sc[0] = __int_as_float(__hip_atomic_fetch_add(comb_cnt + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    __syncthreads();
    const int ns = pos / DEC_KS + 1;
    if (__float_as_int(sc[0]) != ns - 1) return;
    if (threadIdx.x == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); comb_cnt[b] = 0; }
    __syncthreads();
    for (int h = wave; h < H; h += 8) {                                   // lane = dim (the arithmetic of dec_attn_combine_kernel)
        const float* pb = parts + ((size_t)b * nsplit * H + h) * DEC_PART;
        float m = -3.0e38f, l = 0.f, o = 0.f;
        for (int s0 = 0; s0 < ns; s0 += 8) {
            float pm[8], pl[8], po[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* p = pb + (size_t)min(s0 + j, ns - 1) * H * DEC_PART;
                pm[j] = p[0]; pl[j] = p[1]; po[j] = p[2 + lane];
            }
            float mb = m;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (s0 + j < ns) mb = fmaxf(mb, pm[j]);
            const float resc = __expf(m - mb);
            l *= resc; o *= resc; m = mb;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (s0 + j < ns) { const float w = __expf(pm[j] - m); l += w * pl[j]; o += w * po[j]; }
        }
        comb_out[((size_t)b * H + h) * 64 + lane] = round_if(o / l, round_bf16);
    }
}

// ---- B2: out[b, n] = sum_k act[b, k] W[n, k] (+ res[b, n]); 16 output features per workgroup.
//      act = LN?(in[b, :]) or (parts != null) the combined attention output ----
template <typename TW>
__global__ __launch_bounds__(DEC_T) void dec_gemv_kernel(const float* __restrict__ in, int ldin, const float* __restrict__ gamma,
                                                        int Kstat, float eps, const float* __restrict__ parts, int nsplit, int H,
                                                        const int* __restrict__ pos_dev, const TW* __restrict__ W, long long ldw,
                                                        int K, int Nout, const float* __restrict__ res, int ldres,
                                                        float* __restrict__ out, int ldout, int B, int round_bf16,
                                                        const TW* __restrict__ Wlo = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* xs = dsm;
    float* vals = xs + (size_t)B * K;
    float* red = vals + DEC_ROWS * DEC_BMAX;
    if (parts) stage_attention(parts, nsplit, H, *pos_dev, B, round_bf16, xs);
    else stage_activation(in, ldin, K, Kstat, gamma, eps, B, round_bf16, xs, red);
    const int n0 = blockIdx.x * DEC_ROWS;
    const int rows = min(DEC_ROWS, Nout - n0);
    wg_gemv<TW>(W + (size_t)n0 * ldw, ldw, K, rows, xs, K, B, vals, Wlo ? Wlo + (size_t)n0 * ldw : nullptr);
    __syncthreads();
    for (int idx = threadIdx.x; idx < B * rows; idx += DEC_T) {
        const int b = idx / rows, r = idx - b * rows;
        float v = vals[r * DEC_BMAX + b];
        if (res) v += res[(size_t)b * ldres + n0 + r];
        out[(size_t)b * ldout + n0 + r] = v;
    }
}

// erf by Abramowitz-Stegun 7.1.26, identical to ffmid.hip (the batched path's GELU)
__device__ __forceinline__ float dec_gelu(float x) {
    const float z = x * 0.70710678118654752f, ax = fabsf(z);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    const float poly = ((((1.061405429f * t - 1.453152027f) * t + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f) * t;
    return 0.5f * x * (1.0f + copysignf(1.0f - poly * __expf(-ax * ax), z));
}

// ---- C1: h1 = LN(x1) W1^T (value + gate rows of 8 channels), causal depthwise conv over (p-2, p-1, p) with the conv
//          state, GEGLU; u[b, c] = value_conv * gelu(gate_conv); conv state advanced ----
template <typename TW>
__global__ __launch_bounds__(DEC_T) void dec_ffin_kernel(const float* __restrict__ x1, const float* __restrict__ gamma,
                                                        const TW* __restrict__ W1p, const float* __restrict__ convw,
                                                        float* __restrict__ hist, float* __restrict__ u,
                                                        int B, int D, int Fp, float eps, int round_bf16, const TW* __restrict__ W1lo = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* xs = dsm;
    float* vals = xs + (size_t)B * D;
    float* red = vals + DEC_ROWS * DEC_BMAX;
    stage_activation(x1, D, D, D, gamma, eps, B, round_bf16, xs, red);
    const int c0 = blockIdx.x * 8;
    // waves 0,1 -> the 8 value rows, waves 2,3 -> the 8 gate rows
    const int wave = threadIdx.x >> 6;
    {
        const size_t roff = (size_t)((wave < 2 ? c0 : Fp + c0 - 8)) * D;      // row r of this call = W[r]: gate rows start at r = 8
        wg_gemv<TW>(W1p + roff, D, D, DEC_ROWS, xs, D, B, vals, W1lo ? W1lo + roff : nullptr);
    }
    __syncthreads();
    const int ld = 2 * Fp;
    for (int idx = threadIdx.x; idx < B * 8; idx += DEC_T) {
        const int b = idx >> 3, c = idx & 7, col = c0 + c;
        const float hv = round_if(vals[c * DEC_BMAX + b], round_bf16);
        const float hg = round_if(vals[(8 + c) * DEC_BMAX + b], round_bf16);
        float* h0 = hist + (size_t)(b * 2) * ld;        // row p-2
        float* h1 = h0 + ld;                             // row p-1
        const float uv = convw[col] * h0[col] + convw[ld + col] * h1[col] + convw[2 * (size_t)ld + col] * hv;
        const float ug = convw[Fp + col] * h0[Fp + col] + convw[ld + Fp + col] * h1[Fp + col] + convw[2 * (size_t)ld + Fp + col] * hg;
        u[(size_t)b * Fp + col] = dec_gelu(ug) * uv;
        h0[col] = h1[col];       h0[Fp + col] = h1[Fp + col];
        h1[col] = hv;            h1[Fp + col] = hg;
    }
}

// =========================================================================================================================
// Second-generation step kernels (default; OMLM_DECODE_V1=1 selects the ones above).  The first generation measured 8-15 us per
// launch for ~1 us of memory time, because every latency in a workgroup was serialised: stage the activation (global load ->
// two block reductions for LayerNorm -> barrier), THEN start the weight loads, 16 rows per workgroup in a rolled loop of
// load -> wait -> FMA (5.4 round trips for the FF-out rows), then 4 x B serial wave reductions.  Here
//   * a workgroup owns FOUR weight rows (one per wave), so 4x as many workgroups stream at once, and a lane's whole share of
//     its row (<= NI 16-byte pieces) is requested before anything else happens -- one memory round trip per launch;
//   * the activation is staged raw, with its LayerNorm statistics taken in ONE pass (sum and sum of squares, one barrier);
//     each lane normalises the pieces it multiplies on the fly;
//   * the attention partials are combined with all their loads in flight at once.
#define DEC2_ROWS 4

template <typename TW> struct dec_wreg;
#ifndef OMLM_DEC_NT
#define OMLM_DEC_NT 0          // 1: weight rows of the B = 1 kernels as non-temporal loads (A/B: tools/build_ab.sh nt decode.hip -DOMLM_DEC_NT=1)
#endif
template <> struct dec_wreg<h16_t> { u32x4 r;
    __device__ __forceinline__ void load(const h16_t* p) {
#if OMLM_DEC_NT
        r = __builtin_nontemporal_load((const u32x4*)p);
#else
        r = *(const u32x4*)p;
#endif
    }
    __device__ __forceinline__ void zero() { r[0] = r[1] = r[2] = r[3] = 0u; }
    __device__ __forceinline__ void unpack(float* w) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) { w[2 * i] = h16_lo_to_f(r[i]); w[2 * i + 1] = h16_hi_to_f(r[i]); } } };
template <> struct dec_wreg<float> { float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = ((const float4*)p)[0]; b = ((const float4*)p)[1]; }
    __device__ __forceinline__ void zero() { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
    __device__ __forceinline__ void unpack(float* w) const { w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w; } };

// activation staging with one-pass LayerNorm statistics: xs[b][0..K) <- in[b][0..K) (raw); stat[b] = (mean, rstd) over Kstat
__device__ __forceinline__ void dec2_stage(const float* __restrict__ in, int ldin, int K, int Kstat, bool want_stats, float eps,
                                           int B, float* xs, float* red /* [B][4][2] */, float* stat /* [B][2] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = 0; b < B; ++b) {
        float s = 0.f, q = 0.f;
        for (int i = threadIdx.x * 4; i < K; i += DEC_T * 4) {
            const float4 v = *(const float4*)(in + (size_t)b * ldin + i);
            *(float4*)(xs + (size_t)b * K + i) = v;
            if (i + 0 < Kstat) { s += v.x; q += v.x * v.x; }
            if (i + 1 < Kstat) { s += v.y; q += v.y * v.y; }
            if (i + 2 < Kstat) { s += v.z; q += v.z * v.z; }
            if (i + 3 < Kstat) { s += v.w; q += v.w * v.w; }
        }
        if (want_stats) {
            s = wave_sum(s); q = wave_sum(q);
            if (lane == 0) { red[(b * 4 + wave) * 2] = s; red[(b * 4 + wave) * 2 + 1] = q; }
        }
    }
    __syncthreads();
    if (want_stats && threadIdx.x < B) {
        const int b = threadIdx.x;
        float s = 0.f, q = 0.f;
        for (int w = 0; w < 4; ++w) { s += red[(b * 4 + w) * 2]; q += red[(b * 4 + w) * 2 + 1]; }
        const float mean = s / (float)Kstat;
        const float var = fmaxf(q / (float)Kstat - mean * mean, 0.f);
        stat[2 * b] = mean; stat[2 * b + 1] = rsqrtf(var + eps);
    }
    if (want_stats) __syncthreads();
}

// attention output of the new row from the per-split partials, all loads issued before the first use: xs[b][h * 64 + d]
__device__ __forceinline__ void dec2_stage_attention(const float* __restrict__ parts, int nsplit, int H, int pos, int B, int round_bf16, float* xs) {
    const int ns = pos / DEC_KS + 1;
    for (int idx = threadIdx.x; idx < B * H * 64; idx += DEC_T) {
        const int b = idx / (H * 64), hd = idx - b * H * 64, h = hd >> 6, d = hd & 63;
        const float* pb = parts + ((size_t)b * nsplit * H + h) * DEC_PART;
        float m = -3.0e38f, l = 0.f, o = 0.f;
        for (int s0 = 0; s0 < ns; s0 += 8) {                    // 8 splits (512 keys) per batch of loads
            float pm[8], pl[8], po[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float* p = pb + (size_t)min(s0 + j, ns - 1) * H * DEC_PART;
                pm[j] = p[0]; pl[j] = p[1]; po[j] = p[2 + d];
            }
            float mb = m;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (s0 + j < ns) mb = fmaxf(mb, pm[j]);
            const float resc = __expf(m - mb);
            l *= resc; o *= resc; m = mb;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (s0 + j < ns) { const float w = __expf(pm[j] - m); l += w * pl[j]; o += w * po[j]; }
        }
        xs[idx] = round_if(o / l, round_bf16);
    }
    __syncthreads();
}

enum { DEC2_QKV = 0, DEC2_OUT = 1, DEC2_FFIN = 2, DEC2_LNGEMV = 3 };

struct dec2_args {
    const float* in; int ldin; int K, Kstat; const float* gamma; float eps;       // activation (+ LayerNorm over Kstat when gamma)
    const float* parts; int nsplit, H; const int* pos_dev;                        // DEC2_OUT: attention partials instead
    const void* W; const void* W2; long long ldw; int Nout;                       // weight rows (W2: the Wkv rows of DEC2_QKV)
    const void* Wlo;                                                              // PL instantiations ("fp16ff"): lo plane of W, same layout -- the row is W + Wlo
    const float* res; int ldres; float* out; int ldout;                           // out = dot (+ res)
    float* q; float* Kc; float* Vc; int Nmax;                                     // DEC2_QKV destinations
    const float* convw; float* hist; float* u; int Fp;                            // DEC2_FFIN
    int B, round_bf16;
    int* adv_pos; int* adv_step;                                                  // head launch only: counters bumped by its workgroup 0
    // matrix-core kernels: LayerNorm statistics from the producers' per-workgroup partial sums [partial][8 samples][2] instead of a
    // reduction over every sample's row in every workgroup (stat_in, nstat_in partials); this launch's own partials (stat_out)
    const float* stat_in; int nstat_in; float* stat_out;
    // matrix-core kernels, DEC2_LNGEMV: the k-range of a row cut into nsl slices (grid = tiles * nsl); slabs [tile][slice][16 rows][16 samples]
    // in sk_ws, one arrival counter per tile in sk_cnt (zero on entry and exit)
    int nsl; float* sk_ws; int* sk_cnt;
};

template <typename TW, int NI, int MODE>
__global__ __launch_bounds__(DEC_T) void dec2_kernel(dec2_args a) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int B = a.B, K = a.K;
    float* xs = dsm;                                   // [B][K]
    float* red = xs + (size_t)B * K;                   // [B][4][2]
    float* stat = red + DEC_BMAX * 8;                  // [B][2]
    float* vals = stat + DEC_BMAX * 2;                 // [4][DEC_BMAX]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = K >> 3;
    // ---- this wave's weight row, requested in full before anything else ----
    int row;                                           // logical output index of this wave
    const TW* wrow;
    bool ln_row = a.gamma != nullptr;
    if (MODE == DEC2_FFIN) {                           // waves 0,1: value rows c0, c0 + 1; waves 2,3: the gate rows of the same channels
        const int c0 = blockIdx.x * 2;
        row = c0 + (wave & 1);
        wrow = (const TW*)a.W + (size_t)((wave < 2 ? 0 : a.Fp) + row) * a.ldw;
    } else if (MODE == DEC2_QKV) {
        row = blockIdx.x * DEC2_ROWS + wave;
        const int HD = a.H * 64;
        ln_row = row < HD;                             // K / V are projected from the un-normalised residual (transformer.py:228)
        wrow = row < HD ? (const TW*)a.W + (size_t)row * a.ldw : (const TW*)a.W2 + (size_t)(row - HD) * a.ldw;
    } else {
        row = blockIdx.x * DEC2_ROWS + wave;
        wrow = (const TW*)a.W + (size_t)min(row, a.Nout - 1) * a.ldw;
    }
    dec_wreg<TW> wr[NI];
    float4 g0[NI], g1[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) wr[i].load(wrow + c * 8); else wr[i].zero();
        if (a.gamma && c < nch) { g0[i] = *(const float4*)(a.gamma + c * 8); g1[i] = *(const float4*)(a.gamma + c * 8 + 4); }
        else { g0[i] = make_float4(1.f, 1.f, 1.f, 1.f); g1[i] = g0[i]; }
    }
    // ---- activation ----
    if (MODE == DEC2_OUT) dec2_stage_attention(a.parts, a.nsplit, a.H, *a.pos_dev, B, a.round_bf16, xs);
    else dec2_stage(a.in, a.ldin, K, a.Kstat, a.gamma != nullptr, a.eps, B, xs, red, stat);
    // ---- dot products: this lane's pieces of the row against every sample ----
    float acc[DEC_BMAX];
#pragma unroll
    for (int b = 0; b < DEC_BMAX; ++b) acc[b] = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            float w[8];
            wr[i].unpack(w);
#pragma unroll
            for (int b = 0; b < DEC_BMAX; ++b) {
                if (b < B) {
                    const float4 x0 = *(const float4*)(xs + (size_t)b * K + c * 8), x1 = *(const float4*)(xs + (size_t)b * K + c * 8 + 4);
                    float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    if (MODE != DEC2_OUT) {
                        if (ln_row) {
                            const float mean = stat[2 * b], rstd = stat[2 * b + 1];
                            const float gm[8] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w, g1[i].x, g1[i].y, g1[i].z, g1[i].w};
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] = round_if((x[e] - mean) * rstd * gm[e], a.round_bf16);
                        } else if (a.round_bf16) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] = round_if(x[e], 1);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[b] += w[e] * x[e];
                }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < DEC_BMAX; ++b) {
        if (b < B) {
            const float s = wave_sum(acc[b]);
            if (lane == 0) vals[wave * DEC_BMAX + b] = s;
        }
    }
    __syncthreads();
    // ---- epilogue ----
    if (MODE == DEC2_FFIN) {
        const int ld = 2 * a.Fp;
        for (int idx = threadIdx.x; idx < B * 2; idx += DEC_T) {
            const int b = idx >> 1, cc = idx & 1, col = blockIdx.x * 2 + cc;
            const float hv = round_if(vals[cc * DEC_BMAX + b], a.round_bf16);
            const float hg = round_if(vals[(2 + cc) * DEC_BMAX + b], a.round_bf16);
            float* h0 = a.hist + (size_t)(b * 2) * ld;       // row p-2
            float* h1 = h0 + ld;                              // row p-1
            const float uv = a.convw[col] * h0[col] + a.convw[ld + col] * h1[col] + a.convw[2 * (size_t)ld + col] * hv;
            const float ug = a.convw[a.Fp + col] * h0[a.Fp + col] + a.convw[ld + a.Fp + col] * h1[a.Fp + col] + a.convw[2 * (size_t)ld + a.Fp + col] * hg;
            a.u[(size_t)b * a.Fp + col] = dec_gelu(ug) * uv;
            h0[col] = h1[col];       h0[a.Fp + col] = h1[a.Fp + col];
            h1[col] = hv;            h1[a.Fp + col] = hg;
        }
    } else if (MODE == DEC2_QKV) {
        const int pos = *a.pos_dev, HD = a.H * 64;
        for (int idx = threadIdx.x; idx < B * DEC2_ROWS; idx += DEC_T) {
            const int b = idx / DEC2_ROWS, r = idx - b * DEC2_ROWS, n = blockIdx.x * DEC2_ROWS + r;
            const float v = vals[r * DEC_BMAX + b];
            if (n < HD) a.q[(size_t)b * HD + n] = v;
            else if (n < HD + 64) a.Kc[((size_t)b * a.Nmax + pos) * 64 + (n - HD)] = v;                  // raw: normalised by dec_attn
            else a.Vc[((size_t)b * a.Nmax + pos) * 64 + (n - HD - 64)] = round_if(v, a.round_bf16);
        }
    } else {
        for (int idx = threadIdx.x; idx < B * DEC2_ROWS; idx += DEC_T) {
            const int b = idx / DEC2_ROWS, r = idx - b * DEC2_ROWS, n = blockIdx.x * DEC2_ROWS + r;
            if (n < a.Nout) {
                float v = vals[r * DEC_BMAX + b];
                if (a.res) v += a.res[(size_t)b * a.ldres + n];
                a.out[(size_t)b * a.ldout + n] = v;
            }
        }
        if (MODE == DEC2_LNGEMV && blockIdx.x == 0 && threadIdx.x == 0 && a.adv_pos) {      // see dec3_kernel
            a.adv_pos[0] += 1;
            if (a.adv_step) a.adv_step[0] += 1;
        }
    }
}

template <typename TW, int NI, int MODE>
static void dec2_launch(const dec2_args& a, int grid, hipStream_t st) {
    const size_t lds = ((size_t)a.B * a.K + DEC_BMAX * 8 + DEC_BMAX * 2 + 4 * DEC_BMAX) * sizeof(float);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)dec2_kernel<TW, NI, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((dec2_kernel<TW, NI, MODE>), dim3(grid), dim3(DEC_T), lds, st, a);
}

// ---- B == 1 fast path: no LDS, no barrier.  Every wave loads the whole activation row itself (L2 hits; exactly the pieces its
// dot product multiplies), takes the LayerNorm statistics with two wave reductions, and finishes its own output element:
// one memory round trip between launch and store.  FF-in: a wave owns one CHANNEL (its value row and its gate row), so the
// conv / GEGLU epilogue needs no exchange either.

template <typename TW, int NI, int MODE, bool PL = false>
__global__ __launch_bounds__(DEC_T) void dec3_kernel(dec2_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = a.K, nch = K >> 3;
    const int unit = blockIdx.x * 4 + wave;           // output row (FF-in: channel)
    const int HD = a.H * 64;
    if (unit >= (MODE == DEC2_FFIN ? a.Fp : a.Nout)) return;
    bool ln_row = a.gamma != nullptr;
    const TW* wrow;
    const TW* wrow2 = nullptr;
    if (MODE == DEC2_FFIN) { wrow = (const TW*)a.W + (size_t)unit * a.ldw; wrow2 = (const TW*)a.W + (size_t)(a.Fp + unit) * a.ldw; }
    else if (MODE == DEC2_QKV) { ln_row = unit < HD; wrow = unit < HD ? (const TW*)a.W + (size_t)unit * a.ldw : (const TW*)a.W2 + (size_t)(unit - HD) * a.ldw; }
    else wrow = (const TW*)a.W + (size_t)unit * a.ldw;
    static_assert(!PL || MODE == DEC2_LNGEMV, "lo planes: the LayerNorm + row-product launches (FF-out, head); FF-in has dec3_ffin_kernel");
    dec_wreg<TW> wr[NI], wr2[MODE == DEC2_FFIN ? NI : 1], wl[PL ? NI : 1];
    float4 g0[NI], g1[NI], x0[NI], x1[NI];
    // activation row and gamma FIRST, the weight rows behind them: vector-memory results retire in issue order, so the LayerNorm
    // statistics (two wave reductions) start when the L2-resident row has landed and run under the weight fetch from HBM
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < nch;
        const int cc = ok ? c : 0;
        x0[i] = *(const float4*)(a.in + cc * 8); x1[i] = *(const float4*)(a.in + cc * 8 + 4);
        if (a.gamma) { g0[i] = *(const float4*)(a.gamma + cc * 8); g1[i] = *(const float4*)(a.gamma + cc * 8 + 4); }
        else { g0[i] = make_float4(1.f, 1.f, 1.f, 1.f); g1[i] = g0[i]; }
        if (!ok) { x0[i] = make_float4(0.f, 0.f, 0.f, 0.f); x1[i] = x0[i]; }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < nch;
        const int cc = ok ? c : 0;
        wr[i].load(wrow + cc * 8);
        if (PL) { wl[i].load((const TW*)a.Wlo + (size_t)unit * a.ldw + cc * 8); if (!ok) wl[i].zero(); }
        if (MODE == DEC2_FFIN) wr2[i].load(wrow2 + cc * 8);
        if (!ok) { wr[i].zero(); if (MODE == DEC2_FFIN) wr2[i].zero(); }
    }
    __builtin_amdgcn_sched_barrier(0);
    // epilogue operands requested now as well (lane 0 consumes them)
    float resv = 0.f, h0v = 0.f, h0g = 0.f, h1v = 0.f, h1g = 0.f, cw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int pos = 0;
    const int ld = 2 * a.Fp;
    if (lane == 0) {
        if (MODE == DEC2_LNGEMV && a.res) resv = a.res[unit];
        if (MODE == DEC2_QKV) pos = *a.pos_dev;
        if (MODE == DEC2_FFIN) {
            h0v = a.hist[unit]; h0g = a.hist[a.Fp + unit]; h1v = a.hist[ld + unit]; h1g = a.hist[ld + a.Fp + unit];
#pragma unroll
            for (int t = 0; t < 3; ++t) { cw[t] = a.convw[(size_t)t * ld + unit]; cw[3 + t] = a.convw[(size_t)t * ld + a.Fp + unit]; }
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (ln_row) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int e0 = (lane + 64 * i) * 8;
            const float xv[8] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w, x1[i].x, x1[i].y, x1[i].z, x1[i].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) if (e0 + e < a.Kstat) { s += xv[e]; q += xv[e] * xv[e]; }
        }
        s = wave_sum(s); q = wave_sum(q);
        mean = s / (float)a.Kstat;
        rstd = rsqrtf(fmaxf(q / (float)a.Kstat - mean * mean, 0.f) + a.eps);
    }
    float acc = 0.f, acc2 = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float w[8], w2[8];
        wr[i].unpack(w);
        if (PL) {
            float wlo[8];
            wl[i].unpack(wlo);
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] += wlo[e];
        }
        if (MODE == DEC2_FFIN) wr2[i].unpack(w2);
        float x[8] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w, x1[i].x, x1[i].y, x1[i].z, x1[i].w};
        if (ln_row) {
            const float gm[8] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w, g1[i].x, g1[i].y, g1[i].z, g1[i].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = round_if((x[e] - mean) * rstd * gm[e], a.round_bf16);
        } else if (a.round_bf16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = round_if(x[e], 1);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc += w[e] * x[e]; if (MODE == DEC2_FFIN) acc2 += w2[e] * x[e]; }
    }
    acc = wave_sum(acc);
    if (MODE == DEC2_FFIN) acc2 = wave_sum(acc2);
    if (lane != 0) return;
    if (MODE == DEC2_FFIN) {
        const float hv = round_if(acc, a.round_bf16), hg = round_if(acc2, a.round_bf16);
        const float uv = cw[0] * h0v + cw[1] * h1v + cw[2] * hv;
        const float ug = cw[3] * h0g + cw[4] * h1g + cw[5] * hg;
        a.u[unit] = dec_gelu(ug) * uv;
        a.hist[unit] = h1v;            a.hist[a.Fp + unit] = h1g;
        a.hist[ld + unit] = hv;        a.hist[ld + a.Fp + unit] = hg;
    } else if (MODE == DEC2_QKV) {
        if (unit < HD) a.q[unit] = acc;
        else if (unit < HD + 64) a.Kc[(size_t)pos * 64 + (unit - HD)] = acc;                     // raw: normalised by dec_attn
        else a.Vc[(size_t)pos * 64 + (unit - HD - 64)] = round_if(acc, a.round_bf16);
    } else {
        a.out[unit] = acc + resv;
        // last kernel of a step (the head): move the row index / sampler step on here instead of in a launch of its own.  Nothing in
        // this kernel reads them, and every later kernel is ordered behind this one on the stream.
        if (unit == 0 && a.adv_pos) { a.adv_pos[0] += 1; if (a.adv_step) a.adv_step[0] += 1; }
    }
}

// FF-in rows for B == 1 with CPW channels per wave: the wave's copy of LN(x1) (registers) is multiplied into 2 CPW weight rows, so the
// L2 traffic of the activation / gamma reloads drops from 2x the weight bytes (one channel per wave) to 0.5x.
template <typename TW, int CPW, bool PL = false>
__global__ __launch_bounds__(DEC_T) void dec3_ffin_kernel(dec2_args a) {
    constexpr int NI = 2;                              // D = 1024: two 8-element pieces per lane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (blockIdx.x * 4 + wave) * CPW;      // first channel of this wave
    if (c0 >= a.Fp) return;
    const int ld = 2 * a.Fp;
    dec_wreg<TW> wv[CPW][NI], wg[CPW][NI], wvl[PL ? CPW : 1][NI], wgl[PL ? CPW : 1][NI];
    float4 g0[NI], g1[NI], x0[NI], x1[NI];
    // activation row, gamma and the epilogue operands first, the weight rows behind them (see dec3_kernel)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        x0[i] = *(const float4*)(a.in + c * 8); x1[i] = *(const float4*)(a.in + c * 8 + 4);
        g0[i] = *(const float4*)(a.gamma + c * 8); g1[i] = *(const float4*)(a.gamma + c * 8 + 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
#pragma unroll
        for (int ch = 0; ch < CPW; ++ch) {
            const int cc = min(c0 + ch, a.Fp - 1);
            wv[ch][i].load((const TW*)a.W + (size_t)cc * a.ldw + c * 8);
            wg[ch][i].load((const TW*)a.W + (size_t)(a.Fp + cc) * a.ldw + c * 8);
            if (PL) {
                wvl[ch][i].load((const TW*)a.Wlo + (size_t)cc * a.ldw + c * 8);
                wgl[ch][i].load((const TW*)a.Wlo + (size_t)(a.Fp + cc) * a.ldw + c * 8);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // epilogue operands: lane ch finishes channel c0 + ch
    float h0v = 0.f, h0g = 0.f, h1v = 0.f, h1g = 0.f, cw[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int myc = c0 + lane;
    const bool fin = lane < CPW && myc < a.Fp;
    if (fin) {
        h0v = a.hist[myc]; h0g = a.hist[a.Fp + myc]; h1v = a.hist[ld + myc]; h1g = a.hist[ld + a.Fp + myc];
#pragma unroll
        for (int t = 0; t < 3; ++t) { cw[t] = a.convw[(size_t)t * ld + myc]; cw[3 + t] = a.convw[(size_t)t * ld + a.Fp + myc]; }
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        s += (x0[i].x + x0[i].y) + (x0[i].z + x0[i].w) + (x1[i].x + x1[i].y) + (x1[i].z + x1[i].w);
        q += x0[i].x * x0[i].x + x0[i].y * x0[i].y + x0[i].z * x0[i].z + x0[i].w * x0[i].w +
             x1[i].x * x1[i].x + x1[i].y * x1[i].y + x1[i].z * x1[i].z + x1[i].w * x1[i].w;
    }
    s = wave_sum(s); q = wave_sum(q);
    const float mean = s / (float)a.K;
    const float rstd = rsqrtf(fmaxf(q / (float)a.K - mean * mean, 0.f) + a.eps);
    float accv[CPW], accg[CPW];
#pragma unroll
    for (int ch = 0; ch < CPW; ++ch) { accv[ch] = 0.f; accg[ch] = 0.f; }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        float x[8] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w, x1[i].x, x1[i].y, x1[i].z, x1[i].w};
        const float gm[8] = {g0[i].x, g0[i].y, g0[i].z, g0[i].w, g1[i].x, g1[i].y, g1[i].z, g1[i].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = round_if((x[e] - mean) * rstd * gm[e], a.round_bf16);
#pragma unroll
        for (int ch = 0; ch < CPW; ++ch) {
            float w[8], w2[8];
            wv[ch][i].unpack(w); wg[ch][i].unpack(w2);
            if (PL) {
                float l1[8], l2[8];
                wvl[ch][i].unpack(l1); wgl[ch][i].unpack(l2);
#pragma unroll
                for (int e = 0; e < 8; ++e) { w[e] += l1[e]; w2[e] += l2[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { accv[ch] += w[e] * x[e]; accg[ch] += w2[e] * x[e]; }
        }
    }
    float myv = 0.f, myg = 0.f;
#pragma unroll
    for (int ch = 0; ch < CPW; ++ch) {
        const float tv = wave_sum(accv[ch]), tg = wave_sum(accg[ch]);
        if (lane == ch) { myv = tv; myg = tg; }
    }
    if (!fin) return;
    const float hv = round_if(myv, a.round_bf16), hg = round_if(myg, a.round_bf16);
    const float uv = cw[0] * h0v + cw[1] * h1v + cw[2] * hv;
    const float ug = cw[3] * h0g + cw[4] * h1g + cw[5] * hg;
    a.u[myc] = dec_gelu(ug) * uv;
    a.hist[myc] = h1v;            a.hist[a.Fp + myc] = h1g;
    a.hist[ld + myc] = hv;        a.hist[ld + a.Fp + myc] = hg;
}

// =========================================================================================================================
// Batched steps (2 <= B <= 16) on the matrix cores.  With B samples the VALU kernels above repeat every dot product, every
// LayerNorm normalisation and every cross-lane reduction B times per weight row (measured: 506 us per step at B = 8 against 142 us
// at B = 1 for the same weight bytes).  Here a workgroup owns 16 weight rows; the activations of all samples are normalised and
// rounded ONCE into an LDS image [16][K + 8] of the operand type, and out[row, sample] = sum_k W[row, k] x[sample, k] is a chain of
// v_mfma_f32_16x16x32 per wave: A = 16 rows x 32 k straight from the weight rows (one 16-byte load per lane: the MFMA operand layout
// IS 8 consecutive k of one row), B = the samples (padded to 16 columns with zero operands).  The four waves take interleaved
// 32-wide k-steps of the same rows (all of a wave's loads requested before anything else, as above) and their partial tiles meet in
// LDS -- no cross-lane reductions at all.  Same products (16-bit x 16-bit, exact in fp32), fp32 accumulation in a different order.
#define DEC4_T 256
#define DEC4_ROWS 16
#define DEC4_NB 16
typedef __attribute__((ext_vector_type(4))) float dec4_acc;

// Attention output of the new row from the per-split partials, ONCE per step: [B][H * 64] fp32 (rounded to the operand type).  The
// VALU kernels combine inside the to_out launch, every workgroup for itself -- 64 workgroups x B x H x 64 elements x splits x 3
// loads was 33 us of the B = 8 step; this is one launch of B x H wave-sized workgroups.
__global__ __launch_bounds__(64) void dec_attn_combine_kernel(const float* __restrict__ parts, float* __restrict__ out, int nsplit, int H,
                                                              const int* __restrict__ pos_dev, int round_bf16) {
    const int b = blockIdx.y, h = blockIdx.x, d = threadIdx.x;
    const int ns = *pos_dev / DEC_KS + 1;
    const float* pb = parts + ((size_t)b * nsplit * H + h) * DEC_PART;
    float m = -3.0e38f, l = 0.f, o = 0.f;
    for (int s0 = 0; s0 < ns; s0 += 8) {
        float pm[8], pl[8], po[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* p = pb + (size_t)min(s0 + j, ns - 1) * H * DEC_PART;
            pm[j] = p[0]; pl[j] = p[1]; po[j] = p[2 + d];
        }
        float mb = m;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (s0 + j < ns) mb = fmaxf(mb, pm[j]);
        const float resc = __expf(m - mb);
        l *= resc; o *= resc; m = mb;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (s0 + j < ns) { const float w = __expf(pm[j] - m); l += w * pl[j]; o += w * po[j]; }
    }
    out[((size_t)b * H + h) * 64 + d] = round_if(o / l, round_bf16);
}

// PL ("fp16ff": FF-in, FF-out, head): the weights are hi + lo planes and the normalised activations keep a lo image next to the hi one -- three
// MFMAs per k-step (hi hi + hi lo + lo hi, the batched forward's omlm_gemm_planes16 arithmetic).  The images hold DEC4_IMG(NS) samples; a
// larger batch (FF-out rows at B > 8) runs its k-loop twice.  Needs the producers' LayerNorm partials (stat_in).
#define DEC4_IMG(NS) ((((NS) + 7) / 8) == 1 ? 16 : 8)
template <int NS, int MODE, bool PL = false>
__global__ __launch_bounds__(DEC4_T) void dec4_kernel(dec2_args a) {
    extern __shared__ __attribute__((aligned(16))) char dsm4[];
    // split-K (DEC2_LNGEMV with a.nsl > 1: the FF-out launch, 16 output rows x Fp per tile): 64 workgroups each walking a 2752-long row at
    // B = 16 were 21.6 us of the step (14.1 us without the lo planes) -- a quarter of the CUs, every one of them normalising all 16 x 2752
    // activations in two passes.  Here a tile's row is cut into a.nsl slices (one workgroup each, one pass over 16 samples); the slices
    // meet through fp32 slabs and the last to arrive adds them in slice order (residual, output and LayerNorm partials as before).
    int tile = blockIdx.x, slice = 0, k0 = 0, Kown = a.K;
    if (MODE == DEC2_LNGEMV && a.nsl > 1) {
        const int ntiles = gridDim.x / a.nsl;
        if ((ntiles & 7) == 0) {                                // a tile's slices on ONE XCD (workgroup id % 8): the last arriver reads same-XCD slabs
            const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
            tile = (idx / a.nsl) * 8 + xcd; slice = idx % a.nsl;
        } else { tile = blockIdx.x / a.nsl; slice = blockIdx.x % a.nsl; }
        const int steps = a.K >> 5, base = steps / a.nsl, rem = steps % a.nsl;
        k0 = 32 * (slice * base + min(slice, rem));
        Kown = 32 * (base + (slice < rem ? 1 : 0));
    }
    const int B = a.B, K = Kown, KP = K + 8;
    const float* a_in = a.in + k0;
    const float* a_gamma = a.gamma ? a.gamma + k0 : nullptr;
    h16_t* xs = (h16_t*)dsm4;                                   // [B][KP] operand image (PL: [DEC4_IMG][KP] hi, then the lo image)
    h16_t* xs_lo = xs + (size_t)DEC4_IMG(NS) * KP;
    float* red = (float*)(dsm4 + (((size_t)(PL ? 2 * DEC4_IMG(NS) : B) * KP * 2 + 15) & ~(size_t)15));   // [DEC4_NB][4][2]
    float* stat = red + DEC4_NB * 8;                            // [DEC4_NB][2]
    float* part = stat + DEC4_NB * 2;                           // [4][16][16]
    float* vals = part + 4 * 256;                               // [16 rows][16 samples]
    float* xraw = vals + 256;                                   // [B][K] fp32 (LayerNorm inputs between the statistics and the rounding)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    const int HD = a.H * 64;
    static_assert(MODE != DEC2_OUT, "the matrix-core path takes the combined attention output as a plain input (dec_attn_combine_kernel)");
    // ---- this lane's weight row and its pieces ----
    // WHERE they are requested matters: vector-memory results retire in issue order, so with the weights (64 KB of HBM reads per
    // workgroup, twice that with lo planes) requested first, the wait for the activation pieces -- L2 hits -- sat behind the whole weight
    // fetch and the normalisation phase started when the weights had landed (FF-in 14 us with lo planes, 10 us without).  The
    // LayerNorm-by-partials path and the plain path request them right AFTER their activation loads: those retire first, the weights
    // stay in flight under the statistics / normalisation / image phase and are waited for at the k-loop.
    int grow;                                                   // global output row of MFMA row r
    const h16_t* wrow;
    bool ln_rows = a.gamma != nullptr;                          // uniform per workgroup
    if (MODE == DEC2_FFIN) {                                    // rows 0..7: value rows of channels c0..c0+7, rows 8..15: their gate rows
        const int c = min(blockIdx.x * 8 + (r & 7), a.Fp - 1);
        grow = c;
        wrow = (const h16_t*)a.W + (size_t)((r < 8 ? 0 : a.Fp) + c) * a.ldw;
    } else if (MODE == DEC2_QKV) {
        grow = blockIdx.x * DEC4_ROWS + r;
        ln_rows = blockIdx.x * DEC4_ROWS < HD;                  // HD is a multiple of 16: a workgroup holds q rows or k/v rows, never both
        wrow = grow < HD ? (const h16_t*)a.W + (size_t)grow * a.ldw : (const h16_t*)a.W2 + (size_t)(grow - HD) * a.ldw;
    } else {
        grow = tile * DEC4_ROWS + r;
        wrow = (const h16_t*)a.W + (size_t)min(grow, a.Nout - 1) * a.ldw + k0;
    }
    const int S = K >> 5;                                       // 32-wide k-steps (K is a multiple of 32)
    u32x4 wr[NS], wl[PL ? NS : 1];
    static_assert(!PL || MODE == DEC2_FFIN || MODE == DEC2_LNGEMV, "lo planes: FF-in, FF-out, head");
    // what the epilogue reads from memory is requested now (one (sample, row) per thread: B <= 16): the conv history and taps of FF-in,
    // the residual of the row products -- at the end of the launch they were a dependent round trip in front of the last stores
    float pre_h[4] = {0.f, 0.f, 0.f, 0.f}, pre_w[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pre_res = 0.f;
    if (MODE == DEC2_FFIN) {
        const int idx = threadIdx.x, b = idx >> 3, col = blockIdx.x * 8 + (idx & 7), ld = 2 * a.Fp;
        if (idx < B * 8 && col < a.Fp) {
            const float* h0 = a.hist + (size_t)(b * 2) * ld;
            pre_h[0] = h0[col]; pre_h[1] = h0[ld + col]; pre_h[2] = h0[a.Fp + col]; pre_h[3] = h0[ld + a.Fp + col];
#pragma unroll
            for (int k = 0; k < 3; ++k) { pre_w[k] = a.convw[(size_t)k * ld + col]; pre_w[3 + k] = a.convw[(size_t)k * ld + a.Fp + col]; }
        }
    } else if (MODE == DEC2_LNGEMV) {
        const int idx = threadIdx.x, b = idx / DEC4_ROWS, n = tile * DEC4_ROWS + (idx - b * DEC4_ROWS);
        if (a.res && idx < B * DEC4_ROWS && n < a.Nout) pre_res = a.res[(size_t)b * a.ldres + n];
    }
    auto load_weights = [&]() {
        __builtin_amdgcn_sched_barrier(0);                      // (pinned: behind the activation requests above, ahead of everything below)
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int s = wave + 4 * j;
            if (s < S) wr[j] = *(const u32x4*)(wrow + 32 * s + 8 * kq);
            else { wr[j][0] = 0u; wr[j][1] = 0u; wr[j][2] = 0u; wr[j][3] = 0u; }
            if constexpr (PL) {
                const h16_t* wlrow = (const h16_t*)a.Wlo + (wrow - (const h16_t*)a.W);
                if (s < S) wl[j] = *(const u32x4*)(wlrow + 32 * s + 8 * kq);
                else { wl[j][0] = 0u; wl[j][1] = 0u; wl[j][2] = 0u; wl[j][3] = 0u; }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the k-loop: one MFMA per 32-wide step (PL: three); samples >= nhave are zero operands
    auto kloop = [&](dec4_acc& acc, const int nhave) {
        const bool have = r < nhave;                            // this lane's B-operand column is a real sample
        const h16_t* xrow = xs + (size_t)(have ? r : 0) * KP + 8 * kq;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int s = wave + 4 * j;
            if (s < S) {
                u32x4 xb = {0u, 0u, 0u, 0u};
                if (have) xb = *(const u32x4*)(xrow + 32 * s);
                acc = OMLM_MFMA_16x16x32(__builtin_bit_cast(h16x8, wr[j]), __builtin_bit_cast(h16x8, xb), acc);
                if constexpr (PL) {
                    u32x4 xl = {0u, 0u, 0u, 0u};
                    if (have) xl = *(const u32x4*)(xrow + (size_t)DEC4_IMG(NS) * KP + 32 * s);
                    acc = OMLM_MFMA_16x16x32(__builtin_bit_cast(h16x8, wr[j]), __builtin_bit_cast(h16x8, xl), acc);
                    acc = OMLM_MFMA_16x16x32(__builtin_bit_cast(h16x8, wl[j]), __builtin_bit_cast(h16x8, xb), acc);
                }
            }
        }
    };
    dec4_acc acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    // ---- activations: normalised / rounded once per workgroup into the operand image.  All samples' loads of a chunk are in flight
    // together (a per-sample loop of load -> reduce was 2 B dependent L2 round trips per launch: 12-20 us at B = 8) ----
    {
        constexpr int NBR = 8;                                  // samples per register batch
        if (ln_rows && a.stat_in) {
            // statistics from the producers' partial sums: every load of this launch (weights above, gamma, all samples' pieces of the
            // row, the partials) is in flight together; the partials meet through LDS and every thread normalises its pieces straight from
            // registers -- no reduction over the rows, no fp32 staging copy
            constexpr int NCH = (NS + 7) / 8;
            constexpr int NB2 = NCH == 1 ? 16 : 8;                  // samples whose pieces a thread holds at once (register budget)
            float4 v[NCH][NB2], gq[NCH];
            auto load_half = [&](int h0) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int i = threadIdx.x * 4 + ch * DEC4_T * 4;
                    const bool in = i < K;
#pragma unroll
                    for (int b = 0; b < NB2; ++b)
                        v[ch][b] = (in && h0 + b < B) ? *(const float4*)(a_in + (size_t)(h0 + b) * a.ldin + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            auto norm_half = [&](int h0) {
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int i = threadIdx.x * 4 + ch * DEC4_T * 4;
                    if (i >= K) break;
                    const float4 g = gq[ch];
#pragma unroll
                    for (int b = 0; b < NB2; ++b) {
                        if (h0 + b < B) {
                            const float mean = stat[2 * (h0 + b)], rstd = stat[2 * (h0 + b) + 1];
                            const float4 x = v[ch][b];
                            const float y0 = (x.x - mean) * rstd * g.x, y1 = (x.y - mean) * rstd * g.y;
                            const float y2 = (x.z - mean) * rstd * g.z, y3 = (x.w - mean) * rstd * g.w;
                            u32x2 o;
                            o[0] = pack_h16_rne(y0, y1);
                            o[1] = pack_h16_rne(y2, y3);
                            const int rb = PL ? b : h0 + b;                       // (PL: the image holds one batch of NB2 samples at a time)
                            *(u32x2*)(xs + (size_t)rb * KP + i) = o;
                            if constexpr (PL) {
                                u32x2 l;
                                l[0] = pack_h16_rne(y0 - h16_lo_to_f(o[0]), y1 - h16_hi_to_f(o[0]));
                                l[1] = pack_h16_rne(y2 - h16_lo_to_f(o[1]), y3 - h16_hi_to_f(o[1]));
                                *(u32x2*)(xs_lo + (size_t)rb * KP + i) = l;
                            }
                        }
                    }
                }
            };
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int i = threadIdx.x * 4 + ch * DEC4_T * 4;
                gq[ch] = i < K ? *(const float4*)(a_gamma + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            load_half(0);
            // all of a thread's partials requested at once (a rolled loop here was one L2 round trip per iteration: 43 of them for the
            // FF-out launch); lane = sample + SB * slice with SB = 8 (B <= 8: 32 slices per workgroup) or 16
            float2 pv[22];                                          // 352 partials / 16 slices
            auto load_parts = [&](auto sb_tag) {
                constexpr int SB = decltype(sb_tag)::value, NSL = 256 / SB, NPI = (352 + NSL - 1) / NSL;
                const int b = lane & (SB - 1), sl = wave * (64 / SB) + lane / SB;
#pragma unroll
                for (int it = 0; it < NPI; ++it) {
                    const int pi = sl + NSL * it;
                    pv[it] = pi < a.nstat_in ? *(const float2*)(a.stat_in + (pi * DEC4_NB + b) * 2) : make_float2(0.f, 0.f);
                }
            };
            auto reduce_parts = [&](auto sb_tag) {
                constexpr int SB = decltype(sb_tag)::value, NSL = 256 / SB, NPI = (352 + NSL - 1) / NSL;
                float ps = 0.f, pq = 0.f;
#pragma unroll
                for (int it = 0; it < NPI; ++it) { ps += pv[it].x; pq += pv[it].y; }
#pragma unroll
                for (int m = SB; m < 64; m <<= 1) { ps += __shfl_xor(ps, m, 64); pq += __shfl_xor(pq, m, 64); }
                if (lane < SB) { red[(lane * 4 + wave) * 2] = ps; red[(lane * 4 + wave) * 2 + 1] = pq; }
            };
            if (B <= 8) load_parts(std::integral_constant<int, 8>{});
            else        load_parts(std::integral_constant<int, 16>{});
            load_weights();                                         // behind every activation-side request of this launch
            if (B <= 8) reduce_parts(std::integral_constant<int, 8>{});
            else        reduce_parts(std::integral_constant<int, 16>{});
            __syncthreads();
            if (threadIdx.x < DEC4_NB) {
                const int b = threadIdx.x;
                float ps = 0.f, pq = 0.f;
                for (int w = 0; w < 4; ++w) { ps += red[(b * 4 + w) * 2]; pq += red[(b * 4 + w) * 2 + 1]; }
                const float mean = ps / (float)a.Kstat;
                stat[2 * b] = mean; stat[2 * b + 1] = rsqrtf(fmaxf(pq / (float)a.Kstat - mean * mean, 0.f) + a.eps);
            }
            __syncthreads();
            static_assert(NB2 == DEC4_IMG(NS), "image rows");
            norm_half(0);
            if constexpr (PL) {
                __syncthreads();
                kloop(acc, B < NB2 ? B : NB2);
                if (B > NB2) {                                    // the second batch of samples through the same images
                    load_half(NB2);
                    __syncthreads();
                    norm_half(NB2);
                    __syncthreads();
                    kloop(acc2, B - NB2);
                }
            } else {
                if (B > NB2) { load_half(NB2); norm_half(NB2); }  // long rows (FF-out) at B > 8: the second eight samples in a second pass
            }
        } else if (ln_rows) {
            load_weights();
            float s8[NBR], q8[NBR];
#pragma unroll
            for (int b = 0; b < NBR; ++b) { s8[b] = 0.f; q8[b] = 0.f; }
            constexpr int NCH = (NS + 7) / 8;                       // 1024-element chunks of a row: K <= 128 NS
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int i = threadIdx.x * 4 + ch * DEC4_T * 4;
                if (i >= K) break;
                float4 v[NBR];
#pragma unroll
                for (int b = 0; b < NBR; ++b) v[b] = b < B ? *(const float4*)(a_in + (size_t)b * a.ldin + i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int b = 0; b < NBR; ++b) {
                    if (b < B) *(float4*)(xraw + (size_t)b * K + i) = v[b];
                    if (i + 0 < a.Kstat) { s8[b] += v[b].x; q8[b] += v[b].x * v[b].x; }
                    if (i + 1 < a.Kstat) { s8[b] += v[b].y; q8[b] += v[b].y * v[b].y; }
                    if (i + 2 < a.Kstat) { s8[b] += v[b].z; q8[b] += v[b].z * v[b].z; }
                    if (i + 3 < a.Kstat) { s8[b] += v[b].w; q8[b] += v[b].w * v[b].w; }
                }
            }
#pragma unroll
            for (int b = 0; b < NBR; ++b) {
                const float s = wave_sum(s8[b]), q = wave_sum(q8[b]);
                if (lane == 0 && b < B) { red[(b * 4 + wave) * 2] = s; red[(b * 4 + wave) * 2 + 1] = q; }
            }
            __syncthreads();
            if (threadIdx.x < B) {
                const int b = threadIdx.x;
                float s = 0.f, q = 0.f;
                for (int w = 0; w < 4; ++w) { s += red[(b * 4 + w) * 2]; q += red[(b * 4 + w) * 2 + 1]; }
                const float mean = s / (float)a.Kstat;
                stat[2 * b] = mean; stat[2 * b + 1] = rsqrtf(fmaxf(q / (float)a.Kstat - mean * mean, 0.f) + a.eps);
            }
            __syncthreads();
            for (int i = threadIdx.x * 4; i < K; i += DEC4_T * 4) {
                const float4 g = *(const float4*)(a_gamma + i);
#pragma unroll
                for (int b = 0; b < NBR; ++b) {
                    if (b < B) {
                        const float mean = stat[2 * b], rstd = stat[2 * b + 1];
                        const float4 v = *(const float4*)(xraw + (size_t)b * K + i);
                        u32x2 o;
                        o[0] = pack_h16_rne((v.x - mean) * rstd * g.x, (v.y - mean) * rstd * g.y);
                        o[1] = pack_h16_rne((v.z - mean) * rstd * g.z, (v.w - mean) * rstd * g.w);
                        *(u32x2*)(xs + (size_t)b * KP + i) = o;
                    }
                }
            }
        } else {
            bool wreq = false;
            for (int h0 = 0; h0 < B; h0 += NBR)                 // (B > 8: the second eight samples in a second pass)
            for (int i = threadIdx.x * 4; i < K; i += DEC4_T * 4) {
                float4 v[NBR];
#pragma unroll
                for (int b = 0; b < NBR; ++b) v[b] = h0 + b < B ? *(const float4*)(a_in + (size_t)(h0 + b) * a.ldin + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (!wreq) { load_weights(); wreq = true; }     // behind the first batch of activation requests
#pragma unroll
                for (int b = 0; b < NBR; ++b) {
                    if (h0 + b < B) {
                        u32x2 o;
                        o[0] = pack_h16_rne(v[b].x, v[b].y);
                        o[1] = pack_h16_rne(v[b].z, v[b].w);
                        *(u32x2*)(xs + (size_t)(h0 + b) * KP + i) = o;
                    }
                }
            }
            if (!wreq) load_weights();                          // (threads without an activation piece: K < 1024)
        }
    }
    if constexpr (!PL) {
        __syncthreads();
        kloop(acc, B);
    }
    // C layout: acc[e] = out[row 4 kq + e][sample r]
    if constexpr (PL && DEC4_IMG(NS) == 8) {
        if (r < 8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { part[(wave * 16 + 4 * kq + e) * 16 + r] = acc[e]; part[(wave * 16 + 4 * kq + e) * 16 + 8 + r] = acc2[e]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) part[(wave * 16 + 4 * kq + e) * 16 + r] = acc[e];
    }
    __syncthreads();
    {
        const int t = threadIdx.x;                              // (row, sample) = (t >> 4, t & 15)
        vals[t] = part[t] + part[256 + t] + part[512 + t] + part[768 + t];
    }
    if (MODE == DEC2_LNGEMV && a.nsl > 1) {
        // publish this slice's tile, take a ticket; the last arriver adds the slabs in slice order.  (cdna_hip_programming.md, split-K
        // reduction: plain slab stores -> every wave vmcnt(0) -> barrier -> one lane: agent-scope release fence, vmcnt(0), relaxed ticket;
        // last arriver: agent-scope acquire fence -> barrier -> plain loads.  Placement-independent; the XCD mapping above is speed only.)
        float* slab = a.sk_ws + ((size_t)tile * a.nsl) * 256;
        slab[(size_t)slice * 256 + threadIdx.x] = vals[threadIdx.x];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(a.sk_cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            red[0] = __int_as_float(ticket);
        }
        __syncthreads();
        if (__float_as_int(red[0]) != a.nsl - 1) return;
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            a.sk_cnt[tile] = 0;                                 // the next launch finds it zero (stream order)
        }
        __syncthreads();
        float sum = 0.f;
        for (int sl = 0; sl < a.nsl; ++sl) sum += slab[(size_t)sl * 256 + threadIdx.x];
        vals[threadIdx.x] = sum;
    }
    __syncthreads();
    // ---- epilogue ----
    if (MODE == DEC2_FFIN) {
        const int ld = 2 * a.Fp;
        for (int idx = threadIdx.x; idx < B * 8; idx += DEC4_T) {
            const int b = idx >> 3, cc = idx & 7, col = blockIdx.x * 8 + cc;
            if (col >= a.Fp) continue;
            const float hv = round_if(vals[cc * 16 + b], a.round_bf16);
            const float hg = round_if(vals[(8 + cc) * 16 + b], a.round_bf16);
            float* h0 = a.hist + (size_t)(b * 2) * ld;          // row p-2
            float* h1 = h0 + ld;                                 // row p-1
            // (pre_h = {h0[col], h1[col], h0[Fp + col], h1[Fp + col]}, pre_w = the six taps: requested at the top; idx == threadIdx.x)
            const float uv = pre_w[0] * pre_h[0] + pre_w[1] * pre_h[1] + pre_w[2] * hv;
            const float ug = pre_w[3] * pre_h[2] + pre_w[4] * pre_h[3] + pre_w[5] * hg;
            const float uo = dec_gelu(ug) * uv;
            a.u[(size_t)b * a.Fp + col] = uo;
            h0[col] = pre_h[1];      h0[a.Fp + col] = pre_h[3];
            h1[col] = hv;            h1[a.Fp + col] = hg;
            part[idx] = uo;                                      // `part` is free again (its sums are in `vals`): read back below for the partial sums
        }
        if (a.stat_out) {                                       // this workgroup's 8 channels of every sample: LayerNorm partial of u
            __syncthreads();
            if (threadIdx.x < B) {
                float ps = 0.f, pq = 0.f;
                for (int cc = 0; cc < 8; ++cc)
                    if (blockIdx.x * 8 + cc < a.Fp) { const float t = part[threadIdx.x * 8 + cc]; ps += t; pq += t * t; }
                a.stat_out[(blockIdx.x * DEC4_NB + threadIdx.x) * 2] = ps; a.stat_out[(blockIdx.x * DEC4_NB + threadIdx.x) * 2 + 1] = pq;
            }
        }
    } else if (MODE == DEC2_QKV) {
        const int pos = *a.pos_dev;
        for (int idx = threadIdx.x; idx < B * DEC4_ROWS; idx += DEC4_T) {
            const int b = idx / DEC4_ROWS, rr = idx - b * DEC4_ROWS, n = blockIdx.x * DEC4_ROWS + rr;
            const float v = vals[rr * 16 + b];
            if (n < HD) a.q[(size_t)b * HD + n] = v;
            else if (n < HD + 64) a.Kc[((size_t)b * a.Nmax + pos) * 64 + (n - HD)] = v;                  // raw: normalised by dec_attn
            else if (n < HD + 128) a.Vc[((size_t)b * a.Nmax + pos) * 64 + (n - HD - 64)] = round_if(v, a.round_bf16);
        }
    } else {
        for (int idx = threadIdx.x; idx < B * DEC4_ROWS; idx += DEC4_T) {
            const int b = idx / DEC4_ROWS, rr = idx - b * DEC4_ROWS, n = tile * DEC4_ROWS + rr;
            float v = 0.f;
            if (n < a.Nout) {
                v = vals[rr * 16 + b];
                if (a.res) v += MODE == DEC2_LNGEMV ? pre_res : a.res[(size_t)b * a.ldres + n];      // (requested at the top; idx == threadIdx.x)
                a.out[(size_t)b * a.ldout + n] = v;
            }
            if (a.stat_out) part[idx] = v;                      // (b, row) order; `part` is free again: its sums are in `vals`
        }
        if (a.stat_out) {                                       // this workgroup's 16 output rows of every sample: LayerNorm partial
            __syncthreads();
            if (threadIdx.x < B) {
                float ps = 0.f, pq = 0.f;
                for (int rr = 0; rr < DEC4_ROWS; ++rr) { const float t = part[threadIdx.x * DEC4_ROWS + rr]; ps += t; pq += t * t; }
                a.stat_out[(tile * DEC4_NB + threadIdx.x) * 2] = ps; a.stat_out[(tile * DEC4_NB + threadIdx.x) * 2 + 1] = pq;
            }
        }
        if (MODE == DEC2_LNGEMV && blockIdx.x == 0 && threadIdx.x == 0 && a.adv_pos) {      // see dec3_kernel
            a.adv_pos[0] += 1;
            if (a.adv_step) a.adv_step[0] += 1;
        }
    }
}

template <int NS, int MODE, bool PL = false>
static void dec4_launch(const dec2_args& a, int grid, hipStream_t st) {
    const int kmax = a.nsl > 1 ? 32 * (((a.K >> 5) + a.nsl - 1) / a.nsl) : a.K;       // longest k-range of a workgroup
    const size_t lds = (((size_t)(PL ? 2 * DEC4_IMG(NS) : a.B) * (kmax + 8) * 2 + 15) & ~(size_t)15) + (size_t)(DEC4_NB * 8 + DEC4_NB * 2 + 4 * 256 + 256) * sizeof(float) +
                       ((a.gamma && !a.stat_in) ? (size_t)a.B * kmax * sizeof(float) : 0);      // fp32 staging copy: own-reduction path only (B <= 8)
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)dec4_kernel<NS, MODE, PL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((dec4_kernel<NS, MODE, PL>), dim3(grid), dim3(DEC4_T), lds, st, a);
}
// the matrix-core step kernels serve 16-bit weights with D, H * 64, Fp multiples of 32 and k-loops of at most 4 x 24 steps
static bool dec4_ok(const omlm_decode_args& a) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("OMLM_DECODE_MFMA"); off = (e && e[0] == '0') ? 1 : 0; }
    // B > 8: every LayerNorm in front of a matrix-core kernel must find its statistics in the producers' partials (the own-reduction
    // path stages fp32 rows for at most 8 samples): a.ln_parts given
    const bool wide_ok = a.B <= 8 || a.ln_parts != nullptr;
    return !off && a.B >= 2 && a.B <= DEC4_NB && wide_ok && a.D % 32 == 0 && a.D <= 1024 && (a.H * 64) % 32 == 0 && a.H * 64 <= 1024 &&
           a.Fp % 32 == 0 && a.Fp <= 3072 && a.Fp % 8 == 0;
}

template <typename TW, int NI, int MODE>
static void dec3_launch(const dec2_args& a, int units, hipStream_t st) {
    hipLaunchKernelGGL((dec3_kernel<TW, NI, MODE>), dim3((units + 3) / 4), dim3(DEC_T), 0, st, a);
}

template <typename TW>
static int decode_step2_t(const omlm_decode_args& a, const long long* ids, hipStream_t st) {
    const int B = a.B, D = a.D, H = a.H, Fp = a.Fp, HD = H * 64;
    const size_t lds_at2 = (size_t)(64 * 65 + 64 * 64 + H * 64 + 8 * 64) * sizeof(float);
    const bool mfma = sizeof(TW) == 2 && dec4_ok(a);              // 16-bit weights, B >= 2: the matrix-core step kernels
    // LayerNorm partial sums of the matrix-core kernels (see dec2_args::stat_in): three regions of a.ln_parts -- x (written by the embedding
    // gather and by every FF-out launch, read by the q rows and the head), x1 (to_out -> FF-in), u (FF-in -> FF-out)
    const int npd = (D + DEC4_ROWS - 1) / DEC4_ROWS, npf = (Fp + 7) / 8, region = (npd > npf ? npd : npf) * 2 * DEC4_NB;
    float* st_x = (mfma && a.ln_parts) ? a.ln_parts : nullptr;
    float* st_x1 = st_x ? st_x + region : nullptr;
    float* st_u = st_x ? st_x + 2 * region : nullptr;
    int n_x = 0;                                                  // partials of x that are valid right now
    if (a.emb_table) {
        hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(DEC_T), 0, st, ids, a.emb_table, a.emb_row_offset, a.emb_rows, a.x, D, st_x);
        n_x = 1;
    } else if (st_x && B > DEC_BMAX) {
        // the caller embedded the ids itself (sampler + gather kernel): one small launch leaves the rows' sums where the first q rows look for them
        hipLaunchKernelGGL(dec_rowstat_kernel, dim3(B), dim3(DEC_T), 0, st, a.x, D, st_x);
        n_x = 1;
    }
    dec2_args g;
    memset(&g, 0, sizeof(g));
    g.B = B; g.round_bf16 = a.round_bf16; g.eps = a.eps; g.H = H; g.nsplit = a.nsplit; g.pos_dev = a.pos_dev; g.Nmax = a.Nmax; g.Fp = Fp;
    // "fp16ff": FF-in / FF-out / head read W = hi + lo and keep their activations and h1 un-rounded (round_bf16 = 0 for those launches)
    const bool pl = a.W1p_lo != nullptr;
    // FF-out rows cut into four k-slices (see dec4_kernel): the matrix-core kernels with the producers' LayerNorm partials, scratch given
    static int sk_off = -1;
    if (sk_off < 0) { const char* e = getenv("OMLM_DECODE_SPLITK"); sk_off = (e && e[0] == '0') ? 1 : 0; }
    const bool split = !sk_off && mfma && st_x && a.splitk_ws && a.splitk_cnt && (Fp >> 5) >= 8;
    if (pl) {
        OMLM_CHECK_ARG(sizeof(TW) == 2 && a.W2p_lo && (!a.head_W || a.head_W_lo), "lo planes: 16-bit weights, all three families");
        OMLM_CHECK_ARG(B == 1 || (mfma && st_x), "lo planes at B >= 2 run on the matrix-core step kernels (ln_parts given, OMLM_DECODE_MFMA unset)");
        OMLM_CHECK_ARG(Fp <= 3072, "lo planes: feed-forward width <= 3072");
    }
    for (int l = 0; l < a.L; ++l) {
        dec2_args q = g;                                                               // q / k / v rows of the new token
        q.in = a.x; q.ldin = D; q.K = D; q.Kstat = D; q.gamma = a.attn_gamma[l]; q.W = a.Wq[l]; q.W2 = a.Wkv[l]; q.ldw = D;
        q.Nout = HD + 128; q.q = a.q; q.Kc = a.Kc[l]; q.Vc = a.Vc[l];
        if (st_x && n_x > 0) { q.stat_in = st_x; q.nstat_in = n_x; }
        if (B == 1) dec3_launch<TW, 2, DEC2_QKV>(q, HD + 128, st);
        else if (mfma) dec4_launch<8, DEC2_QKV>(q, (HD + 128) / DEC4_ROWS, st);
        else        dec2_launch<TW, 2, DEC2_QKV>(q, (HD + 128) / DEC2_ROWS, st);
        const bool comb_in_attn = mfma && a.splitk_cnt != nullptr;                     // the last workgroup of a sample combines its partials
        hipLaunchKernelGGL(dec_attn2_kernel, dim3(a.nsplit, B), dim3(DEC_AT2), lds_at2, st, a.q, a.Kc[l], a.Vc[l], a.q_scale[l], a.k_scale[l],
                           a.bias_table, a.bias_ld, a.parts, H, a.Nmax, a.nsplit, a.pos_dev, a.scale, a.round_bf16,
                           comb_in_attn ? a.q : (float*)nullptr, comb_in_attn ? a.splitk_cnt : (int*)nullptr);
        dec2_args o = g;                                                               // x1 = x + attn Wo^T
        o.K = HD; o.parts = a.parts; o.W = a.Wo[l]; o.ldw = HD; o.Nout = D; o.res = a.x; o.ldres = D; o.out = a.x1; o.ldout = D;
        if (mfma) {                                                                    // combine once, then a plain (no LayerNorm) row product
            if (!comb_in_attn) hipLaunchKernelGGL(dec_attn_combine_kernel, dim3(H, B), dim3(64), 0, st, a.parts, a.q, a.nsplit, H, a.pos_dev, a.round_bf16);
            o.in = a.q; o.ldin = HD; o.Kstat = HD; o.gamma = nullptr; o.parts = nullptr;     // a.q is free again: the attention kernel consumed it
            o.stat_out = st_x1;
            dec4_launch<8, DEC2_LNGEMV>(o, (D + DEC4_ROWS - 1) / DEC4_ROWS, st);
        } else if (HD <= 512) dec2_launch<TW, 1, DEC2_OUT>(o, (D + DEC2_ROWS - 1) / DEC2_ROWS, st);
        else                dec2_launch<TW, 2, DEC2_OUT>(o, (D + DEC2_ROWS - 1) / DEC2_ROWS, st);
        dec2_args f = g;                                                               // FF-in rows + conv + GEGLU
        f.in = a.x1; f.ldin = D; f.K = D; f.Kstat = D; f.gamma = a.ffin_gamma[l]; f.W = a.W1p[l]; f.ldw = D; f.convw = a.convw[l];
        f.hist = a.hist[l]; f.u = a.u;
        if (st_x1) { f.stat_in = st_x1; f.nstat_in = npd; f.stat_out = st_u; }
        if (pl) { f.Wlo = a.W1p_lo[l]; f.round_bf16 = 0; }
        if constexpr (sizeof(TW) == 2) {
            if (pl && B == 1) {
                static int cpwl = -1;
                if (cpwl < 0) { const char* e = getenv("OMLM_DECODE_CPW_PL"); cpwl = e ? atoi(e) : 2; }
                if (cpwl == 4) hipLaunchKernelGGL((dec3_ffin_kernel<TW, 4, true>), dim3((Fp + 15) / 16), dim3(DEC_T), 0, st, f);
                else           hipLaunchKernelGGL((dec3_ffin_kernel<TW, 2, true>), dim3((Fp + 7) / 8), dim3(DEC_T), 0, st, f);
            }
            else if (pl)      dec4_launch<8, DEC2_FFIN, true>(f, (Fp + 7) / 8, st);
        }
        if (pl) {
        } else if (B == 1) {
            static int cpw = -1;
            if (cpw < 0) { const char* e = getenv("OMLM_DECODE_CPW"); cpw = e ? atoi(e) : 4; }
            if (cpw == 4)      hipLaunchKernelGGL((dec3_ffin_kernel<TW, 4>), dim3((Fp + 15) / 16), dim3(DEC_T), 0, st, f);
            else if (cpw == 2) hipLaunchKernelGGL((dec3_ffin_kernel<TW, 2>), dim3((Fp + 7) / 8), dim3(DEC_T), 0, st, f);
            else               dec3_launch<TW, 2, DEC2_FFIN>(f, Fp, st);
        } else if (mfma) dec4_launch<8, DEC2_FFIN>(f, (Fp + 7) / 8, st);
        else dec2_launch<TW, 2, DEC2_FFIN>(f, Fp / 2, st);
        dec2_args w = g;                                                               // x = x1 + LN(u) W2^T
        w.in = a.u; w.ldin = Fp; w.K = Fp; w.Kstat = a.F; w.gamma = a.mid_gamma[l]; w.W = a.W2p[l]; w.ldw = Fp; w.Nout = D;
        w.res = a.x1; w.ldres = D; w.out = a.x; w.ldout = D;
        if (st_u) { w.stat_in = st_u; w.nstat_in = npf; w.stat_out = st_x; n_x = npd; }
        if (pl) { w.Wlo = a.W2p_lo[l]; w.round_bf16 = 0; }
        if constexpr (sizeof(TW) == 2) {
            if (pl && B == 1) hipLaunchKernelGGL((dec3_kernel<TW, 6, DEC2_LNGEMV, true>), dim3((D + 3) / 4), dim3(DEC_T), 0, st, w);
            else if (pl && split) { w.nsl = 4; w.sk_ws = a.splitk_ws; w.sk_cnt = a.splitk_cnt; dec4_launch<6, DEC2_LNGEMV, true>(w, 4 * ((D + DEC4_ROWS - 1) / DEC4_ROWS), st); }
            else if (pl)      dec4_launch<24, DEC2_LNGEMV, true>(w, (D + DEC4_ROWS - 1) / DEC4_ROWS, st);
        }
        if (pl) {
        } else if (B == 1 && Fp <= 3072) dec3_launch<TW, 6, DEC2_LNGEMV>(w, D, st);
        else if (mfma && split) { w.nsl = 4; w.sk_ws = a.splitk_ws; w.sk_cnt = a.splitk_cnt; dec4_launch<6, DEC2_LNGEMV>(w, 4 * ((D + DEC4_ROWS - 1) / DEC4_ROWS), st); }
        else if (mfma) dec4_launch<24, DEC2_LNGEMV>(w, (D + DEC4_ROWS - 1) / DEC4_ROWS, st);
        else if (Fp <= 3072) dec2_launch<TW, 6, DEC2_LNGEMV>(w, (D + DEC2_ROWS - 1) / DEC2_ROWS, st);
        else                 dec2_launch<TW, 8, DEC2_LNGEMV>(w, (D + DEC2_ROWS - 1) / DEC2_ROWS, st);
    }
    if (a.head_W) {
        dec2_args h = g;
        h.in = a.x; h.ldin = D; h.K = D; h.Kstat = D; h.gamma = a.final_gamma; h.W = a.head_W; h.ldw = D; h.Nout = a.V1;
        h.out = a.logits; h.ldout = a.ldV;
        h.adv_pos = a.advance_pos; h.adv_step = a.advance_step;
        if (st_x && n_x > 0) { h.stat_in = st_x; h.nstat_in = n_x; }
        if (pl) {
            OMLM_CHECK_ARG(B == 1 || h.stat_in, "lo planes: the head needs the LayerNorm partials of the last FF-out launch (L >= 1)");
            h.Wlo = a.head_W_lo; h.round_bf16 = 0;
        }
        if constexpr (sizeof(TW) == 2) {
            if (pl && B == 1) hipLaunchKernelGGL((dec3_kernel<TW, 2, DEC2_LNGEMV, true>), dim3((a.V1 + 3) / 4), dim3(DEC_T), 0, st, h);
            else if (pl)      dec4_launch<8, DEC2_LNGEMV, true>(h, (a.V1 + DEC4_ROWS - 1) / DEC4_ROWS, st);
        }
        if (pl) {
        } else if (B == 1) dec3_launch<TW, 2, DEC2_LNGEMV>(h, a.V1, st);
        else if (mfma) dec4_launch<8, DEC2_LNGEMV>(h, (a.V1 + DEC4_ROWS - 1) / DEC4_ROWS, st);
        else        dec2_launch<TW, 2, DEC2_LNGEMV>(h, (a.V1 + DEC2_ROWS - 1) / DEC2_ROWS, st);
    }
    return omlm_post_launch("omlm_decode_step");
}

// ---- end of step: the row index (and the sampler's step counter) move on, on the device ----
__global__ void dec_advance_kernel(int* pos_dev, int* step_dev) {
    if (threadIdx.x == 0) { if (pos_dev) pos_dev[0] += 1; if (step_dev) step_dev[0] += 1; }
}
#if OMLM_FP16
extern "C" int omlm_decode_advance(int* pos_dev, int* step_dev, void* stream);      // bf16 copy of this file
#else
extern "C" int omlm_decode_advance(int* pos_dev, int* step_dev, void* stream) {
    hipLaunchKernelGGL(dec_advance_kernel, dim3(1), dim3(64), 0, as_stream(stream), pos_dev, step_dev);
    return omlm_post_launch("omlm_decode_advance");
}
#endif

template <typename TW>
static int decode_step_t(const omlm_decode_args& a, const long long* ids, hipStream_t st) {
    const int B = a.B, D = a.D, H = a.H, Fp = a.Fp;
    const size_t tail = (DEC_ROWS * DEC_BMAX + 16) * sizeof(float);
    const size_t lds_d = (size_t)B * D * sizeof(float) + tail;
    const size_t lds_hd = (size_t)B * H * 64 * sizeof(float) + tail;
    const size_t lds_fp = (size_t)B * Fp * sizeof(float) + tail;
    const size_t lds_at = (size_t)(64 * 65 + 64 * 64 + 2 * H * 64) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)dec_gemv_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)dec_qkv_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)dec_ffin_kernel<TW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)dec_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int HD = H * 64;
    const bool pl = a.W1p_lo != nullptr;
    if (pl) OMLM_CHECK_ARG(a.W2p_lo && (!a.head_W || a.head_W_lo), "lo planes: all three families");
    if (a.emb_table)
        hipLaunchKernelGGL(dec_embed_kernel, dim3(B), dim3(DEC_T), 0, st, ids, a.emb_table, a.emb_row_offset, a.emb_rows, a.x, D, (float*)nullptr);
    for (int l = 0; l < a.L; ++l) {
        hipLaunchKernelGGL((dec_qkv_kernel<TW>), dim3((HD + 128) / DEC_ROWS), dim3(DEC_T), lds_d, st, a.x, a.attn_gamma[l],
                           (const TW*)a.Wq[l], (const TW*)a.Wkv[l], a.q, a.Kc[l], a.Vc[l], B, D, H, a.Nmax, a.pos_dev, a.eps, a.round_bf16);
        hipLaunchKernelGGL(dec_attn_kernel, dim3(a.nsplit, B), dim3(DEC_T), lds_at, st, a.q, a.Kc[l], a.Vc[l], a.q_scale[l], a.k_scale[l],
                           a.bias_table, a.bias_ld, a.parts, H, a.Nmax, a.nsplit, a.pos_dev, a.scale, a.round_bf16);
        hipLaunchKernelGGL((dec_gemv_kernel<TW>), dim3((D + DEC_ROWS - 1) / DEC_ROWS), dim3(DEC_T), lds_hd, st, (const float*)nullptr, 0,
                           (const float*)nullptr, 0, a.eps, a.parts, a.nsplit, H, a.pos_dev, (const TW*)a.Wo[l], (long long)HD, HD, D,
                           a.x, D, a.x1, D, B, a.round_bf16);
        // (pl, "fp16ff": W = hi + lo for FF-in / FF-out / head, their activations and h1 un-rounded)
        hipLaunchKernelGGL((dec_ffin_kernel<TW>), dim3(Fp / 8), dim3(DEC_T), lds_d, st, a.x1, a.ffin_gamma[l], (const TW*)a.W1p[l],
                           a.convw[l], a.hist[l], a.u, B, D, Fp, a.eps, pl ? 0 : a.round_bf16, pl ? (const TW*)a.W1p_lo[l] : (const TW*)nullptr);
        hipLaunchKernelGGL((dec_gemv_kernel<TW>), dim3((D + DEC_ROWS - 1) / DEC_ROWS), dim3(DEC_T), lds_fp, st, a.u, Fp, a.mid_gamma[l],
                           a.F, a.eps, (const float*)nullptr, 0, 0, (const int*)nullptr, (const TW*)a.W2p[l], (long long)Fp, Fp, D,
                           a.x1, D, a.x, D, B, pl ? 0 : a.round_bf16, pl ? (const TW*)a.W2p_lo[l] : (const TW*)nullptr);
    }
    if (a.head_W)
        hipLaunchKernelGGL((dec_gemv_kernel<TW>), dim3((a.V1 + DEC_ROWS - 1) / DEC_ROWS), dim3(DEC_T), lds_d, st, a.x, D, a.final_gamma,
                           D, a.eps, (const float*)nullptr, 0, 0, (const int*)nullptr, (const TW*)a.head_W, (long long)D, D, a.V1,
                           (const float*)nullptr, 0, a.logits, a.ldV, B, pl ? 0 : a.round_bf16, pl ? (const TW*)a.head_W_lo : (const TW*)nullptr);
    return omlm_post_launch("omlm_decode_step");
}

// One decode step for the row at index *pos_dev (see include/omlm.h).  ids: [B] int64 sampled in the previous step (used
// when a->emb_table is set; otherwise a->x already holds the new row's embedding).  *pos_dev is advanced only through
// a->advance_pos / a->advance_step (optional device counters bumped by the step's last kernel; pass pos_dev there to move on).
// a->w_dtype: 0 = fp32 weights, 1 = bf16, 2 = fp16 (served by the fp16 copy of this file; a->round_bf16 then rounds to fp16).
#if !OMLM_FP16
extern "C" int omlm_decode_step_h(const omlm_decode_args* a, const long long* ids, void* stream);
#endif
extern "C" int OMLM_API(omlm_decode_step)(const omlm_decode_args* a, const long long* ids, void* stream) {
    OMLM_CHECK_ARG(a != nullptr, "null argument block");
#if !OMLM_FP16
    if (a->w_dtype == OMLM_DT_F16) {
        omlm_decode_args b = *a;
        b.w_dtype = 1;
        return omlm_decode_step_h(&b, ids, stream);
    }
#else
    OMLM_CHECK_ARG(a->w_dtype == 1, "the fp16 copy serves fp16 weights only");
#endif
    OMLM_CHECK_ARG(a->B >= 1 && a->B <= DEC4_NB, "decode batch must be 1..16");
    OMLM_CHECK_ARG(a->B <= DEC_BMAX || (a->w_dtype != 0 && a->D == 1024 && dec4_ok(*a)),
                   "decode batches of 9..16 run on the matrix-core kernels only: 16-bit weights, D = 1024, ln_parts given");
    OMLM_CHECK_ARG(a->D % 8 == 0 && a->Fp % 8 == 0 && a->pos_dev && a->parts, "decode geometry");
    OMLM_CHECK_ARG(a->H >= 1 && a->H <= 16 && (a->H * 64 + 128) % DEC_ROWS == 0, "heads");
    OMLM_CHECK_ARG(a->nsplit * DEC_KS >= a->Nmax, "nsplit must cover Nmax keys");
    OMLM_CHECK_ARG(a->B > DEC_BMAX || (size_t)a->B * a->Fp * sizeof(float) + 1024 <= 150 * 1024, "B * Fp exceeds the LDS budget");
    OMLM_CHECK_ARG(!a->emb_table || ids, "ids required with an embedding table");
    static int v1 = -1;
    if (v1 < 0) { const char* e = getenv("OMLM_DECODE_V1"); v1 = (e && e[0] == '1') ? 1 : 0; }
    const bool v2_ok = a->D == 1024 && a->H * 64 <= 1024 && a->Fp <= 4096 && a->Fp % 2 == 0 && (a->H * 64 + 128) % DEC2_ROWS == 0;
    // batches of 9..16 exist only on the second-generation path's matrix-core kernels: the first-generation kernels below are sized for
    // DEC_BMAX samples (LDS B * Fp floats, DEC_ROWS * DEC_BMAX tail) -- refuse instead of overrunning them (OMLM_DECODE_V1=1, odd geometries)
    OMLM_CHECK_ARG(a->B <= DEC_BMAX || (!v1 && v2_ok), "decode batches above 8 need the second-generation step kernels (D = 1024, OMLM_DECODE_V1 unset)");
    if (!v1 && v2_ok) {
#if OMLM_FP16
        const int rc = decode_step2_t<h16_t>(*a, ids, as_stream(stream));
#else
        const int rc = a->w_dtype == 0 ? decode_step2_t<float>(*a, ids, as_stream(stream)) : decode_step2_t<h16_t>(*a, ids, as_stream(stream));
#endif
        if (rc == OMLM_OK && a->advance_pos && !a->head_W) return omlm_decode_advance(a->advance_pos, a->advance_step, stream);
        return rc;
    }
#if OMLM_FP16
    const int rc = decode_step_t<h16_t>(*a, ids, as_stream(stream));
#else
    const int rc = a->w_dtype == 0 ? decode_step_t<float>(*a, ids, as_stream(stream)) : decode_step_t<h16_t>(*a, ids, as_stream(stream));
#endif
    if (rc == OMLM_OK && a->advance_pos) return omlm_decode_advance(a->advance_pos, a->advance_step, stream);    // first-generation kernels: own launch
    return rc;
}

}   // namespace OMLM_NS
