// Token-side kernels of TokenConditionedTransformer (reference open_musiclm.py:116-145, :389-410):
//   * fused embedding gather: per-quantizer offset ids -> rows of the per-sequence tables, pad -> 0,
//     learned start tokens interleaved, optional absolute position rows added, written straight into the
//     concatenated [B, N, D] trunk input (the reference builds 2*n_seq tensors and torch.cat's them).
//   * its transpose (scatter-add with the grad_shrink factor, utils.py:60-61).
//   * cross-entropy forward (per-row log-sum-exp + NLL sum) and backward (softmax - onehot) over the padded
//     logits rows produced by the head GEMMs.
#include "common.h"

#define MAX_SEQ 4
struct EmbedTables {
    const float* table[MAX_SEQ];
    const float* start[MAX_SEQ];
    const float* pos[MAX_SEQ];      // absolute position tables or null
    float* dtable[MAX_SEQ];
    float* dstart[MAX_SEQ];
    float* dpos[MAX_SEQ];
    long long rows[MAX_SEQ];        // table rows / position rows per sequence (0: unchecked)
    long long prows[MAX_SEQ];
    int* err;                       // device flag: bit 0 = token id past its table, bit 1 = position past its table, bit 2 = label >= V
};

// ids [B, N] int32: >= 0 table row (offsets already applied), -1 pad (zero row), -2 start token.
// seg [N] int32: sequence index of the position; posidx [N]: index inside its sequence (for abs-pos tables).
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ ids, const int* __restrict__ seg,
                                                        const int* __restrict__ posidx, EmbedTables t,
                                                        float* __restrict__ out, int B, int N, int D) {
    const int nv = D / 4;
    for (long long row = blockIdx.x; row < (long long)B * N; row += gridDim.x) {
        const int n = (int)(row % N);
        int id = ids[row];
        const int s = seg[n];
        // an id / position past its table is a corrupted input (wrong codebook_size, damaged token store): torch's embedding
        // raises a device assert there; here the row is treated as a pad and the caller's flag is raised (never an OOB access)
        if (id >= 0 && t.rows[s] > 0 && id >= t.rows[s]) { if (t.err && threadIdx.x == 0) atomicOr(t.err, 1); id = -1; }
        const float4* src = nullptr;
        if (id >= 0) src = (const float4*)(t.table[s] + (size_t)id * D);
        else if (id == -2) src = (const float4*)t.start[s];
        bool pos_ok = (id >= 0 || id == -1) && t.pos[s];
        if (pos_ok && t.prows[s] > 0 && posidx[n] >= t.prows[s]) { if (t.err && threadIdx.x == 0) atomicOr(t.err, 2); pos_ok = false; }
        const float4* ps = pos_ok ? (const float4*)(t.pos[s] + (size_t)posidx[n] * D) : nullptr;
        float4* dst = (float4*)(out + (size_t)row * D);
        for (int c = threadIdx.x; c < nv; c += 256) {
            float4 v = src ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (ps) { const float4 p = ps[c]; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
            dst[c] = v;
        }
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ ids, const int* __restrict__ seg,
                                                        const int* __restrict__ posidx, EmbedTables t,
                                                        const float* __restrict__ dx, int B, int N, int D, float alpha) {
    for (long long row = blockIdx.x; row < (long long)B * N; row += gridDim.x) {
        const int n = (int)(row % N);
        int id = ids[row];
        const int s = seg[n];
        if (id >= 0 && t.rows[s] > 0 && id >= t.rows[s]) { if (t.err && threadIdx.x == 0) atomicOr(t.err, 1); id = -1; }
        float* dst = nullptr;
        if (id >= 0) dst = t.dtable[s] ? t.dtable[s] + (size_t)id * D : nullptr;
        else if (id == -2) dst = t.dstart[s];
        bool pos_ok = (id >= 0 || id == -1) && t.dpos[s];
        if (pos_ok && t.prows[s] > 0 && posidx[n] >= t.prows[s]) { if (t.err && threadIdx.x == 0) atomicOr(t.err, 2); pos_ok = false; }
        float* pd = pos_ok ? t.dpos[s] + (size_t)posidx[n] * D : nullptr;
        const float* src = dx + (size_t)row * D;
        for (int c = threadIdx.x; c < D; c += 256) {
            const float g = src[c] * alpha;
            if (dst) unsafeAtomicAdd(dst + c, g);
            if (pd) unsafeAtomicAdd(pd + c, g);
        }
    }
}

static int fill_tables(EmbedTables& t, const float* const* tables, const float* const* starts, const float* const* pos,
                       float* const* dtables, float* const* dstarts, float* const* dpos, int nseq,
                       const long long* table_rows, const long long* pos_rows, int* err_flag) {
    memset(&t, 0, sizeof(t));
    t.err = err_flag;
    for (int i = 0; i < nseq; ++i) {
        if (table_rows) t.rows[i] = table_rows[i];
        if (pos_rows) t.prows[i] = pos_rows[i];
        if (tables) t.table[i] = tables[i];
        if (starts) t.start[i] = starts[i];
        if (pos) t.pos[i] = pos[i];
        if (dtables) t.dtable[i] = dtables[i];
        if (dstarts) t.dstart[i] = dstarts[i];
        if (dpos) t.dpos[i] = dpos[i];
    }
    return 0;
}

// tables/starts/pos: host arrays (length nseq) of DEVICE pointers.  table_rows / pos_rows: host arrays (length nseq) of the tables'
// row counts (null: unchecked); err_flag: device int (null: out-of-range rows are still skipped, silently).
extern "C" int omlm_embed_gather_fwd(const int* ids, const int* seg, const int* posidx,
                                     const float* const* tables, const float* const* starts, const float* const* pos,
                                     int nseq, float* out, int B, int N, int D,
                                     const long long* table_rows, const long long* pos_rows, int* err_flag, void* stream) {
    if (B <= 0 || N <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(ids && seg && tables && starts && out, "null pointer");
    OMLM_CHECK_ARG(nseq >= 1 && nseq <= MAX_SEQ, "1..4 token sequences supported");
    OMLM_CHECK_ARG(D % 4 == 0, "D % 4");
    OMLM_CHECK_ARG(!pos || posidx, "posidx required with position tables");
    EmbedTables t;
    fill_tables(t, tables, starts, pos, nullptr, nullptr, nullptr, nseq, table_rows, pos_rows, err_flag);
    long long rows = (long long)B * N;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)(rows < 16384 ? rows : 16384)), dim3(256), 0, as_stream(stream), ids, seg, posidx, t, out, B, N, D);
    return omlm_post_launch("omlm_embed_gather_fwd");
}

// accumulates (+=) alpha * dx rows into the table / start-token / position gradients.
extern "C" int omlm_embed_gather_bwd(const int* ids, const int* seg, const int* posidx,
                                     float* const* dtables, float* const* dstarts, float* const* dpos,
                                     int nseq, const float* dx, int B, int N, int D, float alpha,
                                     const long long* table_rows, const long long* pos_rows, int* err_flag, void* stream) {
    if (B <= 0 || N <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(ids && seg && dtables && dstarts && dx, "null pointer");
    OMLM_CHECK_ARG(nseq >= 1 && nseq <= MAX_SEQ, "1..4 token sequences supported");
    EmbedTables t;
    fill_tables(t, nullptr, nullptr, nullptr, dtables, dstarts, dpos, nseq, table_rows, pos_rows, err_flag);
    long long rows = (long long)B * N;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((unsigned)(rows < 16384 ? rows : 16384)), dim3(256), 0, as_stream(stream), ids, seg, posidx, t, dx, B, N, D, alpha);
    return omlm_post_launch("omlm_embed_gather_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// cross entropy over rows of padded logits [R, ld] (first V columns valid).  Rows are addressed through an
// optional row map so the labels can stay in [B, n] order while logits live in the head-GEMM layout.
// label < 0 -> row ignored (ignore_index semantics).
// One WAVE per row (round 6): a row of V <= 64 * CE_NV logits sits in registers (element c = lane + 64 j; every load unconditional on a clamped
// index and consumed together -- predicated loads were serialised by hipcc, one wait each), maximum and sum are two wave reductions, no LDS,
// no barrier.  The workgroup-per-row form below walks its rows through three barriers each with 4-byte loads (83 us for 118 MB at coarse-small:
// 1.5 TB/s); it stays for wider rows.
#define CE_NV 17
__global__ __launch_bounds__(256) void ce_fwd_wave_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                          float* __restrict__ row_lse, float* __restrict__ nll_sum,
                                                          int R, int V, int ld, int* __restrict__ err) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float local = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const float* lr = logits + (size_t)row * ld;
        float v[CE_NV];
#pragma unroll
        for (int j = 0; j < CE_NV; ++j) { const int c = lane + 64 * j; v[j] = lr[c < V ? c : V - 1]; }
        const int lb = labels[row];
        const float own = lr[(lb >= 0 && lb < V) ? lb : 0];
#pragma unroll
        for (int j = 0; j < CE_NV; ++j) asm volatile("" : "+v"(v[j]));
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < CE_NV; ++j) { if (lane + 64 * j >= V) v[j] = -INFINITY; mx = fmaxf(mx, v[j]); }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < CE_NV; ++j) s += __expf(v[j] - mx);          // (exp(-inf) = 0 for the slots past V)
        s = wave_sum(s);
        const float lse = mx + __logf(s);
        if (lane == 0) {
            row_lse[row] = lse;
            if (lb >= V) { if (err) atomicOr(err, 4); }            // a label past the vocabulary: row ignored, flag raised
            else if (lb >= 0) local += lse - own;
        }
    }
    // one atomic per WORKGROUP, and few workgroups: every add lands on the same word, and same-address atomics retire one at a time
    // (one per wave -- 16 k of them -- made this launch 370 us)
    __shared__ float wsum[4];
    if (lane == 0) wsum[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (t != 0.f) unsafeAtomicAdd(nll_sum, t);
    }
}

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                     float* __restrict__ row_lse, float* __restrict__ nll_sum,
                                                     int R, int V, int ld, int* __restrict__ err) {
    __shared__ float red[4];
    float local = 0.f;
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const float* lr = logits + (size_t)row * ld;
        float mx = -INFINITY;
        for (int c = threadIdx.x; c < V; c += 256) mx = fmaxf(mx, lr[c]);
        mx = wave_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 0.f;
        for (int c = threadIdx.x; c < V; c += 256) s += __expf(lr[c] - mx);
        s = block_sum<256>(s, red);
        const float lse = mx + __logf(s);
        if (threadIdx.x == 0) {
            row_lse[row] = lse;
            const int lb = labels[row];
            if (lb >= V) { if (err) atomicOr(err, 4); }            // a label past the vocabulary: row ignored, flag raised
            else if (lb >= 0) local += lse - lr[lb];
        }
    }
    if (threadIdx.x == 0 && local != 0.f) unsafeAtomicAdd(nll_sum, local);
}

// dlogits[row, c] = coef * g * (softmax - onehot) for c < V, 0 for V <= c < ldd
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                     const float* __restrict__ row_lse, const float* __restrict__ gscale,
                                                     float coef, T* __restrict__ dlogits, int R, int V, int ld, int ldd) {
    const float g = coef * (gscale ? gscale[0] : 1.0f);
    for (int row = blockIdx.x; row < R; row += gridDim.x) {
        const float* lr = logits + (size_t)row * ld;
        T* dr = dlogits + (size_t)row * ldd;
        const int lb = labels[row];
        const float lse = row_lse[row];
        for (int c = threadIdx.x; c < ldd; c += 256) {
            float v = 0.f;
            if (c < V && lb >= 0 && lb < V) v = g * (__expf(lr[c] - lse) - (c == lb ? 1.0f : 0.0f));
            store_from_float(dr + c, v);
        }
    }
}

// 16-bit outputs, one WAVE per row (round 6): a lane owns 8 consecutive columns per trip (two 16-byte loads, one 16-byte store) instead of one
// column per thread per trip with 2-byte stores (50 us for 125 MB in + 60 MB out at coarse-small)
template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_wave_kernel(const float* __restrict__ logits, const int* __restrict__ labels,
                                                          const float* __restrict__ row_lse, const float* __restrict__ gscale,
                                                          float coef, T* __restrict__ dlogits, int R, int V, int ld, int ldd) {
    const float g = coef * (gscale ? gscale[0] : 1.0f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const float* lr = logits + (size_t)row * ld;
        T* dr = dlogits + (size_t)row * ldd;
        const int lb = labels[row];
        const float lse = row_lse[row];
        const bool live = lb >= 0 && lb < V;
        for (int c = lane * 8; c < ldd; c += 512) {
            float x[8];
            if (c + 8 <= V) {
                const float4 a = *(const float4*)(lr + c), b = *(const float4*)(lr + c + 4);
                x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = c + e < V ? lr[c + e] : 0.f;
            }
            union { T h[8]; uint4 u; } pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = 0.f;
                if (c + e < V && live) v = g * (__expf(x[e] - lse) - (c + e == lb ? 1.0f : 0.0f));
                pk.h[e] = (T)v;
            }
            *(uint4*)(dr + c) = pk.u;
        }
    }
}

extern "C" int omlm_cross_entropy_fwd(const float* logits, const int* labels, float* row_lse, float* nll_sum,
                                      int R, int V, int ld, int* err_flag, void* stream) {
    if (R <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(logits && labels && row_lse && nll_sum && ld >= V, "cross entropy arguments");
    if (V <= 64 * CE_NV) {
        const int blocks = (R + 3) / 4;
        hipLaunchKernelGGL(ce_fwd_wave_kernel, dim3(blocks < 1024 ? blocks : 1024), dim3(256), 0, as_stream(stream), logits, labels, row_lse, nll_sum, R, V, ld, err_flag);
    } else
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(R < 4096 ? R : 4096), dim3(256), 0, as_stream(stream), logits, labels, row_lse, nll_sum, R, V, ld, err_flag);
    return omlm_post_launch("omlm_cross_entropy_fwd");
}

extern "C" int omlm_cross_entropy_bwd(const float* logits, const int* labels, const float* row_lse, const float* gscale,
                                      float coef, void* dlogits, int R, int V, int ld, int ldd, int out_dtype, void* stream) {
    if (R <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(logits && labels && row_lse && dlogits && ld >= V && ldd >= V, "cross entropy arguments");
    dim3 grid(R < 8192 ? R : 8192), block(256);
    OMLM_CHECK_ARG(out_dtype >= 0 && out_dtype <= 2, "out_dtype: 0 = fp32, 1 = bf16, 2 = fp16");
    // 16-bit outputs with 16-byte friendly pitches: the wave-per-row kernel
    const bool wave_ok = out_dtype != 0 && (ldd & 7) == 0 && (ld & 3) == 0 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0;
    if (wave_ok) {
        const int blocks = (R + 3) / 4;
        dim3 wgrid(blocks < 8192 ? blocks : 8192);
        if (out_dtype == OMLM_DT_F16)
            hipLaunchKernelGGL(ce_bwd_wave_kernel<f16_t>, wgrid, block, 0, as_stream(stream), logits, labels, row_lse, gscale, coef, (f16_t*)dlogits, R, V, ld, ldd);
        else
            hipLaunchKernelGGL(ce_bwd_wave_kernel<h16_t>, wgrid, block, 0, as_stream(stream), logits, labels, row_lse, gscale, coef, (h16_t*)dlogits, R, V, ld, ldd);
        return omlm_post_launch("omlm_cross_entropy_bwd");
    }
    if (out_dtype == 0)
        hipLaunchKernelGGL(ce_bwd_kernel<float>, grid, block, 0, as_stream(stream), logits, labels, row_lse, gscale, coef, (float*)dlogits, R, V, ld, ldd);
    else if (out_dtype == OMLM_DT_F16)
        hipLaunchKernelGGL(ce_bwd_kernel<f16_t>, grid, block, 0, as_stream(stream), logits, labels, row_lse, gscale, coef, (f16_t*)dlogits, R, V, ld, ldd);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<h16_t>, grid, block, 0, as_stream(stream), logits, labels, row_lse, gscale, coef, (h16_t*)dlogits, R, V, ld, ldd);
    return omlm_post_launch("omlm_cross_entropy_bwd");
}

// ---------------------------------------------------------------------------------------------------------
// Training-batch preparation of TokenConditionedTransformerWrapper.forward (open_musiclm.py:340-376) + the id flattening of
// TokenConditionedTransformer.forward (:116-130) + generate_mask_with_prob (utils.py:49-56) as ONE launch: the reference's (and, until
// round 4, this repo's) ~45 small torch kernels per step -- eos append, label clones, conditioning-pad / eos masking, per-quantizer
// offsets, start markers, concatenations, the top-k of a randn row and its scatter -- were 0.3-0.6 ms of launch latency inside the
// captured step.  One WAVE per sample:
//   ids32 [B, N]   -2 start marker | conditioning sequences: (raw id, eos appended) with pad (-1) / eos positions -> 0, + codebook * (p mod Q)
//                  | last sequence: raw ids (+ offsets), its appended eos dropped (:356)
//   labels_s [B, L_s + 1] int32  raw ids with eos appended (:347,:355), per sequence (null pointer: not wanted)
//   keymask [B, N] uint8  1 start tokens, live conditioning ids, every position of the last sequence; AND the forgetful mask: the n_drop
//                  largest of scores[b, 1:] are dropped (position 0 never is; ties by lowest index, like the sampler)
#define PREP_NV_MAX 64                               /* register slots per lane: N <= 4096 */
struct PrepSeq { const long long* ids; int* labels; int len; int eos; int Q; int codebook; int start; };
struct PrepArgs { PrepSeq s[MAX_SEQ]; int nseq; int B; int N; int pad_id; const float* scores; int n_drop; int* ids32; unsigned char* keymask; };

__device__ __forceinline__ unsigned prep_ord(float v) { const unsigned u = f2u(v); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

// PREP_NV: slots per lane as a compile-time constant (a lone wave is a serial instruction stream: with 64 slots behind `j < nv` tests the
// radix descent alone was ~70 us at N = 1116); the descent stops at the first threshold that cuts exactly n_drop keys.
template <int PREP_NV>
__global__ __launch_bounds__(64) void prepare_train_batch_kernel(PrepArgs a) {
    const int b = blockIdx.x, lane = threadIdx.x, N = a.N;
    int* idrow = a.ids32 + (size_t)b * N;
    unsigned char* mrow = a.keymask + (size_t)b * N;
    // ---- forgetful mask: threshold of the n_drop largest scores (radix descent on ballots, all in registers) ----
    unsigned keys[PREP_NV];
    constexpr int nv = PREP_NV;
    unsigned thr = 0xFFFFFFFFu;
    int n_eq_keep = 0;
    const bool forget = a.scores != nullptr && a.n_drop > 0;
    if (forget) {
        const float* sr = a.scores + (size_t)b * N;
        float sv[PREP_NV];
#pragma unroll
        for (int j = 0; j < PREP_NV; ++j) { const int c = lane + 64 * j; sv[j] = sr[c < N ? c : N - 1]; }
#pragma unroll
        for (int j = 0; j < PREP_NV; ++j) {
            const int c = lane + 64 * j;
            keys[j] = (c < N && c > 0) ? prep_ord(sv[j]) : 0u;          // position 0 (and the tail) can never be among the top scores
        }
        unsigned t = 0;
        bool exact = false;
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = t | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < PREP_NV; ++j) cnt += __popcll(__ballot(keys[j] >= cand));
            if (cnt >= a.n_drop) t = cand;
            if (cnt == a.n_drop) { exact = true; break; }
        }
        int ng = 0;
        if (!exact) {
#pragma unroll
            for (int j = 0; j < PREP_NV; ++j) ng += __popcll(__ballot(keys[j] > t));
        }
        thr = t; n_eq_keep = exact ? 0x7fffffff : a.n_drop - ng;
    }
    // ---- ids / labels / mask ----
    int seen_eq = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < PREP_NV; ++j) {
        const int n = lane + 64 * j;
        bool drop = false;
        if (forget) {
            const bool eq = keys[j] == thr && n < N && n > 0;
            const unsigned long long eqmask = __ballot(eq);
            const int rank = seen_eq + __popcll(eqmask & below);
            drop = n < N && n > 0 && (keys[j] > thr || (eq && rank < n_eq_keep));
            seen_eq += __popcll(eqmask);
        }
        if (n >= N) continue;
        int si = 0;
#pragma unroll
        for (int q = 1; q < MAX_SEQ; ++q) if (q < a.nseq && n >= a.s[q].start) si = q;
        const PrepSeq& sq = a.s[si];
        const int p = n - sq.start - 1;                                  // index inside the sequence (-1: its start token)
        const bool last = si == a.nseq - 1;
        int id = -2;
        unsigned char live = 1;
        if (p >= 0) {
            long long raw = p < sq.len ? sq.ids[(size_t)b * sq.len + p] : (long long)sq.eos;      // p == len: the appended eos (conditioning sequences only)
            if (!last) {
                live = (raw != a.pad_id && raw != sq.eos) ? 1 : 0;
                if (!live) raw = 0;
            }
            id = (int)raw + (sq.Q > 1 ? sq.codebook * (p % sq.Q) : 0);
        }
        idrow[n] = id;
        mrow[n] = (live && !drop) ? 1 : 0;
    }
    // labels: every sequence's ids with the eos appended
    for (int q = 0; q < a.nseq; ++q) {
        const PrepSeq& sq = a.s[q];
        if (!sq.labels) continue;
        int* lr = sq.labels + (size_t)b * (sq.len + 1);
        for (int p = lane; p <= sq.len; p += 64) lr[p] = p < sq.len ? (int)sq.ids[(size_t)b * sq.len + p] : sq.eos;
    }
}

// ids[s]: int64 [B, len[s]] (flattened 'b ... -> b (...)'); labels[s]: int32 [B, len[s] + 1] or null; ids32 / keymask: [B, N] with
// N = sum_s (len[s] + 1) + (nseq - 1) ... i.e. one start token per sequence, an appended eos for every sequence but the last.
extern "C" int omlm_prepare_train_batch(const long long* const* ids, int* const* labels, const int* len, const int* eos, const int* Q,
                                        const int* codebook, int nseq, int B, int pad_id, const float* scores, int n_drop,
                                        int* ids32, unsigned char* keymask, int N, void* stream) {
    if (B <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(ids && len && eos && Q && codebook && ids32 && keymask && nseq >= 1 && nseq <= MAX_SEQ, "prepare_train_batch arguments");
    OMLM_CHECK_ARG(N >= 1 && N <= 64 * PREP_NV_MAX, "prepare_train_batch: N must be 1..4096");
    PrepArgs a;
    memset(&a, 0, sizeof(a));
    int pos = 0;
    for (int s = 0; s < nseq; ++s) {
        OMLM_CHECK_ARG(ids[s] && len[s] >= 0 && Q[s] >= 1, "prepare_train_batch: bad sequence");
        a.s[s].ids = ids[s]; a.s[s].labels = labels ? labels[s] : nullptr; a.s[s].len = len[s]; a.s[s].eos = eos[s]; a.s[s].Q = Q[s];
        a.s[s].codebook = codebook[s]; a.s[s].start = pos;
        pos += 1 + len[s] + (s < nseq - 1 ? 1 : 0);
    }
    OMLM_CHECK_ARG(pos == N, "prepare_train_batch: N does not match the sequence lengths");
    OMLM_CHECK_ARG(!scores || (n_drop >= 0 && n_drop < N), "prepare_train_batch: n_drop");
    a.nseq = nseq; a.B = B; a.N = N; a.pad_id = pad_id; a.scores = scores; a.n_drop = scores ? n_drop : 0; a.ids32 = ids32; a.keymask = keymask;
    const int nv = (N + 63) / 64;
    if (nv <= 9)       hipLaunchKernelGGL(prepare_train_batch_kernel<9>, dim3(B), dim3(64), 0, as_stream(stream), a);
    else if (nv <= 18) hipLaunchKernelGGL(prepare_train_batch_kernel<18>, dim3(B), dim3(64), 0, as_stream(stream), a);
    else if (nv <= 32) hipLaunchKernelGGL(prepare_train_batch_kernel<32>, dim3(B), dim3(64), 0, as_stream(stream), a);
    else               hipLaunchKernelGGL(prepare_train_batch_kernel<64>, dim3(B), dim3(64), 0, as_stream(stream), a);
    return omlm_post_launch("omlm_prepare_train_batch");
}
