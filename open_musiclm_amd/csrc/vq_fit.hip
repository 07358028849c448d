// Fitting side of the CLAP residual vector quantizer (reference: ClapRVQTrainer.train_step, trainer.py:689-736, calls
// ClapQuantized.quantize(embeds, return_rvq_loss=True) with rq.train(True), clap_quantized.py:75-82; the arithmetic is the
// third-party vector-quantize-pytorch EuclideanCodebook, constructed at clap_quantized.py:38-46 with kmeans_init=True,
// decay=0.95, commitment_weight=0, threshold_ema_dead_code -- un-vendored and un-pinned: PARITY UNPINNED, the published
// algorithm is restated in oracle/musiclm_oracle.py::rvq_fit_step and these kernels are checked against that).
//
// Per quantizer layer, on the running residual r [n, D]:
//   assign      idx_i = argmin_k |r_i - e_k|^2                       (omlm_rvq_encode, one stage, also yields r - e_idx)
//   accumulate  count_k = #{i: idx_i = k},  sum_k = sum_{idx_i = k} r_i                                   (this file)
//   k-means     e_k <- sum_k / count_k where count_k > 0             (the 10 Lloyd iterations of the first batch)
//   EMA         cs_k <- d cs_k + (1-d) count_k;  avg_k <- d avg_k + (1-d) sum_k;
//               e_k <- avg_k / ((cs_k + eps) / (sum cs + K eps) * sum cs)                      (Laplace-smoothed cluster sizes)
// All HBM-trivial (n x D and K x D floats); the point is that the whole fit step stays on the device.
#include "common.h"

// counts / sums must be zeroed by the caller.  One workgroup per row; fp32 atomics (order-dependent rounding only).
__global__ __launch_bounds__(256) void vq_accumulate_kernel(const float* __restrict__ x, const int* __restrict__ idx, int idx_stride,
                                                            float* __restrict__ counts, float* __restrict__ sums, int n, int D, int K) {
    const int row = blockIdx.x;
    const int k = idx[(size_t)row * idx_stride];
    if (k < 0 || k >= K) return;
    for (int d = threadIdx.x; d < D; d += 256) unsafeAtomicAdd(sums + (size_t)k * D + d, x[(size_t)row * D + d]);
    if (threadIdx.x == 0) unsafeAtomicAdd(counts + k, 1.0f);
}

// Lloyd update: codes that received points move to their mean, the others stay.  Also refreshes the transposed copy [D, K].
__global__ __launch_bounds__(256) void vq_kmeans_update_kernel(float* __restrict__ means, float* __restrict__ means_T,
                                                               const float* __restrict__ counts, const float* __restrict__ sums, int K, int D) {
    const int k = blockIdx.x;
    const float c = counts[k];
    for (int d = threadIdx.x; d < D; d += 256) {
        float m = means[(size_t)k * D + d];
        if (c > 0.f) m = sums[(size_t)k * D + d] / c;
        means[(size_t)k * D + d] = m;
        means_T[(size_t)d * K + k] = m;
    }
}

// cluster sizes first (one workgroup: the total over K is needed by every code), then the codes.
__global__ __launch_bounds__(1024) void vq_cluster_size_kernel(float* __restrict__ cluster_size, const float* __restrict__ counts,
                                                               float* __restrict__ total_out, int K, float decay) {
    __shared__ float red[16];
    float part = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float v = cluster_size[k] * decay + counts[k] * (1.0f - decay);
        cluster_size[k] = v;
        part += v;
    }
    const float tot = block_sum<1024>(part, red);
    if (threadIdx.x == 0) total_out[0] = tot;
}
__global__ __launch_bounds__(256) void vq_embed_kernel(const float* __restrict__ cluster_size, float* __restrict__ embed_avg,
                                                       float* __restrict__ embed, float* __restrict__ embed_T,
                                                       const float* __restrict__ sums, const float* __restrict__ total, int K, int D,
                                                       float decay, float eps) {
    const int k = blockIdx.x;
    const float tot = total[0];
    const float smoothed = (cluster_size[k] + eps) / (tot + (float)K * eps) * tot;
    for (int d = threadIdx.x; d < D; d += 256) {
        const float a = embed_avg[(size_t)k * D + d] * decay + sums[(size_t)k * D + d] * (1.0f - decay);
        embed_avg[(size_t)k * D + d] = a;
        const float e = a / smoothed;
        embed[(size_t)k * D + d] = e;
        embed_T[(size_t)d * K + k] = e;
    }
}

extern "C" int omlm_vq_accumulate(const float* x, const int* indices, int idx_stride, float* counts, float* sums,
                                  int n, int D, int K, void* stream) {
    if (n <= 0) return OMLM_OK;
    OMLM_CHECK_ARG(x && indices && counts && sums && D > 0 && K > 0 && idx_stride >= 1, "vq_accumulate arguments");
    hipLaunchKernelGGL(vq_accumulate_kernel, dim3(n), dim3(256), 0, as_stream(stream), x, indices, idx_stride, counts, sums, n, D, K);
    return omlm_post_launch("omlm_vq_accumulate");
}

extern "C" int omlm_vq_kmeans_update(float* means, float* means_T, const float* counts, const float* sums, int K, int D, void* stream) {
    OMLM_CHECK_ARG(means && means_T && counts && sums && K > 0 && D > 0, "vq_kmeans_update arguments");
    hipLaunchKernelGGL(vq_kmeans_update_kernel, dim3(K), dim3(256), 0, as_stream(stream), means, means_T, counts, sums, K, D);
    return omlm_post_launch("omlm_vq_kmeans_update");
}

// total_scratch: one float of device scratch
extern "C" int omlm_vq_ema_update(float* cluster_size, float* embed_avg, float* embed, float* embed_T, const float* counts,
                                  const float* sums, float* total_scratch, int K, int D, float decay, float eps, void* stream) {
    OMLM_CHECK_ARG(cluster_size && embed_avg && embed && embed_T && counts && sums && total_scratch && K > 0 && D > 0,
                   "vq_ema_update arguments");
    OMLM_CHECK_ARG(decay >= 0.f && decay <= 1.f, "decay");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(vq_cluster_size_kernel, dim3(1), dim3(1024), 0, st, cluster_size, counts, total_scratch, K, decay);
    hipLaunchKernelGGL(vq_embed_kernel, dim3(K), dim3(256), 0, st, (const float*)cluster_size, embed_avg, embed, embed_T, sums,
                       (const float*)total_scratch, K, D, decay, eps);
    return omlm_post_launch("omlm_vq_ema_update");
}
