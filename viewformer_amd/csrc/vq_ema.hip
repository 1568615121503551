// Training branch of the codebook quantizer: the EMA codebook update of QuantizeEMA.forward
// (viewformer/models/utils_th.py:46-64), gfx950.  HBM-bound integer / segment work, no matrix pipe:
//
//   vf_vq_ema_accumulate_f32: counts[k] = #rows with idx == k (:47), embed_sum[d][k] = sum of those rows of z (:48).
//     One workgroup per code scans the index vector in order, compacts the matching row ids (ballot + prefix) and adds the
//     rows in THAT order: no atomics, bit-reproducible, independent of scheduling.  z is read once in total.
//   (the two all-reduces of :50-52 are the caller's, on counts / embed_sum: RCCL through torch.distributed)
//   vf_vq_ema_update_f32: ema buffers += (new - ema)(1 - decay) (:55-56), bias correction 1 - decay^counter (:24-30, passed in as
//     `corr`), n = sum of corrected cluster sizes (:59), Laplace smoothing (:60-62), embeddings = ema_dw / cluster_size (:63-64).
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

__global__ __launch_bounds__(256) void vq_ema_accumulate_kernel(const float* __restrict__ z, const long long* __restrict__ idx,
                                                                long long M, int D, int Kc, float* __restrict__ counts,
                                                                float* __restrict__ embed_sum) {
    __shared__ int rows[256];
    __shared__ int wave_cnt[4];
    const int k = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                 // features tid, tid + 256, ... (D <= 1024)
    long long total = 0;
    for (long long base = 0; base < M; base += 256) {
        const long long r = base + tid;
        const bool hit = r < M && idx[r] == k;
        const unsigned long long bal = __ballot(hit);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) off += wave_cnt[w]; n += wave_cnt[w]; }
        if (hit) rows[off + before] = (int)(r - base);
        __syncthreads();
        for (int i = 0; i < n; ++i) {                     // rows in index order
            const float* zr = z + (base + rows[i]) * (long long)D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = tid + 256 * q;
                if (d < D) acc[q] += zr[d];
            }
        }
        total += n;
        __syncthreads();
    }
    if (tid == 0) counts[k] = (float)total;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int d = tid + 256 * q;
        if (d < D) embed_sum[(long long)d * Kc + k] = acc[q];
    }
}

// one workgroup (1024 threads): thread -> codes k, k + 1024, ...
__global__ __launch_bounds__(1024) void vq_ema_update_kernel(const float* __restrict__ counts, const float* __restrict__ embed_sum,
                                                             float* __restrict__ cs_hidden, float* __restrict__ dw_hidden,
                                                             float* __restrict__ emb, int D, int Kc, float decay, float eps,
                                                             float corr) {
    __shared__ float red[1024];
    const int tid = threadIdx.x;
    const float a = 1.0f - decay;
    float part = 0.f;
    for (int k = tid; k < Kc; k += 1024) {
        const float c = cs_hidden[k] + (counts[k] - cs_hidden[k]) * a;      // Tensor.add_(other - self, alpha = 1 - decay)
        cs_hidden[k] = c;
        part += c / corr;
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {                   // fixed tree: deterministic
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float n = red[0];
    for (int k = tid; k < Kc; k += 1024) {
        const float cluster = (cs_hidden[k] / corr + eps) / (n + (float)Kc * eps) * n;
        for (int d = 0; d < D; ++d) {
            const long long i = (long long)d * Kc + k;
            const float w = dw_hidden[i] + (embed_sum[i] - dw_hidden[i]) * a;
            dw_hidden[i] = w;
            emb[i] = w / corr / cluster;
        }
    }
}

}  // namespace

extern "C" {

int vf_vq_ema_accumulate_f32(const float* z, const int64_t* idx, int64_t M, int D, int Kc, float* counts, float* embed_sum,
                             void* stream) {
    if (!z || !idx || !counts || !embed_sum || M <= 0 || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (D > 1024) return VF_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(vq_ema_accumulate_kernel, dim3((unsigned)Kc), dim3(256), 0, (hipStream_t)stream, z,
                       reinterpret_cast<const long long*>(idx), (long long)M, D, Kc, counts, embed_sum);
    return vf_last_status();
}

int vf_vq_ema_update_f32(const float* counts, const float* embed_sum, float* cluster_size_hidden, float* dw_hidden,
                         float* embeddings, int D, int Kc, float decay, float eps, float corr, void* stream) {
    if (!counts || !embed_sum || !cluster_size_hidden || !dw_hidden || !embeddings || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (!(decay > 0.f && decay < 1.f) || !(corr > 0.f)) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(vq_ema_update_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, counts, embed_sum, cluster_size_hidden,
                       dw_hidden, embeddings, D, Kc, decay, eps, corr);
    return vf_last_status();
}

}  // extern "C"
