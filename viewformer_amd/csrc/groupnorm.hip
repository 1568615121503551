// GroupNorm(32 groups, eps) statistics + standalone apply, NHWC fp32, gfx950.
//
// Replaces torch.nn.GroupNorm(32, C, eps=1e-6, affine=True) (viewformer/models/vqgan_th.py:16-17).
// HBM-bound: the statistics pass reads the activation once (coalesced float4 rows); the apply is
// normally folded into the consuming conv's A-operand staging (igemm prologue), so the normalised
// tensor is never written.  Deterministic: fixed-order fp32 partial sums per (image, split, group),
// combined in fp64 by the finalize kernel (no atomics).
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int MAX_SPLIT = 64;

__host__ inline int gn_nsplit(int HW) {
    int s = HW / 256;
    if (s < 1) s = 1;
    if (s > MAX_SPLIT) s = MAX_SPLIT;
    return s;
}

// grid (nsplit, n_img), 256 threads.  Thread t owns channel quad cq = t % (C/4) and pixel lane
// pl = t / (C/4); it strides over the split's pixels with step ppp = 256 / (C/4).
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                         int HW, int C, int groups, int nsplit) {
    __shared__ float s_sum[1024];
    __shared__ float s_sq[1024];
    const int tid = threadIdx.x;
    const int cquads = C >> 2;
    const int ppp = 256 / cquads;
    const int cq = tid % cquads;
    const int pl = tid / cquads;
    const int img = blockIdx.y;
    const int split = blockIdx.x;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per;
    const int p1 = min(HW, p0 + per);

    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    const float* base = x + ((size_t)img * HW) * C + cq * 4;
    for (int p = p0 + pl; p < p1; p += ppp) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * C);
        s += v;
        q += v * v;
    }
    // [pl][C] in LDS (ppp * C == 1024 floats)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s_sum[pl * C + cq * 4 + e] = s[e];
        s_sq[pl * C + cq * 4 + e] = q[e];
    }
    __syncthreads();
    // per-channel sums over the pixel lanes (fixed order), C <= 1024 channels over 256 threads
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < ppp; ++k) { a += s_sum[k * C + c]; b += s_sq[k * C + c]; }
        s_sum[c] = a;   // row 0 is overwritten only by the thread that just consumed column c
        s_sq[c] = b;
    }
    __syncthreads();
    if (tid < groups) {
        const int cg = C / groups;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < cg; ++k) { a += s_sum[tid * cg + k]; b += s_sq[tid * cg + k]; }
        float* dst = part + (((size_t)img * nsplit + split) * groups + tid) * 2;
        dst[0] = a;
        dst[1] = b;
    }
}

// one wave per (image, group): lane l adds slots l, l+64, ... in fp64, then a fixed xor-shuffle tree combines the 64 lanes
// (deterministic; the fused conv epilogues hand over up to 256 slots per image and a serial loop took 23 us per call)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                          float* __restrict__ mean_c, float* __restrict__ scale_c, int n_img,
                                                          int HW, int C, int groups, int nsplit, float eps) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_img * groups) return;                     // wave-uniform
    const int img = i / groups, g = i - img * groups;
    double a = 0.0, b = 0.0;
    for (int s = lane; s < nsplit; s += 64) {
        const float* src = part + (((size_t)img * nsplit + s) * groups + g) * 2;
        a += (double)src[0];
        b += (double)src[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
    }
    const int cg = C / groups;
    const double cnt = (double)HW * cg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;   // biased variance, as torch
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int k = lane; k < cg; k += 64) {
        const int c = g * cg + k;
        mean_c[(size_t)img * C + c] = fmean;
        scale_c[(size_t)img * C + c] = rstd * gamma[c];
    }
}

// The same reduction for nsplit <= 512 on EIGHT lanes per (image, group) instead of a wave (896 images x 32 groups = 28 672 waves per launch, 67 launches per
// inference step at 11 us each, most of it waves whose lanes hold one value or none): lane j of an item holds, for k < 8, the wave kernel's lane j + 8 k (its slots j + 8 k + 64 r added in ascending order) and adds the eight in the wave kernel's
// butterfly order — (k, k + 4) is its xor-32 level, (k, k + 2) xor 16, (k, k + 1) xor 8 — then three shuffles finish levels xor 4, 2, 1: the same additions of
// the same doubles (addition commutes), so mean and scale are bit-identical.
__global__ __launch_bounds__(256) void gn_finalize8_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                           float* __restrict__ mean_c, float* __restrict__ scale_c, int n_img,
                                                           int HW, int C, int groups, int nsplit, float eps) {
    const int j = threadIdx.x & 7;
    const int i = blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = i < n_img * groups;                  // (dead items keep their lanes in the shuffles)
    const int ii = live ? i : 0;
    const int img = ii / groups, g = ii - img * groups;
    double a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = 0.0; b[k] = 0.0; }
    for (int s0 = j; s0 < nsplit; s0 += 64) {              // (a round = the wave kernel's lanes j + 8 k, k < 8, at their next slot: eight loads in flight)
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = s0 + 8 * k;
            v[k] = make_float2(0.f, 0.f);
            if (s < nsplit) v[k] = *reinterpret_cast<const float2*>(part + (((size_t)img * nsplit + s) * groups + g) * 2);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (s0 + 8 * k < nsplit) { a[k] += (double)v[k].x; b[k] += (double)v[k].y; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] += a[k + 4]; b[k] += b[k + 4]; }
#pragma unroll
    for (int k = 0; k < 2; ++k) { a[k] += a[k + 2]; b[k] += b[k + 2]; }
    double sa = a[0] + a[1], sb = b[0] + b[1];
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
        sa += __shfl_xor(sa, o, 64);
        sb += __shfl_xor(sb, o, 64);
    }
    if (!live) return;
    const int cg = C / groups;
    const double cnt = (double)HW * cg;
    const double mean = sa / cnt;
    double var = sb / cnt - mean * mean;   // biased variance, as torch
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int k = j; k < cg; k += 8) {
        const int c = g * cg + k;
        mean_c[(size_t)img * C + c] = fmean;
        scale_c[(size_t)img * C + c] = rstd * gamma[c];
    }
}
static void gn_finalize_launch(hipStream_t s, const float* part, const float* gamma, float* mean_c, float* scale_c, int n_img, int HW, int C, int groups,
                               int nsplit, float eps) {
    const int n = n_img * groups;
    if (nsplit <= 512)
        hipLaunchKernelGGL(gn_finalize8_kernel, dim3((n + 31) / 32), dim3(256), 0, s, part, gamma, mean_c, scale_c, n_img, HW, C, groups, nsplit, eps);
    else
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((n + 3) / 4), dim3(256), 0, s, part, gamma, mean_c, scale_c, n_img, HW, C, groups, nsplit, eps);
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean_c,
                                                       const float* __restrict__ scale_c, const float* __restrict__ beta,
                                                       float* __restrict__ out, long long total4, int HW, int C,
                                                       int swish) {
    const int cquads = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % cquads);
        const long long pix = i / cquads;
        const int img = (int)(pix / HW);
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        const f32x4 m = *reinterpret_cast<const f32x4*>(mean_c + (size_t)img * C + cq * 4);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale_c + (size_t)img * C + cq * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(beta + cq * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = (v[e] - m[e]) * sc[e] + b[e];
            if (swish) t = vf_swish(t);
            o[e] = t;
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = o;
    }
}

bool gn_shape_ok(int C, int groups) {
    if (C <= 0 || groups <= 0 || C % groups != 0 || (C & 3)) return false;
    const int cquads = C >> 2;
    if (cquads > 256 || 256 % cquads != 0) return false;   // C in {4,8,...,1024} powers of two
    return groups <= 256;
}

}  // namespace

extern "C" {

size_t vf_groupnorm_workspace_bytes(int n_img, int HW, int C) {
    (void)C;
    if (n_img <= 0 || HW <= 0) return 0;
    return (size_t)n_img * gn_nsplit(HW) * 256 /* >= groups */ * 2 * sizeof(float);
}

int vf_groupnorm_stats_f32(const float* x, const float* gamma, int n_img, int HW, int C, int groups, float eps,
                           float* mean_c, float* scale_c, void* ws, void* stream) {
    if (!x || !gamma || !mean_c || !scale_c || !ws || n_img <= 0 || HW <= 0) return VF_ERR_BAD_ARG;
    if (!gn_shape_ok(C, groups)) return VF_ERR_UNSUPPORTED;
    const int nsplit = gn_nsplit(HW);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nsplit, n_img), dim3(256), 0, s, x, (float*)ws, HW, C, groups, nsplit);
    int st = vf_last_status();
    if (st) return st;
    gn_finalize_launch(s, (const float*)ws, gamma, mean_c, scale_c, n_img, HW, C, groups, nsplit, eps);
    return vf_last_status();
}

int vf_groupnorm_finalize_f32(const float* part, const float* gamma, int n_img, int HW, int C, int groups, int nslots,
                              float eps, float* mean_c, float* scale_c, void* stream) {
    if (!part || !gamma || !mean_c || !scale_c || n_img <= 0 || HW <= 0 || nslots <= 0) return VF_ERR_BAD_ARG;
    if (C <= 0 || groups <= 0 || C % groups != 0) return VF_ERR_UNSUPPORTED;
    gn_finalize_launch((hipStream_t)stream, part, gamma, mean_c, scale_c, n_img, HW, C, groups, nslots, eps);
    return vf_last_status();
}

int vf_groupnorm_apply_f32(const float* x, const float* mean_c, const float* scale_c, const float* beta, float* out,
                           int n_img, int HW, int C, int swish, void* stream) {
    if (!x || !mean_c || !scale_c || !beta || !out || n_img <= 0 || HW <= 0 || C <= 0 || (C & 3)) return VF_ERR_BAD_ARG;
    const long long total4 = (long long)n_img * HW * (C >> 2);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, mean_c, scale_c,
                       beta, out, total4, HW, C, swish);
    return vf_last_status();
}

}  // extern "C"
