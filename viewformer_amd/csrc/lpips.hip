// Perceptual (LPIPS-VGG) loss pieces of the codebook training step (vqgan_th.py:337,402-404: lpips.LPIPS(net='vgg') on (inputs,
// reconstructions)), gfx950.  The VGG-16 convolutions run on the convolution kernels of this library; what is here is the HBM-bound
// rest: ScalingLayer, ReLU, 2x2 max-pool (forward/backward) and the per-layer LPIPS head
//     d(pixel) = sum_c w_c (f0_c / (|f0| + eps) - f1_c / (|f1| + eps))^2 ,  eps = 1e-10      (lpips normalize_tensor + lin layer)
// with its backward w.r.t. f1 (the reconstruction's features; the VGG and lin weights are frozen, vqgan_th.py:338-339).
// One wavefront per pixel in the head kernels: the C <= 512 channels of a pixel are contiguous in NHWC, each lane owns C/64 of them.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

inline unsigned lp_grid(long long total, int per_block, unsigned cap = 16384) {
    long long b = (total + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

__global__ void scaling_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float s0, float s1, float s2,
                               float c0, float c1, float c2, int bwd) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const float sh = c == 0 ? s0 : c == 1 ? s1 : s2, sc = c == 0 ? c0 : c == 1 ? c1 : c2;
        y[i] = bwd ? x[i] / sc : (x[i] - sh) / sc;
    }
}

__global__ void relu_kernel(float* __restrict__ x, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<f32x4*>(x)[i];
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        reinterpret_cast<f32x4*>(x)[i] = v;
    }
}

__global__ void relu_bwd_kernel(float* __restrict__ dy, const float* __restrict__ y, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 d = reinterpret_cast<f32x4*>(dy)[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        d.x = v.x > 0.f ? d.x : 0.f; d.y = v.y > 0.f ? d.y : 0.f; d.z = v.z > 0.f ? d.z : 0.f; d.w = v.w > 0.f ? d.w : 0.f;
        reinterpret_cast<f32x4*>(dy)[i] = d;
    }
}

// NHWC 2x2/2 max-pool; H, W are the OUTPUT sizes
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, long long total4, int H, int W, int C) {
    const int cq = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cq);
        long long t = i / cq;
        const int xo = (int)(t % W); t /= W;
        const int yo = (int)(t % H);
        const long long img = t / H;
        const float* b = x + (((img * 2 * H + 2 * yo) * 2 * W) + 2 * xo) * C + c4 * 4;
        const long long row = (long long)2 * W * C;
        const f32x4 a = *reinterpret_cast<const f32x4*>(b), bb = *reinterpret_cast<const f32x4*>(b + C);
        const f32x4 c = *reinterpret_cast<const f32x4*>(b + row), d = *reinterpret_cast<const f32x4*>(b + row + C);
        f32x4 m;
        m.x = fmaxf(fmaxf(a.x, bb.x), fmaxf(c.x, d.x)); m.y = fmaxf(fmaxf(a.y, bb.y), fmaxf(c.y, d.y));
        m.z = fmaxf(fmaxf(a.z, bb.z), fmaxf(c.z, d.z)); m.w = fmaxf(fmaxf(a.w, bb.w), fmaxf(c.w, d.w));
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
    }
}

// backward: the gradient of an output goes to the FIRST maximum of its window in row-major order (torch max_pool2d); every input
// element belongs to exactly one window, so dx is written once, no atomics.  H, W = output sizes.
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long long total,
                                   int H, int W, int C) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int xo = (int)(t % W); t /= W;
        const int yo = (int)(t % H);
        const long long img = t / H;
        const long long b = (((img * 2 * H + 2 * yo) * 2 * W) + 2 * xo) * C + c;
        const long long row = (long long)2 * W * C;
        const float v0 = x[b], v1 = x[b + C], v2 = x[b + row], v3 = x[b + row + C];
        int k = 0;
        float m = v0;
        if (v1 > m) { m = v1; k = 1; }
        if (v2 > m) { m = v2; k = 2; }
        if (v3 > m) { m = v3; k = 3; }
        const float g = dy[i];
        dx[b] = k == 0 ? g : 0.f;
        dx[b + C] = k == 1 ? g : 0.f;
        dx[b + row] = k == 2 ? g : 0.f;
        dx[b + row + C] = k == 3 ? g : 0.f;
    }
}

constexpr int LP_PIX_PER_BLOCK = 64;      // 4 waves x 16 pixels each

// per-(image, block) partial sums of d(pixel): part[blk * n_img + img] (block-major so a column sum gives per-image totals)
__global__ __launch_bounds__(256) void lpips_head_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                         const float* __restrict__ w, float* __restrict__ part, int HW, int C,
                                                         int n_img) {
    __shared__ float red[4];
    const int img = blockIdx.y, blk = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int i = 0; i < LP_PIX_PER_BLOCK / 4; ++i) {
        const int px = blk * LP_PIX_PER_BLOCK + wave * (LP_PIX_PER_BLOCK / 4) + i;
        if (px >= HW) break;
        const float* a = f0 + ((long long)img * HW + px) * C;
        const float* b = f1 + ((long long)img * HW + px) * C;
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < C; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
        sa = vf_wave_sum(sa); sb = vf_wave_sum(sb);
        const float na = sqrtf(sa) + 1e-10f, nb = sqrtf(sb) + 1e-10f;
        float d = 0.f;
        for (int c = lane; c < C; c += 64) { const float t = a[c] / na - b[c] / nb; d += w[c] * t * t; }
        acc += vf_wave_sum(d);
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[(long long)blk * n_img + img] = (red[0] + red[1]) + (red[2] + red[3]);
}

// df1 (+)= gscale * d d(pixel) / d f1
__global__ __launch_bounds__(256) void lpips_head_bwd_kernel(const float* __restrict__ f0, const float* __restrict__ f1,
                                                             const float* __restrict__ w, float* __restrict__ df1, long long npix,
                                                             int C, float gscale, int accumulate) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (long long px = (long long)blockIdx.x * 4 + wave; px < npix; px += (long long)gridDim.x * 4) {
        const float* a = f0 + px * C;
        const float* b = f1 + px * C;
        float sa = 0.f, sb = 0.f;
        for (int c = lane; c < C; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
        sa = vf_wave_sum(sa); sb = vf_wave_sum(sb);
        const float s = sqrtf(sb);
        const float na = sqrtf(sa) + 1e-10f, nb = s + 1e-10f;
        // g_c = d d / d bhat_c = -2 w_c (ahat_c - bhat_c);  bhat = b / nb, nb = |b| + eps:
        // d d / d b_j = g_j / nb - (sum_c g_c b_c) b_j / (nb^2 |b|)
        float dot = 0.f;
        for (int c = lane; c < C; c += 64) dot += -2.f * w[c] * (a[c] / na - b[c] / nb) * b[c];
        dot = vf_wave_sum(dot);
        const float k = s > 0.f ? dot / (nb * nb * s) : 0.f;
        float* o = df1 + px * C;
        for (int c = lane; c < C; c += 64) {
            const float g = -2.f * w[c] * (a[c] / na - b[c] / nb);
            const float v = gscale * (g / nb - k * b[c]);
            o[c] = accumulate ? o[c] + v : v;
        }
    }
}

}  // namespace

extern "C" {

int vf_lpips_scaling_f32(const float* x, float* y, int64_t npix, const float* shift3, const float* scale3, int backward, void* stream) {
    if (!x || !y || npix <= 0 || !shift3 || !scale3) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(scaling_kernel, dim3(lp_grid(npix * 3, 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)npix * 3,
                       shift3[0], shift3[1], shift3[2], scale3[0], scale3[1], scale3[2], backward);
    return vf_last_status();
}

int vf_relu_f32(float* x, int64_t n, void* stream) {
    if (!x || n <= 0 || (n & 3)) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(relu_kernel, dim3(lp_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, (long long)(n / 4));
    return vf_last_status();
}

int vf_relu_bwd_f32(float* dy, const float* y, int64_t n, void* stream) {
    if (!dy || !y || n <= 0 || (n & 3)) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(lp_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, dy, y, (long long)(n / 4));
    return vf_last_status();
}

int vf_maxpool2_f32(const float* x, float* y, int n_img, int Hout, int Wout, int C, void* stream) {
    if (!x || !y || n_img <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || (C & 3)) return VF_ERR_BAD_ARG;
    const long long total4 = (long long)n_img * Hout * Wout * (C / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(lp_grid(total4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, total4, Hout, Wout, C);
    return vf_last_status();
}

int vf_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int n_img, int Hout, int Wout, int C, void* stream) {
    if (!x || !dy || !dx || n_img <= 0 || Hout <= 0 || Wout <= 0 || C <= 0) return VF_ERR_BAD_ARG;
    const long long total = (long long)n_img * Hout * Wout * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(lp_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, total, Hout, Wout, C);
    return vf_last_status();
}

int vf_lpips_head_blocks(int HW) { return HW <= 0 ? 0 : (HW + LP_PIX_PER_BLOCK - 1) / LP_PIX_PER_BLOCK; }

int vf_lpips_head_f32(const float* f0, const float* f1, const float* w, float* part, int n_img, int HW, int C, void* stream) {
    if (!f0 || !f1 || !w || !part || n_img <= 0 || HW <= 0 || C <= 0) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(lpips_head_kernel, dim3(vf_lpips_head_blocks(HW), n_img), dim3(256), 0, (hipStream_t)stream, f0, f1, w, part, HW,
                       C, n_img);
    return vf_last_status();
}

int vf_lpips_head_bwd_f32(const float* f0, const float* f1, const float* w, float* df1, int64_t npix, int C, float gscale,
                          int accumulate, void* stream) {
    if (!f0 || !f1 || !w || !df1 || npix <= 0 || C <= 0) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(lpips_head_bwd_kernel, dim3(lp_grid(npix, 4, 65536)), dim3(256), 0, (hipStream_t)stream, f0, f1, w, df1,
                       (long long)npix, C, gscale, accumulate);
    return vf_last_status();
}

}  // extern "C"
