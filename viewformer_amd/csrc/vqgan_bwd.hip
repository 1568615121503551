// Backward-pass helpers of the codebook (VQGAN) training step (vqgan_th.py:321-447: forward :349-352, loss :354-368, Adam :427-429),
// gfx950.  All contractions of the backward pass are launches of the forward GEMM/conv kernels on re-packed operands (dX = conv with
// the 180-degree-rotated, channel-transposed weight; dW = gathered-transposed activations x packed dY, split-K); what is left is
// HBM-bound elementwise / reduction work, collected here.  First version: correct, unfused.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

// dst[c][p] = src[img][y*stride + oy][x*stride + ox][c] (0 outside the image), p = (img, y, x) over the Hout x Wout output grid:
// the tap-shifted, channel-major view of an NHWC activation that turns conv dW into one GEMM per tap (dW_tap = dst . dY).
__global__ __launch_bounds__(256) void gather_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int n_img,
                                                               int Hin, int Win, int C, int Hout, int Wout, int stride, int oy,
                                                               int ox, long long ld_dst) {
    __shared__ float tile[32][33];
    const long long P = (long long)n_img * Hout * Wout;
    const long long p0 = (long long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long long p = p0 + ty + 8 * r;
        float v = 0.f;
        if (p < P && c0 + tx < C) {
            const int x = (int)(p % Wout);
            const int y = (int)((p / Wout) % Hout);
            const long long img = p / ((long long)Wout * Hout);
            const int sy = y * stride + oy, sx = x * stride + ox;
            if (sy >= 0 && sy < Hin && sx >= 0 && sx < Win) v = src[((img * Hin + sy) * Win + sx) * C + c0 + tx];
        }
        tile[ty + 8 * r][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + ty + 8 * r;
        const long long p = p0 + tx;
        if (c < C && p < P) dst[(long long)c * ld_dst + p] = tile[tx][ty + 8 * r];
    }
}

// nearest-x2 upsample backward: dx[img][y][x][c] = sum of the 2x2 block of du
__global__ void sum2x2_kernel(const float* __restrict__ du, float* __restrict__ dx, long long total4, int H, int W, int C) {
    const int cq = C >> 2;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % cq);
        long long t = i / cq;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const long long img = t / H;
        const float* b = du + (((img * 2 * H + 2 * y) * 2 * W) + 2 * x) * C + c4 * 4;
        const long long row = (long long)2 * W * C;
        const f32x4 s = *reinterpret_cast<const f32x4*>(b) + *reinterpret_cast<const f32x4*>(b + C) +
                        *reinterpret_cast<const f32x4*>(b + row) + *reinterpret_cast<const f32x4*>(b + row + C);
        *reinterpret_cast<f32x4*>(dx + i * 4) = s;
    }
}

// GroupNorm(+swish) backward.  Forward: xhat = (x - mean) * rstd, t = xhat * gamma + beta, a = swish ? t * sigmoid(t) : t.
// Given da: dt = da * da/dt;  dgamma[c] = sum dt * xhat, dbeta[c] = sum dt;  per (image, group) m1 = mean(gamma dt),
// m2 = mean(gamma dt xhat);  dx = rstd * (gamma dt - m1 - xhat m2).   mean_c / scale_c are the forward's [Nimg][C] (scale = rstd*gamma).
__device__ __forceinline__ float dswish(float t) {
    const float s = 1.0f / (1.0f + expf(-t));
    return s * (1.0f + t * (1.0f - s));
}

// stage 1: grid (nsplit, n_img), per-channel partial sums over the split's pixels: part[img][split][C][2] = {sum dt, sum dt*xhat}
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ da,
                                                             const float* __restrict__ mean_c, const float* __restrict__ scale_c,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ part, int HW, int C, int nsplit, int swish) {
    const int img = blockIdx.y, split = blockIdx.x;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mu = mean_c[(size_t)img * C + c], sg = scale_c[(size_t)img * C + c], g = gamma[c], b = beta[c];
        const float rstd = sg / g;
        float s1 = 0.f, s2 = 0.f;
        for (int p = p0; p < p1; ++p) {
            const size_t i = ((size_t)img * HW + p) * C + c;
            const float xh = (x[i] - mu) * rstd;
            float dt = da[i];
            if (swish) dt *= dswish(xh * g + b);
            s1 += dt;
            s2 += dt * xh;
        }
        float* d = part + ((((size_t)img * nsplit + split) * C) + c) * 2;
        d[0] = s1;
        d[1] = s2;
    }
}

// stage 2: one block per image: reduce the splits (fixed order), per-group means -> gm[img][groups][2]; per-image channel sums
// -> chan[img][C][2] (dgamma / dbeta contributions, summed over images by the caller's column sum)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ gamma,
                                                            float* __restrict__ chan, float* __restrict__ gm, int HW, int C,
                                                            int groups, int nsplit) {
    __shared__ float s1s[1024], s2s[1024];
    const int img = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int s = 0; s < nsplit; ++s) {
            const float* d = part + ((((size_t)img * nsplit + s) * C) + c) * 2;
            a += d[0];
            b += d[1];
        }
        s1s[c] = a;
        s2s[c] = b;
        chan[((size_t)img * C + c) * 2] = b;          // dgamma contribution
        chan[((size_t)img * C + c) * 2 + 1] = a;      // dbeta contribution
    }
    __syncthreads();
    const int cg = C / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        float m1 = 0.f, m2 = 0.f;
        for (int k = 0; k < cg; ++k) {
            const int c = g * cg + k;
            m1 += gamma[c] * s1s[c];
            m2 += gamma[c] * s2s[c];
        }
        const float inv = 1.0f / ((float)HW * cg);
        gm[((size_t)img * groups + g) * 2] = m1 * inv;
        gm[((size_t)img * groups + g) * 2 + 1] = m2 * inv;
    }
}

// stage 3: dx (+= when accumulate)
__global__ void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ da, const float* __restrict__ mean_c,
                                    const float* __restrict__ scale_c, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ gm, float* __restrict__ dx,
                                    long long total, int HW, int C, int groups, int swish, int accumulate) {
    const int cg = C / groups;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long img = i / ((long long)HW * C);
        const float mu = mean_c[img * C + c], sg = scale_c[img * C + c], g = gamma[c];
        const float rstd = sg / g;
        const float xh = (x[i] - mu) * rstd;
        float dt = da[i];
        if (swish) dt *= dswish(xh * g + beta[c]);
        const float* m = gm + (img * groups + c / cg) * 2;
        const float v = rstd * (g * dt - m[0] - xh * m[1]);
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// row softmax backward (VQGAN AttnBlock, vqgan_th.py:132-134): ds = scale * p * (dp - sum_j p_j dp_j), in place on dp
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long long rows,
                                                               int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + (size_t)row * n;
    float* dr = dp + (size_t)row * n;
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += pr[c] * dr[c];
    s = vf_wave_sum(s);
    for (int c = lane; c < n; c += 64) dr[c] = scale * pr[c] * (dr[c] - s);
}

// L1 reconstruction loss (vqgan_th.py:355,361): partial sums of |x - y| and d/dy = sign(y - x) * w
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ dy,
                                                 float* __restrict__ part, long long n, float w) {
    float s = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = y[i] - x[i];
        s += fabsf(d);
        dy[i] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
    }
    __shared__ float red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

inline unsigned grid1(long long n, int per, unsigned cap = 8192) {
    long long b = (n + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace

extern "C" {

int vf_gather_transpose_f32(const float* src, float* dst, int n_img, int Hin, int Win, int C, int Hout, int Wout, int stride, int oy,
                            int ox, int64_t ld_dst, void* stream) {
    if (!src || !dst || n_img <= 0 || Hin <= 0 || Win <= 0 || C <= 0 || Hout <= 0 || Wout <= 0 || stride <= 0) return VF_ERR_BAD_ARG;
    const long long P = (long long)n_img * Hout * Wout;
    if (ld_dst < P) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)((P + 31) / 32), (unsigned)((C + 31) / 32));
    hipLaunchKernelGGL(gather_transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, n_img, Hin, Win, C, Hout, Wout,
                       stride, oy, ox, (long long)ld_dst);
    return vf_last_status();
}

int vf_upsample2_bwd_f32(const float* du, float* dx, int n_img, int H, int W, int C, void* stream) {
    if (!du || !dx || n_img <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return VF_ERR_BAD_ARG;
    const long long total4 = (long long)n_img * H * W * (C >> 2);
    hipLaunchKernelGGL(sum2x2_kernel, dim3(grid1(total4, 256)), dim3(256), 0, (hipStream_t)stream, du, dx, total4, H, W, C);
    return vf_last_status();
}

size_t vf_groupnorm_bwd_workspace_bytes(int n_img, int HW, int C, int groups) {
    if (n_img <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
    int nsplit = HW / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
    return ((size_t)n_img * nsplit * C * 2 + (size_t)n_img * C * 2 + (size_t)n_img * groups * 2) * sizeof(float);
}

/* chan_sums out: [n_img][C][2] = per-image {dgamma, dbeta} contributions (the caller sums over images) */
int vf_groupnorm_bwd_f32(const float* x, const float* da, const float* mean_c, const float* scale_c, const float* gamma,
                         const float* beta, float* dx, float* chan_sums, int n_img, int HW, int C, int groups, int swish,
                         int accumulate, void* ws, void* stream) {
    if (!x || !da || !mean_c || !scale_c || !gamma || !beta || !dx || !chan_sums || !ws) return VF_ERR_BAD_ARG;
    if (n_img <= 0 || HW <= 0 || C <= 0 || groups <= 0 || C % groups != 0 || C > 1024) return VF_ERR_UNSUPPORTED;
    int nsplit = HW / 64;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 64) nsplit = 64;
    float* part = (float*)ws;
    float* gm = part + (size_t)n_img * nsplit * C * 2;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nsplit, n_img), dim3(256), 0, s, x, da, mean_c, scale_c, gamma, beta, part, HW, C,
                       nsplit, swish);
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(n_img), dim3(256), 0, s, (const float*)part, gamma, chan_sums, gm, HW, C, groups,
                       nsplit);
    const long long total = (long long)n_img * HW * C;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid1(total, 256)), dim3(256), 0, s, x, da, mean_c, scale_c, gamma, beta,
                       (const float*)gm, dx, total, HW, C, groups, swish, accumulate);
    return vf_last_status();
}

int vf_softmax_rows_bwd_f32(const float* p, float* dp, int64_t rows, int n, float scale, void* stream) {
    if (!p || !dp || rows < 0 || n <= 0) return VF_ERR_BAD_ARG;
    if (rows == 0) return VF_OK;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp,
                       (long long)rows, n, scale);
    return vf_last_status();
}

/* number of partial sums vf_l1_loss_f32 writes for n elements (<= 1024) */
int vf_l1_loss_partials(int64_t n) { return n > 0 ? (int)grid1(n, 256 * 8, 1024) : 0; }

/* part[0 .. vf_l1_loss_partials(n)) = partial sums of |y - x| (fixed order: deterministic); dy = sign(y - x) * grad_weight */
int vf_l1_loss_f32(const float* x, const float* y, float* dy, float* part, int64_t n, float grad_weight, void* stream) {
    if (!x || !y || !dy || !part || n <= 0) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(l1_kernel, dim3(grid1(n, 256 * 8, 1024)), dim3(256), 0, (hipStream_t)stream, x, y, dy, part, (long long)n,
                       grad_weight);
    return vf_last_status();
}

}  // extern "C"
