// HBM-bound glue kernels of the hot path (gfx950): LayerNorm, embedding sum, tiny dense, row
// argmax / softmax, uint8 pre/post-processing and the 3-channel conv_in.  One wave per row where a
// row reduction is needed (64-lane shuffle reductions), float4 accesses, grid-stride loops.
#include "vf_common.h"
#include <string.h>
#include "../../include/vf_hip.h"

namespace {

// ---------------------------------------------------------------- LayerNorm (migt.py:225,227,292)
// one wave per row; two-pass (mean, then centred variance) on register-resident data; d <= 64*4*MAXV
constexpr int LN_MAXV = 8;   // up to 2048 features
template <int MAXV, bool OUT16 = false>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out,
                                                        long long rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = d >> 2;   // float4 per row
    const float* xr = x + (size_t)row * d;
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = *reinterpret_cast<const f32x4*>(xr + c * 4);
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = vf_wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float t = v[i][e] - mean; q += t * t; }
        }
    }
    const float var = vf_wave_sum(q) / (float)d;
    const float rstd = 1.0f / sqrtf(var + eps);
    float* orow = out + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + c * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
            if (OUT16) {                               // bf16 row for a bf16-MFMA consumer (rounded exactly as it would round on load)
                typedef __bf16 ln_bf16x4 __attribute__((ext_vector_type(4)));
                ln_bf16x4 ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) ob[e] = (__bf16)o[e];
                *reinterpret_cast<ln_bf16x4*>(reinterpret_cast<__bf16*>(out) + (size_t)row * d + c * 4) = ob;
            } else
            *reinterpret_cast<f32x4*>(orow + c * 4) = o;
        }
    }
}

// ---------------------------------------------------------------- embedding sum (migt.py:358-368,392)
__global__ __launch_bounds__(256) void embed_sum_kernel(const int* __restrict__ ids, const float* __restrict__ wte,
                                                        const float* __restrict__ wpe, const float* __restrict__ add,
                                                        float* __restrict__ out, long long BS, int L, int d, int vocab) {
    const int dq = d >> 2;
    const long long total = BS * L * dq;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % dq);
        const long long tok = i / dq;
        const int l = (int)(tok % L);
        const long long bs = tok / L;
        int id = ids[tok];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4 a = *reinterpret_cast<const f32x4*>(wte + (size_t)id * d + c * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(wpe + (size_t)l * d + c * 4);
        const f32x4 p = *reinterpret_cast<const f32x4*>(add + (size_t)bs * d + c * 4);
        // reference order: sum([tok, pos, pose]) = (tok + pos) + pose   (migt.py:332-333)
        *reinterpret_cast<f32x4*>(out + i * 4) = (a + b) + p;
    }
}

// ---------------------------------------------------------------- tiny dense, K <= 16 (pose c_fc, K=7)
__global__ __launch_bounds__(256) void dense_small_k_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ out,
                                                            long long rows, int K, int N, int gelu) {
    const long long total = rows * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        const long long r = i / N;
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(x[r * K + k], W[(size_t)k * N + n], acc);
        acc += b ? b[n] : 0.f;
        out[i] = gelu ? vf_gelu_erf(acc) : acc;
    }
}

// ---------------------------------------------------------------- first-max argmax, one wave per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long long rows, int n, int ld,
                                                          long long* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * ld;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < n; c += 64) {
        const float v = xr[c];
        if (v > bv || bi == 0x7fffffff) { bv = v; bi = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) idx[row] = bi == 0x7fffffff ? 0 : bi;
}

// ---------------------------------------------------------------- row softmax (in place), one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, long long rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + (size_t)row * n;
    float mx = -INFINITY;
    for (int c = lane; c < n; c += 64) mx = fmaxf(mx, xr[c] * scale);
    mx = vf_wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < n; c += 64) { const float e = expf(xr[c] * scale - mx); xr[c] = e; s += e; }
    s = vf_wave_sum(s);
    for (int c = lane; c < n; c += 64) xr[c] = xr[c] / s;
}

// ---------------------------------------------------------------- uint8 post-process
__global__ void postprocess_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v = fminf(fmaxf(x[i], -1.0f), 1.0f);   // tf.clip_by_value
        v = v / 2.0f + 0.5f;                          // :129
        int q = (int)(v * 255.5f);                    // convert_image_dtype: scale = max + 0.5, truncating cast
        q = q < 0 ? 0 : (q > 255 ? 255 : q);
        out[i] = (unsigned char)q;
    }
}

// ---------------------------------------------------------------- conv_in: u8/f32 NHWC(3) -> 3x3 pad1 -> Cout
// thread = (pixel, CPT output channels): the 27 inputs of the pixel are loaded / converted ONCE per thread and
// reused for CPT (32 when Cout % 32 == 0) channels; weights [27][Cout] + bias in LDS (same-address reads across
// the pixels of a wave broadcast).  Write-bound by design (Cout*4 B per pixel): the first version converted the 27
// inputs for every 4 channels and ran at 0.65 TB/s.
template <int CPT>
__global__ __launch_bounds__(256) void conv_in_kernel(const unsigned char* __restrict__ img_u8,
                                                      const float* __restrict__ img_f32, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int n_img, int H, int W, int Cout) {
    extern __shared__ float sw[];   // [27][Cout] then [Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) {
        const int co = i % Cout, t = i / Cout;           // t = ci*9 + ky*3 + kx  (OIHW inner order)
        sw[i] = w[(size_t)co * 27 + t];
    }
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[27 * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    // CPT channels per thread as CPT/4 float4 chunks strided by 32 channels: the 8 threads of a pixel then write one
    // full 128-byte line per store instruction (chunk k: channels k*32 + t8*4 .. +3)
    const int cg = Cout / CPT;                            // threads per pixel (8 when CPT = Cout/8)
    const int cstride = (CPT == 4) ? 0 : cg * 4;          // channel stride between a thread's float4 chunks
    const long long total = (long long)n_img * H * W * cg;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % cg) * 4;
        const long long pix = i / cg;
        const int ox = (int)(pix % W);
        const int oy = (int)((pix / W) % H);
        const long long img = pix / ((long long)W * H);
        float in[27];                                      // [ky][kx][ci], zero outside the image (pad 1)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const size_t p = ok ? ((size_t)img * H + iy) * W + ix : 0;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    float v;
                    if (img_f32) v = img_f32[p * 3 + ci];
                    else v = ((float)img_u8[p * 3 + ci] * (1.0f / 255.0f)) * 2.0f - 1.0f;   // TF convert_image_dtype, *2-1
                    in[(ky * 3 + kx) * 3 + ci] = ok ? v : 0.f;
                }
            }
        float* o = out + (size_t)pix * Cout + c0;
#pragma unroll
        for (int ck = 0; ck < CPT / 4; ++ck) {
            const int c4 = ck * cstride;
            f32x4 acc = *reinterpret_cast<const f32x4*>(sw + 27 * Cout + c0 + c4);
            // same accumulation order as before: bias, then (ky, kx, ci)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ci = 0; ci < 3; ++ci) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + (ci * 9 + ky * 3 + kx) * Cout + c0 + c4);
                        acc += in[(ky * 3 + kx) * 3 + ci] * wv;
                    }
            *reinterpret_cast<f32x4*>(o + c4) = acc;
        }
    }
}

inline unsigned grid_for(long long total, int per_block, unsigned cap = 16384) {
    long long b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// image resize of the evaluators' pre-process (data/_common.py:19-61): uint8 -> /255 -> nearest (enlarging) or bilinear,
// align_corners = False (shrinking) -> clamp -> *255 -> truncate to uint8.  The reference's CPU interpolation evaluates
// fma(l0, a, l1 * b) along x, then along y, in fp32; reproduced operation by operation so that the uint8 result (a truncation,
// hence sensitive to the last bit) is identical (tests/golden/resize.npz).
__device__ __forceinline__ void resize_axis(int dst, int in, float scale, int& i0, int& i1, float& l0, float& l1) {
    float src = __fsub_rn(__fmul_rn(scale, __fadd_rn((float)dst, 0.5f)), 0.5f);
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = __fsub_rn(src, (float)i0);
    l0 = __fsub_rn(1.f, l1);
}

__global__ void resize_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long long total, int Hin, int Win,
                                 int Hout, int Wout, int C, int bilinear) {
    const float sy = __fdiv_rn((float)Hin, (float)Hout), sx = __fdiv_rn((float)Win, (float)Wout);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int x = (int)(t % Wout); t /= Wout;
        const int y = (int)(t % Hout);
        const long long img = t / Hout;
        const uint8_t* b = src + img * Hin * Win * C + c;
        float v;
        if (!bilinear) {
            int iy = (int)floorf(__fmul_rn((float)y, sy)), ix = (int)floorf(__fmul_rn((float)x, sx));
            iy = iy < Hin - 1 ? iy : Hin - 1;
            ix = ix < Win - 1 ? ix : Win - 1;
            v = __fdiv_rn((float)b[((long long)iy * Win + ix) * C], 255.f);
        } else {
            int y0, y1, x0, x1;
            float ly0, ly1, lx0, lx1;
            resize_axis(y, Hin, sy, y0, y1, ly0, ly1);
            resize_axis(x, Win, sx, x0, x1, lx0, lx1);
            const float p00 = __fdiv_rn((float)b[((long long)y0 * Win + x0) * C], 255.f);
            const float p01 = __fdiv_rn((float)b[((long long)y0 * Win + x1) * C], 255.f);
            const float p10 = __fdiv_rn((float)b[((long long)y1 * Win + x0) * C], 255.f);
            const float p11 = __fdiv_rn((float)b[((long long)y1 * Win + x1) * C], 255.f);
            const float top = __fmaf_rn(lx0, p00, __fmul_rn(lx1, p01));
            const float bot = __fmaf_rn(lx0, p10, __fmul_rn(lx1, p11));
            v = __fmaf_rn(ly0, top, __fmul_rn(ly1, bot));
        }
        v = fminf(fmaxf(v, 0.f), 1.f);
        dst[i] = (uint8_t)__fmul_rn(v, 255.f);
    }
}

}  // namespace

// registry behind vf_build_flags(): filled by the static initialisers of translation units built with a developer switch (vf_common.h)
namespace {
constexpr int VF_MAX_FLAGS = 32;
const char* g_flag_names[VF_MAX_FLAGS];
int g_flag_count = 0;
}

extern "C" {

int vf_register_build_flag(const char* name) {
    for (int i = 0; i < g_flag_count; ++i)
        if (strcmp(g_flag_names[i], name) == 0) return i;
    if (g_flag_count < VF_MAX_FLAGS) g_flag_names[g_flag_count++] = name;
    return g_flag_count - 1;
}

int vf_build_flags(void) { return g_flag_count; }

const char* vf_build_flag_name(int i) { return (i >= 0 && i < g_flag_count) ? g_flag_names[i] : nullptr; }

int vf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out, int64_t rows, int d, float eps,
                     void* stream) {
    if (!x || !gamma || !beta || !out || rows < 0 || d <= 0) return VF_ERR_BAD_ARG;
    if ((d & 3) || d > 64 * 4 * LN_MAXV) return VF_ERR_UNSUPPORTED;
    if (rows == 0) return VF_OK;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (d <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(256), 0, s, x, gamma, beta, out, (long long)rows, d, eps);
    else if (d <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, s, x, gamma, beta, out, (long long)rows, d, eps);
    else if (d <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, s, x, gamma, beta, out, (long long)rows, d, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, s, x, gamma, beta, out, (long long)rows, d, eps);
    return vf_last_status();
}

int vf_layernorm_bf16out_f32(const float* x, const float* gamma, const float* beta, void* out_bf16, int64_t rows, int d, float eps,
                             void* stream) {
    if (!x || !gamma || !beta || !out_bf16 || rows < 0 || d <= 0) return VF_ERR_BAD_ARG;
    if ((d & 3) || d > 64 * 4 * LN_MAXV) return VF_ERR_UNSUPPORTED;
    if (rows == 0) return VF_OK;
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    float* o = reinterpret_cast<float*>(out_bf16);
    if (d <= 256) hipLaunchKernelGGL((layernorm_kernel<1, true>), grid, dim3(256), 0, s, x, gamma, beta, o, (long long)rows, d, eps);
    else if (d <= 512) hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, dim3(256), 0, s, x, gamma, beta, o, (long long)rows, d, eps);
    else if (d <= 1024) hipLaunchKernelGGL((layernorm_kernel<4, true>), grid, dim3(256), 0, s, x, gamma, beta, o, (long long)rows, d, eps);
    else hipLaunchKernelGGL((layernorm_kernel<8, true>), grid, dim3(256), 0, s, x, gamma, beta, o, (long long)rows, d, eps);
    return vf_last_status();
}

int vf_embed_sum_f32(const int32_t* ids, const float* wte, const float* wpe, const float* add, float* out, int64_t BS,
                     int L, int d, int vocab, void* stream) {
    if (!ids || !wte || !wpe || !add || !out || BS < 0 || L <= 0 || d <= 0 || (d & 3) || vocab <= 0) return VF_ERR_BAD_ARG;
    if (BS == 0) return VF_OK;
    const long long total = (long long)BS * L * (d >> 2);
    hipLaunchKernelGGL(embed_sum_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ids, wte, wpe,
                       add, out, (long long)BS, L, d, vocab);
    return vf_last_status();
}

int vf_dense_small_k_gelu_f32(const float* x, const float* W, const float* b, float* out, int64_t rows, int K, int N,
                              int gelu, void* stream) {
    if (!x || !W || !out || rows < 0 || K <= 0 || N <= 0) return VF_ERR_BAD_ARG;
    if (K > 16) return VF_ERR_UNSUPPORTED;
    if (rows == 0) return VF_OK;
    hipLaunchKernelGGL(dense_small_k_kernel, dim3(grid_for((long long)rows * N, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, W, b, out, (long long)rows, K, N, gelu);
    return vf_last_status();
}

int vf_argmax_rows_f32(const float* x, int64_t rows, int n, int ld, int64_t* idx, void* stream) {
    if (!x || !idx || rows < 0 || n <= 0 || ld < n) return VF_ERR_BAD_ARG;
    if (rows == 0) return VF_OK;
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)rows, n, ld, reinterpret_cast<long long*>(idx));
    return vf_last_status();
}

int vf_softmax_rows_f32(float* x, int64_t rows, int n, float scale, void* stream) {
    if (!x || rows < 0 || n <= 0) return VF_ERR_BAD_ARG;
    if (rows == 0) return VF_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)rows, n, scale);
    return vf_last_status();
}

int vf_resize_u8(const uint8_t* src, uint8_t* dst, int n_img, int Hin, int Win, int Hout, int Wout, int C, int bilinear,
                 void* stream) {
    if (n_img < 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0) return VF_ERR_BAD_ARG;
    if (n_img == 0) return VF_OK;
    if (!src || !dst) return VF_ERR_BAD_ARG;
    const long long total = (long long)n_img * Hout * Wout * C;
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, total, Hin, Win,
                       Hout, Wout, C, bilinear);
    return vf_last_status();
}

int vf_postprocess_u8(const float* x, uint8_t* out, int64_t n, void* stream) {
    if (!x || !out || n < 0) return VF_ERR_BAD_ARG;
    if (n == 0) return VF_OK;
    hipLaunchKernelGGL(postprocess_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       (long long)n);
    return vf_last_status();
}

int vf_conv_in_u8_f32(const uint8_t* img_u8, const float* img_f32, const float* w_oihw, const float* bias, float* out,
                      int n_img, int H, int W, int Cout, void* stream) {
    if ((!img_u8 && !img_f32) || !w_oihw || !out || n_img <= 0 || H <= 0 || W <= 0 || Cout <= 0) return VF_ERR_BAD_ARG;
    if (Cout & 3) return VF_ERR_UNSUPPORTED;
    const size_t smem = (size_t)28 * Cout * sizeof(float);
    if (smem > 64 * 1024) return VF_ERR_UNSUPPORTED;
    if (Cout == 128) {          // 8 threads per pixel x 16 channels: full-line stores
        const long long total = (long long)n_img * H * W * 8;
        hipLaunchKernelGGL(conv_in_kernel<16>, dim3(grid_for(total, 256, 16384)), dim3(256), smem, (hipStream_t)stream, img_u8,
                           img_f32, w_oihw, bias, out, n_img, H, W, Cout);
    } else {
        const long long total = (long long)n_img * H * W * (Cout >> 2);
        hipLaunchKernelGGL(conv_in_kernel<4>, dim3(grid_for(total, 256, 8192)), dim3(256), smem, (hipStream_t)stream, img_u8,
                           img_f32, w_oihw, bias, out, n_img, H, W, Cout);
    }
    return vf_last_status();
}

/* host-side CRC-32C (Castagnoli, reflected; slicing-by-8) for the TFRecord / TensorBundle wire formats the reference's
 * datasets and Keras checkpoints use (tensorflow/core/lib/hash/crc32c.h).  crc = 0 starts a new checksum. */
uint32_t vf_crc32c(const void* data, size_t n, uint32_t crc) {
    static uint32_t T[8][256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
            T[0][i] = c;
        }
        for (int t = 1; t < 8; ++t)
            for (uint32_t i = 0; i < 256; ++i) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        init = true;
    }
    const unsigned char* p = static_cast<const unsigned char*>(data);
    crc = ~crc;
    while (n >= 8) {
        const uint32_t lo = crc ^ ((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
        crc = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^ T[3][p[4]] ^ T[2][p[5]] ^
              T[1][p[6]] ^ T[0][p[7]];
        p += 8;
        n -= 8;
    }
    while (n--) crc = T[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

}  // extern "C"
