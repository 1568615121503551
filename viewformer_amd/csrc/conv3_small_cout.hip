// 3x3 stride-1 convolution to a handful of output channels (the decoder's conv_out: 128 -> 3 @128x128,
// vqgan_th.py:285-289,316-318 with the norm_out GroupNorm + swish of :313-315 fused), gfx950.
//
// With 3 output channels the implicit-GEMM kernels pad N to 32 and waste 10x of the matrix pipe (0.45 ms per 32 images
// at 79 "TF" of padded work).  The algorithmic work is tiny (3456 MAC per pixel), so this is a plain VALU kernel bounded by
// the one read of the activation: one workgroup = an 8x32 pixel tile; per 32-channel chunk the (8+2)x(32+2) patch goes
// through the GroupNorm-apply(+swish) ONCE into LDS (fp32, 36-float pixel stride: conflict-free b128 reads), the chunk's
// [tap][c][4] weights sit next to it (wave-uniform broadcast reads), and each thread accumulates its pixel's outputs in
// fp32 fmaf order (chunk, tap, channel).
#include <type_traits>
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int CK = 32, TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, P_LD = 36;
constexpr int SLOTS = (NPIX * 8 + 255) / 256;          // float4 staging slots per thread (11)
constexpr int MAXCO = 4;

// IN16 (round 4): the input activation is bf16 NHWC (the bf16-activation decoder, conv3_halo_bf16.hip's IO16): 8 bytes per thread and chunk, widened exactly
template <bool PRO, bool SWISH, bool IN16 = false>
__global__ __launch_bounds__(256) void conv3_small_cout_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ pro_mean,
                                                               const float* __restrict__ pro_scale,
                                                               const float* __restrict__ pro_beta, float* __restrict__ out,
                                                               int H, int W, int Cin, int Cout) {
    __shared__ __attribute__((aligned(16))) float patch[NPIX * P_LD];
    extern __shared__ __attribute__((aligned(16))) float wl[];      // ALL weights, once: [chunk][tap][c][MAXCO] (Cin / 32 x 4.6 KB)
    const int tid = threadIdx.x;
    const int tilesX = W / TW, tilesY = H / TH;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* __restrict__ X = IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(x) + (size_t)img * H * W * Cin)
                                       : x + (size_t)img * H * W * Cin;
    const int c4 = tid & 7;
    const int px = tid & 31, py = tid >> 5;
    const int nchunks = Cin / CK;

    float acc[MAXCO];
#pragma unroll
    for (int co = 0; co < MAXCO; ++co) acc[co] = (co < Cout && bias) ? bias[co] : 0.f;

    // weights: wl[chunk][tap][c][co] from OIHW w[co][Cin][3][3] (the first version re-gathered a chunk's 1152 values from global per chunk)
    for (int i = tid; i < nchunks * 9 * CK * MAXCO; i += 256) {
        const int co = i & (MAXCO - 1), c = (i >> 2) & (CK - 1), tap = (i / (CK * MAXCO)) % 9, chunk = i / (9 * CK * MAXCO);
        wl[i] = co < Cout ? w[((size_t)co * Cin + chunk * CK + c) * 9 + tap] : 0.f;
    }
    // patch slots of this thread (the next chunk's raw values are fetched one chunk ahead: the first version loaded and used them in
    // the same chunk, exposing the HBM latency four times per workgroup)
    int s_off[SLOTS];
    unsigned ok_mask = 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int pix = (tid >> 3) + 32 * q;
        const int pixc = pix < NPIX ? pix : 0;
        const int pr = pixc / PW, pc = pixc - pr * PW;
        const int sy = y0 - 1 + pr, sx = x0 - 1 + pc;
        const bool ok = pix < NPIX && sy >= 0 && sy < H && sx >= 0 && sx < W;
        ok_mask |= (unsigned)ok << q;
        s_off[q] = ok ? (sy * W + sx) * Cin + c4 * 4 : c4 * 4;
    }
    f32x4 preg[SLOTS];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            if constexpr (IN16) {
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(X) + s_off[q] + chunk * CK);
                preg[q][0] = __builtin_bit_cast(float, w[0] << 16);
                preg[q][1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
                preg[q][2] = __builtin_bit_cast(float, w[1] << 16);
                preg[q][3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
            } else {
                preg[q] = *reinterpret_cast<const f32x4*>(X + s_off[q] + chunk * CK);
            }
        }
    };
    fetch(0);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                                   // the previous chunk's readers are done (chunk 0: the weights are in place)
        f32x4 pm = {0.f, 0.f, 0.f, 0.f}, ps = pm, pb = pm;
        if (PRO) {
            pm = *reinterpret_cast<const f32x4*>(pro_mean + (size_t)img * Cin + chunk * CK + c4 * 4);
            ps = *reinterpret_cast<const f32x4*>(pro_scale + (size_t)img * Cin + chunk * CK + c4 * 4);
            pb = *reinterpret_cast<const f32x4*>(pro_beta + chunk * CK + c4 * 4);
        }
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int pix = (tid >> 3) + 32 * q;
            if (pix < NPIX) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((ok_mask >> q) & 1u) {
                    v = preg[q];
                    if (PRO) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = (v[e] - pm[e]) * ps[e] + pb[e];
                            if (SWISH) t = vf_swish_1ulp(t);
                            v[e] = t;
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(patch + pix * P_LD + c4 * 4) = v;
            }
        }
        if (chunk + 1 < nchunks) fetch(chunk + 1);
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* a = patch + ((py + tap / 3) * PW + px + tap % 3) * P_LD;
            const float* wt = wl + (chunk * 9 + tap) * CK * MAXCO;
#pragma unroll
            for (int k4 = 0; k4 < CK / 4; ++k4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(a + k4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wt + (k4 * 4 + e) * MAXCO);
#pragma unroll
                    for (int co = 0; co < MAXCO; ++co) acc[co] = __builtin_fmaf(av[e], wv[co], acc[co]);
                }
            }
        }
    }
    float* o = out + (((size_t)img * H + y0 + py) * W + x0 + px) * Cout;
    for (int co = 0; co < Cout; ++co) o[co] = acc[co];
}

}  // namespace

extern "C" {

int vf_conv3_small_cout_f32(const float* x, const float* w_oihw, const float* bias, const float* pro_mean,
                            const float* pro_scale, const float* pro_beta, int pro_swish, float* out, int n_img, int H, int W,
                            int Cin, int Cout, int x_bf16, void* stream) {
    if (!x || !w_oihw || !out || n_img <= 0 || H <= 0 || W <= 0) return VF_ERR_BAD_ARG;
    if (Cout < 1 || Cout > MAXCO || Cin % CK != 0 || H % TH != 0 || W % TW != 0) return VF_ERR_UNSUPPORTED;
    if ((pro_mean || pro_scale || pro_beta) && !(pro_mean && pro_scale && pro_beta)) return VF_ERR_BAD_ARG;
    const dim3 grid((unsigned)((long long)n_img * (H / TH) * (W / TW)));
    hipStream_t s = (hipStream_t)stream;
    const size_t wsm = (size_t)(Cin / CK) * 9 * CK * MAXCO * sizeof(float);
    if (wsm > 96 * 1024) return VF_ERR_UNSUPPORTED;
    auto launch = [&](auto pro, auto swish, auto in16) -> int {
        constexpr bool P = decltype(pro)::value, S = decltype(swish)::value, I = decltype(in16)::value;
        static unsigned long long attr_devs = 0;      // per instantiation; bit d: raised on device d (static patch 49 KB + dynamic weights > 64 KB)
        if (vf_attr_needed(&attr_devs)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_small_cout_kernel<P, S, I>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr_devs);
        }
        hipLaunchKernelGGL((conv3_small_cout_kernel<P, S, I>), grid, dim3(256), wsm, s, x, w_oihw, bias, pro_mean, pro_scale, pro_beta, out, H, W, Cin, Cout);
        return vf_last_status();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (!pro_mean) return x_bf16 ? launch(F_{}, F_{}, T_{}) : launch(F_{}, F_{}, F_{});
    if (pro_swish) return x_bf16 ? launch(T_{}, T_{}, T_{}) : launch(T_{}, T_{}, F_{});
    return x_bf16 ? launch(T_{}, F_{}, T_{}) : launch(T_{}, F_{}, F_{});
}

}  // extern "C"
