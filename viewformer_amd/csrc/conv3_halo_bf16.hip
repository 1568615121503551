// bf16-MFMA sibling of conv3_halo_f32.hip (same halo-tile structure, same pixel permutation, same epilogue):
// 3x3 stride-1 / nearest-x2-upsample convolutions of the DECODER on v_mfma_f32_32x32x16_bf16, with fp32
// activations in HBM, the GroupNorm-apply(+swish) prologue evaluated in fp32 and rounded to bf16 (RNE) when the
// patch is parked in LDS, bf16 fragment-packed weights streamed L2 -> VGPR, fp32 accumulation and the fp32
// bias/residual epilogue.  Decoded pixels are bounded by a tolerance in the north star (not bit-exact), so this
// arm trades 2^-9 operand rounding for the 16x faster matrix pipe; the encoder never uses it.
// Patch pixel stride 80 B (64 B of bf16 + 16 B pad): 16 consecutive patch pixels -> 16 distinct 16-byte LDS
// slots for every ds_read_b128 lane group, for every tap shift.
#include "halo_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef VF_BF16_CONV_BUF
#define VF_BF16_CONV_BUF 1
#endif
#ifndef VF_BF16_CONV_TALL
#define VF_BF16_CONV_TALL 0     // A/B on MI355X: 694 vs 702 TF (128 ch @128^2), 994 vs 1004 (512 ch @16^2) — the weight stream is not this kernel's bound
#endif

constexpr int CK = 32;
constexpr int P_LDB = 80;           // bytes per patch pixel in LDS
constexpr int TH = 8, TW = 16;
constexpr int BN = 128;
constexpr int TAP_BYTES = CK * BN * 2;   // one (chunk, tap, n-block) weight tile: 8 KB

// PAIR: one 8x16 tile = two 8x8 images side by side, each with its own 10x10 halo patch (patch width 20)
template <bool UP2, bool PAIR = false>
struct Geo {
    static constexpr int PH = UP2 ? (TH / 2 + 2) : (TH + 2);
    static constexpr int PW = PAIR ? 20 : (UP2 ? (TW / 2 + 2) : (TW + 2));
    static constexpr int NPIX = PH * PW;
    static constexpr int SLOTS = (NPIX * 8 + 255) / 256;
    static constexpr int BUF = (NPIX + 1) * P_LDB;             // bytes, +1 dummy pixel
};

__device__ __forceinline__ float fast_swish(float t) {
    // bf16 arm only: hardware exp2 + rcp (the result is rounded to 8 mantissa bits right after)
    return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * t));
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv3_halo_bf16_kernel(vf_igemm_args p) {
    using G = Geo<UP2, PAIR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile: TALL = all 128 pixels x 32 channels (4 x 1 MFMA tiles, half the weight bytes per wave) instead of 64 pixels x 64
    // channels (2 x 2).  Unlike the x3h kernels it buys nothing here (see VF_BF16_CONV_TALL): at one MFMA per product the fp32
    // GroupNorm + swish prologue of the patch (two transcendentals per element) costs about as much as the MFMAs.
    constexpr bool TALL = VF_BF16_CONV_TALL != 0;
    constexpr int MI = TALL ? 4 : 2, NJ = TALL ? 1 : 2;
    const int wave_m = TALL ? 0 : wave >> 1, wave_n = TALL ? wave : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();                        // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = bid % nb; bid /= nb;
    int tx = 0, ty = 0, img;
    if (PAIR) { img = bid * 2; }
    else { tx = bid % tilesX; bid /= tilesX; ty = bid % tilesY; img = bid / tilesY; }
    const int n_img_total = p.M / (p.Hout * p.Wout);
    const int img1 = PAIR ? min(img + 1, n_img_total - 1) : img;   // 2nd image of the pair (= the 1st when n_img is odd: same values rewritten)
    const int pair_pix = (img1 - img) * p.Hin * p.Win;            // pixel offset of the 2nd image
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = UP2 ? (y0 / 2 - 1) : (y0 - 1);
    const int sx0 = UP2 ? (x0 / 2 - 1) : (x0 - 1);

    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / CK;
    const int last_stage = nchunks * 9 - 1;

    const int c4 = tid & 7;
    int s_off[G::SLOTS];
    bool s_ok[G::SLOTS];
    int s_sel[G::SLOTS];
    int s_lds[G::SLOTS];
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
        const int pix = (tid >> 3) + 32 * q;
        const int pixc = pix < G::NPIX ? pix : G::NPIX;
        const int pr = pixc / G::PW, pc0 = pixc - pr * G::PW;
        const int sel = PAIR ? (pc0 >= 10) : 0;
        const int pc = pc0 - 10 * sel;
        s_sel[q] = sel;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool ok = pix < G::NPIX && sy >= 0 && sy < p.Hin && sx >= 0 && sx < p.Win;
        s_ok[q] = ok;
        s_off[q] = ok ? (sel * pair_pix + sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        s_lds[q] = pixc * P_LDB + c4 * 8;
    }

    f32x4 preg[G::SLOTS];
    f32x4 pmean, pscale, pbeta, pmean1, pscale1;
#if VF_BF16_CONV_BUF      // buffer resources (SGPR base, scalar chunk / tap offset, 32-bit lane offsets) for the patch and the weight stream
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
#endif
    auto patch_load = [&](int chunk) {
#if VF_BF16_CONV_BUF
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q)
            preg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (unsigned)s_off[q] * 4u, (unsigned)(chunk * CK * 4), 0));
#else
        const float* xc = X + chunk * CK;
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
#endif
        if (PRO) {
            pmean = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pscale = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
            if (PAIR) {
                pmean1 = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
                pscale1 = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
            }
        }
    };
    auto patch_store_slot = [&](int buf, int q) {
        unsigned char* dst = smem_h + buf * G::BUF;
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = preg[q][e];
            if (PRO) {
                const float mu = (PAIR && s_sel[q]) ? pmean1[e] : pmean[e];
                const float sc = (PAIR && s_sel[q]) ? pscale1[e] : pscale[e];
                t = (t - mu) * sc + pbeta[e];
                if (SWISH) t = fast_swish(t);
            }
            o[e] = (__bf16)(s_ok[q] ? t : 0.f);
        }
        *reinterpret_cast<bf16x4*>(dst + s_lds[q]) = o;
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[MI], a_r[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int a0 = wave_m * 4 + mi * 2 + trow;
        a_r[mi] = a0;
        const int tcol = PAIR ? (tpx >> 3) * 10 + (tpx & 7) : tpx;
        a_base[mi] = (a0 * G::PW + tcol) * P_LDB + half * 16;
    }

    // packed weights [chunk][tap][nblk][ks(2)][half(2)][n(128)][8 bf16]
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    // software pipeline over taps (g = chunk*9 + tap): weight fragments BD taps ahead in a register ring, LDS fragments one
    // tap ahead; sched_barrier pins the distances (the scheduler otherwise sinks the loads next to their uses)
#ifndef VF_BF16_BD
#define VF_BF16_BD 2
#endif
    constexpr int BD = VF_BF16_BD, RING = BD + 1;
    static_assert(9 % RING == 0, "ring indices must repeat per chunk");
    bf16x8 bring[RING][2 * NJ];
    bf16x8 aring[2][MI][2];
    const int last_g = nchunks * 9 - 1;
#if VF_BF16_CONV_BUF
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
#endif
    auto b_load = [&](bf16x8 (&dst)[2 * NJ], int g) {
#if VF_BF16_CONV_BUF
        const unsigned soff = (unsigned)((size_t)min(g, last_g) * tap_stride);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dst[ks * NJ + j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (unsigned)((ks * 2 * BN + j * 32) * 16 + b_lane), soff, 0));
#else
        const unsigned char* src = Wb + (size_t)min(g, last_g) * tap_stride;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                dst[ks * NJ + j] = *reinterpret_cast<const bf16x8*>(src + (ks * 2 * BN + j * 32) * 16 + b_lane);
#endif
    };
    auto a_load = [&](bf16x8 (&dst)[MI][2], const unsigned char* patch, int tap) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int aoff;
            if (UP2) {
                const int pr = (a_r[mi] + dy + 1) >> 1, pc = (tpx + dx + 1) >> 1;
                aoff = (pr * G::PW + pc) * P_LDB + half * 16;
            } else {
                aoff = a_base[mi] + (dy * G::PW + dx) * P_LDB;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) dst[mi][ks] = *reinterpret_cast<const bf16x8*>(patch + aoff + ks * 32);
        }
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    patch_load(0);
#pragma unroll
    for (int g = 0; g < BD; ++g) b_load(bring[g], g);
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned char* patch = smem_h + (chunk & 1) * G::BUF;
        patch_load(min(chunk + 1, nchunks - 1));
        a_load(aring[0], patch, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            b_load(bring[(tap + BD) % RING], chunk * 9 + tap + BD);
            if (tap + 1 < 9) a_load(aring[(tap + 1) & 1], patch, tap + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[tap & 1][mi][ks], bring[tap % RING][ks * NJ + j], acc[mi][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap >= 1 && tap <= G::SLOTS) patch_store_slot((chunk + 1) & 1, tap - 1);
        }
        __syncthreads();
    }

    vf_halo_epilogue_t<PAIR, MI, NJ>(p, acc, img, img1, y0, x0, PAIR ? 0 : (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
}

__global__ void pack_conv_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Cin, int Cout, int nb, int nchunks) {
    // dst [chunk][tap][nblk][ks(2)][half(2)][n(128)][8]; src OIHW [Cout][Cin][3][3]
    const long long total = (long long)nchunks * 9 * nb * CK * BN;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        long long t = idx >> 3;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int ks = (int)((t >> 1) & 1);
        t >>= 2;
        const int nblk = (int)(t % nb); t /= nb;
        const int tap = (int)(t % 9);
        const int chunk = (int)(t / 9);
        const int c = chunk * CK + ks * 16 + half * 8 + e;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[((size_t)n * Cin + c) * 9 + tap];
        dst[idx] = (__bf16)v;
    }
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR>
int launch_halo(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, PAIR>;
    const size_t smem = (size_t)2 * G::BUF;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = PAIR ? (long long)((n_img + 1) / 2) * (a.Cout / BN)
                                  : (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<UP2, PRO, SWISH, PAIR>), dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PAIR>
int dispatch_pro(const vf_igemm_args& a, hipStream_t s) {
    if (!a.pro_mean) return launch_halo<UP2, false, false, PAIR>(a, s);
    return a.pro_swish ? launch_halo<UP2, true, true, PAIR>(a, s) : launch_halo<UP2, true, false, PAIR>(a, s);
}

}  // namespace

extern "C" {

size_t vf_conv3_bf16_packed_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cin + CK - 1) / CK) * 9 * ((Cout + BN - 1) / BN) * CK * BN;
}

int vf_conv3_bf16_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream) {
    if (!w_oihw || !dst || Cin <= 0 || Cout <= 0) return VF_ERR_BAD_ARG;
    const int nb = (Cout + BN - 1) / BN, nchunks = (Cin + CK - 1) / CK;
    const long long total = (long long)nchunks * 9 * nb * CK * BN;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (__bf16*)dst, Cin, Cout, nb,
                       nchunks);
    return vf_last_status();
}

int vf_conv3_halo_bf16(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0) return VF_ERR_BAD_ARG;
    if (a.drop_rate != 0.f || a.out_aux) return VF_ERR_UNSUPPORTED;      // fused output dropout / second output: vf_gemm_bf16 only
    if (a.mode != VF_MODE_CONV3_S1 && a.mode != VF_MODE_CONV3_UP2) return VF_ERR_UNSUPPORTED;
    const bool pair = a.mode == VF_MODE_CONV3_S1 && a.Hout == 8 && a.Wout == 8;     // two 8x8 images per tile
    if (a.Cout % BN != 0 || a.Cin % CK != 0 || (!pair && (a.Hout % TH != 0 || a.Wout % TW != 0))) return VF_ERR_UNSUPPORTED;
    if (a.Hin <= 0 || a.Win <= 0 || a.M % (a.Hout * a.Wout) != 0) return VF_ERR_BAD_ARG;
    if (a.mode == VF_MODE_CONV3_S1 && (a.Hout != a.Hin || a.Wout != a.Win)) return VF_ERR_BAD_ARG;
    if (a.mode == VF_MODE_CONV3_UP2 && (a.Hout != a.Hin * 2 || a.Wout != a.Win * 2)) return VF_ERR_BAD_ARG;
    if (a.batch > 1 || a.epilogue != VF_EPI_NONE || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
    if ((a.pro_mean || a.pro_scale || a.pro_beta) && !(a.pro_mean && a.pro_scale && a.pro_beta)) return VF_ERR_BAD_ARG;
    if ((long long)a.Hin * a.Win * a.Cin >= (1ll << 31)) return VF_ERR_UNSUPPORTED;
    if (int st = vf_halo_gn_check(a)) return st;
    hipStream_t s = (hipStream_t)stream;
    if (pair) return dispatch_pro<false, true>(a, s);
    return (a.mode == VF_MODE_CONV3_UP2) ? dispatch_pro<true, false>(a, s) : dispatch_pro<false, false>(a, s);
}

}  // extern "C"
