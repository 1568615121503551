// bf16-MFMA sibling of conv3_halo_f32.hip (same halo-tile structure, same pixel permutation, same epilogue):
// 3x3 stride-1 / nearest-x2-upsample convolutions of the DECODER on v_mfma_f32_32x32x16_bf16, with fp32
// activations in HBM, the GroupNorm-apply(+swish) prologue evaluated in fp32 and rounded to bf16 (RNE) when the
// patch is parked in LDS, bf16 fragment-packed weights streamed L2 -> VGPR, fp32 accumulation and the fp32
// bias/residual epilogue.  Decoded pixels are bounded by a tolerance in the north star (not bit-exact), so this
// arm trades 2^-9 operand rounding for the 16x faster matrix pipe; the encoder never uses it.
// IO16 (round 4): the activations BETWEEN decoder layers can be bf16 in HBM (vf_igemm_args.reserved0 bit 0: x is bf16 NHWC; bit 1: out — and the
// residual, which is the same stream — are bf16).  At the decoder's dominant shape (128 -> 128 @128^2, 128 images) the kernel moved 3.2 GB of fp32
// activations per 0.87 ms launch (3.7 TB/s); the operand rounding itself is unchanged (the patch was rounded to bf16 on its way into LDS before,
// after the fp32 GroupNorm + swish — which it still is; only the value the prologue starts from, and the stored sum, now carry 8 mantissa bits).
// GroupNorm partial statistics are taken from the fp32 values before the output rounding.  bf16 stores / residual loads go by lane PAIRS (adjacent
// lanes hold adjacent channels of the same 16 pixels: each lane handles two channels of half the pixels, 4-byte accesses — gemm_bf16's scheme).
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

// the bf16-output epilogue (IO16 bit 1): bias, bf16 residual, fp32 GroupNorm partials, lane-pair bf16 stores
template <int MI, int NJ>
__device__ __forceinline__ void halo_epilogue16(const vf_igemm_args& p, const f32x16 (&acc)[MI][NJ], int img, int y0, int x0, int tile_slot, int nblk,
                                                int wave_m, int wave_n, int half, int l31) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    __bf16* __restrict__ Out = reinterpret_cast<__bf16*>(p.out) + (size_t)img * p.Hout * p.Wout * p.ldc;
    const __bf16* __restrict__ Res = p.res ? reinterpret_cast<const __bf16*>(p.res) + (size_t)img * p.Hout * p.Wout * p.ldr : nullptr;
    const bool stats = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;
    const int odd = l31 & 1;
    auto pix_of = [&](int mi, int r) {
        const int i0 = (r & 3) + 8 * (r >> 2) + 4 * half;
        return (y0 + wave_m * 4 + mi * 2 + vf_perm_row(i0)) * p.Wout + x0 + vf_perm_px(i0);
    };
    // the residual words of ALL the wave's tiles first: one memory round trip instead of one per tile (eight of them cost a quarter of the
    // kernel's time at 128 -> 128 @128^2: 0.18 of 0.72 ms).  The even lane fetches channels (n, n + 1) of pixel-row r, the odd lane (n - 1, n) of
    // row r + 1; one exchange hands each lane its own channel of both rows
    unsigned rw[NJ][MI][8];
    if (Res) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; r += 2)
                    rw[j][mi][r >> 1] = *reinterpret_cast<const unsigned*>(Res + (size_t)pix_of(mi, r + odd) * p.ldr + (n - odd));
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int sl = 0; sl < MI / 2; ++sl) {                       // one statistics slot = 4 tile rows = 2 accumulator tiles
            vf_gn_acc_t s = 0, q = 0;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int mi = sl * 2 + mm;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[mi][j][r] + bias;
                if (Res) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned mine = rw[j][mi][r >> 1];
                        const unsigned give = odd ? (mine << 16) : (mine & 0xffff0000u);          // as fp32 bits
                        const unsigned got = vf_lane_xor1(give);
                        v[r] += __builtin_bit_cast(float, odd ? got : (mine << 16));
                        v[r + 1] += __builtin_bit_cast(float, odd ? (mine & 0xffff0000u) : got);
                    }
                }
                if (stats) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s += (vf_gn_acc_t)v[r]; q += (vf_gn_acc_t)v[r] * (vf_gn_acc_t)v[r]; }
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float give = odd ? v[r] : v[r + 1];
                    const float got = vf_lane_xor1(give);
                    bf16x2_t h;
                    h[0] = (__bf16)(odd ? got : v[r]);
                    h[1] = (__bf16)(odd ? v[r + 1] : got);
                    *reinterpret_cast<bf16x2_t*>(Out + (size_t)pix_of(mi, r + odd) * p.ldc + (n - odd)) = h;
                }
            }
            if (stats) {
                vf_gn_group_reduce(s, q, cg);
                if (half == 0 && (l31 & (cg - 1)) == 0) {
                    float* dst = p.gn_part + ((((size_t)img * p.gn_slots) + tile_slot + wave_m + sl) * 32 + n / cg) * 2;
                    dst[0] = (float)s;
                    dst[1] = (float)q;
                }
            }
        }
    }
}

// IO16: bit 0 = the input activation is bf16, bit 1 = the output (and the residual) are bf16
template <bool UP2, bool PRO, bool SWISH, bool PAIR = false, int IO16 = 0>
__global__ __launch_bounds__(256, 2) void conv3_halo_bf16_kernel(vf_igemm_args p) {
    static_assert(!(PAIR && IO16), "the 8x8 pair tiles stay fp32");
    constexpr bool IN16 = (IO16 & 1) != 0, OUT16 = (IO16 & 2) != 0;
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

    const float* __restrict__ X = IN16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(p.x) + (size_t)img * p.Hin * p.Win * p.Cin)
                                       : p.x + (size_t)img * p.Hin * p.Win * p.Cin;
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
    f32x4 pa, pb, pa1, pb1;          // GroupNorm-apply folded to one fma: a = scale, b = beta - mean * scale (pa1 / pb1: the pair tile's 2nd image)
#if VF_BF16_CONV_BUF      // buffer resources (SGPR base, scalar chunk / tap offset, 32-bit lane offsets) for the patch and the weight stream
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
#endif
    auto patch_load = [&](int chunk) {
#if VF_BF16_CONV_BUF
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) {
            if constexpr (IN16) {                                   // four bf16 channels = 8 bytes per thread, widened exactly
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(x_rs, (unsigned)s_off[q] * 2u, (unsigned)(chunk * CK * 2), 0));
                preg[q][0] = __builtin_bit_cast(float, w[0] << 16);
                preg[q][1] = __builtin_bit_cast(float, w[0] & 0xffff0000u);
                preg[q][2] = __builtin_bit_cast(float, w[1] << 16);
                preg[q][3] = __builtin_bit_cast(float, w[1] & 0xffff0000u);
            } else {
                preg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (unsigned)s_off[q] * 4u, (unsigned)(chunk * CK * 4), 0));
            }
        }
#else
        static_assert(!IN16, "bf16 input needs the buffer-resource loads");
        const float* xc = X + chunk * CK;
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
#endif
        if (PRO) {
            const f32x4 pmean = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            const f32x4 pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
            pa = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img * p.Cin + chunk * CK + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) pb[e] = __builtin_fmaf(-pmean[e], pa[e], pbeta[e]);
            if (PAIR) {
                const f32x4 pmean1 = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
                pa1 = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) pb1[e] = __builtin_fmaf(-pmean1[e], pa1[e], pbeta[e]);
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
                t = __builtin_fmaf(t, (PAIR && s_sel[q]) ? pa1[e] : pa[e], (PAIR && s_sel[q]) ? pb1[e] : pb[e]);
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

    if constexpr (OUT16) halo_epilogue16<MI, NJ>(p, acc, img, y0, x0, (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
    else vf_halo_epilogue_t<PAIR, MI, NJ>(p, acc, img, img1, y0, x0, PAIR ? 0 : (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
}


#ifndef VF_BF16_CONV_W2
#define VF_BF16_CONV_W2 1       // bf16-activation launches (IO16 == 3) on the two-wave kernel below (0: the four-wave kernel's IO16 = 3 form)
#endif
#ifndef VF_W2_ABL
#define VF_W2_ABL 0             // ablation bits (WRONG results; tools/variants.sh builds only): 1 weight loads pinned to stage 0 (L1 hits), 2 LDS fragment
#endif                          // reads once per chunk, 4 no epilogue stores, 8 no MFMAs, 16 no patch loads / stores after chunk 0
#ifndef VF_BF16_W2_BD
#define VF_BF16_W2_BD 2         // weight fragments this many (tap, k-step) stages ahead; ring of BD + 1 (must divide 18)
#endif

// The two-wave form for bf16 activations (round 4).  The four-wave kernel above gives a wave 64 pixels x 64 channels (2 x 2 MFMA tiles): per
// 16-channel step 2 activation fragments (LDS) + 2 weight fragments (L2 -> L1 -> VGPR) feed 4 MFMAs, i.e. 0.5 KB through the vector-memory return
// path per MFMA.  That path delivers 64 B / clk / CU and the CU's four matrix pipes retire one 32x32x16 MFMA per 8 clk: 0.5 KB / MFMA IS the return
// path's peak, so the matrix pipe idled behind it (0.28 - 0.31 of the bf16 peak; the tall 4 x 1 tile moves the same problem to the LDS read path:
// 1 KB / MFMA = its 128 B / clk).  Here a workgroup is TWO waves and a wave owns all 128 pixels x 64 channels (4 x 2 tiles): 4 activation + 2 weight
// fragments per 8 MFMAs = 0.5 KB / MFMA from LDS (half its rate) and 0.25 KB / MFMA through the return path (half its rate).  Same tile, same pixel
// permutation, same k order (chunk, tap, k-step) -> bit-identical accumulators; four workgroups per CU keep two waves per SIMD.
// The patch is staged 8 channels per thread (one 16-byte bf16 load), the GroupNorm-apply is folded to one fma per element (a = scale,
// b = beta - mean * scale; both kernels of this file use the fold so that they round alike).
template <bool UP2, bool PRO, bool SWISH>
__global__ __launch_bounds__(128, 2) void conv3_halo_bf16_w2_kernel(vf_igemm_args p) {
    using G = Geo<UP2, false>;
    constexpr int SLOTS = (G::NPIX + 31) / 32;
    constexpr int MI = 4, NJ = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][BUF]
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_n = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = UP2 ? (y0 / 2 - 1) : (y0 - 1);
    const int sx0 = UP2 ? (x0 / 2 - 1) : (x0 - 1);
    const __bf16* __restrict__ X = reinterpret_cast<const __bf16*>(p.x) + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / CK;

    const int c8 = tid & 3;
    int s_off[SLOTS], s_lds[SLOTS];
    unsigned ok_mask = 0;
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int pix = (tid >> 2) + 32 * q;
        const int pixc = pix < G::NPIX ? pix : G::NPIX;
        const int pr = pixc / G::PW, pc = pixc - pr * G::PW;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool ok = pix < G::NPIX && sy >= 0 && sy < p.Hin && sx >= 0 && sx < p.Win;
        ok_mask |= (unsigned)ok << q;
        s_off[q] = (ok ? (sy * p.Win + sx) * p.Cin + c8 * 8 : c8 * 8) * 2;       // bytes
        s_lds[q] = pixc * P_LDB + c8 * 16;
    }
    u32x4 preg[SLOTS];
    float pa[8], pb[8];
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(X), 0, 0x7fffffff, 0x00020000);
    auto patch_load = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q)
            preg[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (unsigned)s_off[q], (unsigned)(chunk * CK * 2), 0));
        if (PRO) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = chunk * CK + c8 * 8 + h * 4;
                const f32x4 mu = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img * p.Cin + c);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img * p.Cin + c);
                const f32x4 be = *reinterpret_cast<const f32x4*>(p.pro_beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { pa[h * 4 + e] = sc[e]; pb[h * 4 + e] = __builtin_fmaf(-mu[e], sc[e], be[e]); }
            }
        }
    };
    auto patch_store_slot = [&](int buf, int q) {
        unsigned char* dst = smem_h + buf * G::BUF;
        const bool ok = (ok_mask >> q) & 1u;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned w = preg[q][e >> 1];
            float t = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
            if (PRO) {
                t = __builtin_fmaf(t, pa[e], pb[e]);
                if (SWISH) t = fast_swish(t);
            }
            o[e] = (__bf16)t;
        }
        u32x4 ow = __builtin_bit_cast(u32x4, o);              // zero padding outside the image: select on the packed words
#pragma unroll
        for (int e = 0; e < 4; ++e) ow[e] = ok ? ow[e] : 0u;
        *reinterpret_cast<u32x4*>(dst + s_lds[q]) = ow;
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a_base[mi] = ((mi * 2 + trow) * G::PW + tpx) * P_LDB + half * 16;

    // packed weights [chunk][tap][nblk][ks(2)][half(2)][n(128)][8 bf16]; a stage is one (tap, k-step): 8 MFMAs
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    constexpr int BD = VF_BF16_W2_BD, RB = BD + 1;
    static_assert(18 % RB == 0, "ring indices must repeat per chunk");
    bf16x8 bring[RB][NJ];
    bf16x8 aring[2][MI];
    const int last_g = nchunks * 9 - 1;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
    auto b_load = [&](bf16x8 (&dst)[NJ], int g2) {
        const unsigned soff = (VF_W2_ABL & 1) ? 0u : (unsigned)((size_t)min(g2 >> 1, last_g) * tap_stride);
        const int ks = g2 & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            dst[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(w_rs, (unsigned)((ks * 2 * BN + j * 32) * 16 + b_lane), soff, 0));
    };
    auto a_load = [&](bf16x8 (&dst)[MI], const unsigned char* patch, int s) {
        const int tap = s >> 1, ks = s & 1;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int aoff;
            if (UP2) {
                const int pr = (mi * 2 + trow + dy + 1) >> 1, pc = (tpx + dx + 1) >> 1;
                aoff = (pr * G::PW + pc) * P_LDB + half * 16;
            } else {
                aoff = a_base[mi] + (dy * G::PW + dx) * P_LDB;
            }
            dst[mi] = *reinterpret_cast<const bf16x8*>(patch + aoff + ks * 32);
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
    for (int s = 0; s < BD; ++s) b_load(bring[s], s);
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned char* patch = smem_h + (chunk & 1) * G::BUF;
        const bool more = chunk + 1 < nchunks;               // (the four-wave kernel re-stages the last chunk once more instead of branching)
        if (!(VF_W2_ABL & 16) && more) patch_load(chunk + 1);
        a_load(aring[0], patch, 0);
        if (VF_W2_ABL & 2) a_load(aring[1], patch, 1);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            b_load(bring[(s + BD) % RB], chunk * 18 + s + BD);
            if (s + 1 < 18 && !(VF_W2_ABL & 2)) a_load(aring[(s + 1) & 1], patch, s + 1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (VF_W2_ABL & 8) { acc[mi][j][0] += (float)aring[s & 1][mi][0] * (float)bring[s % RB][j][0]; }
                    else acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[s & 1][mi], bring[s % RB][j], acc[mi][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (!(VF_W2_ABL & 16) && (s & 1) && (s >> 1) < SLOTS && more) patch_store_slot((chunk + 1) & 1, s >> 1);
        }
        __syncthreads();
    }
    if ((VF_W2_ABL & 4) && acc[0][0][0] != 12345.678f) return;
    halo_epilogue16<MI, NJ>(p, acc, img, y0, x0, (ty * tilesX + tx) * 2, nblk, 0, wave_n, half, l31);
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

template <bool UP2, bool PRO, bool SWISH, bool PAIR, int IO16 = 0>
int launch_halo(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, PAIR>;
    const size_t smem = (size_t)2 * G::BUF;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = PAIR ? (long long)((n_img + 1) / 2) * (a.Cout / BN)
                                  : (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    hipLaunchKernelGGL((conv3_halo_bf16_kernel<UP2, PRO, SWISH, PAIR, IO16>), dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PRO, bool SWISH>
int launch_halo_w2(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, false>;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    hipLaunchKernelGGL((conv3_halo_bf16_w2_kernel<UP2, PRO, SWISH>), dim3((unsigned)blocks), dim3(128), (size_t)2 * G::BUF, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PAIR, int IO16 = 0>
int dispatch_pro(const vf_igemm_args& a, hipStream_t s) {
    if (!a.pro_mean) return launch_halo<UP2, false, false, PAIR, IO16>(a, s);
    return a.pro_swish ? launch_halo<UP2, true, true, PAIR, IO16>(a, s) : launch_halo<UP2, true, false, PAIR, IO16>(a, s);
}
template <bool UP2>
int dispatch_io(const vf_igemm_args& a, hipStream_t s) {
    switch (a.reserved0 & 3) {                                      // bit 0: bf16 in; bit 1: bf16 out (+ bf16 residual)
        case 0: return dispatch_pro<UP2, false, 0>(a, s);
        case 2: return dispatch_pro<UP2, false, 2>(a, s);
#if VF_BF16_CONV_W2
        case 3:
            if (a.Cin & 7) return VF_ERR_UNSUPPORTED;
            if (!a.pro_mean) return launch_halo_w2<UP2, false, false>(a, s);
            return a.pro_swish ? launch_halo_w2<UP2, true, true>(a, s) : launch_halo_w2<UP2, true, false>(a, s);
#else
        case 3: return dispatch_pro<UP2, false, 3>(a, s);
#endif
        default: return VF_ERR_UNSUPPORTED;                        // (bf16 in, fp32 out: no caller)
    }
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
    if (a.reserved0 & ~3) return VF_ERR_BAD_ARG;
    if (a.reserved0 && (pair || (a.ldc & 1) || (a.res && (a.ldr & 1)) || (a.Cin & 3))) return VF_ERR_UNSUPPORTED;      // bf16 activations: 4-byte lane pairs
    if (pair) return dispatch_pro<false, true>(a, s);
    return (a.mode == VF_MODE_CONV3_UP2) ? dispatch_io<true>(a, s) : dispatch_io<false>(a, s);
}

}  // extern "C"
