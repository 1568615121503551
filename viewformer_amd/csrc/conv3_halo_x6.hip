// fp32-EQUIVALENT 3x3 convolution on the bf16 matrix pipe ("x6": 6-term split-bf16 products), gfx950.
// Same halo-tile structure, pixel permutation and epilogue as conv3_halo_f32.hip, but every fp32 operand x is split
// EXACTLY into three bf16 pieces x = h + m + l (8+8+8 mantissa bits) and every fp32 product a*b is evaluated as
//     al*bh + ah*bl + am*bm + am*bh + ah*bm + ah*bh        (each bf16 x bf16 product is exact in fp32)
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16, small terms first.  The dropped terms (am*bl, al*bm, al*bl) are
// below 2^-24 of the product, i.e. below the rounding of a native fp32 multiply.  Measured (tools/split_bf16_probe.hip,
// profiles/r1_split_bf16_probe.txt): error vs fp64 relative to sum|a*b| over K=1152: x6 rms 2.4e-8 / max 1.5e-7,
// native v_mfma_f32_32x32x2_f32 rms 2.7e-8 / max 1.5e-7, host fmaf chain identical to the f32 MFMA.  The bf16 pipe is
// 16x the f32 pipe (measured 2.0-2.2 PF vs 0.15 PF), so 6 MFMAs per fp32 product still leave a 2.2x higher ceiling
// (~350 TF fp32-equivalent) than the native f32 MFMA (157 TF).
// Data path: fp32 activations in HBM; the GroupNorm-apply(+swish) prologue in exact fp32 (same code as the f32 kernel);
// the split happens ONCE per patch element when the patch is parked in LDS as [pixel][plane h|m|l][32 ch] bf16
// (208-byte pixel stride = 13 x 16 B: conflict-free ds_read_b128 for every tap shift); weights are pre-split at pack
// time into three fragment-packed bf16 planes and streamed L2 -> VGPR one (tap, k-step) stage ahead.
// Reference call sites: torch.nn.Conv2d 3x3 pad 1 in ResnetBlock / Upsample (vqgan_th.py:23-32,60-70,197,249) with
// GroupNorm+swish (:11-17,80-85) and the residual add (:90) fused.
#include "halo_common.h"

// tuning knobs (tools/variants.sh builds side-by-side libraries with different values for A/B timing on the GPU)
#ifndef VF_X6_BD
#define VF_X6_BD 2        // weight fragments are fetched this many stages ahead (register ring of BD + 1);
                          // measured 1/2/5: 192/198/204 TF @128^2 and 98/128/158 TF on the 8x8 pair tiles (L2-miss bound)
#endif
#ifndef VF_X6_AD
#define VF_X6_AD 1        // LDS activation fragments are read this many stages ahead (0 or 1)
#endif
#ifndef VF_X6_STORE
#define VF_X6_STORE 0     // 0: staging slot q is transformed+parked in stage 2q+1; 1: in stage 2q+3
#endif
#ifndef VF_X6_SB
#define VF_X6_SB 1        // 1: __builtin_amdgcn_sched_barrier(0) at every stage boundary (pins the prefetch distance: the
                          // scheduler otherwise sinks the ring loads next to their uses); 2: also between loads and MFMAs
#endif
#ifndef VF_X6_AMID
#define VF_X6_AMID 2      // VF_X6_SB == 3: the next stage's LDS fragments are issued before product t = AMID of this stage
#endif
#ifndef VF_X6_PRECISE_SWISH
#define VF_X6_PRECISE_SWISH 0
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;
constexpr int P_LDB = 208;          // bytes per patch pixel in LDS: 3 planes x 64 B + 16 B pad
constexpr int TH = 8, TW = 16;
constexpr int BN = 128;
constexpr int PLANE_BYTES = 2 * BN * 16;      // one (k-step, plane): [half(2)][n(128)][8 bf16] = 4 KB
constexpr int KS_BYTES = 3 * PLANE_BYTES;     // one k-step of 16 channels: 3 planes
constexpr int TAP_BYTES = 2 * KS_BYTES;       // one (chunk, tap, n-block) weight tile: 24 KB

// PAIR: one 8x16 tile = two 8x8 images side by side, each with its own 10x10 halo patch (patch width 20)
template <bool UP2, bool PAIR = false>
struct Geo {
    static constexpr int PH = UP2 ? (TH / 2 + 2) : (TH + 2);
    static constexpr int PW = PAIR ? 20 : (UP2 ? (TW / 2 + 2) : (TW + 2));
    static constexpr int NPIX = PH * PW;
    static constexpr int SLOTS = (NPIX * 8 + 255) / 256;
    static constexpr int BUF = (NPIX + 1) * P_LDB;             // bytes, +1 dummy pixel
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    // exact 3-way split: h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m); both subtractions are exact
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv3_halo_x6_kernel(vf_igemm_args p) {
    using G = Geo<UP2, PAIR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
#ifdef VF_X6_CLOCKPROBE      // developer build: shader-clock / 100 MHz wall-clock stamps of one mid-grid workgroup into p.bias[0..3] (as raw bits)
    const long long cp_c0 = __builtin_readcyclecounter(), cp_w0 = __builtin_amdgcn_s_memrealtime();
#endif

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
    auto patch_load = [&](int chunk) {
        const float* xc = X + chunk * CK;
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
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
        unsigned char* dst = smem_h + buf * G::BUF + s_lds[q];
        bf16x4 oh, om, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = preg[q][e];
            if (PRO) {
                const float mu = (PAIR && s_sel[q]) ? pmean1[e] : pmean[e];
                const float sc = (PAIR && s_sel[q]) ? pscale1[e] : pscale[e];
                t = (t - mu) * sc + pbeta[e];
                if (SWISH) t = VF_X6_PRECISE_SWISH ? vf_swish(t) : vf_swish_1ulp(t);
            }
            __bf16 h, m, l;
            split3(s_ok[q] ? t : 0.f, h, m, l);
            oh[e] = h; om[e] = m; ol[e] = l;
        }
        *reinterpret_cast<bf16x4*>(dst) = oh;
        *reinterpret_cast<bf16x4*>(dst + 64) = om;
        *reinterpret_cast<bf16x4*>(dst + 128) = ol;
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[2], a_r[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int a0 = wave_m * 4 + mi * 2 + trow;
        a_r[mi] = a0;
        const int tcol = PAIR ? (tpx >> 3) * 10 + (tpx & 7) : tpx;
        a_base[mi] = (a0 * G::PW + tcol) * P_LDB + half * 16;
    }

    // packed weights [chunk][tap][nblk][ks(2)][plane(3)][half(2)][n(128)][8 bf16]; one pipeline stage = one (tap, ks)
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * 64 + l31) * 16;
    // software pipeline over stages g = chunk*18 + tap*2 + ks: B two stages ahead in a 3-deep register ring (an L2 hit
    // costs about one stage of MFMA time, so one-ahead left the matrix pipe waiting), A one stage ahead (2-deep)
    constexpr int BD = VF_X6_BD, RING = BD + 1, AD = VF_X6_AD;
    static_assert(18 % RING == 0 && (AD == 0 || AD == 1), "ring indices must repeat per chunk");
    bf16x8 bring[RING][3][2];
    bf16x8 aring[2][2][3];
    const int last_g = nchunks * 18 - 1;
    auto b_load = [&](bf16x8 (&dst)[3][2], int g) {
        g = min(g, last_g);
        const unsigned char* src = Wb + (size_t)(g >> 1) * tap_stride + (g & 1) * KS_BYTES + b_lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[pl][j] = *reinterpret_cast<const bf16x8*>(src + pl * PLANE_BYTES + j * 32 * 16);
    };
    auto a_load = [&](bf16x8 (&dst)[2][3], const unsigned char* patch, int s) {
        const int tap = s >> 1, ks = s & 1;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            int aoff;
            if (UP2) {
                const int pr = (a_r[mi] + dy + 1) >> 1, pc = (tpx + dx + 1) >> 1;
                aoff = (pr * G::PW + pc) * P_LDB + half * 16;
            } else {
                aoff = a_base[mi] + (dy * G::PW + dx) * P_LDB;
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[mi][pl] = *reinterpret_cast<const bf16x8*>(patch + aoff + pl * 64 + ks * 32);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        if (AD) a_load(aring[0], patch, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            b_load(bring[(s + BD) % RING], chunk * 18 + s + BD);
            if (AD == 0) a_load(aring[s & 1], patch, s);
            else if (VF_X6_SB != 3 && s + 1 < 18) a_load(aring[(s + 1) & 1], patch, s + 1);
            if (VF_X6_SB == 2) __builtin_amdgcn_sched_barrier(0);      // loads are issued before this stage's MFMAs
            // the six partial products of a*b, smallest magnitude first (plane 0 = h, 1 = m, 2 = l)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (VF_X6_SB == 3 && t == VF_X6_AMID) {                // LDS fragments of the next stage issued mid-stage
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < 18) a_load(aring[(s + 1) & 1], patch, s + 1);
                }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[s & 1][mi][PA[t]], bring[s % RING][PB[t]][j], acc[mi][j], 0, 0, 0);
            }
            // the next chunk's patch: one staging slot per odd stage (transform + split in the MFMA shadow)
            if (VF_X6_SB == 1 || VF_X6_SB == 3) __builtin_amdgcn_sched_barrier(0);
            if ((s & 1) && (s >> 1) >= VF_X6_STORE && (s >> 1) - VF_X6_STORE < G::SLOTS) patch_store_slot((chunk + 1) & 1, (s >> 1) - VF_X6_STORE);
            if (VF_X6_SB == 2) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    vf_halo_epilogue<PAIR>(p, acc, img, img1, y0, x0, PAIR ? 0 : (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
#ifdef VF_X6_CLOCKPROBE
    if (blockIdx.x == gridDim.x / 2 && tid == 0 && p.gn_part) {
        long long* dbg = reinterpret_cast<long long*>(p.gn_part);
        dbg[0] = __builtin_readcyclecounter() - cp_c0;
        dbg[1] = __builtin_amdgcn_s_memrealtime() - cp_w0;
    }
#endif
}

// ---- 16x16-pixel tile form (stride 1, H % 16 == 0, W % 16 == 0); opt-in (-DVF_X6_BIG=1), see the A/B note at the dispatch -------
// One workgroup per CU (4 waves, one per SIMD, up to 512 registers each): a wave owns 8 rows x 16 px x 64 output channels
// = 8 accumulator tiles, so every weight fragment it loads feeds 4 MFMAs per product instead of 2 (half the L1 traffic per
// MFMA), the halo overhead drops from 1.41 to 1.27 patch pixels per output pixel, and a (tap, k-step) stage is 48 MFMAs =
// 1536 matrix-pipe cycles — long enough that one wave per SIMD with the loads pinned one / two stages ahead keeps the pipe fed
// on its own.  Same LDS layout, weight packing, epilogue and GroupNorm partial slots as the 8x16 kernel above.
constexpr int BTH = 16;
constexpr int B_PH = BTH + 2, B_PW = TW + 2, B_NPIX = B_PH * B_PW;          // 18 x 18 = 324
constexpr int B_SLOTS = (B_NPIX * 8 + 255) / 256;                           // 11
constexpr int B_BUF = (B_NPIX + 1) * P_LDB;

template <bool PRO, bool SWISH>
__global__ __launch_bounds__(256, 1) void conv3_halo_x6_big_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][B_BUF]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / BTH;
    int bid = (int)vf_xcd_bid();                        // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * BTH, x0 = tx * TW;
    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / CK;

    const int c4 = tid & 7;
    int s_off[B_SLOTS], s_lds[B_SLOTS];
    bool s_ok[B_SLOTS];
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) {
        const int pix = (tid >> 3) + 32 * q;
        const int pixc = pix < B_NPIX ? pix : B_NPIX;
        const int pr = pixc / B_PW, pc = pixc - pr * B_PW;
        const int sy = y0 - 1 + pr, sx = x0 - 1 + pc;
        const bool ok = pix < B_NPIX && sy >= 0 && sy < p.Hin && sx >= 0 && sx < p.Win;
        s_ok[q] = ok;
        s_off[q] = ok ? (sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        s_lds[q] = pixc * P_LDB + c4 * 8;
    }
    f32x4 preg[B_SLOTS];
    f32x4 pmean, pscale, pbeta;
    auto patch_load = [&](int chunk) {
        const float* xc = X + chunk * CK;
#pragma unroll
        for (int q = 0; q < B_SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
        if (PRO) {
            pmean = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pscale = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
        }
    };
    auto patch_store_slot = [&](int buf, int q) {
        unsigned char* dst = smem_h + buf * B_BUF + s_lds[q];
        bf16x4 oh, om, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = preg[q][e];
            if (PRO) {
                t = (t - pmean[e]) * pscale[e] + pbeta[e];
                if (SWISH) t = vf_swish_1ulp(t);
            }
            __bf16 h, m, l;
            split3(s_ok[q] ? t : 0.f, h, m, l);
            oh[e] = h; om[e] = m; ol[e] = l;
        }
        *reinterpret_cast<bf16x4*>(dst) = oh;
        *reinterpret_cast<bf16x4*>(dst + 64) = om;
        *reinterpret_cast<bf16x4*>(dst + 128) = ol;
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a_base[mi] = ((wave_m * 8 + mi * 2 + trow) * B_PW + tpx) * P_LDB + half * 16;

    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * 64 + l31) * 16;
    bf16x8 bring[3][3][2];
    bf16x8 aring[2][4][3];
    const int last_g = nchunks * 18 - 1;
    auto b_load = [&](bf16x8 (&dst)[3][2], int g) {
        g = min(g, last_g);
        const unsigned char* src = Wb + (size_t)(g >> 1) * tap_stride + (g & 1) * KS_BYTES + b_lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[pl][j] = *reinterpret_cast<const bf16x8*>(src + pl * PLANE_BYTES + j * 32 * 16);
    };
    auto a_load = [&](bf16x8 (&dst)[4][3], const unsigned char* patch, int s) {
        const int tap = s >> 1, ks = s & 1;
        const int off = ((tap / 3) * B_PW + tap % 3) * P_LDB + ks * 32;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[mi][pl] = *reinterpret_cast<const bf16x8*>(patch + a_base[mi] + off + pl * 64);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    patch_load(0);
    b_load(bring[0], 0);
    b_load(bring[1], 1);
#pragma unroll
    for (int q = 0; q < B_SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const unsigned char* patch = smem_h + (chunk & 1) * B_BUF;
        patch_load(min(chunk + 1, nchunks - 1));
        a_load(aring[0], patch, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            b_load(bring[(s + 2) % 3], chunk * 18 + s + 2);
#ifndef VF_X6_BIG_AMID
#define VF_X6_BIG_AMID 1      // the next stage's LDS fragments are issued before product t = AMID (one wave per SIMD: nothing else hides
#endif                        // their latency; left to the scheduler they sink to the end of the stage)
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (t == VF_X6_BIG_AMID) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < 18) a_load(aring[(s + 1) & 1], patch, s + 1);
                }
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[s & 1][mi][PA[t]], bring[s % 3][PB[t]][j], acc[mi][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s >= 1 && s <= B_SLOTS) patch_store_slot((chunk + 1) & 1, s - 1);
        }
        __syncthreads();
    }
    // the wave's 8 rows are exactly one 8x16 tile of the epilogue's (and the GroupNorm slots') geometry: rows 0-3 then 4-7
    const int ty8 = ty * 2 + wave_m;
    const f32x16 (&lo)[2][2] = *reinterpret_cast<const f32x16 (*)[2][2]>(&acc[0]);
    const f32x16 (&hi)[2][2] = *reinterpret_cast<const f32x16 (*)[2][2]>(&acc[2]);
    vf_halo_epilogue<false>(p, lo, img, img, ty8 * TH, x0, (ty8 * tilesX + tx) * 2, nblk, 0, wave_n, half, l31);
    vf_halo_epilogue<false>(p, hi, img, img, ty8 * TH, x0, (ty8 * tilesX + tx) * 2, nblk, 1, wave_n, half, l31);
}

// ---- stride-2 form: Downsample = pad (right, bottom) by one, 3x3 stride 2 (vqgan_th.py:45-49) ------------------------
// One 8x16 OUTPUT tile needs a 17x33 input patch; it is staged 16 channels at a time (112-byte pixel stride) in ONE LDS
// buffer (62 KB -> two workgroups per CU, which cover each other's staging), stored by column parity
// ([row][even columns | odd columns]) so that the stride-2 reads of a tap are again 16 consecutive LDS pixels.
// Weights: the same packing as the stride-1 kernel (a 16-channel chunk is one k-step of a packed 32-channel chunk).
constexpr int S2_PH = 2 * TH + 1, S2_PW = 2 * TW + 1;        // 17 x 33
constexpr int S2_NPIX = S2_PH * S2_PW;                       // 561
constexpr int S2_LDB = 112;                                  // 3 planes x 32 B + 16 B pad = 7 x 16 B
constexpr int S2_SLOTS = (S2_NPIX * 4 + 255) / 256;          // float4 staging slots per thread (9)
constexpr int S2_BUF = (S2_NPIX + 1) * S2_LDB;

__global__ __launch_bounds__(256, 2) void conv3_s2_x6_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [S2_BUF]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();                        // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / 16;

    // staging slots: thread -> (patch pixel, float4 column of the 16-channel chunk)
    const int c4 = tid & 3;
    int s_off[S2_SLOTS], s_lds[S2_SLOTS];
    bool s_ok[S2_SLOTS];
#pragma unroll
    for (int q = 0; q < S2_SLOTS; ++q) {
        const int pix = (tid >> 2) + 64 * q;
        const int pixc = pix < S2_NPIX ? pix : 0;
        const int pr = pixc / S2_PW, pc = pixc - pr * S2_PW;
        const int sy = 2 * y0 + pr, sx = 2 * x0 + pc;
        const bool ok = pix < S2_NPIX && sy < p.Hin && sx < p.Win;            // right / bottom zero padding
        s_ok[q] = ok;
        s_off[q] = ok ? (sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        const int slot = pix < S2_NPIX ? pr * S2_PW + (pc & 1) * (TW + 1) + (pc >> 1) : S2_NPIX;   // parity-major row
        s_lds[q] = slot * S2_LDB + c4 * 8;
    }
    f32x4 preg[S2_SLOTS];
    auto patch_load = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < S2_SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(X + chunk * 16 + s_off[q]);
    };
    auto patch_park = [&]() {
#pragma unroll
        for (int q = 0; q < S2_SLOTS; ++q) {
            bf16x4 oh, om, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h, m, l;
                split3(s_ok[q] ? preg[q][e] : 0.f, h, m, l);
                oh[e] = h; om[e] = m; ol[e] = l;
            }
            unsigned char* dst = smem_h + s_lds[q];
            *reinterpret_cast<bf16x4*>(dst) = oh;
            *reinterpret_cast<bf16x4*>(dst + 32) = om;
            *reinterpret_cast<bf16x4*>(dst + 64) = ol;
        }
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a_base[mi] = (2 * (wave_m * 4 + mi * 2 + trow) * S2_PW + tpx) * S2_LDB + half * 16;

    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * 64 + l31) * 16;
    const int last_g = nchunks * 9 - 1;
    bf16x8 bring[3][3][2];
    bf16x8 aring[2][2][3];
    auto b_load = [&](bf16x8 (&dst)[3][2], int g) {            // g = chunk16 * 9 + tap
        g = min(g, last_g);
        const int c = g / 9, tap = g - c * 9;
        const unsigned char* src = Wb + (size_t)((c >> 1) * 9 + tap) * tap_stride + (c & 1) * KS_BYTES + b_lane;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[pl][j] = *reinterpret_cast<const bf16x8*>(src + pl * PLANE_BYTES + j * 32 * 16);
    };
    auto a_load = [&](bf16x8 (&dst)[2][3], int tap) {
        const int dy = tap / 3, dx = tap % 3;
        const int off = (dy * S2_PW + (dx & 1) * (TW + 1) + (dx >> 1)) * S2_LDB;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) dst[mi][pl] = *reinterpret_cast<const bf16x8*>(smem_h + a_base[mi] + off + pl * 32);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    patch_load(0);
    b_load(bring[0], 0);
    b_load(bring[1], 1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                                        // every wave is done reading the previous chunk
        patch_park();
        patch_load(min(chunk + 1, nchunks - 1));
        __syncthreads();
        a_load(aring[0], 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            b_load(bring[(t + 2) % 3], chunk * 9 + t + 2);
            if (t + 1 < 9) a_load(aring[(t + 1) & 1], t + 1);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aring[t & 1][mi][PA[k]], bring[t % 3][PB[k]][j], acc[mi][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    vf_halo_epilogue<false>(p, acc, img, img, y0, x0, (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
}

__global__ void pack_conv_x6_kernel(const float* __restrict__ w, __bf16* __restrict__ dst, int Cin, int Cout, int nb, int nchunks) {
    // dst [chunk][tap][nblk][ks(2)][plane(3)][half(2)][n(128)][8]; src OIHW [Cout][Cin][3][3]
    const long long total = (long long)nchunks * 9 * nb * CK * BN;       // fp32 weights (each becomes 3 bf16)
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
        __bf16 h, m, l;
        split3(v, h, m, l);
        const size_t base = ((((size_t)(chunk * 9 + tap) * nb + nblk) * 2 + ks) * 3) * (2 * BN * 8) + ((size_t)half * BN + nl) * 8 + e;
        dst[base] = h;
        dst[base + 2 * BN * 8] = m;
        dst[base + 2 * (2 * BN * 8)] = l;
    }
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR>
int launch_halo(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, PAIR>;
    const size_t smem = (size_t)2 * G::BUF;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = PAIR ? (long long)((n_img + 1) / 2) * (a.Cout / BN)
                                  : (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    hipLaunchKernelGGL((conv3_halo_x6_kernel<UP2, PRO, SWISH, PAIR>), dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PAIR>
int dispatch_pro(const vf_igemm_args& a, hipStream_t s) {
    if (!a.pro_mean) return launch_halo<UP2, false, false, PAIR>(a, s);
    return a.pro_swish ? launch_halo<UP2, true, true, PAIR>(a, s) : launch_halo<UP2, true, false, PAIR>(a, s);
}

}  // namespace

extern "C" {

int vf_conv3_halo_gn_slots(int Hout, int Wout) { return vf_halo_gn_slots(Hout, Wout); }

size_t vf_conv3_x6_packed_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cin + CK - 1) / CK) * 9 * ((Cout + BN - 1) / BN) * CK * BN * 3;
}

int vf_conv3_x6_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream) {
    if (!w_oihw || !dst || Cin <= 0 || Cout <= 0) return VF_ERR_BAD_ARG;
    const int nb = (Cout + BN - 1) / BN, nchunks = (Cin + CK - 1) / CK;
    const long long total = (long long)nchunks * 9 * nb * CK * BN;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_x6_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, (__bf16*)dst, Cin, Cout, nb,
                       nchunks);
    return vf_last_status();
}

int vf_conv3_halo_x6(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0) return VF_ERR_BAD_ARG;
    if (a.drop_rate != 0.f || a.out_aux) return VF_ERR_UNSUPPORTED;      // fused output dropout / second output: vf_gemm_bf16 only
    if (a.mode == VF_MODE_CONV3_S2PAD) {
        if (a.Cout % BN != 0 || a.Cin % CK != 0 || a.Hout % TH != 0 || a.Wout % TW != 0 || a.pro_mean) return VF_ERR_UNSUPPORTED;
        if (a.Hin != a.Hout * 2 || a.Win != a.Wout * 2 || a.M % (a.Hout * a.Wout) != 0) return VF_ERR_BAD_ARG;
        if (a.batch > 1 || a.epilogue != VF_EPI_NONE || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
        if ((long long)a.Hin * a.Win * a.Cin >= (1ll << 31)) return VF_ERR_UNSUPPORTED;
        if (int st = vf_halo_gn_check(a)) return st;
        const long long blocks = (long long)(a.M / (a.Hout * a.Wout)) * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
        hipLaunchKernelGGL(conv3_s2_x6_kernel, dim3((unsigned)blocks), dim3(256), (size_t)S2_BUF, (hipStream_t)stream, a);
        return vf_last_status();
    }
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
#ifndef VF_X6_BIG
#define VF_X6_BIG 0      // A/B result (profiles/r1_x6_feed_probe.txt): 216-225 TF, the same as the 8x16 kernel -> kept opt-in
#endif
    if (VF_X6_BIG && a.mode == VF_MODE_CONV3_S1 && !pair && a.Hout % BTH == 0) {
        const long long blocks = (long long)(a.M / (a.Hout * a.Wout)) * (a.Hout / BTH) * (a.Wout / TW) * (a.Cout / BN);
        const size_t smem = (size_t)2 * B_BUF;
        static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
        if (vf_attr_needed(&attr_devs)) {
            for (const void* f : {reinterpret_cast<const void*>(conv3_halo_x6_big_kernel<false, false>),
                                  reinterpret_cast<const void*>(conv3_halo_x6_big_kernel<true, false>),
                                  reinterpret_cast<const void*>(conv3_halo_x6_big_kernel<true, true>)}) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
                if (e != hipSuccess) return (int)e;
            }
            vf_attr_done(&attr_devs);
        }
        if (!a.pro_mean) hipLaunchKernelGGL((conv3_halo_x6_big_kernel<false, false>), dim3((unsigned)blocks), dim3(256), smem, s, a);
        else if (a.pro_swish) hipLaunchKernelGGL((conv3_halo_x6_big_kernel<true, true>), dim3((unsigned)blocks), dim3(256), smem, s, a);
        else hipLaunchKernelGGL((conv3_halo_x6_big_kernel<true, false>), dim3((unsigned)blocks), dim3(256), smem, s, a);
        return vf_last_status();
    }
    if (pair) return dispatch_pro<false, true>(a, s);
    return (a.mode == VF_MODE_CONV3_UP2) ? dispatch_pro<true, false>(a, s) : dispatch_pro<false, false>(a, s);
}

}  // extern "C"
