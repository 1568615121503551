// 3x3 convolution (stride 1, and nearest-x2-upsample + stride 1) as a halo-tile implicit GEMM on
// exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.  Second-generation kernel for the layers that
// carry ~90 % of the VQGAN FLOPs; igemm_f32.hip remains the generic path (stride-2, 8x8 maps,
// narrow channel counts, dense layers).
//
// Why: profiling the per-tap gather kernel (profiles/r1_conv_pertap.txt) showed the matrix pipe only 51 %
// busy with 10 VALU instructions per MFMA — the GroupNorm+swish prologue was re-evaluated for each of
// the 9 taps and every tap recomputed gather addresses and re-staged A through LDS.  Here:
//   * one workgroup = an 8x16 output-pixel tile x 128 output channels; per 32-channel chunk the
//     (8+2)x(16+2) input patch is fetched ONCE, pushed through the fused GroupNorm-apply(+swish)
//     ONCE (1.4 evaluations per output pixel instead of 9) and parked in LDS (double-buffered);
//   * the 9 taps are shifted windows of that patch: A fragments are ds_read_b128 at compile-time
//     offsets.  MFMA row i is mapped to the pixel that makes each hardware 16-lane ds_read_b128 group
//     ({0-3,12-15,20-27} / {4-11,16-19,28-31}) cover 16 CONSECUTIVE patch pixels -> bank-conflict free
//     with a 36-float pixel stride for every tap shift;
//   * weight fragments (pre-packed fragment-major) stream L2 -> VGPR directly, one tap ahead in
//     registers: no LDS staging, no VALU, and only ONE workgroup barrier per chunk (= per 576 MFMAs);
//   * the chunk body is ONE branch-free basic block (compile-time prologue flags, clamped prefetch
//     addresses, selects instead of predicated loads) with the patch transform of staging slot q issued
//     in tap q+1, so the scheduler can slot the VALU work into the 64-cycle MFMA shadows instead of
//     leaving it as one clump that both co-resident workgroups hit at the same time.
// Reference call sites: torch.nn.Conv2d 3x3 pad 1 in ResnetBlock / Upsample (vqgan_th.py:23-32,60-70,
// 197,249) with GroupNorm+swish (:11-17,80-85) and the residual add (:90) fused.
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int CK = 32;
constexpr int P_LD = 36;            // floats per patch pixel in LDS
constexpr int TH = 8, TW = 16;      // output tile
constexpr int BN = 128;

// MFMA row i (0..31) -> (tile row 0/1, pixel 0..15) such that each ds_read_b128 lane group is contiguous
__host__ __device__ constexpr int perm_row(int i) { return (i < 4) ? 0 : (i < 12) ? 1 : (i < 16) ? 0 : (i < 20) ? 1 : (i < 28) ? 0 : 1; }
__host__ __device__ constexpr int perm_px(int i) {
    return (i < 4) ? i : (i < 12) ? i - 4 : (i < 16) ? i - 8 : (i < 20) ? i - 8 : (i < 28) ? i - 12 : i - 16;
}

// PAIR: one 8x16 tile = two 8x8 images side by side, each with its own 10x10 halo patch (patch width 20)
template <bool UP2, bool PAIR = false>
struct Geo {
    static constexpr int PH = UP2 ? (TH / 2 + 2) : (TH + 2);   // patch extent in source pixels
    static constexpr int PW = PAIR ? 20 : (UP2 ? (TW / 2 + 2) : (TW + 2));
    static constexpr int NPIX = PH * PW;
    static constexpr int SLOTS = (NPIX * 8 + 255) / 256;       // float4 staging slots per thread
    static constexpr int BUF = (NPIX + 1) * P_LD;              // +1 dummy pixel: sink for idle staging lanes
};

template <bool UP2, bool PRO, bool SWISH, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv3_halo_kernel(vf_igemm_args p) {
    using G = Geo<UP2, PAIR>;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1;
    const int wave_n = wave & 1;
    const int half = lane >> 5;
    const int l31 = lane & 31;

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
    const int y0 = ty * TH, x0 = tx * TW;          // output-tile origin
    const int sy0 = UP2 ? (y0 / 2 - 1) : (y0 - 1); // source-patch origin
    const int sx0 = UP2 ? (x0 / 2 - 1) : (x0 - 1);

    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;   // wave-uniform base
    const int nchunks = p.Cin / CK;
    const int last_stage = nchunks * 9 - 1;

    // ---- staging slots: thread -> SLOTS (pixel, float4 column) pairs of the patch ---------------------
    const int c4 = tid & 7;
    int s_off[G::SLOTS];          // source offset in floats inside the image (clamped to 0 when invalid)
    bool s_ok[G::SLOTS];
    int s_sel[G::SLOTS];
    int s_lds[G::SLOTS];
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
        const int pix = (tid >> 3) + 32 * q;
        const int pixc = pix < G::NPIX ? pix : G::NPIX;     // idle lanes write the dummy pixel
        const int pr = pixc / G::PW, pc0 = pixc - pr * G::PW;
        const int sel = PAIR ? (pc0 >= 10) : 0;
        const int pc = pc0 - 10 * sel;
        s_sel[q] = sel;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool ok = pix < G::NPIX && sy >= 0 && sy < p.Hin && sx >= 0 && sx < p.Win;
        s_ok[q] = ok;
        s_off[q] = ok ? (sel * pair_pix + sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        s_lds[q] = pixc * P_LD + c4 * 4;
    }

    f32x4 preg[G::SLOTS];
    f32x4 pmean, pscale, pbeta, pmean1, pscale1;
    auto patch_load = [&](int chunk) {
        const float* xc = X + chunk * CK;                    // uniform
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
        if (PRO) {
            const float* pm = p.pro_mean + (size_t)img * p.Cin + chunk * CK;
            const float* ps = p.pro_scale + (size_t)img * p.Cin + chunk * CK;
            const float* pb = p.pro_beta + chunk * CK;
            pmean = *reinterpret_cast<const f32x4*>(pm + c4 * 4);
            pscale = *reinterpret_cast<const f32x4*>(ps + c4 * 4);
            pbeta = *reinterpret_cast<const f32x4*>(pb + c4 * 4);
            if (PAIR) {
                pmean1 = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
                pscale1 = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
            }
        }
    };
    auto patch_store_slot = [&](int buf, int q) {
        float* dst = smem + buf * G::BUF;
        f32x4 v = preg[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (PRO) {
                const float mu = (PAIR && s_sel[q]) ? pmean1[e] : pmean[e];
                const float sc = (PAIR && s_sel[q]) ? pscale1[e] : pscale[e];
                t = (t - mu) * sc + pbeta[e];
                if (SWISH) t = vf_swish(t);
            }
            v[e] = s_ok[q] ? t : 0.f;                        // zero padding lives in the conv's input space
        }
        *reinterpret_cast<f32x4*>(dst + s_lds[q]) = v;
    };

    // ---- A-fragment base: this lane's pixel inside each of the wave's two 2x16 m-tiles -----------------
    const int trow = perm_row(l31), tpx = perm_px(l31);
    int a_base[2], a_r[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int a0 = wave_m * 4 + mi * 2 + trow;
        a_r[mi] = a0;
        const int tcol = PAIR ? (tpx >> 3) * 10 + (tpx & 7) : tpx;
        a_base[mi] = (a0 * G::PW + tcol) * P_LD + half * 4;
    }

    // ---- B fragments straight from L2: packed [chunk][tap][nblk][g][half][n][4] -----------------------
    const float* __restrict__ Wb = p.w_packed + (size_t)nblk * (CK * BN);       // uniform
    const size_t tap_stride = (size_t)nb * CK * BN;
    const int b_lane = (half * BN + wave_n * 64 + l31) * 4;                     // per-lane float offset
    f32x4 bc[8], bn[8];
    auto b_load = [&](f32x4 (&dst)[8], int stage) {
        const float* src = Wb + (size_t)stage * tap_stride;                     // uniform
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[g * 2 + j] = *reinterpret_cast<const f32x4*>(src + (g * 2 * BN + j * 32) * 4 + b_lane);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    patch_load(0);
    b_load(bc, 0);
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const float* patch = smem + (chunk & 1) * G::BUF;
        // prefetch the next chunk's patch (clamped: the last chunk re-fetches itself into the idle buffer)
        patch_load(min(chunk + 1, nchunks - 1));
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            b_load(bn, min(chunk * 9 + tap + 1, last_stage));
            int aoff[2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                if (UP2) {
                    const int pr = (a_r[mi] + dy + 1) >> 1, pc = (tpx + dx + 1) >> 1;
                    aoff[mi] = (pr * G::PW + pc) * P_LD + half * 4;
                } else {
                    aoff[mi] = a_base[mi] + (dy * G::PW + dx) * P_LD;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a[2];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(patch + aoff[mi] + g * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][e], bc[g * 2 + j][e], acc[mi][j], 0, 0, 0);
            }
            // transform + park one staging slot of the NEXT chunk's patch per tap (taps 1..SLOTS)
            if (tap >= 1 && tap <= G::SLOTS) patch_store_slot((chunk + 1) & 1, tap - 1);
#pragma unroll
            for (int q = 0; q < 8; ++q) bc[q] = bn[q];
        }
        __syncthreads();
    }

    // ---- epilogue -------------------------------------------------------------------------------------
    float* __restrict__ Out = p.out + (size_t)img * p.Hout * p.Wout * p.ldc;
    const float* __restrict__ Res = p.res ? p.res + (size_t)img * p.Hout * p.Wout * p.ldr : nullptr;
    // batched, branch-free tile epilogue (epilogue.h): tiles are always full here
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * BN + wave_n * 64 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int py = y0 + wave_m * 4 + mi * 2;
            // accumulator row r -> pixel index inside the image (perm_* are compile-time per r, selected by half)
            auto pix = [&](int r) {
                const int i0 = (r & 3) + 8 * (r >> 2);
                const int prow = half ? perm_row(i0 + 4) : perm_row(i0);
                const int ppx = half ? perm_px(i0 + 4) : perm_px(i0);
                if (PAIR) return (ppx >> 3) * (img1 - img) * p.Hout * p.Wout + (py + prow) * p.Wout + (ppx & 7);
                return (py + prow) * p.Wout + x0 + ppx;
            };
            auto oo = [&](int r) { return pix(r) * p.ldc; };
            auto ro = [&](int r) { return pix(r) * p.ldr; };
            if (Res) vf_store_tile<0, true>(acc[mi][j], bias, Out + n, Res + n, oo, ro);
            else vf_store_tile<0, false>(acc[mi][j], bias, Out + n, Res, oo, ro);
        }
    }
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR>
int launch_halo(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, PAIR>;
    const size_t smem = (size_t)2 * G::BUF * sizeof(float);
    auto kern = conv3_halo_kernel<UP2, PRO, SWISH, PAIR>;
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = PAIR ? (long long)((n_img + 1) / 2) * (a.Cout / BN)
                                  : (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PAIR>
int dispatch_pro(const vf_igemm_args& a, hipStream_t s) {
    if (!a.pro_mean) return launch_halo<UP2, false, false, PAIR>(a, s);
    return a.pro_swish ? launch_halo<UP2, true, true, PAIR>(a, s) : launch_halo<UP2, true, false, PAIR>(a, s);
}

}  // namespace

// eligibility + launch; called from vf_igemm_f32 (igemm_f32.hip).  Returns 1 if the shape is not handled here.
int vf_conv3_halo_try(const vf_igemm_args& a, hipStream_t stream, int* status) {
    if (a.mode != VF_MODE_CONV3_S1 && a.mode != VF_MODE_CONV3_UP2) return 1;
    if (a.Cout % BN != 0 || a.Cin % CK != 0) return 1;
    const bool pair = a.mode == VF_MODE_CONV3_S1 && a.Hout == 8 && a.Wout == 8;     // two 8x8 images per tile
    if (!pair && (a.Hout % TH != 0 || a.Wout % TW != 0)) return 1;
    if (a.batch > 1 || a.epilogue != VF_EPI_NONE) return 1;
    if ((long long)a.Hin * a.Win * a.Cin >= (1ll << 31)) return 1;
    *status = pair ? dispatch_pro<false, true>(a, stream)
                   : (a.mode == VF_MODE_CONV3_UP2) ? dispatch_pro<true, false>(a, stream) : dispatch_pro<false, false>(a, stream);
    return 0;
}
