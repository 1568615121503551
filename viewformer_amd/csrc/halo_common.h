// Shared by the halo-tile convolution kernels (conv3_halo_x6.hip, conv3_halo_bf16.hip): the MFMA-row -> pixel permutation and
// the epilogue (bias, residual, store, optional fused GroupNorm partial statistics of the stored values).
#pragma once
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

#ifndef VF_HALO_RES_PRELOAD
#define VF_HALO_RES_PRELOAD 1   // residual values of all tiles fetched before the first store
#endif
#ifndef VF_GN_STATS_F64
#define VF_GN_STATS_F64 1      // fused GroupNorm partial sums in fp64 (position-independent; 0: fp32, the round-1 form)
#endif
#if VF_GN_STATS_F64
typedef double vf_gn_acc_t;
#else
typedef float vf_gn_acc_t;
#endif

// MFMA row i (0..31) -> (tile row 0/1, pixel 0..15) such that each hardware ds_read_b128 16-lane group
// ({0-3,12-15,20-27} / {4-11,16-19,28-31}) covers 16 CONSECUTIVE patch pixels
__host__ __device__ constexpr int vf_perm_row(int i) { return (i < 4) ? 0 : (i < 12) ? 1 : (i < 16) ? 0 : (i < 20) ? 1 : (i < 28) ? 0 : 1; }
__host__ __device__ constexpr int vf_perm_px(int i) {
    return (i < 4) ? i : (i < 12) ? i - 4 : (i < 16) ? i - 8 : (i < 20) ? i - 8 : (i < 28) ? i - 12 : i - 16;
}

// partial-statistics slots per image: one per (8x16 tile, wave row-half); an 8x8 map (pair tiles) has 2
static inline int vf_halo_gn_slots(int Hout, int Wout) {
    if (Hout == 8 && Wout == 8) return 2;
    if (Hout <= 0 || Wout <= 0 || Hout % 8 || Wout % 16) return 0;
    return (Hout / 8) * (Wout / 16) * 2;
}

// host-side check of the fused-statistics request (0 = ok)
static inline int vf_halo_gn_check(const vf_igemm_args& a) {
    if (!a.gn_part) return VF_OK;
    const int cg = a.Cout / 32;
    if (a.Cout % 32 || !(cg == 4 || cg == 8 || cg == 16 || cg == 32)) return VF_ERR_UNSUPPORTED;
    if (a.gn_slots != vf_halo_gn_slots(a.Hout, a.Wout)) return VF_ERR_BAD_ARG;
    return VF_OK;
}

// One workgroup's epilogue.  A wave owns MI x NJ accumulator tiles: acc[mi][j] = tile rows [wave_m * 4 + 2 mi, +2) x 16 px, output
// channels [wave_n * 32 NJ + 32 j, +32).  (MI, NJ) = (2, 2): the 2 x 2 wave grid of the original kernels (wave_m in {0, 1} = rows
// 0-3 / 4-7); (4, 1): the "tall" form, every wave covers all 8 rows and 32 channels (wave_m = 0, wave_n = wave).  The fused GroupNorm
// partial statistics keep their layout either way: slot tile_slot + (row >> 2) holds the sums over rows 4 (row >> 2) .. +3.
// PAIR: the tile is two 8x8 images (img, img1) side by side.  A lane then sums a different subset of an image's pixels in the left and in
// the right half; with fp32 sums that made an image's GroupNorm statistics depend, in the last bit, on its position in the batch (the batch
// invariance test held on its data until an unrelated epilogue change altered the compiler's fma contraction — tools/debug_invariance.py
// found the layer).  The partial sums are therefore carried in fp64 (vf_gn_acc_t) and rounded once.
template <bool PAIR, int MI, int NJ, bool PRELOAD = (VF_HALO_RES_PRELOAD != 0)>
__device__ __forceinline__ void vf_halo_epilogue_t(const vf_igemm_args& p, const f32x16 (&acc)[MI][NJ], int img, int img1, int y0,
                                                   int x0, int tile_slot, int nblk, int wave_m, int wave_n, int half, int l31) {
    constexpr int BN = 128;
    float* __restrict__ Out = p.out + (size_t)img * p.Hout * p.Wout * p.ldc;
    const float* __restrict__ Res = p.res ? p.res + (size_t)img * p.Hout * p.Wout * p.ldr : nullptr;
    const bool stats = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;
    auto ppx = [&](int r) { const int i0 = (r & 3) + 8 * (r >> 2); return half ? vf_perm_px(i0 + 4) : vf_perm_px(i0); };
    auto pix_of = [&](int mi, int r) {
        const int py = y0 + wave_m * 4 + mi * 2;
        const int i0 = (r & 3) + 8 * (r >> 2);
        const int prow = half ? vf_perm_row(i0 + 4) : vf_perm_row(i0);
        if (PAIR) return (ppx(r) >> 3) * (img1 - img) * p.Hout * p.Wout + (py + prow) * p.Wout + (ppx(r) & 7);
        return (py + prow) * p.Wout + x0 + ppx(r);
    };
    // residual values of ALL the wave's tiles first (PRELOAD): one memory round trip instead of one per tile
    float rr[PRELOAD ? NJ : 1][PRELOAD ? MI : 1][16];
    if (PRELOAD && Res) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) rr[PRELOAD ? j : 0][PRELOAD ? mi : 0][r] = Res[n + pix_of(mi, r) * p.ldr];
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int sl = 0; sl < MI / 2; ++sl) {                       // one statistics slot = 4 tile rows = 2 accumulator tiles
            vf_gn_acc_t s[2] = {0, 0}, q[2] = {0, 0};
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int mi = sl * 2 + mm;
                auto oo = [&](int r) { return pix_of(mi, r) * p.ldc; };
                auto ro = [&](int r) { return pix_of(mi, r) * p.ldr; };
                auto sel = [&](int r) { return PAIR ? (ppx(r) >> 3) : 0; };
                if (PRELOAD && Res) {
                    if (stats) vf_store_tile_stats_pre(acc[mi][j], bias, Out + n, rr[PRELOAD ? j : 0][PRELOAD ? mi : 0], oo, sel, s, q);
                    else vf_store_tile_pre(acc[mi][j], bias, Out + n, rr[PRELOAD ? j : 0][PRELOAD ? mi : 0], oo);
                } else if (stats) {
                    if (Res) vf_store_tile_stats<true>(acc[mi][j], bias, Out + n, Res + n, oo, ro, sel, s, q);
                    else vf_store_tile_stats<false>(acc[mi][j], bias, Out + n, Res, oo, ro, sel, s, q);
                } else {
                    if (Res) vf_store_tile<0, true>(acc[mi][j], bias, Out + n, Res + n, oo, ro);
                    else vf_store_tile<0, false>(acc[mi][j], bias, Out + n, Res, oo, ro);
                }
            }
            if (stats) {
#pragma unroll
                for (int k = 0; k < (PAIR ? 2 : 1); ++k) {
                    vf_gn_acc_t ss = s[k], qq = q[k];
                    vf_gn_group_reduce(ss, qq, cg);
                    // an odd image count duplicates the last image into the second half of its pair tile: that half is not a new image
                    if (half == 0 && (l31 & (cg - 1)) == 0 && (k == 0 || img1 != img)) {
                        const int im = k ? img1 : img;
                        float* dst = p.gn_part + ((((size_t)im * p.gn_slots) + tile_slot + wave_m + sl) * 32 + n / cg) * 2;
                        dst[0] = (float)ss;
                        dst[1] = (float)qq;
                    }
                }
            }
        }
    }
}

// the original 2 x 2 wave grid (conv3_halo_x6 / _bf16 / _f32, the stride-2 kernels)
template <bool PAIR>
__device__ __forceinline__ void vf_halo_epilogue(const vf_igemm_args& p, const f32x16 (&acc)[2][2], int img, int img1, int y0,
                                                 int x0, int tile_slot, int nblk, int wave_m, int wave_n, int half, int l31) {
    vf_halo_epilogue_t<PAIR, 2, 2>(p, acc, img, img1, y0, x0, tile_slot, nblk, wave_m, wave_n, half, l31);
}
