// Shared MFMA-tile epilogue for the f32 GEMM-family kernels (gfx950).
//
// A 32x32 accumulator tile in the C layout holds, per lane, 16 rows (r -> (r&3) + 8*(r>>2) [+4*half, folded
// into the base]) of ONE column.  The first version of the kernels stored element by element inside
// `if (m < M)` / `if (res)` branches: every element became its own basic block ending in
// `s_waitcnt vmcnt(0)`, and since vmcnt also counts stores on CDNA4 each of the 64 stores per lane waited
// for the previous store's memory round trip — ~40 us of serialized latency per workgroup, as long as
// the whole MFMA loop (found with tools/mfma_peak.hip + ISA inspection).  Here a tile is handled as
// ONE basic block: all 16 residual loads are issued back to back, one wait, 16 adds, 16 back-to-back
// stores.  Row offsets come from a functor so the halo conv (permuted pixel rows) shares the code.
#pragma once
#include "vf_common.h"

// off(r) -> element offset of accumulator row r relative to `out` / `res` (already positioned at the lane's
// column and at the +4*half row).  FULL: every row of the tile is in range (wave-uniform fact).
template <int EPI, bool HAS_RES, typename OffOut, typename OffRes>
__device__ __forceinline__ void vf_store_tile(const f32x16& acc, float bias, float* __restrict__ out,
                                              const float* __restrict__ res, OffOut off_out, OffRes off_res) {
    float v[16];
    if (HAS_RES) {
        float rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rr[r] = res[off_res(r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = acc[r] + bias;
            if (EPI == 1) t = vf_gelu_erf(t);
            if (EPI == 2) t = vf_gelu_erf_fast(t);
            v[r] = t + rr[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float t = acc[r] + bias;
            if (EPI == 1) t = vf_gelu_erf(t);
            if (EPI == 2) t = vf_gelu_erf_fast(t);
            v[r] = t;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[off_out(r)] = v[r];
}

// Ragged tile (some rows >= M): same batching, rows clamped for the loads and predicated for the store via
// a select of the destination into a per-lane dummy (the lane's own valid row 0 is rewritten with its own
// value — no branch, no out-of-bounds access).  `nrows` = number of valid rows counted from this lane's row 0.
template <int EPI, bool HAS_RES>
__device__ __forceinline__ void vf_store_tile_ragged(const f32x16& acc, float bias, float* __restrict__ out,
                                                     const float* __restrict__ res, long long ldc, long long ldr,
                                                     int rows_left /* M - (row of r=0 incl. half) */) {
    if (rows_left <= 0) return;      // whole lane out of range (uniform per half-wave; rare tail tile only)
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        float t = acc[r] + bias;
        if (EPI == 1) t = vf_gelu_erf(t);
        if (EPI == 2) t = vf_gelu_erf_fast(t);
        if (HAS_RES) t += res[(long long)(row < rows_left ? row : 0) * ldr];
        v[r] = t;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        if (row < rows_left) out[(long long)row * ldc] = v[r];
    }
}

// ---- fused GroupNorm statistics of the stored tile -------------------------------------------------------------
// Same tile store as vf_store_tile<0, HAS_RES>, additionally accumulating the sum and the sum of squares of the values it
// stores into s[sel(r)] / q[sel(r)] (sel(r) in {0, 1}: which of the two images of a pair tile row r belongs to; always 0
// for ordinary tiles).
// The sums are carried in the accumulator type S of the caller: double (halo_common.h) makes a tile's partial sums independent of the
// order in which a lane and the shuffle tree meet the values — a pair tile's left and right half group an image's pixels differently.
template <bool HAS_RES, typename OffOut, typename OffRes, typename Sel, typename S>
__device__ __forceinline__ void vf_store_tile_stats(const f32x16& acc, float bias, float* __restrict__ out,
                                                    const float* __restrict__ res, OffOut off_out, OffRes off_res, Sel sel,
                                                    S (&s)[2], S (&q)[2]) {
    float v[16];
    if (HAS_RES) {
        float rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rr[r] = res[off_res(r)];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bias + rr[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r] + bias;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = sel(r);
        s[k] += (S)v[r];
        q[k] += (S)v[r] * (S)v[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[off_out(r)] = v[r];
}

// The same two stores with the residual values already in registers (halo_common.h loads them for ALL tiles of the wave up front: the
// per-tile form pays one memory round trip per tile, four in a row for the tall halo tile).
template <typename OffOut>
__device__ __forceinline__ void vf_store_tile_pre(const f32x16& acc, float bias, float* __restrict__ out, const float (&rr)[16], OffOut off_out) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + bias + rr[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) out[off_out(r)] = v[r];
}
template <typename OffOut, typename Sel, typename S>
__device__ __forceinline__ void vf_store_tile_stats_pre(const f32x16& acc, float bias, float* __restrict__ out, const float (&rr)[16], OffOut off_out,
                                                        Sel sel, S (&s)[2], S (&q)[2]) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + bias + rr[r];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k = sel(r);
        s[k] += (S)v[r];
        q[k] += (S)v[r] * (S)v[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[off_out(r)] = v[r];
}

// Reduce a lane's per-column (s, q) over the cg adjacent columns of a GroupNorm group (cg = C/32 in {4, 8, 16, 32},
// columns = the 32 lanes of a half-wave) and over the two half-waves (the other 16 rows of the tile); the first lane of each
// group then owns the group's partial of this wave's 64 x 32 sub-tile.  Fixed shuffle tree -> deterministic.
template <typename S>
__device__ __forceinline__ void vf_gn_group_reduce(S& s, S& q, int cg) {
    s += __shfl_xor(s, 32, 64);
    q += __shfl_xor(q, 32, 64);
    for (int o = 1; o < cg; o <<= 1) {
        s += __shfl_xor(s, o, 64);
        q += __shfl_xor(q, o, 64);
    }
}
