// Dense / 1x1 GEMM on bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 activations in HBM, fp32 accumulate and
// the fp32 bias / exact-erf GELU / residual epilogue — the reduced-precision arm for the transformer and the
// decoder (north star: "transformer logits and decoded pixels within a stated fp tolerance"; the encoder and
// the codebook lookup stay exact fp32 so that token indices remain bit-exact).
//
//   out[m][n] = epi( sum_k bf16(A[m][k]) * bf16(W[k][n]) + bias[n] ) + res[m][n]       (products exact, fp32 sums)
//
// 128x128 tile per 256-thread workgroup (2x2 waves of 64x64), K in stages of 64: A is loaded as fp32 float4,
// rounded to bf16 (v_cvt_pk_bf16_f32, RNE) in registers and parked in LDS with a 144-byte row stride (every
// hardware ds_read_b128 lane group hits 16 distinct 16-byte slots); W is pre-packed fragment-major
// ([k-chunk][n-block][k-step(4)][half(2)][n(128)][8 bf16]) so the B stage is a linear 16 KB copy and a lane's B
// fragment is one ds_read_b128.  Double-buffered LDS (70 KB -> 2 workgroups/CU), one barrier per stage,
// 16 MFMAs (512 matrix-pipe cycles) per wave per stage.
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#ifndef VF_GEMM_BF16_TALL
#define VF_GEMM_BF16_TALL 0     // 1: 128 x 32 wave tile (see gemm_bf16_direct_kernel) — measured SLOWER here (599 -> 507 TF at K = 3072), kept for A/B
#endif

constexpr int CK = 64;
constexpr int BM = 128, BN = 128;
constexpr int A_LDB = 144;                  // bytes per A row in LDS (128 data + 16 pad)
constexpr int A_BYTES = BM * A_LDB;         // 18432
constexpr int B_BYTES = CK * BN * 2;        // 16384

__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                    // [2][A_BYTES]
    unsigned char* Bs = smem_b + 2 * A_BYTES;      // [2][B_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const int bz = blockIdx.z;
    const float* __restrict__ X = p.x + (size_t)bz * p.stride_x;
    const unsigned char* __restrict__ Wp = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)bz * p.stride_w * 2
                                           + (size_t)nblk * B_BYTES;
    const size_t stage_stride = (size_t)nb * B_BYTES;
    const int nstages = p.Cin / CK;

    // A staging: thread -> float4 column (tid & 15), rows (tid >> 4) + 16 q
    const int a_c4 = tid & 15, a_r0 = tid >> 4;
    const float* arow[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int m = mtile * BM + a_r0 + 16 * q;
        m = m < p.M ? m : p.M - 1;                 // clamped: rows past M are never stored
        arow[q] = X + (size_t)m * p.lda + a_c4 * 4;
    }
    f32x4 areg[8];
    f32x4 breg[4];
    auto load_stage = [&](int s) {
#pragma unroll
        for (int q = 0; q < 8; ++q) areg[q] = *reinterpret_cast<const f32x4*>(arow[q] + s * CK);
        const unsigned char* src = Wp + (size_t)s * stage_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) breg[q] = *reinterpret_cast<const f32x4*>(src + (size_t)(tid + 256 * q) * 16);
    };
    auto store_stage = [&](int buf) {
        unsigned char* a_dst = As + buf * A_BYTES;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)areg[q][e];
            *reinterpret_cast<bf16x4*>(a_dst + (a_r0 + 16 * q) * A_LDB + a_c4 * 8) = v;
        }
        unsigned char* b_dst = Bs + buf * B_BYTES;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(b_dst + (size_t)(tid + 256 * q) * 16) = breg[q];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        load_stage(min(s + 1, nstages - 1));
        const unsigned char* a_src = As + (s & 1) * A_BYTES + (wave_m * 64 + l31) * A_LDB + half * 16;
        const unsigned char* b_src = Bs + (s & 1) * B_BYTES + (half * BN + wave_n * 64 + l31) * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(a_src + i * 32 * A_LDB + ks * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(b_src + (ks * 2 * BN + j * 32) * 16);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        store_stage((s + 1) & 1);                 // (the last iteration re-stores the last stage into the idle buffer)
        __syncthreads();
    }

    float* __restrict__ Out = p.out + (size_t)bz * p.stride_out;
    const float* __restrict__ Res = p.res ? p.res + (size_t)bz * p.stride_res : nullptr;
    const bool full = (mtile * BM + BM <= p.M) && (nblk * BN + BN <= p.Cout);
    const long long ldc = p.ldc, ldr = p.ldr;
    const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * BN + wave_n * 64 + j * 32 + l31;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m0 = mtile * BM + wave_m * 64 + i * 32 + 4 * half;
            const int nn = nok ? n : 0;
            float* o = Out + (size_t)(m0 < p.M ? m0 : 0) * ldc + nn;
            const float* rs = Res ? Res + (size_t)(m0 < p.M ? m0 : 0) * ldr + nn : nullptr;
            const int rows_left = nok ? p.M - m0 : 0;
            if (full) {
                auto oo = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldc; };
                auto ro = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldr; };
                if (gelu) {
                    if (Res) vf_store_tile<2, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<2, false>(acc[i][j], bias, o, rs, oo, ro);
                } else {
                    if (Res) vf_store_tile<0, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<0, false>(acc[i][j], bias, o, rs, oo, ro);
                }
            } else if (gelu) {
                if (Res) vf_store_tile_ragged<2, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<2, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            } else {
                if (Res) vf_store_tile_ragged<0, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<0, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            }
        }
    }
}

// Second-generation kernel (K % 128 == 0): the weight fragments no longer pass through LDS — the packed layout already
// is fragment-major, so a lane's B fragment is one 16-byte L2 load — and are fetched one whole 64-deep stage ahead into a
// register ring pinned with sched_barrier (the recipe of conv3_halo_x6.hip).  LDS then carries only the A tile: half the
// LDS traffic of the kernel above, which was LDS-bandwidth-bound (16 ds_read_b128 per 16 MFMAs per wave).
// A16: the activation matrix is already bf16 in HBM ([M][lda] bf16, lda in elements: LayerNorm / GELU / attention outputs written by
// their producers in the format this kernel would round them to anyway) -> a 16 KB stage copied 16 B per thread, no conversion.
// O16: the output is written as bf16 (a value consumed only by another bf16 GEMM; no residual).
template <bool A16, bool O16>
__global__ __launch_bounds__(256, 2) void gemm_bf16_direct_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][A_BYTES]
    unsigned char* As = smem_b;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile: TALL = 128 rows x 32 columns (4 x 1 MFMA tiles): a stage needs 4 weight fragments through the L1 -> VGPR return path
    // instead of 8 and 16 activation fragments from LDS instead of 8.  Unlike the x3h convolution (+3 %) this loses 10-15 % here:
    // one MFMA per fragment pair leaves the doubled LDS reads exposed.  Default off.
    constexpr bool TALL = VF_GEMM_BF16_TALL != 0;
    constexpr int MI = TALL ? 4 : 2, NJ = TALL ? 1 : 2;
    const int wave_m = TALL ? 0 : wave >> 1, wave_n = TALL ? wave : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const float* __restrict__ X = p.x;
    const unsigned char* __restrict__ Wp = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * B_BYTES;
    const size_t stage_stride = (size_t)nb * B_BYTES;
    const int nstages = p.Cin / CK;

    // fp32 A: thread -> float4 column (tid & 15), rows (tid >> 4) + 16 q (8 loads); bf16 A: 16-byte piece (tid & 7), rows (tid >> 3) + 32 q
    const int a_c4 = A16 ? (tid & 7) : (tid & 15), a_r0 = A16 ? (tid >> 3) : (tid >> 4);
    constexpr int NQ = A16 ? 4 : 8, RSTEP = A16 ? 32 : 16;
    const float* arow[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        int m = mtile * BM + a_r0 + RSTEP * q;
        m = m < p.M ? m : p.M - 1;
        arow[q] = A16 ? reinterpret_cast<const float*>(reinterpret_cast<const __bf16*>(X) + (size_t)m * p.lda + a_c4 * 8)
                      : X + (size_t)m * p.lda + a_c4 * 4;
    }
    f32x4 areg[NQ];
    auto a_fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            areg[q] = A16 ? *reinterpret_cast<const f32x4*>(reinterpret_cast<const __bf16*>(arow[q]) + s * CK)
                          : *reinterpret_cast<const f32x4*>(arow[q] + s * CK);
    };
    auto a_park = [&](int buf, int q) {
        if (A16) {
            if (q < NQ) *reinterpret_cast<f32x4*>(As + buf * A_BYTES + (a_r0 + RSTEP * q) * A_LDB + a_c4 * 16) = areg[q < NQ ? q : 0];
        } else {
            bf16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (__bf16)areg[q < NQ ? q : 0][e];
            *reinterpret_cast<bf16x4*>(As + buf * A_BYTES + (a_r0 + RSTEP * q) * A_LDB + a_c4 * 8) = v;
        }
    };
    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    bf16x8 bring[2][4][NJ];            // [stage parity][ks][j]
    auto b_load = [&](bf16x8 (&dst)[4][NJ], int s) {
        const unsigned char* src = Wp + (size_t)min(s, nstages - 1) * stage_stride + b_lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[ks][j] = *reinterpret_cast<const bf16x8*>(src + (ks * 2 * BN + j * 32) * 16);
    };

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    a_fetch(0);
    b_load(bring[0], 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) a_park(0, q);
    __syncthreads();

    auto stage_body = [&](int s, bf16x8 (&bcur)[4][NJ], bf16x8 (&bnext)[4][NJ]) {
        const unsigned char* a_src = As + (s & 1) * A_BYTES + (wave_m * 64 + l31) * A_LDB + half * 16;
        a_fetch(min(s + 1, nstages - 1));
        b_load(bnext, s + 1);
        bf16x8 a[4][MI];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(a_src + i * 32 * A_LDB + ks * 32);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], bcur[ks][j], acc[i][j], 0, 0, 0);
            if (A16) a_park((s + 1) & 1, ks);
            else { a_park((s + 1) & 1, ks * 2); a_park((s + 1) & 1, ks * 2 + 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int s = 0; s < nstages; s += 2) {          // nstages is even (Cin % 128 == 0)
        stage_body(s, bring[0], bring[1]);
        stage_body(s + 1, bring[1], bring[0]);
    }

    if (O16) {
        // bf16 store (bias + optional GELU; no residual).  A lane holds one column of 16 rows; neighbouring lanes hold neighbouring
        // columns, so each pair of lanes swaps half of its rows (one DPP move per row) and every lane stores 8 x 4 bytes — two
        // adjacent columns of one row — instead of 16 x 2 bytes.  Ragged edges fall back to single elements.
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bool gelu16 = p.epilogue == VF_EPI_GELU_ERF;
        const bool gbwd = p.epilogue == VF_EPI_GELU_BWD;             // out = acc * gelu'(u), u = p.res[m][n] (the saved pre-activation)
        __bf16* __restrict__ O = reinterpret_cast<__bf16*>(p.out);
        const bool pairs = (p.ldc & 1) == 0 && (p.Cout & 1) == 0;
        const int odd = l31 & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
            const float bias = (n < p.Cout && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m0 = mtile * BM + wave_m * 64 + i * 32 + 4 * half;
                float t[16];
                if (gbwd) {
                    // gelu_bwd_bf16out_kernel's expression (vf_gelu_grad_fast: explicitly rounded) on the value the un-fused path would have stored
                    // as fp32: same bits
                    float uu[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (r & 3) + 8 * (r >> 2);
                        uu[r] = p.res[(size_t)(m < p.M ? m : 0) * p.ldr + (n < p.Cout ? n : 0)];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = __fmul_rn(__fadd_rn(acc[i][j][r], bias), vf_gelu_grad_fast(uu[r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        t[r] = acc[i][j][r] + bias;
                        if (gelu16) t[r] = vf_gelu_erf_fast(t[r]);
                    }
                }
                if (pairs) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        // even lane: row r of columns (n, n+1); odd lane: row r+1 of columns (n-1, n)
                        const float give = odd ? t[r] : t[r + 1];
                        const float got = vf_lane_xor1(give);
                        const int m = m0 + ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                        bf16x2_t v;
                        v[0] = (__bf16)(odd ? got : t[r]);
                        v[1] = (__bf16)(odd ? t[r + 1] : got);
                        const int n0 = n - odd;
                        if (m < p.M && n0 < p.Cout) *reinterpret_cast<bf16x2_t*>(O + (size_t)m * p.ldc + n0) = v;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (r & 3) + 8 * (r >> 2);
                        if (m < p.M && n < p.Cout) O[(size_t)m * p.ldc + n] = (__bf16)t[r];
                    }
                }
            }
        }
        return;
    }
    float* __restrict__ Out = p.out;
    const float* __restrict__ Res = p.res;
    const bool full = (mtile * BM + BM <= p.M) && (nblk * BN + BN <= p.Cout);
    const long long ldc = p.ldc, ldr = p.ldr;
    const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = nblk * BN + wave_n * (32 * NJ) + j * 32 + l31;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m0 = mtile * BM + wave_m * 64 + i * 32 + 4 * half;
            const int nn = nok ? n : 0;
            float* o = Out + (size_t)(m0 < p.M ? m0 : 0) * ldc + nn;
            const float* rs = Res ? Res + (size_t)(m0 < p.M ? m0 : 0) * ldr + nn : nullptr;
            const int rows_left = nok ? p.M - m0 : 0;
            if (full) {
                auto oo = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldc; };
                auto ro = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldr; };
                if (gelu) {
                    if (Res) vf_store_tile<2, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<2, false>(acc[i][j], bias, o, rs, oo, ro);
                } else {
                    if (Res) vf_store_tile<0, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<0, false>(acc[i][j], bias, o, rs, oo, ro);
                }
            } else if (gelu) {
                if (Res) vf_store_tile_ragged<2, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<2, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            } else {
                if (Res) vf_store_tile_ragged<0, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<0, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            }
        }
    }
}

// 256x128 workgroup tile (wave tile 128 x 64 = 4x2 MFMA tiles, 128 accumulator registers): every B fragment streamed from L2 feeds
// four MFMAs and every A fragment read from LDS feeds two, i.e. 0.75 operand fragments per MFMA instead of 1.0, and the fp32 A tile
// (global float4 -> bf16 -> LDS) is amortised over twice the columns' worth of work per wave: 1.75 memory instructions per MFMA
// instead of 3.0 for the 128x128 kernels above, which is what bounds them (they sit at 420-470 TFLOP/s with the matrix pipe < 25 %
// busy).  K in stages of 32 (two k-steps, 16 MFMAs per wave per barrier), double-buffered LDS (2 x 20 KB), B one stage ahead in a
// 2-deep register ring.  Needs K % 64 == 0 (even stage count) and takes the same bf16 packing.
constexpr int WM = 256;
constexpr int W_CK = 32;
constexpr int W_LDB = 80;                   // bytes per A row in LDS (64 data + 16 pad: 5 x 16 B, conflict-free ds_read_b128)
constexpr int W_ABYTES = WM * W_LDB;        // 20480

template <int EPI, bool HAS_RES, bool FULL>
__device__ __forceinline__ void w256_store(const vf_igemm_args& p, const f32x16 (&acc)[4][2], int mtile, int nblk, int wave_m, int wave_n,
                                           int half, int l31) {
    const long long ldc = p.ldc, ldr = p.ldr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * 128 + wave_n * 64 + j * 32 + l31;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        const int nn = nok ? n : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m0 = mtile * 256 + wave_m * 128 + i * 32 + 4 * half;
            float* o = p.out + (size_t)(m0 < p.M ? m0 : 0) * ldc + nn;
            const float* rs = HAS_RES ? p.res + (size_t)(m0 < p.M ? m0 : 0) * ldr + nn : nullptr;
            if (FULL) {
                auto oo = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldc; };
                auto ro = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldr; };
                vf_store_tile<EPI, HAS_RES>(acc[i][j], bias, o, rs, oo, ro);
            } else {
                vf_store_tile_ragged<EPI, HAS_RES>(acc[i][j], bias, o, rs, ldc, ldr, nok ? p.M - m0 : 0);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void gemm_bf16_w256_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* As = smem_b;                    // [2][W_ABYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const float* __restrict__ X = p.x;
    const unsigned char* __restrict__ Wp = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * B_BYTES;
    const size_t chunk_stride = (size_t)nb * B_BYTES;          // one 64-deep packed chunk = two stages
    const int nstages = p.Cin / W_CK;

    const int a_c4 = tid & 7, a_r0 = tid >> 3;                 // float4 column of the 32-wide stage, rows a_r0 + 32 q
    const float* arow[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        int m = mtile * WM + a_r0 + 32 * q;
        m = m < p.M ? m : p.M - 1;
        arow[q] = X + (size_t)m * p.lda + a_c4 * 4;
    }
    f32x4 areg[8];
    auto a_fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < 8; ++q) areg[q] = *reinterpret_cast<const f32x4*>(arow[q] + s * W_CK);
    };
    auto a_park = [&](int buf, int q) {
        bf16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (__bf16)areg[q][e];
        *reinterpret_cast<bf16x4*>(As + buf * W_ABYTES + (a_r0 + 32 * q) * W_LDB + a_c4 * 8) = v;
    };
    const unsigned b_lane = (unsigned)((half * BN + wave_n * 64 + l31) * 16);
    bf16x8 bring[2][2][2];             // [stage parity][ks][j]
    auto b_load = [&](bf16x8 (&dst)[2][2], int s) {
        s = min(s, nstages - 1);
        const unsigned char* src = Wp + (size_t)(s >> 1) * chunk_stride + (unsigned)((s & 1) * 2 * (2 * BN * 16)) + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[ks][j] = *reinterpret_cast<const bf16x8*>(src + (ks * 2 * BN + j * 32) * 16);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    a_fetch(0);
    b_load(bring[0], 0);
#pragma unroll
    for (int q = 0; q < 8; ++q) a_park(0, q);
    __syncthreads();

    auto stage_body = [&](int s, bf16x8 (&bcur)[2][2], bf16x8 (&bnext)[2][2]) {
        const unsigned char* a_src = As + (s & 1) * W_ABYTES + (wave_m * 128 + l31) * W_LDB + half * 16;
        a_fetch(min(s + 1, nstages - 1));
        b_load(bnext, s + 1);
        bf16x8 a[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) a[ks][i] = *reinterpret_cast<const bf16x8*>(a_src + i * 32 * W_LDB + ks * 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][i], bcur[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) a_park((s + 1) & 1, ks * 4 + q);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int s = 0; s < nstages; s += 2) {          // nstages is even (Cin % 64 == 0)
        stage_body(s, bring[0], bring[1]);
        stage_body(s + 1, bring[1], bring[0]);
    }

    const bool full = (mtile * WM + WM <= p.M) && (nblk * BN + BN <= p.Cout);
    const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
    // one fully unrolled instantiation per (epilogue, residual, full/ragged): the accumulators stay in registers
    if (full) {
        if (gelu) { if (p.res) w256_store<2, true, true>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); else w256_store<2, false, true>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); }
        else { if (p.res) w256_store<0, true, true>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); else w256_store<0, false, true>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); }
    } else {
        if (gelu) { if (p.res) w256_store<2, true, false>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); else w256_store<2, false, false>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); }
        else { if (p.res) w256_store<0, true, false>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); else w256_store<0, false, false>(p, acc, mtile, nblk, wave_m, wave_n, half, l31); }
    }
}

// pack fp32 [K][N] (strided) -> bf16 fragment-major [K/64][nb][ks(4)][half(2)][n(128)][8]
__global__ void pack_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int K, int N, long long sk,
                                 long long sn, int nb, int nchunks, long long src_bstride, long long dst_bstride) {
    const long long total = (long long)nchunks * nb * CK * BN;
    const int bz = blockIdx.y;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        long long t = idx >> 3;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int ks = (int)((t >> 1) & 3);
        t >>= 3;
        const int nblk = (int)(t % nb);
        const int chunk = (int)(t / nb);
        const int k = chunk * CK + ks * 16 + half * 8 + e;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (k < K && n < N) v = src[(size_t)bz * src_bstride + k * sk + n * sn];
        dst[(size_t)bz * dst_bstride + idx] = (__bf16)v;
    }
}

// the same packing for MANY weights in one launch (blockIdx.y = descriptor): the training step re-packs 2 x 48 layer weights after every
// optimizer step, 96 launches of a few microseconds each
__global__ void pack_bf16_multi_kernel(const vf_pack_desc* __restrict__ descs) {
    const vf_pack_desc d = descs[blockIdx.y];
    const int nb = (d.N + BN - 1) / BN, nchunks = (d.K + CK - 1) / CK;
    // one thread = the 8 consecutive k of one packed 16-byte group (round 6: the element-per-thread form read 8 different k rows from 8 neighbouring
    // threads — 32-byte segments, 1.6 TB/s for the step's 88 M weights): neighbouring threads are neighbouring n, so for the [K][N] source (sn == 1)
    // each of the 8 loads is one contiguous 256-byte run per wave; for the [N][K] source (sk == 1) a thread's 8 values are two 16-byte loads
    const long long groups = (long long)nchunks * nb * (CK * BN / 8);
    __bf16* __restrict__ dst = reinterpret_cast<__bf16*>(d.dst);
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    for (long long gi = blockIdx.x * (long long)blockDim.x + threadIdx.x; gi < groups; gi += (long long)gridDim.x * blockDim.x) {
        long long t = gi;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int ks = (int)((t >> 1) & 3);
        t >>= 3;
        const int nblk = (int)(t % nb);
        const int chunk = (int)(t / nb);
        const int k0 = chunk * CK + ks * 16 + half * 8;
        const int n = nblk * BN + nl;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (n < d.N) {
            const float* sp = d.src + (long long)k0 * d.sk + (long long)n * d.sn;
            if (d.sk == 1 && k0 + 8 <= d.K && ((reinterpret_cast<uintptr_t>(sp) & 15) == 0)) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (k0 + e < d.K) v[e] = sp[(long long)e * d.sk];
            }
        }
        bf16x8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
        *reinterpret_cast<bf16x8_t*>(dst + gi * 8) = o;
    }
}

}  // namespace

int vf_gemm_bf16_g256_launch(const vf_igemm_args& a, hipStream_t stream);      // gemm_bf16_g256.hip: 256 x 256 tile, LDS-DMA operands

extern "C" {

size_t vf_gemm_bf16_packed_elems(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (size_t)((K + CK - 1) / CK) * ((N + BN - 1) / BN) * CK * BN;
}

int vf_gemm_bf16_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, int batch, int64_t src_bstride,
                      void* stream) {
    if (!src || !dst || K <= 0 || N <= 0 || batch < 1) return VF_ERR_BAD_ARG;
    const int nb = (N + BN - 1) / BN, nchunks = (K + CK - 1) / CK;
    const long long total = (long long)nchunks * nb * CK * BN;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(blocks, batch), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, K, N,
                       (long long)sk, (long long)sn, nb, nchunks, (long long)src_bstride, total);
    return vf_last_status();
}

int vf_gemm_bf16_pack_multi(const vf_pack_desc* descs_device, int n, void* stream) {
    if (!descs_device || n <= 0) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(pack_bf16_multi_kernel, dim3(128, (unsigned)n), dim3(256), 0, (hipStream_t)stream, descs_device);
    return vf_last_status();
}

int vf_gemm_bf16(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0 || a.Cin <= 0 || a.Cout <= 0) return VF_ERR_BAD_ARG;
    if (a.gn_part) return VF_ERR_UNSUPPORTED;        // fused GroupNorm statistics: halo-tile kernels only
    if (a.mode != VF_MODE_GEMM || a.pro_mean) return VF_ERR_UNSUPPORTED;
    if (a.Cin % CK != 0) return VF_ERR_UNSUPPORTED;
    if (a.epilogue == VF_EPI_GELU_BWD && !(a.reserved0 & 2)) return VF_ERR_UNSUPPORTED;      // (bf16-output form only)
    if ((a.epilogue == VF_EPI_GELU_DUAL) != (a.out_aux != nullptr)) return VF_ERR_BAD_ARG;
    if (a.lda < a.Cin || (a.lda & 3) || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
    if (a.epilogue == VF_EPI_GELU_DUAL) {             // the 256-tile kernel's fp32 epilogue only: no other kernel of this file writes out_aux
        if (!(a.reserved0 & 1) || a.batch > 1 || a.Cin % (2 * CK) != 0) return VF_ERR_UNSUPPORTED;      // (bf16 activations in; out fp32 or bf16)
        const int rc = vf_gemm_bf16_g256_launch(a, (hipStream_t)stream);
        return rc;                                    // (VF_ERR_UNSUPPORTED for shapes that kernel does not tile: the caller runs the two passes)
    }
    if (a.drop_rate != 0.f) {                         // fused output dropout: the 256-tile kernel's fp32 epilogue only
        if (!(a.reserved0 & 1) || (a.reserved0 & 2) || a.batch > 1 || a.Cin % (2 * CK) != 0) return VF_ERR_UNSUPPORTED;
        return vf_gemm_bf16_g256_launch(a, (hipStream_t)stream);      // (VF_ERR_UNSUPPORTED for shapes it does not tile: GEMM, then vf_dropout_add_f32)
    }
    const size_t smem = (size_t)2 * (A_BYTES + B_BYTES);
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const int nb = (a.Cout + BN - 1) / BN, mt = (a.M + BM - 1) / BM;
#ifndef VF_GEMM_BF16_DIRECT
#define VF_GEMM_BF16_DIRECT 1
#endif
#ifndef VF_GEMM_BF16_W256
#define VF_GEMM_BF16_W256 0       // A/B on MI355X: 433-450 TF vs 440-458 for the 128x128 direct kernel at M = 65536 (ties) -> opt-in
#endif
    if (VF_GEMM_BF16_W256 && a.batch <= 1 && a.M >= 4 * WM) {       // Cin % 64 == 0 holds (checked above)
        const int mt2 = (a.M + WM - 1) / WM;
        hipLaunchKernelGGL(gemm_bf16_w256_kernel, dim3((unsigned)(mt2 * nb)), dim3(256), (size_t)2 * W_ABYTES, (hipStream_t)stream, a);
        return vf_last_status();
    }
    const bool a16 = a.reserved0 & 1, o16 = a.reserved0 & 2;      // bf16 activations in / out (see gemm_bf16_direct_kernel)
    if (a16 || o16) {
        if (a.Cin % (2 * CK) != 0 || a.batch > 1) return VF_ERR_UNSUPPORTED;
        if ((a16 && (a.lda & 7)) || (o16 && a.res && a.epilogue != VF_EPI_GELU_BWD)) return VF_ERR_BAD_ARG;
        if (a.epilogue == VF_EPI_GELU_BWD && !(o16 && a.res && a.ldr >= a.Cout)) return VF_ERR_BAD_ARG;
        // large token matrices with 256-aligned widths: the 256 x 256 LDS-DMA kernel (bit-identical results; vf_select(VF_SEL_GEMM_G256, 0)
        // keeps the 128 x 128 kernel for A/B runs)
        if (a.reserved0 & 4) {                                  // a bf16 pre-activation behind `res` (VF_EPI_GELU_BWD): the 256-tile kernel only
            if (a.epilogue != VF_EPI_GELU_BWD) return VF_ERR_BAD_ARG;
            return vf_gemm_bf16_g256_launch(a, (hipStream_t)stream);
        }
        if (vf_selected(VF_SEL_GEMM_G256)) {                    // (0: the 128-tile kernel — the parity test flips it in-process)
            const int rc = vf_gemm_bf16_g256_launch(a, (hipStream_t)stream);
            if (rc != VF_ERR_UNSUPPORTED) return rc;
        }
        const dim3 g((unsigned)(mt * nb));
        hipStream_t st = (hipStream_t)stream;
        if (a16 && o16) hipLaunchKernelGGL((gemm_bf16_direct_kernel<true, true>), g, dim3(256), (size_t)2 * A_BYTES, st, a);
        else if (a16) hipLaunchKernelGGL((gemm_bf16_direct_kernel<true, false>), g, dim3(256), (size_t)2 * A_BYTES, st, a);
        else hipLaunchKernelGGL((gemm_bf16_direct_kernel<false, true>), g, dim3(256), (size_t)2 * A_BYTES, st, a);
        return vf_last_status();
    }
    if (VF_GEMM_BF16_DIRECT && a.Cin % (2 * CK) == 0 && a.batch <= 1) {
        hipLaunchKernelGGL((gemm_bf16_direct_kernel<false, false>), dim3((unsigned)(mt * nb)), dim3(256), (size_t)2 * A_BYTES, (hipStream_t)stream, a);
        return vf_last_status();
    }
    dim3 grid((unsigned)(mt * nb), 1, (unsigned)(a.batch > 0 ? a.batch : 1));
    hipLaunchKernelGGL(gemm_bf16_kernel, grid, dim3(256), smem, (hipStream_t)stream, a);
    return vf_last_status();
}

}  // extern "C"
