// fp32-EQUIVALENT dense / 1x1 GEMM with half the matrix instructions of gemm_x6.hip ("x3h"), gfx950 — the dense sibling of
// conv3_halo_x3h.hip (see its header for the arithmetic: two fp16 pieces per operand, low piece carried at 2^11, three products, the
// cross terms in their own fp32 accumulator, weights pre-scaled by a power of two at pack time; range condition |x| in [~6e-5, 65504)).
//   out[m][n] = epi( sum_k A[m][k] * W[k][n] + bias[n] ) + res[m][n]
// Same tile / pipeline as gemm_x6_kernel: 128x128 tile per 256-thread workgroup, K in chunks of 32, the fp32 A tile loaded as float4,
// optionally pushed through the GroupNorm-apply(+swish) prologue, split ONCE and parked in LDS as [row][plane h|l'][32 k] f16 with a
// 144-byte row stride; two weight planes ([k-chunk][n-block][k-step(2)][plane(2)][half(2)][n(128)][8]) streamed L2 -> VGPR one chunk
// ahead.  Used for the forward dense layers (LayerNorm / GELU / attention outputs are O(1)); gradients stay on gemm_x6.
// Replaces: Conv1D.call (migt.py:89-96), SharedEmbeddings._linear (:51-56), gelu (:70), the 1x1 convolutions of vqgan_th.py.
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef VF_GEMM_X3H_TALL
#define VF_GEMM_X3H_TALL 0     // A/B on the encoder's 1x1 shapes (tools/microbench.py gemm_1x1*): 143 vs 137, 164 vs 172, 180 vs 181 TF — a tie
#endif

constexpr int CK = 32;
constexpr int BM = 128, BN = 128;
constexpr int A_LDB = 144;                        // bytes per A row in LDS: 2 planes x 64 B + 16 B pad
constexpr int A_BYTES = BM * A_LDB;               // 18432
constexpr int PLANE_BYTES = 2 * BN * 16;          // one (k-step, plane): [half(2)][n(128)][8 f16] = 4 KB
constexpr int KS_BYTES = 2 * PLANE_BYTES;
constexpr int CHUNK_BYTES = 2 * KS_BYTES;         // one (k-chunk, n-block): 16 KB
constexpr int TAIL_BYTES = 16;                    // behind the planes: float 1/S, uint32 max|w| bits (pack scratch)

__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * 2048.f);
}

// PIMG: distinct images among the 4 rows a thread stages (rows r0 + 32 q of a 128-row tile): 1 when an image has a multiple of 128 rows,
// 2 for a multiple of 64, else 4 — the per-image GroupNorm mean / scale registers of the prologue shrink with it (32 -> 8 / 16 registers:
// the 4-image form spilled 54 registers into the staging path of every chunk)
template <bool PRO, bool SWISH, int PIMG = 4>
__global__ __launch_bounds__(256, 2) void gemm_x3h_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];   // [2][A_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile: TALL = all 128 rows x 32 columns (4 x 1 MFMA tiles) instead of 64 x 64 (2 x 2): no two waves stream the same weight
    // fragment (half the weight bytes through the CU's vector-memory path; twice the A fragments from LDS, which has headroom)
    constexpr bool TALL = VF_GEMM_X3H_TALL != 0;
    constexpr int MI = TALL ? 4 : 2, NJ = TALL ? 1 : 2;
    const int wave_m = TALL ? 0 : wave >> 1, wave_n = TALL ? wave : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const float* __restrict__ X = p.x;
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * CHUNK_BYTES;
    const size_t chunk_stride = (size_t)nb * CHUNK_BYTES;
    // split-K (reserved0 = number of splits, blockIdx.y = split): each split reduces an even-sized range of the 32-deep chunks
    // into its own [M][ldc] slab at out + split * stride_out (the caller sums the slabs in a fixed order: deterministic)
    const int nsplit = p.reserved0 > 1 ? p.reserved0 : 1;
    const int total_chunks = p.Cin / CK;
    const int per = ((total_chunks + nsplit - 1) / nsplit + 1) & ~1;
    const int c0 = min((int)blockIdx.y * per, total_chunks);
    const int nchunks = min(c0 + per, total_chunks);            // exclusive end of this split's range

    // A staging: thread -> float4 column (tid & 7) of rows (tid >> 3) + 32 q
    const int c4 = tid & 7, r0 = tid >> 3;
    const float* arow[4];
    int aimg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m = mtile * BM + r0 + 32 * q;
        m = m < p.M ? m : p.M - 1;                 // clamped: rows past M are never stored
        arow[q] = X + (size_t)m * p.lda + c4 * 4;
        aimg[q] = PRO ? m / p.pro_rows_per_img : 0;
    }
    f32x4 areg[4];
    f32x4 pbeta;
    f32x4 pmean[PIMG], pscale[PIMG];
    auto a_fetch = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) areg[q] = *reinterpret_cast<const f32x4*>(arow[q] + chunk * CK);
        if (PRO) {
            pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
#pragma unroll
            for (int i = 0; i < PIMG; ++i) {
                pmean[i] = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)aimg[i * (4 / PIMG)] * p.Cin + chunk * CK + c4 * 4);
                pscale[i] = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)aimg[i * (4 / PIMG)] * p.Cin + chunk * CK + c4 * 4);
            }
        }
    };
    auto a_park = [&](int buf, int q) {
        unsigned char* dst = smem_g + buf * A_BYTES + (r0 + 32 * q) * A_LDB + c4 * 8;
        f16x4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = areg[q][e];
            if (PRO) {
                t = (t - pmean[q * PIMG / 4][e]) * pscale[q * PIMG / 4][e] + pbeta[e];
                if (SWISH) t = vf_swish_1ulp(t);
            }
            _Float16 h, l;
            split2(t, h, l);
            oh[e] = h; ol[e] = l;
        }
        *reinterpret_cast<f16x4*>(dst) = oh;
        *reinterpret_cast<f16x4*>(dst + 64) = ol;
    };

    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    f16x8 bring[2][2][2][NJ];         // [chunk parity][ks][plane][j]
    auto b_load = [&](f16x8 (&dst)[2][2][NJ], int chunk) {
        chunk = min(chunk, nchunks - 1);
        const unsigned char* src = Wb + (size_t)chunk * chunk_stride + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    dst[ks][pl][j] = *reinterpret_cast<const f16x8*>(src + ks * KS_BYTES + pl * PLANE_BYTES + j * 32 * 16);
    };
    const int a_lane = (wave_m * 64 + l31) * A_LDB + half * 16;

    f32x16 acc[MI][NJ], accx[MI][NJ];      // main products ah*bh; cross products (al*2^11)*bh + ah*(bl*2^11)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    a_fetch(min(c0, total_chunks - 1));
    b_load(bring[0], c0);
#pragma unroll
    for (int q = 0; q < 4; ++q) a_park(0, q);
    __syncthreads();

    auto chunk_body = [&](int chunk, f16x8 (&bcur)[2][2][NJ], f16x8 (&bnext)[2][2][NJ]) {
        const unsigned char* a_src = smem_g + ((chunk - c0) & 1) * A_BYTES + a_lane;
        a_fetch(min(chunk + 1, nchunks - 1));
        b_load(bnext, chunk + 1);
        f16x8 a[2][MI][2];            // [ks][mi][plane]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
                    a[ks][mi][pl] = *reinterpret_cast<const f16x8*>(a_src + mi * 32 * A_LDB + pl * 64 + ks * 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][mi][1], bcur[ks][0][j], accx[mi][j], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][mi][0], bcur[ks][1][j], accx[mi][j], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][mi][0], bcur[ks][0][j], acc[mi][j], 0, 0, 0);
            a_park((chunk - c0 + 1) & 1, ks * 2);
            a_park((chunk - c0 + 1) & 1, ks * 2 + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int chunk = c0; chunk < nchunks; chunk += 2) {          // ranges are even-sized (Cin % 64 == 0, per is even)
        chunk_body(chunk, bring[0], bring[1]);
        chunk_body(chunk + 1, bring[1], bring[0]);
    }

    // out = (acc + accx * 2^-11) / S, S = the power-of-two weight scale stored behind the packed planes
    const float inv_s = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)total_chunks * chunk_stride);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accx[i][j][r], 4.8828125e-4f, acc[i][j][r]) * inv_s;
    float* __restrict__ Out = p.out + (size_t)blockIdx.y * p.stride_out;
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
                    if (Res) vf_store_tile<1, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<1, false>(acc[i][j], bias, o, rs, oo, ro);
                } else {
                    if (Res) vf_store_tile<0, true>(acc[i][j], bias, o, rs, oo, ro);
                    else vf_store_tile<0, false>(acc[i][j], bias, o, rs, oo, ro);
                }
            } else if (gelu) {
                if (Res) vf_store_tile_ragged<1, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<1, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            } else {
                if (Res) vf_store_tile_ragged<0, true>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
                else vf_store_tile_ragged<0, false>(acc[i][j], bias, o, rs, ldc, ldr, rows_left);
            }
        }
    }
}


__global__ void absmax_strided_kernel(const float* __restrict__ w, int K, int N, long long sk, long long sn, unsigned* __restrict__ out) {
    float m = 0.f;
    const long long n = (long long)K * N;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[(i / N) * sk + (i % N) * sn]));
    vf_block_max_atomic(m, out);
}

// fp32 [K][N] (strided) -> two f16 planes of w * S, fragment-major [K/32][nb][ks(2)][plane(2)][half(2)][n(128)][8]
__global__ void pack_gemm_x3h_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int K, int N, long long sk,
                                     long long sn, int nb, int nchunks, unsigned char* __restrict__ tail) {
    const float amax = __uint_as_float(*reinterpret_cast<const unsigned*>(tail + 4));
    const int ex = amax > 0.f ? ilogbf(amax) : 13;
    const float S = ldexpf(1.f, 13 - ex);
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(tail) = ldexpf(1.f, ex - 13);
    const long long total = (long long)nchunks * nb * CK * BN;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        long long t = idx >> 3;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int ks = (int)((t >> 1) & 1);
        t >>= 2;
        const int nblk = (int)(t % nb);
        const int chunk = (int)(t / nb);
        const int k = chunk * CK + ks * 16 + half * 8 + e;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (k < K && n < N) v = src[k * sk + n * sn] * S;
        _Float16 h, l;
        split2(v, h, l);
        const size_t base = ((((size_t)chunk * nb + nblk) * 2 + ks) * 2) * (2 * BN * 8) + ((size_t)half * BN + nl) * 8 + e;
        dst[base] = h;
        dst[base + 2 * BN * 8] = l;
    }
}

template <bool PRO, bool SWISH, int PIMG = 4>
int launch(const vf_igemm_args& a, hipStream_t stream) {
    const size_t smem = (size_t)2 * A_BYTES;
    const int nb = (a.Cout + BN - 1) / BN, mt = (a.M + BM - 1) / BM;
    const int nsplit = a.reserved0 > 1 ? a.reserved0 : 1;
    hipLaunchKernelGGL((gemm_x3h_kernel<PRO, SWISH, PIMG>), dim3((unsigned)(mt * nb), (unsigned)nsplit), dim3(256), smem, stream, a);
    return vf_last_status();
}

}  // namespace

extern "C" {

size_t vf_gemm_x3h_packed_elems(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (size_t)((K + CK - 1) / CK) * ((N + BN - 1) / BN) * CK * BN * 2 + TAIL_BYTES / 2;
}

int vf_gemm_x3h_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, void* stream) {
    if (!src || !dst || K <= 0 || N <= 0) return VF_ERR_BAD_ARG;
    const int nb = (N + BN - 1) / BN, nchunks = (K + CK - 1) / CK;
    const long long total = (long long)nchunks * nb * CK * BN;
    unsigned char* tail = reinterpret_cast<unsigned char*>(dst) + (size_t)total * 2 * sizeof(_Float16);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(tail, 0, TAIL_BYTES, s) != hipSuccess) return vf_last_status();
    const long long nw = (long long)K * N;
    hipLaunchKernelGGL(absmax_strided_kernel, dim3((unsigned)((nw + 2047) / 2048 > 128 ? 128 : (nw + 2047) / 2048)), dim3(256), 0, s, src, K,
                       N, (long long)sk, (long long)sn, reinterpret_cast<unsigned*>(tail + 4));
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_gemm_x3h_kernel, dim3(blocks), dim3(256), 0, s, src, (_Float16*)dst, K, N, (long long)sk, (long long)sn, nb,
                       nchunks, tail);
    return vf_last_status();
}

int vf_gemm_x3h(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0 || a.Cin <= 0 || a.Cout <= 0) return VF_ERR_BAD_ARG;
    if (a.epilogue != VF_EPI_NONE && a.epilogue != VF_EPI_GELU_ERF) return VF_ERR_UNSUPPORTED;      // (VF_EPI_GELU_BWD: vf_gemm_bf16 only)
    if (a.gn_part) return VF_ERR_UNSUPPORTED;        // fused GroupNorm statistics: halo-tile kernels only
    if (a.drop_rate != 0.f || a.out_aux) return VF_ERR_UNSUPPORTED;      // fused output dropout / second output: vf_gemm_bf16 only
    if (a.mode != VF_MODE_GEMM || a.batch > 1) return VF_ERR_UNSUPPORTED;
    if (a.Cin % 64 != 0) return VF_ERR_UNSUPPORTED;  // two 32-deep chunks per pipeline round
    if (a.reserved0 > 1 && (a.bias || a.res || a.epilogue != VF_EPI_NONE || a.pro_mean || a.stride_out < (int64_t)a.M * a.ldc))
        return VF_ERR_BAD_ARG;                       // split-K writes raw partial slabs
    if (a.lda < a.Cin || (a.lda & 3) || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
    if ((a.pro_mean || a.pro_scale || a.pro_beta) && !(a.pro_mean && a.pro_scale && a.pro_beta && a.pro_rows_per_img > 0))
        return VF_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (!a.pro_mean) return launch<false, false>(a, s);
    // (a clamped tail row of the last tile may belong to a later image than its group's first row; it is never stored)
    if (a.pro_rows_per_img % 128 == 0) return a.pro_swish ? launch<true, true, 1>(a, s) : launch<true, false, 1>(a, s);
    if (a.pro_rows_per_img % 64 == 0) return a.pro_swish ? launch<true, true, 2>(a, s) : launch<true, false, 2>(a, s);
    return a.pro_swish ? launch<true, true>(a, s) : launch<true, false>(a, s);
}

}  // extern "C"
