// Tied LM head with the arg-max fused into its epilogue, bf16 arm, gfx950.
//
// Replaces SharedEmbeddings._linear (viewformer/models/migt.py:51-56) + the slice to n_embeddings (:417) + tf.argmax over the codes
// (viewformer/evaluate/evaluate_transformer.py:123) for the rows the evaluator consumes: idx[m] = first arg-max_n sum_k bf16(h[m][k]) *
// bf16(wte[n][k]).  The [M][n_embeddings] logits (33.5 MB at the bench's 8192 rows) are never written.  Same operands, same k order
// and same single fp32 accumulation chain per (row, code) as vf_gemm_bf16 on the same packing, so the index equals the arg-max of that
// kernel's logits bit for bit (ties -> lowest index, like vf_argmax_rows_f32).
//
// One workgroup = 32 rows x all codes: the 32 rows stay in registers as bf16 A fragments (K/16 fragments per lane), each of the 4
// waves walks a quarter of the codes in 32-wide tiles with the packed weight fragments streamed L2 -> VGPR two k-steps ahead, and
// keeps a per-lane running (max, first index); one shuffle reduction per row at the end, then across the 4 waves through LDS.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BN = 128;                   // the bf16 weight packing's chunk / n-block (csrc/gemm_bf16.hip)

template <int KSTEPS>
__global__ __launch_bounds__(256, 1) void lmhead_argmax_kernel(const void* __restrict__ hrows, int h16, long long ldh,
                                                               const unsigned char* __restrict__ Wp, long long M, int N,
                                                               long long* __restrict__ idx_out, float* __restrict__ max_out) {
    __shared__ float red_v[4][32];
    __shared__ int red_i[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long m0 = (long long)blockIdx.x * 32;
    long long row = m0 + l31;
    row = row < M ? row : M - 1;

    bf16x8 a[KSTEPS];
    if (h16) {
        const __bf16* src = reinterpret_cast<const __bf16*>(hrows) + (size_t)row * ldh + half * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(src + ks * 16);
    } else {
        const float* src = reinterpret_cast<const float*>(hrows) + (size_t)row * ldh + half * 8;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + ks * 16);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(src + ks * 16 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[ks][e] = (__bf16)t0[e]; a[ks][4 + e] = (__bf16)t1[e]; }
        }
    }

    const int nb = N / BN;
    const int tiles_per_wave = N / 4 / 32;
    float best[16];
    int besti[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { best[r] = -INFINITY; besti[r] = 0; }
    for (int nt = 0; nt < tiles_per_wave; ++nt) {
        const int n0 = wave * (N / 4) + nt * 32;
        const int nblk = n0 / BN, nl = (n0 % BN) + l31;
        // fragment (chunk, ks) of column n: ((((chunk*nb + nblk)*4 + ks)*2 + half)*128 + nl) * 16 bytes
        const unsigned char* wsrc = Wp + ((size_t)nblk * 8 + half) * (BN * 16) + (size_t)nl * 16;
        const size_t chunk_stride = (size_t)nb * 8 * BN * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bf16x8 b = *reinterpret_cast<const bf16x8*>(wsrc + (size_t)(ks >> 2) * chunk_stride + (size_t)(ks & 3) * (2 * BN * 16));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b, acc, 0, 0, 0);
        }
        const int n = n0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (acc[r] > best[r]) { best[r] = acc[r]; besti[r] = n; }           // strict: the first maximum of this lane's (ascending) codes
    }
    // row r of a lane = row (r&3) + 8 (r>>2) + 4 half of the tile; reduce over the 32 lanes (codes) of the half-wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = best[r];
        int i = besti[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(i, o, 64);
            if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (l31 == 0) {
            const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
            red_v[wave][rl] = v;
            red_i[wave][rl] = i;
        }
    }
    __syncthreads();
    if (tid < 32) {
        float v = red_v[0][tid];
        int i = red_i[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = red_v[w][tid];
            const int oi = red_i[w][tid];
            if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (m0 + tid < M) {
            idx_out[m0 + tid] = (long long)i;
            if (max_out) max_out[m0 + tid] = v;
        }
    }
}

}  // namespace

extern "C" {

int vf_lmhead_argmax_bf16(const void* h, int h_bf16, int64_t ldh, const void* w_packed, int64_t M, int K, int N, int64_t* idx,
                          float* max_logit, void* stream) {
    if (M == 0) return VF_OK;
    if (!h || !w_packed || !idx || M < 0 || K <= 0 || N <= 0 || ldh < K) return VF_ERR_BAD_ARG;
    if (N % BN != 0 || (K != 768 && K != 128) || (h_bf16 ? (ldh & 7) : (ldh & 3))) return VF_ERR_UNSUPPORTED;
    const unsigned grid = (unsigned)((M + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(w_packed);
    if (K == 768)
        hipLaunchKernelGGL(lmhead_argmax_kernel<48>, dim3(grid), dim3(256), 0, s, h, h_bf16, (long long)ldh, wp, (long long)M, N,
                           reinterpret_cast<long long*>(idx), max_logit);
    else
        hipLaunchKernelGGL(lmhead_argmax_kernel<8>, dim3(grid), dim3(256), 0, s, h, h_bf16, (long long)ldh, wp, (long long)M, N,
                           reinterpret_cast<long long*>(idx), max_logit);
    return vf_last_status();
}

}  // extern "C"
