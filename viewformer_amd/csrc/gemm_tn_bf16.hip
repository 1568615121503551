// Weight-gradient GEMM of the bf16 training arm: dW[K][N] = sum_m X[m][K] * dY[m][N], with the bias gradient sum_m dY[m][N] on the
// side, gfx950.  Both operands are read ROW-MAJOR IN THE REDUCTION DIMENSION, as the training step holds them — X = a saved bf16
// activation [M][ldx], dY = the fp32 gradient [M][ldy] — so the three passes the first version spent per weight gradient disappear:
// the widening transpose of X ([K][M] fp32 copy), the fp32 -> bf16 fragment packing of dY, and the column-sum pass over dY.
//   * tile 256 (K) x 256 (N) per workgroup of 8 waves (wave tile 128 x 64 = 4 x 2 MFMA tiles, 128 accumulator registers), reduction
//     in 64-row chunks, double-buffered LDS (2 x 64 KB), one raw barrier per chunk;
//   * X chunk [64 m][256 k] bf16 by LDS-DMA into the "tr" image of attention_dma.hip ([32-column block][m][32 columns], 64-byte rows):
//     ds_read_b64_tr_b16 hands every lane 4 consecutive m of ITS column — the A fragment X^T[k][m] of v_mfma_f32_32x32x16_bf16
//     straight from the row-major activation;
//   * dY chunk [64 m][256 n] fp32 through registers (a wave reads whole 1 KB rows), rounded to bf16 once (the rounding the packed
//     operand had before) and parked in the same image layout: the B fragment dY[m][n] is the same transposed read; the values pass
//     through registers anyway, so each thread also keeps the running column sums of what it loads -> the bias gradient;
//   * split-K over row ranges (gridDim.y): every split writes its own fp32 slab (and bias slab); vf_sum_slabs_f32 folds them in slab
//     order (deterministic), as for the other weight-gradient paths.
// Arithmetic intensity 87 FLOP per byte moved through the CU (96 KB per 8.4 MFLOP chunk) against 44 for the 128 x 128 batched form
// it replaces.  Reference: the autograd of Conv1D.call (x @ W + b, viewformer/models/migt.py:89-96) inside MIGT.train_step
// (migt.py:464-505) under mixed_float16.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 256, TN_ = 256, CM = 64;          // tile rows (K of the layer), tile columns (N), reduction rows per chunk
constexpr int IMG = CM * 256 * 2;                    // one operand chunk as bf16: 32 KB
constexpr int STAGE = 2 * IMG;                       // X image | dY image

__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p)));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p + 8 * 64)));
    const bf16x4 v0 = __builtin_bit_cast(bf16x4, r0), v1 = __builtin_bit_cast(bf16x4, r1);
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = v0[e]; a[4 + e] = v1[e]; }
    return a;
}

// Y16: dY arrives as bf16 (its producer — gelu_bwd, the attention backward — wrote the rounding this kernel would apply): it takes the DMA
// path of X, and the bias sums are read back from the landed LDS image instead of the staging registers.
template <bool Y16>
__global__ __launch_bounds__(512, 1) void gemm_tn_bf16_kernel(const __bf16* __restrict__ x, const void* __restrict__ dy_,
                                                              float* __restrict__ w_slabs, float* __restrict__ b_slabs, int M, int K, int N,
                                                              int ldx, int ldy, long long w_stride, long long b_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 2 x (X image | dY image)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int ntn = N / TN_;
    // 1-D grid of nsplit x tiles workgroups in XCD-contiguous logical order (vf_common.h), split-major: the ~32 workgroups an XCD runs are then
    // (nearly) all tiles of ONE row range, whose X and dY chunks its L2 fetches once.  With (tile, split) as a 2-D grid the dispatcher dealt the
    // tiles of a split round-robin over the 8 XCDs: FETCH_SIZE 217 MB per launch (x 2 per the guide's gfx950 note) for 118 MB of operands
    // (profiles/r3_train_step_pmc.txt).
    const int ntiles_kn = (K / TK) * ntn;
    const int lbid = (int)vf_xcd_bid();
    const int split = lbid / ntiles_kn, nsplit = (int)gridDim.x / ntiles_kn;
    const int tile = lbid - split * ntiles_kn;
    const int tk = tile / ntn, tn = tile - tk * ntn;
    const int nchunk_all = M / CM;
    const int c0 = (int)((long long)nchunk_all * split / nsplit), c1 = (int)((long long)nchunk_all * (split + 1) / nsplit);

    const __bf16* xb = x + (size_t)tk * TK;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(xb), 0, 0x7fffffff, 0x00020000);
    const float* dyb = reinterpret_cast<const float*>(dy_) + (size_t)tn * TN_;                          // (!Y16)
    const __bf16* dyb16 = reinterpret_cast<const __bf16*>(dy_) + (size_t)tn * TN_;                      // (Y16)
    const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(dyb16), 0, 0x7fffffff, 0x00020000);

    // X: wave w moves column block w (32 columns) of the chunk: 4 pieces of 16 rows x 64 B
    auto issue_x = [&](int stage, int chunk) {
        unsigned char* img = smem + stage * STAGE + wave * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bufds16(x_rs, img + g * 1024, (unsigned)((g * 16 + (lane >> 2)) * ldx * 2 + wave * 64 + (lane & 3) * 16), (unsigned)(chunk * CM * ldx * 2));
    };
    // dY: thread -> column quad nq (this wave reads whole 1 KB rows), rows mg + 8 j
    const int nq = tid & 63, mg = tid >> 6;
    f32x4 yreg[Y16 ? 1 : 8];
    auto load_y = [&](int stage, int chunk) {
        if constexpr (Y16) {                                          // the X path: wave w moves column block w, 4 pieces of 16 rows x 64 B
            unsigned char* img = smem + stage * STAGE + IMG + wave * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bufds16(y_rs, img + g * 1024, (unsigned)((g * 16 + (lane >> 2)) * ldy * 2 + wave * 64 + (lane & 3) * 16), (unsigned)(chunk * CM * ldy * 2));
        } else {
            const float* src = dyb + (size_t)(chunk * CM + mg) * ldy + nq * 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) yreg[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(8 * j) * ldy);
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto park_y = [&](int stage) {                                    // (!Y16) registers -> bf16 image, column sums on the way
        if constexpr (!Y16) {
            unsigned char* img = smem + stage * STAGE + IMG + (nq >> 3) * 4096 + (nq & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                bsum += yreg[j];
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)yreg[j][e];
                *reinterpret_cast<bf16x4*>(img + (mg + 8 * j) * 64) = o;
            }
        }
    };
    auto sum_y = [&](int stage) {                                     // (Y16) column sums of the landed image: this thread's 8 rows x 4 columns
        if constexpr (Y16) {
            const unsigned char* img = smem + stage * STAGE + IMG + (nq >> 3) * 4096 + (nq & 7) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16x4 t = *reinterpret_cast<const bf16x4*>(img + (mg + 8 * j) * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum[e] += (float)t[e];
            }
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int wk = wave >> 2, wn = wave & 3;                          // wave tile: rows 128 wk .. + 127, columns 64 wn .. + 63
    const unsigned tr_off = (unsigned)((4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);

    if (c0 < c1) {
        issue_x(0, c0);
        load_y(0, c0);
        park_y(0);
    }
    for (int c = c0; c < c1; ++c) {
        const int st = (c - c0) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's X pieces of chunk c have landed, its dY writes are done
        __builtin_amdgcn_s_barrier();                                 // ... for every wave; everyone is done reading stage st ^ 1
        const bool more = c + 1 < c1;
        if (more) {
            issue_x(st ^ 1, c + 1);
            load_y(st ^ 1, c + 1);
        }
        if (b_slabs && tk == 0) sum_y(st);
        const unsigned char* sx = smem + st * STAGE;
        const unsigned char* sy = sx + IMG;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            bf16x8 bf[2];
#ifdef VF_X_TRINTRIN
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(sy + tr_off + (wn * 2 + b) * 4096 + ms * 16 * 64);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bf16x8 af = tr_frag(sx + tr_off + (wk * 4 + a) * 4096 + ms * 16 * 64);
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[b], acc[a][b], 0, 0, 0);
            }
#else
            // transposing reads as inline asm (vf_tr_frag*_wait, vf_common.h): through the intrinsic, hipcc put `s_waitcnt vmcnt(0)` in front of the
            // chunk's first read — i.e. it waited for chunk c + 1's DMA (and dY loads), issued a few instructions earlier, before multiplying chunk c:
            // the double buffer never overlapped anything
            const unsigned ay = vf_lds_addr(sy) + tr_off + (unsigned)(wn * 2 * 4096), ax = vf_lds_addr(sx) + tr_off + (unsigned)(wk * 4 * 4096);
            vf_tr_frag2_wait(bf[0], bf[1], ay, ms * 16 * 64, 4096 + ms * 16 * 64, 8 * 64);
#pragma unroll
            for (int a = 0; a < 4; a += 2) {
                bf16x8 af0, af1;
                vf_tr_frag2_wait(af0, af1, ax, a * 4096 + ms * 16 * 64, (a + 1) * 4096 + ms * 16 * 64, 8 * 64);
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, bf[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a + 1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bf[b], acc[a + 1][b], 0, 0, 0);
            }
#endif
        }
        if (more) park_y(st ^ 1);                                     // (the loads have had the chunk's MFMAs to arrive)
    }

    // ---- the split's slab of dW: row k = 256 tk + 128 wk + 32 a + (r & 3) + 8 (r >> 2) + 4 half, column n = 256 tn + 64 wn + 32 b + l31
    float* slab = w_slabs + (size_t)split * w_stride + (size_t)(tk * TK + wk * 128) * N + tn * TN_ + wn * 64;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                slab[(size_t)(32 * a + (r & 3) + 8 * (r >> 2) + 4 * half) * N + 32 * b + l31] = acc[a][b][r];

    // ---- bias gradient of this (column tile, split): the eight row groups of a column folded in a fixed order (row tile 0 only)
    if (b_slabs && tk == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* red = reinterpret_cast<float*>(smem);                  // [8][256]
#pragma unroll
        for (int e = 0; e < 4; ++e) red[mg * 256 + nq * 4 + e] = bsum[e];
        __syncthreads();
        if (tid < 256) {
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += red[g * 256 + tid];
            b_slabs[(size_t)split * b_stride + tn * TN_ + tid] = s;
        }
    }
}

}  // namespace

extern "C" {

int vf_gemm_tn_bf16(const void* x_bf16, int ldx, const void* dy, int dy_is_bf16, int ldy, int M, int K, int N, int splits, float* w_slabs,
                    float* b_slabs, int64_t slab_stride, void* stream) {
    if (!x_bf16 || !dy || !w_slabs || M <= 0 || K <= 0 || N <= 0 || splits < 1) return VF_ERR_BAD_ARG;
    // slab_stride 0: weight slabs K*N apart, bias slabs N apart (two arrays); > 0: ONE array of `splits` records, the split's weight slab at
    // w_slabs + s * slab_stride and its bias slab at b_slabs + s * slab_stride (b_slabs = w_slabs + K*N: one vf_sum_slabs_f32 folds both)
    if (slab_stride != 0 && (slab_stride < (int64_t)K * N + (b_slabs ? N : 0) || (slab_stride & 3))) return VF_ERR_BAD_ARG;
    const long long w_stride = slab_stride ? slab_stride : (long long)K * N, b_stride = slab_stride ? slab_stride : N;
    if (ldx < K || ldy < N) return VF_ERR_BAD_ARG;
    if (K % TK || N % TN_ || M % CM || (ldx & 7) || (ldy & (dy_is_bf16 ? 7 : 3)) || splits > M / CM) return VF_ERR_UNSUPPORTED;
    if (((uintptr_t)x_bf16 | (uintptr_t)dy | (uintptr_t)w_slabs) & 15) return VF_ERR_UNSUPPORTED;
    if ((size_t)M * ldx * 2 >= (1ull << 31) || (dy_is_bf16 && (size_t)M * ldy * 2 >= (1ull << 31))) return VF_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    static unsigned long long attr_devs = 0;
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const dim3 grid((unsigned)((K / TK) * (N / TN_) * splits));
    if (dy_is_bf16)
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<true>, grid, dim3(512), (size_t)2 * STAGE, (hipStream_t)stream, reinterpret_cast<const __bf16*>(x_bf16), dy,
                           w_slabs, b_slabs, M, K, N, ldx, ldy, w_stride, b_stride);
    else
        hipLaunchKernelGGL(gemm_tn_bf16_kernel<false>, grid, dim3(512), (size_t)2 * STAGE, (hipStream_t)stream, reinterpret_cast<const __bf16*>(x_bf16), dy,
                           w_slabs, b_slabs, M, K, N, ldx, ldy, w_stride, b_stride);
    return vf_last_status();
}

}  // extern "C"
