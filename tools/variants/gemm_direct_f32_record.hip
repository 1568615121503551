// RECORD, not part of libvf_hip.so since round 4: the LDS-free "direct" f32 GEMM (A and B fragments L2 -> VGPR, no barrier): 118 TF asymptotic vs 123 TF for the LDS-staged igemm kernel, 97 vs 102 TF at K = 768 - was opt-in behind VF_ENABLE_DIRECT=1, with an MFMA-only calibration switch (VF_GEMM_NOLOAD=1, wrong results by construction).
// Kept as the source the measurements in DESIGN.md refer to; builds against the round-3 C-ABI (git show 7e8c4c4:include/vf_hip.h).
// Dense / 1x1-conv GEMM on exact-f32 MFMA with NO LDS and NO workgroup barrier (gfx950).
//
//   out[m][n] = epi( sum_k A[m][k] * W[k][n] + bias[n] ) + res[m][n]
//
// The f32 MFMA (v_mfma_f32_32x32x2_f32) retires one instruction per 64 cycles, so a wave needs only two
// 16-byte operand fetches per 4 MFMAs — a rate L1/L2 sustain directly.  Each wave therefore owns a 64x64
// output tile and streams its own fragments straight into VGPRs, one 32-deep K stage ahead:
//   A: lane (row i, half h) reads A[row][8g+4h .. +3] — 32 rows x 32 B per instruction, every byte of
//      each 128-B row segment is consumed within the stage (L1 keeps the line);
//   B: pre-packed fragment-major weights, 512 contiguous bytes per half-wave.
// No staging VALU, no ds_write/ds_read, no s_barrier: the instruction stream per stage is 16 global
// loads + 64 MFMAs, and the four waves of a workgroup (2x2 over a 128x128 tile) share A rows / B
// columns through L1.  Rows past M are clamped to M-1 for loading and masked at the store.
// Replaces Conv1D.call (migt.py:89-96) incl. gelu (:70) / residual (:233,237), SharedEmbeddings._linear
// (:51-56), and the 1x1 convs of vqgan_th.py:72-76,114-118,332-333; igemm_f32.hip stays the fallback
// (GroupNorm prologue, Cout % 128 != 0).
#include "../../viewformer_amd/csrc/vf_common.h"
#include "../../viewformer_amd/csrc/epilogue.h"
#include "../../include/vf_hip.h"
#include <stdlib.h>

namespace {

constexpr int CK = 32;
constexpr int BN = 128;
constexpr int BM = 128;

template <int EPI, bool NOLOAD = false>
__global__ __launch_bounds__(256, 2) void gemm_direct_kernel(vf_igemm_args p) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int nblk = blockIdx.x % nb;
    const int mtile = blockIdx.x / nb;
    const int bz = blockIdx.z;
    const float* __restrict__ X = p.x + (size_t)bz * p.stride_x;
    const float* __restrict__ Wb = p.w_packed + (size_t)bz * p.stride_w + (size_t)nblk * (CK * BN);   // uniform
    const size_t stage_stride = (size_t)nb * CK * BN;
    const int nstages = p.Cin / CK;

    // per-lane A row pointers (clamped) and B lane offset
    const float* arow[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int m = mtile * BM + wave_m * 64 + mi * 32 + l31;
        m = m < p.M ? m : p.M - 1;
        arow[mi] = X + (size_t)m * p.lda + half * 4;
    }
    const int b_lane = (half * BN + wave_n * 64 + l31) * 4;

    f32x4 ac[8], bc[8], an[8], bn[8];
    auto load = [&](f32x4 (&a)[8], f32x4 (&b)[8], int stage) {
        const float* bsrc = Wb + (size_t)stage * stage_stride;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) a[g * 2 + mi] = *reinterpret_cast<const f32x4*>(arow[mi] + stage * CK + g * 8);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[g * 2 + j] = *reinterpret_cast<const f32x4*>(bsrc + (g * 2 * BN + j * 32) * 4 + b_lane);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load(ac, bc, 0);
    for (int s = 0; s < nstages; ++s) {
        if (!NOLOAD) load(an, bn, min(s + 1, nstages - 1));   // clamped prefetch keeps the loop body branch-free
        __builtin_amdgcn_sched_barrier(0);           // pin: next-stage loads are issued BEFORE this stage's MFMAs
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[g * 2 + mi][e], bc[g * 2 + j][e], acc[mi][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) { if (!NOLOAD) { ac[q] = an[q]; bc[q] = bn[q]; } else { asm volatile("" : "+v"(ac[q]), "+v"(bc[q])); } }
    }

    float* __restrict__ Out = p.out + (size_t)bz * p.stride_out;
    const float* __restrict__ Res = p.res ? p.res + (size_t)bz * p.stride_res : nullptr;
    const bool full = (mtile * BM + BM) <= p.M;      // workgroup-uniform
    const long long ldc = p.ldc, ldr = p.ldr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * BN + wave_n * 64 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int m0 = mtile * BM + wave_m * 64 + mi * 32 + 4 * half;
            float* o = Out + (size_t)m0 * ldc + n;
            const float* rs = Res ? Res + (size_t)m0 * ldr + n : nullptr;
            auto oo = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldc; };
            auto ro = [&](int r) { return (long long)((r & 3) + 8 * (r >> 2)) * ldr; };
            if (full) {
                if (Res) vf_store_tile<EPI, true>(acc[mi][j], bias, o, rs, oo, ro);
                else vf_store_tile<EPI, false>(acc[mi][j], bias, o, rs, oo, ro);
            } else {
                if (Res) vf_store_tile_ragged<EPI, true>(acc[mi][j], bias, o, rs, ldc, ldr, p.M - m0);
                else vf_store_tile_ragged<EPI, false>(acc[mi][j], bias, o, rs, ldc, ldr, p.M - m0);
            }
        }
    }
}

}  // namespace

// eligibility + launch; called from vf_igemm_f32.  Returns 1 if the shape is not handled here.
int vf_gemm_direct_try(const vf_igemm_args& a, hipStream_t stream, int* status) {
    if (a.mode != VF_MODE_GEMM || a.pro_mean) return 1;
    if (a.Cout % BN != 0 || a.Cin % CK != 0) return 1;
    const int nb = a.Cout / BN;
    const int mt = (a.M + BM - 1) / BM;
    dim3 grid((unsigned)(mt * nb), 1, (unsigned)(a.batch > 0 ? a.batch : 1));
    static const bool noload = [] { const char* e = getenv("VF_GEMM_NOLOAD"); return e && e[0] == '1'; }();   // MFMA-only calibration (wrong results)
    if (noload)
        hipLaunchKernelGGL((gemm_direct_kernel<VF_EPI_NONE, true>), grid, dim3(256), 0, stream, a);
    else if (a.epilogue == VF_EPI_GELU_ERF)
        hipLaunchKernelGGL(gemm_direct_kernel<VF_EPI_GELU_ERF>, grid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(gemm_direct_kernel<VF_EPI_NONE>, grid, dim3(256), 0, stream, a);
    *status = vf_last_status();
    return 0;
}
