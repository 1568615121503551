// Codebook L2-argmin on exact-f32 MFMA, gfx950.
//
// Replaces QuantizeEMA.forward (eval branch), viewformer/models/utils_th.py:32-44:
//     dist = z.pow(2).sum(1) - 2 * z @ E + E.pow(2).sum(0);  idx = (-dist).max(1)   (ties -> lowest index)
// The [M][Kc] distance matrix (256 KB per image in the reference) is never written: each
// 256-thread workgroup owns 128 rows of z, walks the codebook in 128-code tiles with the same
// LDS-staged 32x32x2 f32 MFMA pipeline as igemm_f32.hip (wave tile = 32 rows x 128 codes, so a row's
// candidates live in one wave), forms dist = (zz - 2*dot) + ee with the reference's association
// in the epilogue of each code tile and keeps a per-lane running (min, first index); one 5-step
// shuffle reduction per row at the end.  zz (row sum of squares) is accumulated by the staging
// threads during the first code tile, so z is read from HBM exactly once (L2 serves the re-reads
// for the later code tiles): algorithmic bytes = M*D*4 (z) + D*Kc*4 (codebook, once) + M*8 (idx).
// Arithmetic intensity is ~Kc/2 FLOP/B, so the binding roof is the f32 MFMA rate, not HBM.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int CK = 32;
constexpr int A_LD = 36;
constexpr int BM = 128;
constexpr int BN = 128;

__global__ __launch_bounds__(256, 2) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ Ep,
                                                           const float* __restrict__ e_sq, long long M, int D, int Kc,
                                                           long long* __restrict__ idx_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][A_LD]
    float* Bs = smem + 2 * BM * A_LD;     // [2][CK*BN]
    float* zz_s = Bs + 2 * CK * BN;       // [BM]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const long long m0 = (long long)blockIdx.x * BM;

    const int nchunks = D / CK;
    const int ntiles = (Kc + BN - 1) / BN;
    const int nstages = nchunks * ntiles;

    const int a_col4 = tid & 7;
    const int a_row0 = tid >> 3;
    f32x4 areg[4], breg[4];
    float zpart[4] = {0.f, 0.f, 0.f, 0.f};

    auto load_stage = [&](int ntile, int chunk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = m0 + a_row0 + 32 * q;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < M) v = *reinterpret_cast<const f32x4*>(z + (size_t)m * D + chunk * CK + a_col4 * 4);
            areg[q] = v;
        }
        // packed layout [chunk][tap=1][nblk][...]
        const float* wsrc = Ep + ((size_t)chunk * ntiles + ntile) * (CK * BN);
#pragma unroll
        for (int q = 0; q < 4; ++q) breg[q] = *reinterpret_cast<const f32x4*>(wsrc + (size_t)(tid + 256 * q) * 4);
    };
    auto store_stage = [&](int buf, bool first_tile) {
        float* a_dst = As + buf * (BM * A_LD);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = areg[q];
            if (first_tile) zpart[q] += vf_vq_sq4(v);
            *reinterpret_cast<f32x4*>(a_dst + (a_row0 + 32 * q) * A_LD + a_col4 * 4) = v;
        }
        float* b_dst = Bs + buf * (CK * BN);
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(b_dst + (size_t)(tid + 256 * q) * 4) = breg[q];
    };

    float bestv[16];
    int besti[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { bestv[r] = INFINITY; besti[r] = 0; }

    f32x16 acc[4];
    int ntile = 0, chunk = 0;
    load_stage(0, 0);
    store_stage(0, true);
    __syncthreads();

    for (int s = 0; s < nstages; ++s) {
        int nchunk = chunk + 1, nntile = ntile;
        if (nchunk == nchunks) { nchunk = 0; nntile = ntile + 1; }
        const bool more = (s + 1) < nstages;
        if (more) load_stage(nntile, nchunk);

        if (chunk == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        const float* a_src = As + (s & 1) * (BM * A_LD) + (wave * 32 + l31) * A_LD + half * 4;
        const float* b_src = Bs + (s & 1) * (CK * BN) + (half * BN + l31) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(a_src + g * 8);
            f32x4 b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_src + (g * 2 * BN + j * 32) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[j][e], acc[j], 0, 0, 0);
        }
        if (more) store_stage((s + 1) & 1, nntile == 0);

        if (ntile == 0 && chunk == nchunks - 1) {
            // all of z's columns have passed through the staging threads: finish zz per row
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = zpart[q];
                v += __shfl_xor(v, 1, 64);
                v += __shfl_xor(v, 2, 64);
                v += __shfl_xor(v, 4, 64);
                if (a_col4 == 0) zz_s[a_row0 + 32 * q] = v;
            }
        }
        __syncthreads();

        if (chunk == nchunks - 1) {
            // epilogue of this code tile: dist = (zz - 2*dot) + ee, running first-min per lane
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = ntile * BN + j * 32 + l31;
                const bool nok = n < Kc;
                const float ee = nok ? e_sq[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float t = zz_s[row] - 2.0f * acc[j][r];
                    const float d = t + ee;
                    if (nok && d < bestv[r]) { bestv[r] = d; besti[r] = n; }
                }
            }
        }
        chunk = nchunk; ntile = nntile;
    }

    // reduce over the 32 lanes that share a row (same half); ties -> lowest index
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = bestv[r];
        int i = besti[r];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(i, o, 64);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        if (l31 == 0) {
            const long long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < M) idx_out[m] = (long long)i;
        }
    }
}

__global__ void colsumsq_kernel(const float* __restrict__ E, float* __restrict__ out, int D, int Kc) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kc) return;
    float s = 0.f;
    for (int d = 0; d < D; ++d) { const float v = E[(size_t)d * Kc + k]; s += v * v; }
    out[k] = s;
}

__global__ void gather_kernel(const float* __restrict__ E, const long long* __restrict__ idx, float* __restrict__ out,
                              long long M, int D, int Kc) {
    // one workgroup per row; E is [D][Kc] so the gather is a strided column read (L2-resident, 1 MB)
    const long long m = blockIdx.x;
    long long k = idx[m];
    if (k < 0) k = 0;
    if (k >= Kc) k = Kc - 1;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(size_t)m * D + d] = E[(size_t)d * Kc + k];
}

}  // namespace

extern "C" {

size_t vf_vq_packed_floats(int D, int Kc) {
    if (D <= 0 || Kc <= 0) return 0;
    return (size_t)((D + CK - 1) / CK) * ((Kc + BN - 1) / BN) * CK * BN;
}

int vf_vq_pack_codebook_f32(const float* E, float* dst, int D, int Kc, void* stream) {
    if (!E || !dst || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    // B[k=d][n=code] = E[d][code]: row-major [K][N]; always the 128-wide code tile
    return vf_pack_b_impl(E, dst, D, Kc, 1, Kc, 1, 0, BN, 1, 0, (hipStream_t)stream);
}

int vf_colsumsq_f32(const float* E, float* e_sq, int D, int Kc, void* stream) {
    if (!E || !e_sq || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    hipLaunchKernelGGL(colsumsq_kernel, dim3((Kc + 255) / 256), dim3(256), 0, (hipStream_t)stream, E, e_sq, D, Kc);
    return vf_last_status();
}

int vf_vq_argmin_f32(const float* z, const float* E_packed, const float* e_sq, int64_t M, int D, int Kc,
                     int64_t* idx, void* stream) {
    if (!z || !E_packed || !e_sq || !idx || M < 0 || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (D % CK != 0) return VF_ERR_UNSUPPORTED;
    if (M == 0) return VF_OK;
    const size_t smem = (size_t)(2 * BM * A_LD + 2 * CK * BN + BM) * sizeof(float);
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const unsigned grid = (unsigned)((M + BM - 1) / BM);
    hipLaunchKernelGGL(vq_argmin_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, z, E_packed, e_sq,
                       (long long)M, D, Kc, reinterpret_cast<long long*>(idx));
    return vf_last_status();
}

int vf_codebook_gather_f32(const float* E, const int64_t* idx, float* out, int64_t M, int D, int Kc, void* stream) {
    if (!E || !idx || !out || M < 0 || D <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (M == 0) return VF_OK;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)M), dim3(64), 0, (hipStream_t)stream, E,
                       reinterpret_cast<const long long*>(idx), out, (long long)M, D, Kc);
    return vf_last_status();
}

}  // extern "C"
