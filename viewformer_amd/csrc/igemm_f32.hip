// Implicit-GEMM conv3x3 / conv1x1 / dense on exact-f32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[m][n] = epi( sum_{chunk,tap,c} pro(A_tap[m][c]) * W[tap][c][n] + bias[n] ) + res[m][n]
//
// Tile: 128 output rows x BN (128/64/32) columns per 256-thread workgroup, K in stages of 32
// channels per (chunk, tap).  A rows are gathered per tap straight from the NHWC activation
// (128 B per pixel per stage -> coalesced; the 9 taps of a chunk re-touch the same lines, which
// L2 absorbs), optionally pushed through the fused GroupNorm-apply(+swish) prologue in
// registers, and staged in LDS with a 36-float row stride (conflict-free ds_read_b128 for the
// 16-lane groups).  Weights are pre-packed fragment-major ([chunk][tap][nblk][g][half][n][4]) so
// the B stage is a linear 16 KB copy and each lane's B fragment is one ds_read_b128.  Each wave
// owns a 64x64 (BN=128) output sub-tile = 2x2 accumulators of 32x32; one ds_read_b128 of A and of
// B feeds four K=2 MFMAs, so LDS traffic is ~1/64 of the MFMA time and the kernel is bound by
// the f32 MFMA issue rate (64 cycles each).  Double-buffered LDS, one barrier per stage, global
// loads for stage s+1 in flight while stage s computes; 2 workgroups per CU (70 KB LDS each).
//
// Reference call sites replaced: see include/vf_hip.h (vqgan_th.py Conv2d sites, migt.py Conv1D).
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"
#include <stdlib.h>

int vf_conv3_halo_try(const vf_igemm_args& a, hipStream_t stream, int* status);   // conv3_halo_f32.hip

namespace {

constexpr int CK = 32;     // K per stage
constexpr int A_LD = 36;   // LDS row stride of the A tile (floats)

__host__ __device__ inline int bn_for(int N) { return N > 64 ? 128 : (N > 32 ? 64 : 32); }

struct RowInfo {
    int pix_base;   // image index * Hin*Win  (conv) | row index m (gemm)
    int oy, ox;     // output coords (conv)
    int img;        // image index for the prologue tables
    int valid;
};

// PBN = column-block width of the weight PACKING (bn_for(Cout)); equals the tile width BN except for the 64x64
// small-tile variant, which walks half of a 128-wide packed block.
template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int PBN = WAVES_N * WN_T * 32>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(vf_igemm_args p) {
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int BM = WAVES_M * WM_T * 32;   // 128, or 64 for the small-tile variant (under-filled grids)
    constexpr int AP = BM / 32;               // A staging passes (32 rows x 8 float4 per pass)
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int B_F4 = CK * BN / 4 / 256;   // float4 per thread for the B stage

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][A_LD]
    float* Bs = smem + 2 * BM * A_LD;     // [2][CK*BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave / WAVES_N;
    const int wave_n = wave % WAVES_N;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const int bz = blockIdx.z;

    const float* __restrict__ X = p.x + (size_t)bz * p.stride_x;
    const float* __restrict__ Wp = p.w_packed + (size_t)bz * p.stride_w;
    float* __restrict__ Out = p.out + (size_t)bz * p.stride_out;
    const float* __restrict__ Res = p.res ? p.res + (size_t)bz * p.stride_res : nullptr;

    const int taps = (p.mode == VF_MODE_GEMM) ? 1 : 9;
    const int nchunks = p.Cin / CK;
    const int nstages = nchunks * taps;
    const bool has_pro = p.pro_mean != nullptr;

    // ---- per-thread staging rows -------------------------------------------------------
    const int a_col4 = tid & 7;
    const int a_row0 = tid >> 3;
    RowInfo rows[AP];
    const int HWo = p.Hout * p.Wout;
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int m = mtile * BM + a_row0 + 32 * q;
        RowInfo r;
        r.valid = m < p.M;
        const int mm = r.valid ? m : 0;
        if (p.mode == VF_MODE_GEMM) {
            r.pix_base = mm;
            r.oy = r.ox = 0;
            r.img = has_pro ? mm / p.pro_rows_per_img : 0;
        } else {
            const int img = mm / HWo;
            const int rem = mm - img * HWo;
            r.oy = rem / p.Wout;
            r.ox = rem - r.oy * p.Wout;
            r.pix_base = img * p.Hin * p.Win;
            r.img = img;
        }
        rows[q] = r;
    }

    f32x4 areg[AP];
    f32x4 breg[B_F4];
    f32x4 pmean[AP], pscale[AP], pbeta;
    bool aok[AP];

    auto load_stage = [&](int chunk, int tap) {
        const int c0 = chunk * CK + a_col4 * 4;
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const RowInfo r = rows[q];
            bool ok = r.valid;
            size_t off;
            if (p.mode == VF_MODE_GEMM) {
                off = (size_t)r.pix_base * p.lda + c0;
            } else {
                const int dy = tap / 3, dx = tap - dy * 3;
                int iy, ix;
                if (p.mode == VF_MODE_CONV3_S1) {
                    iy = r.oy + dy - 1; ix = r.ox + dx - 1;
                    ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                } else if (p.mode == VF_MODE_CONV3_S2PAD) {
                    iy = 2 * r.oy + dy; ix = 2 * r.ox + dx;
                    ok = ok && iy < p.Hin && ix < p.Win;
                } else {  // nearest x2 upsample folded into the gather
                    const int uy = r.oy + dy - 1, ux = r.ox + dx - 1;
                    ok = ok && uy >= 0 && uy < p.Hout && ux >= 0 && ux < p.Wout;
                    iy = uy >> 1; ix = ux >> 1;
                }
                off = ((size_t)(r.pix_base + iy * p.Win + ix)) * p.Cin + c0;
            }
            aok[q] = ok;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(X + off);
            areg[q] = v;
            if (has_pro && tap == 0) {
                const size_t po = (size_t)r.img * p.Cin + c0;
                pmean[q] = *reinterpret_cast<const f32x4*>(p.pro_mean + po);
                pscale[q] = *reinterpret_cast<const f32x4*>(p.pro_scale + po);
            }
        }
        if (has_pro && tap == 0) pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + c0);
        constexpr int SUB = PBN / BN;                       // tile blocks per packed block
        const int nbp = (p.Cout + PBN - 1) / PBN;
        const float* wsrc = Wp + ((size_t)(chunk * taps + tap) * nbp + nblk / SUB) * (CK * PBN) + (nblk % SUB) * BN * 4;
#pragma unroll
        for (int q = 0; q < B_F4; ++q) {
            const int li = tid + 256 * q;                   // float4 index inside the [8][BN] tile
            breg[q] = *reinterpret_cast<const f32x4*>(wsrc + ((size_t)(li / BN) * PBN + (li % BN)) * 4);
        }
    };

    auto store_stage = [&](int buf) {
        float* a_dst = As + buf * (BM * A_LD);
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            f32x4 v = areg[q];
            if (has_pro) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = (v[e] - pmean[q][e]) * pscale[q][e] + pbeta[e];
                    if (p.pro_swish) t = vf_swish(t);
                    v[e] = aok[q] ? t : 0.f;
                }
            }
            *reinterpret_cast<f32x4*>(a_dst + (a_row0 + 32 * q) * A_LD + a_col4 * 4) = v;
        }
        float* b_dst = Bs + buf * (CK * BN);
#pragma unroll
        for (int q = 0; q < B_F4; ++q)
            *reinterpret_cast<f32x4*>(b_dst + (size_t)(tid + 256 * q) * 4) = breg[q];
    };

    f32x16 acc[WM_T][WN_T];
#pragma unroll
    for (int i = 0; i < WM_T; ++i)
#pragma unroll
        for (int j = 0; j < WN_T; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int chunk = 0, tap = 0;
    load_stage(0, 0);
    store_stage(0);
    __syncthreads();

    for (int s = 0; s < nstages; ++s) {
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == taps) { ntap = 0; nchunk = chunk + 1; }
        const bool more = (s + 1) < nstages;
        if (more) load_stage(nchunk, ntap);

        const float* a_src = As + (s & 1) * (BM * A_LD) + (wave_m * WM_T * 32 + l31) * A_LD + half * 4;
        const float* b_src = Bs + (s & 1) * (CK * BN) + (half * BN + wave_n * WN_T * 32 + l31) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a[WM_T], b[WN_T];
#pragma unroll
            for (int i = 0; i < WM_T; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_src + i * 32 * A_LD + g * 8);
#pragma unroll
            for (int j = 0; j < WN_T; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_src + (g * 2 * BN + j * 32) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WM_T; ++i)
#pragma unroll
                    for (int j = 0; j < WN_T; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_stage((s + 1) & 1);
        __syncthreads();
        chunk = nchunk; tap = ntap;
    }

    // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -----------
#pragma unroll
    for (int j = 0; j < WN_T; ++j) {
        const int n = nblk * BN + (wave_n * WN_T + j) * 32 + l31;
        const bool nok = n < p.Cout;
        const float bias = (nok && p.bias) ? p.bias[n] : 0.f;
        // batched tile epilogue (epilogue.h): one basic block per tile, no per-element vmcnt(0) round trips.
        // Columns past Cout (padded weight tiles) are steered to rows_left = 0.
        const long long ldc = p.ldc, ldr = p.ldr;
        const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
#pragma unroll
        for (int i = 0; i < WM_T; ++i) {
            const int m0 = mtile * BM + (wave_m * WM_T + i) * 32 + 4 * half;
            const int nn = nok ? n : 0;
            float* o = Out + (size_t)(m0 < p.M ? m0 : 0) * ldc + nn;
            const float* rs = Res ? Res + (size_t)(m0 < p.M ? m0 : 0) * ldr + nn : nullptr;
            const int rows_left = nok ? p.M - m0 : 0;
            const bool full = (mtile * BM + BM <= p.M) && (nblk * BN + BN <= p.Cout);   // workgroup-uniform
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

__global__ void pack_b_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int N, int taps,
                              long long sk, long long sn, long long st, int BN, int nb, int nchunks,
                              long long src_bstride, long long dst_bstride) {
    const long long total = (long long)nchunks * taps * nb * CK * BN;
    const int bz = blockIdx.y;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        long long t = idx >> 2;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int g = (int)((t >> 1) & 3);
        t >>= 3;
        const int nblk = (int)(t % nb); t /= nb;
        const int tap = (int)(t % taps);
        const int chunk = (int)(t / taps);
        const int k = chunk * CK + g * 8 + half * 4 + e;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (k < K && n < N) v = src[(size_t)bz * src_bstride + k * sk + n * sn + tap * st];
        dst[(size_t)bz * dst_bstride + idx] = v;
    }
}

template <int WAVES_M, int WAVES_N, int WM_T, int WN_T, int PBN = WAVES_N * WN_T * 32>
int launch_igemm(const vf_igemm_args& a, hipStream_t stream) {
    constexpr int BN = WAVES_N * WN_T * 32;
    constexpr int BM = WAVES_M * WM_T * 32;
    const size_t smem = (size_t)(2 * BM * A_LD + 2 * CK * BN) * sizeof(float);
    auto kern = igemm_f32_kernel<WAVES_M, WAVES_N, WM_T, WN_T, PBN>;
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const int nb = (a.Cout + BN - 1) / BN;
    const int mt = (a.M + BM - 1) / BM;
    dim3 grid((unsigned)(mt * nb), 1, (unsigned)(a.batch > 0 ? a.batch : 1));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
    return vf_last_status();
}

}  // namespace

int vf_pack_b_impl(const float* src, float* dst, int K, int N, int taps, long long sk, long long sn, long long st,
                   int BN, int batch, long long src_bstride, hipStream_t stream) {
    const int nb = (N + BN - 1) / BN;
    const int nchunks = (K + CK - 1) / CK;
    const long long total = (long long)nchunks * taps * nb * CK * BN;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_b_kernel, dim3(blocks, batch), dim3(256), 0, stream, src, dst, K, N, taps, sk, sn, st, BN,
                       nb, nchunks, src_bstride, total);
    return vf_last_status();
}

extern "C" {

int vf_abi_version(void) { return 19; }

// ---- kernel selection (include/vf_hip.h): process-wide switches between kernels whose results the tests assert BIT-IDENTICAL.  The library
// reads no environment variable (tests/test_abi.py checks that it does not even import the libc call); a host that wants an environment override
// translates it into vf_select calls (viewformer_amd/_lib.py does, once, at load).
static int g_vf_select[VF_SEL_COUNT] = {1, 1, 1, 1, 1, 0};      // (VF_SEL_GEMM_TAIL: off — faster alone, not in the step: profiles/r6_gemm_tail_ab.txt)
int vf_select(int which, int value) {
    if (which < 0 || which >= VF_SEL_COUNT || (value != 0 && value != 1)) return VF_ERR_BAD_ARG;
    return __atomic_exchange_n(&g_vf_select[which], value, __ATOMIC_RELAXED);
}
int vf_selected(int which) {
    if (which < 0 || which >= VF_SEL_COUNT) return VF_ERR_BAD_ARG;
    return __atomic_load_n(&g_vf_select[which], __ATOMIC_RELAXED);
}
// sizes of the structs that cross the boundary by pointer: a binding checks its mirror against them (a silent mismatch would be memory corruption)
size_t vf_sizeof_igemm_args(void) { return sizeof(vf_igemm_args); }
size_t vf_sizeof_pack_desc(void) { return sizeof(vf_pack_desc); }
const char* vf_build_arch(void) { return "gfx950"; }

size_t vf_igemm_packed_floats(int K, int N, int taps) {
    if (K <= 0 || N <= 0 || taps <= 0) return 0;
    const int BN = bn_for(N);
    const size_t nb = (size_t)(N + BN - 1) / BN;
    const size_t nchunks = (size_t)(K + CK - 1) / CK;
    return nchunks * taps * nb * CK * BN;
}

int vf_igemm_pack_f32(const float* src, float* dst, int K, int N, int taps, int64_t sk, int64_t sn, int64_t st,
                      int batch, int64_t src_bstride, void* stream) {
    if (!src || !dst || K <= 0 || N <= 0 || (taps != 1 && taps != 9) || batch < 1) return VF_ERR_BAD_ARG;
    return vf_pack_b_impl(src, dst, K, N, taps, sk, sn, st, bn_for(N), batch, src_bstride, (hipStream_t)stream);
}

int vf_igemm_f32(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0 || a.Cin <= 0 || a.Cout <= 0) return VF_ERR_BAD_ARG;
    if (a.epilogue != VF_EPI_NONE && a.epilogue != VF_EPI_GELU_ERF) return VF_ERR_UNSUPPORTED;      // (VF_EPI_GELU_BWD: vf_gemm_bf16 only)
    if (a.gn_part) return VF_ERR_UNSUPPORTED;        // fused GroupNorm statistics: halo-tile kernels only
    if (a.drop_rate != 0.f || a.out_aux) return VF_ERR_UNSUPPORTED;      // fused output dropout / second output: vf_gemm_bf16 only
    if (a.Cin % CK != 0) return VF_ERR_UNSUPPORTED;
    if (a.mode < VF_MODE_GEMM || a.mode > VF_MODE_CONV3_UP2) return VF_ERR_BAD_ARG;
    if (a.mode != VF_MODE_GEMM) {
        if (a.Hin <= 0 || a.Win <= 0 || a.Hout <= 0 || a.Wout <= 0) return VF_ERR_BAD_ARG;
        if (a.M % (a.Hout * a.Wout) != 0) return VF_ERR_BAD_ARG;
        if (a.mode == VF_MODE_CONV3_S1 && (a.Hout != a.Hin || a.Wout != a.Win)) return VF_ERR_BAD_ARG;
        if (a.mode == VF_MODE_CONV3_S2PAD && (a.Hout != a.Hin / 2 || a.Wout != a.Win / 2)) return VF_ERR_BAD_ARG;
        if (a.mode == VF_MODE_CONV3_UP2 && (a.Hout != a.Hin * 2 || a.Wout != a.Win * 2)) return VF_ERR_BAD_ARG;
    } else {
        if (a.lda < a.Cin || (a.lda & 3)) return VF_ERR_BAD_ARG;
    }
    if (a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
    const bool pro = a.pro_mean || a.pro_scale || a.pro_beta;
    if (pro && !(a.pro_mean && a.pro_scale && a.pro_beta)) return VF_ERR_BAD_ARG;
    if (pro && a.mode == VF_MODE_GEMM && a.pro_rows_per_img <= 0) return VF_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    {
        // 3x3 stride-1 / upsample layers with wide channels go to the halo-tile kernel (conv3_halo_f32.hip)
        int st = 0;
        if (vf_conv3_halo_try(a, s, &st) == 0) return st;
    }
    const int BN = bn_for(a.Cout);
    if (BN == 128) {
        // under-filled grid (few rows, e.g. the decoder's 8x8 maps: 2048 rows x 512 channels = 64 tiles of 128x128 on
        // 256 CUs): the 64x64-tile variant quadruples the workgroup count
        const long long tiles128 = (long long)((a.M + 127) / 128) * ((a.Cout + 127) / 128) * (a.batch > 0 ? a.batch : 1);
        if (tiles128 < 384 && a.Cout % 64 == 0) return launch_igemm<2, 2, 1, 1, 128>(a, s);
        return launch_igemm<2, 2, 2, 2>(a, s);
    }
    if (BN == 64) return launch_igemm<4, 1, 1, 2>(a, s);
    return launch_igemm<4, 1, 1, 1>(a, s);
}

}  // extern "C"
