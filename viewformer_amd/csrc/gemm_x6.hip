// fp32-EQUIVALENT dense / 1x1 GEMM on the bf16 matrix pipe ("x6"), gfx950 — the dense sibling of conv3_halo_x6.hip.
//
//   out[m][n] = epi( sum_k A[m][k] * W[k][n] + bias[n] ) + res[m][n]
//
// with every fp32 product evaluated as the six bf16 partial products of the exact splits a = ah+am+al, w = wh+wm+wl
// (al*wh + ah*wl + am*wm + am*wh + ah*wm + ah*wh, fp32 accumulation, small terms first): the same error against fp64 as
// the native f32 MFMA (tests/test_hip_x6.py), at ~1.7x its speed.
// 128x128 tile per 256-thread workgroup (2x2 waves of 64x64), K in chunks of 32: the fp32 A tile is loaded as float4,
// optionally pushed through the GroupNorm-apply(+swish) prologue, split ONCE and parked in LDS as
// [row][plane h|m|l][32 k] bf16 with a 208-byte row stride (conflict-free ds_read_b128); the weights are pre-split at pack
// time into three fragment-packed planes ([k-chunk][n-block][k-step(2)][plane(3)][half(2)][n(128)][8]) and streamed
// L2 -> VGPR one whole chunk ahead (sched_barrier-pinned, like the conv kernel).  Double-buffered LDS, one barrier per
// chunk (48 MFMAs per wave).
// Replaces: Conv1D.call (migt.py:89-96), SharedEmbeddings._linear (:51-56), gelu (:70), the 1x1 convolutions of
// vqgan_th.py (:72-76 nin_shortcut, :99-118 AttnBlock q/k/v/proj_out, :332-333 quant_conv / post_quant_conv).
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 32;
constexpr int BM = 128, BN = 128;
constexpr int A_LDB = 208;                        // bytes per A row in LDS: 3 planes x 64 B + 16 B pad
constexpr int A_BYTES = BM * A_LDB;               // 26624
constexpr int PLANE_BYTES = 2 * BN * 16;          // one (k-step, plane): [half(2)][n(128)][8 bf16] = 4 KB
constexpr int KS_BYTES = 3 * PLANE_BYTES;
constexpr int CHUNK_BYTES = 2 * KS_BYTES;         // one (k-chunk, n-block): 24 KB

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

template <bool PRO, bool SWISH>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];   // [2][A_BYTES]

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
    f32x4 pmean[4], pscale[4];
    auto a_fetch = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) areg[q] = *reinterpret_cast<const f32x4*>(arow[q] + chunk * CK);
        if (PRO) {
            pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pmean[q] = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)aimg[q] * p.Cin + chunk * CK + c4 * 4);
                pscale[q] = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)aimg[q] * p.Cin + chunk * CK + c4 * 4);
            }
        }
    };
    auto a_park = [&](int buf, int q) {
        unsigned char* dst = smem_g + buf * A_BYTES + (r0 + 32 * q) * A_LDB + c4 * 8;
        bf16x4 oh, om, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = areg[q][e];
            if (PRO) {
                t = (t - pmean[q][e]) * pscale[q][e] + pbeta[e];
                if (SWISH) t = vf_swish_1ulp(t);
            }
            __bf16 h, m, l;
            split3(t, h, m, l);
            oh[e] = h; om[e] = m; ol[e] = l;
        }
        *reinterpret_cast<bf16x4*>(dst) = oh;
        *reinterpret_cast<bf16x4*>(dst + 64) = om;
        *reinterpret_cast<bf16x4*>(dst + 128) = ol;
    };

    const int b_lane = (half * BN + wave_n * 64 + l31) * 16;
    bf16x8 bring[2][2][3][2];          // [chunk parity][ks][plane][j]
    auto b_load = [&](bf16x8 (&dst)[2][3][2], int chunk) {
        chunk = min(chunk, nchunks - 1);
        const unsigned char* src = Wb + (size_t)chunk * chunk_stride + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    dst[ks][pl][j] = *reinterpret_cast<const bf16x8*>(src + ks * KS_BYTES + pl * PLANE_BYTES + j * 32 * 16);
    };
    const int a_lane = (wave_m * 64 + l31) * A_LDB + half * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    a_fetch(min(c0, total_chunks - 1));
    b_load(bring[0], c0);
#pragma unroll
    for (int q = 0; q < 4; ++q) a_park(0, q);
    __syncthreads();

    auto chunk_body = [&](int chunk, bf16x8 (&bcur)[2][3][2], bf16x8 (&bnext)[2][3][2]) {
        const unsigned char* a_src = smem_g + ((chunk - c0) & 1) * A_BYTES + a_lane;
        a_fetch(min(chunk + 1, nchunks - 1));
        b_load(bnext, chunk + 1);
        bf16x8 a[2][2][3];             // [ks][mi][plane]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[ks][mi][pl] = *reinterpret_cast<const bf16x8*>(a_src + mi * 32 * A_LDB + pl * 64 + ks * 32);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};       // plane 0 = h, 1 = m, 2 = l; smallest products first
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][mi][PA[t]], bcur[ks][PB[t]][j], acc[mi][j], 0, 0, 0);
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

    float* __restrict__ Out = p.out + (size_t)blockIdx.y * p.stride_out;
    const float* __restrict__ Res = p.res;
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

// fp32 [K][N] (strided) -> three bf16 planes, fragment-major [K/32][nb][ks(2)][plane(3)][half(2)][n(128)][8]
__global__ void pack_gemm_x6_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int K, int N, long long sk,
                                    long long sn, int nb, int nchunks) {
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
        if (k < K && n < N) v = src[k * sk + n * sn];
        __bf16 h, m, l;
        split3(v, h, m, l);
        const size_t base = ((((size_t)chunk * nb + nblk) * 2 + ks) * 3) * (2 * BN * 8) + ((size_t)half * BN + nl) * 8 + e;
        dst[base] = h;
        dst[base + 2 * BN * 8] = m;
        dst[base + 2 * (2 * BN * 8)] = l;
    }
}

__global__ void sum_slabs_kernel(const float* __restrict__ slabs, int nslabs, long long stride, long long n4,
                                 float* __restrict__ dst, int accumulate) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        // eight slabs' loads in flight, added in slab order (the sums are the one-load-at-a-time loop's bit for bit: that loop's trip count is a
        // run-time value, so the compiler kept it rolled — six dependent load latencies per element for the training step's five slabs + dst)
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        if (accumulate) acc = *reinterpret_cast<const f32x4*>(dst + i * 4);
        for (int s0 = 0; s0 < nslabs; s0 += 8) {
            f32x4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nslabs) x[u] = *reinterpret_cast<const f32x4*>(slabs + (s0 + u) * stride + i * 4);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + u < nslabs) acc += x[u];
        }
        *reinterpret_cast<f32x4*>(dst + i * 4) = acc;
    }
}

// ---- weight gradient of a 3x3 convolution as ONE GEMM: dW[(tap, ci)][co] = sum_p X[p + tap][ci] * dY[p][co] --------------------
// Same tile / pipeline as gemm_x6_kernel with K = pixels; the A operand is gathered straight from the NHWC activation: workgroup
// row tile = 128 input channels of one tap, thread = (channel r, pixel quad): 4 dword loads along consecutive output pixels of one
// channel (a wavefront reads 64 consecutive channels of a pixel = 256 contiguous bytes), so the values arrive k-contiguous and are
// split and parked exactly like the dense kernel's float4 — no transposed copy of the activation in HBM (the first version's
// vf_gather_transpose_f32 x 9 cost as much as all the GEMM work).  One extra row tile (tap 9) has a single row of ones: its output
// row is the bias gradient sum_p dY[p][co].  Split-K slabs as in gemm_x6_kernel.
struct wgrad_args {
    const float* x;
    const void* dyp;
    float* out;
    int n_img, Hin, Win, Cin, Hout, Wout, Cout, lw, lhw, nsplit, M;
    long long P, stride_out;
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void wgrad_x6_kernel(wgrad_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];   // [2][A_BYTES]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = (p.Cout + BN - 1) / BN;
    const unsigned lbid = vf_xcd_bid();                 // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = lbid % nb;
    const int mtile = lbid / nb;
    const int tpt = p.Cin / BM;                      // row tiles per tap
    const int tap = mtile / tpt, cblk = mtile % tpt; // tap == 9: the ones row (bias gradient)
    const int ky = tap / 3, kx = tap % 3;
    const bool ones_tile = tap >= 9;
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.dyp) + (size_t)nblk * CHUNK_BYTES;
    const size_t chunk_stride = (size_t)nb * CHUNK_BYTES;
    const int nsplit = p.nsplit;
    const int total_chunks = (int)(p.P / CK);
    const int per = ((total_chunks + nsplit - 1) / nsplit + 1) & ~1;
    const int c0 = min((int)blockIdx.y * per, total_chunks);
    const int nchunks = min(c0 + per, total_chunks);

    const int r = tid & 127, kg = tid >> 7;
    const float* __restrict__ xc = p.x + cblk * BM + r;
    f32x4 areg[4];
    auto a_fetch = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (ones_tile) {
                const float o = r == 0 ? 1.f : 0.f;
                areg[q] = f32x4{o, o, o, o};
                continue;
            }
            const int p0 = chunk * CK + 4 * (kg + 2 * q);                     // 4 consecutive output pixels of one row
            const int x0 = p0 & (p.Wout - 1), y = (p0 >> p.lw) & (p.Hout - 1), img = p0 >> p.lhw;
            int sy;
            bool oky;
            if (MODE == VF_MODE_CONV3_S1) { sy = y + ky - 1; oky = sy >= 0 && sy < p.Hin; }
            else if (MODE == VF_MODE_CONV3_S2PAD) { sy = 2 * y + ky; oky = sy < p.Hin; }
            else { const int uy = y + ky - 1; oky = uy >= 0 && uy < p.Hout; sy = uy >> 1; }
            const long long rowbase = ((long long)img * p.Hin + (oky ? sy : 0)) * p.Win;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int sx;
                bool okx;
                if (MODE == VF_MODE_CONV3_S1) { sx = x0 + e + kx - 1; okx = sx >= 0 && sx < p.Win; }
                else if (MODE == VF_MODE_CONV3_S2PAD) { sx = 2 * (x0 + e) + kx; okx = sx < p.Win; }
                else { const int ux = x0 + e + kx - 1; okx = ux >= 0 && ux < p.Wout; sx = ux >> 1; }
                const bool ok = oky && okx;
                const float v = xc[(rowbase + (ok ? sx : 0)) * p.Cin];
                areg[q][e] = ok ? v : 0.f;
            }
        }
    };
    auto a_park = [&](int buf, int q) {
        unsigned char* dst = smem_g + buf * A_BYTES + r * A_LDB + (kg + 2 * q) * 8;
        bf16x4 oh, om, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, m, l;
            split3(areg[q][e], h, m, l);
            oh[e] = h; om[e] = m; ol[e] = l;
        }
        *reinterpret_cast<bf16x4*>(dst) = oh;
        *reinterpret_cast<bf16x4*>(dst + 64) = om;
        *reinterpret_cast<bf16x4*>(dst + 128) = ol;
    };

    const int b_lane = (half * BN + wave_n * 64 + l31) * 16;
    bf16x8 bring[2][2][3][2];
    auto b_load = [&](bf16x8 (&dst)[2][3][2], int chunk) {
        chunk = min(chunk, nchunks - 1);
        const unsigned char* src = Wb + (size_t)chunk * chunk_stride + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    dst[ks][pl][j] = *reinterpret_cast<const bf16x8*>(src + ks * KS_BYTES + pl * PLANE_BYTES + j * 32 * 16);
    };
    const int a_lane = (wave_m * 64 + l31) * A_LDB + half * 16;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = 0.f;

    if (c0 < nchunks) {
        a_fetch(c0);
        b_load(bring[0], c0);
#pragma unroll
        for (int q = 0; q < 4; ++q) a_park(0, q);
    }
    __syncthreads();

    auto chunk_body = [&](int chunk, bf16x8 (&bcur)[2][3][2], bf16x8 (&bnext)[2][3][2]) {
        const unsigned char* a_src = smem_g + ((chunk - c0) & 1) * A_BYTES + a_lane;
        a_fetch(min(chunk + 1, nchunks - 1));
        b_load(bnext, chunk + 1);
        bf16x8 a[2][2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[ks][mi][pl] = *reinterpret_cast<const bf16x8*>(a_src + mi * 32 * A_LDB + pl * 64 + ks * 32);
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
        constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks][mi][PA[t]], bcur[ks][PB[t]][j], acc[mi][j], 0, 0, 0);
            a_park((chunk - c0 + 1) & 1, ks * 2);
            a_park((chunk - c0 + 1) & 1, ks * 2 + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    };
    for (int chunk = c0; chunk < nchunks; chunk += 2) {
        chunk_body(chunk, bring[0], bring[1]);
        chunk_body(chunk + 1, bring[1], bring[0]);
    }

    float* __restrict__ Out = p.out + (size_t)blockIdx.y * p.stride_out;
    const bool full = (mtile * BM + BM <= p.M) && (nblk * BN + BN <= p.Cout);
    const long long ldc = p.Cout;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nblk * BN + wave_n * 64 + j * 32 + l31;
        const bool nok = n < p.Cout;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m0 = mtile * BM + wave_m * 64 + i * 32 + 4 * half;
            float* o = Out + (size_t)(m0 < p.M ? m0 : 0) * ldc + (nok ? n : 0);
            const int rows_left = nok ? p.M - m0 : 0;
            if (full) {
                auto oo = [&](int rr) { return (long long)((rr & 3) + 8 * (rr >> 2)) * ldc; };
                vf_store_tile<0, false>(acc[i][j], 0.f, o, nullptr, oo, oo);
            } else {
                vf_store_tile_ragged<0, false>(acc[i][j], 0.f, o, nullptr, ldc, ldc, rows_left);
            }
        }
    }
}

template <bool PRO, bool SWISH>
int launch(const vf_igemm_args& a, hipStream_t stream) {
    const size_t smem = (size_t)2 * A_BYTES;
    const int nb = (a.Cout + BN - 1) / BN, mt = (a.M + BM - 1) / BM;
    const int nsplit = a.reserved0 > 1 ? a.reserved0 : 1;
    hipLaunchKernelGGL((gemm_x6_kernel<PRO, SWISH>), dim3((unsigned)(mt * nb), (unsigned)nsplit), dim3(256), smem, stream, a);
    return vf_last_status();
}

}  // namespace

extern "C" {

size_t vf_gemm_x6_packed_elems(int K, int N) {
    if (K <= 0 || N <= 0) return 0;
    return (size_t)((K + CK - 1) / CK) * ((N + BN - 1) / BN) * CK * BN * 3;
}

int vf_gemm_x6_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, void* stream) {
    if (!src || !dst || K <= 0 || N <= 0) return VF_ERR_BAD_ARG;
    const int nb = (N + BN - 1) / BN, nchunks = (K + CK - 1) / CK;
    const long long total = (long long)nchunks * nb * CK * BN;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_gemm_x6_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, K, N,
                       (long long)sk, (long long)sn, nb, nchunks);
    return vf_last_status();
}

int vf_sum_slabs_f32(const float* slabs, int nslabs, int64_t stride, int64_t n, float* dst, int accumulate, void* stream) {
    if (!slabs || !dst || nslabs <= 0 || n <= 0 || (n & 3) || (stride & 3)) return VF_ERR_BAD_ARG;
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslabs,
                       (long long)stride, n4, dst, accumulate);
    return vf_last_status();
}

size_t vf_conv3_wgrad_x6_rows(int Cin) { return Cin > 0 ? (size_t)9 * Cin + 1 : 0; }

int vf_conv3_wgrad_x6(const float* x, const void* dy_packed, float* slabs, int n_img, int Hin, int Win, int Cin, int Hout, int Wout,
                      int Cout, int mode, int splits, void* stream) {
    if (!x || !dy_packed || !slabs || n_img <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Hout <= 0 || Wout <= 0 || Cout <= 0 ||
        splits < 1)
        return VF_ERR_BAD_ARG;
    if (mode == VF_MODE_CONV3_S1 && (Hout != Hin || Wout != Win)) return VF_ERR_BAD_ARG;
    if (mode == VF_MODE_CONV3_S2PAD && (Hout != Hin / 2 || Wout != Win / 2)) return VF_ERR_BAD_ARG;
    if (mode == VF_MODE_CONV3_UP2 && (Hout != Hin * 2 || Wout != Win * 2)) return VF_ERR_BAD_ARG;
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    const long long P = (long long)n_img * Hout * Wout;
    if (Cin % BM || !pow2(Hout) || !pow2(Wout) || (Wout & 3) || P % 64 || P > 0x7fffffffLL) return VF_ERR_UNSUPPORTED;
    wgrad_args a;
    a.x = x; a.dyp = dy_packed; a.out = slabs;
    a.n_img = n_img; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = Hout; a.Wout = Wout; a.Cout = Cout;
    a.lw = __builtin_ctz((unsigned)Wout); a.lhw = a.lw + __builtin_ctz((unsigned)Hout);
    a.nsplit = splits; a.M = 9 * Cin + 1; a.P = P; a.stride_out = (long long)a.M * Cout;
    const int nb = (Cout + BN - 1) / BN, mt = 9 * (Cin / BM) + 1;
    const size_t smem = (size_t)2 * A_BYTES;
    const dim3 grid((unsigned)(mt * nb), (unsigned)splits);
    hipStream_t s = (hipStream_t)stream;
    if (mode == VF_MODE_CONV3_S1) hipLaunchKernelGGL(wgrad_x6_kernel<VF_MODE_CONV3_S1>, grid, dim3(256), smem, s, a);
    else if (mode == VF_MODE_CONV3_S2PAD) hipLaunchKernelGGL(wgrad_x6_kernel<VF_MODE_CONV3_S2PAD>, grid, dim3(256), smem, s, a);
    else if (mode == VF_MODE_CONV3_UP2) hipLaunchKernelGGL(wgrad_x6_kernel<VF_MODE_CONV3_UP2>, grid, dim3(256), smem, s, a);
    else return VF_ERR_BAD_ARG;
    return vf_last_status();
}

int vf_gemm_x6(const vf_igemm_args* args, void* stream) {
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
    return a.pro_swish ? launch<true, true>(a, s) : launch<true, false>(a, s);
}

}  // extern "C"
