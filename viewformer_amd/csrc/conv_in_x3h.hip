// encoder.conv_in (3 -> Cout 3x3 convolution on the pre-processed image) on the fp16 matrix pipe with the x3h arithmetic of
// conv3_halo_x3h.hip (fp32-equivalent: two fp16 pieces per operand, low piece carried at 2^11, three products, cross terms in their
// own accumulator, weights pre-scaled by a power of two).  The inputs are pixels in [-1, 1] — squarely inside the arithmetic's range.
// The VALU form (conv_in_kernel, misc.hip) spends 432 FMAs per pixel-thread with its weights in LDS and writes 1.8 TB/s; here the
// 27-deep reduction is padded to K = 48 = 9 taps x (3 channels + 1 zero) + 3 zero taps (three 16-deep MFMA steps, 36 MFMAs per wave
// for 32 pixels x 128 channels) and the kernel is bound by its 512 B/pixel of output.  Same 8x16-pixel tile, pixel permutation and
// epilogue (bias, store, optional fused GroupNorm partial statistics of the output) as the halo-tile kernels.
// Replaces: tf.image.convert_image_dtype * 2 - 1 (evaluate_transformer.py:105-108) + Encoder.conv_in (vqgan_th.py:161-165).
#include "halo_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8, TW = 16, BN = 128;
constexpr int PW = TW + 2, PH = TH + 2, NPIX = PH * PW;        // 10 x 18 patch
constexpr int P_LDB = 16;                                      // bytes per patch pixel: [h: 4 f16 | l': 4 f16], 4th channel zero
constexpr int PLANE_BYTES = 2 * BN * 16;                       // [half(2)][n(128)][8 f16]
constexpr int KS_BYTES = 2 * PLANE_BYTES;                      // 2 planes
constexpr int BLK_BYTES = 3 * KS_BYTES;                        // 3 k-steps: 24 KB per 128 output channels
constexpr int TAIL_BYTES = 16;

__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * 2048.f);
}

template <bool U8>
__global__ __launch_bounds__(256) void conv_in_x3h_kernel(vf_igemm_args p, const unsigned char* __restrict__ img_u8) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[NPIX * P_LDB];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * TH, x0 = tx * TW;

    // ---- stage the 10x18 patch: pre-process (uint8 -> [-1, 1] exactly as conv_in_kernel), split, park ------------------------
    if (tid < NPIX) {
        const int pr = tid / PW, pc = tid - pr * PW;
        const int sy = y0 - 1 + pr, sx = x0 - 1 + pc;
        const bool ok = sy >= 0 && sy < p.Hout && sx >= 0 && sx < p.Wout;
        const size_t px = ((size_t)img * p.Hout + (ok ? sy : 0)) * p.Wout + (ok ? sx : 0);
        f16x4 oh, ol;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v;
            if (U8) v = ((float)img_u8[px * 3 + c] * (1.0f / 255.0f)) * 2.0f - 1.0f;      // TF convert_image_dtype, then * 2 - 1
            else v = p.x[px * 3 + c];
            _Float16 h, l;
            split2(ok ? v : 0.f, h, l);
            oh[c] = h; ol[c] = l;
        }
        oh[3] = (_Float16)0.f; ol[3] = (_Float16)0.f;
        *reinterpret_cast<f16x4*>(patch + tid * P_LDB) = oh;
        *reinterpret_cast<f16x4*>(patch + tid * P_LDB + 8) = ol;
    }

    // ---- weight fragments: [nblk][ks(3)][plane(2)][half(2)][n(128)][8] -------------------------------------------------------
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * BLK_BYTES +
                                           (half * BN + wave_n * 64 + l31) * 16;
    f16x8 b[3][2][2];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) b[ks][pl][j] = *reinterpret_cast<const f16x8*>(Wb + ks * KS_BYTES + pl * PLANE_BYTES + j * 32 * 16);

    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }
    __syncthreads();

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
        // lane's 8 k-values of this step: taps 2g and 2g + 1 (4 channels each), g = ks * 2 + half; taps >= 9 are zero padding
        const int g = ks * 2 + half;
        f16x8 a[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int a0 = wave_m * 4 + mi * 2 + trow;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                f16x4 lo = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}, hi = lo;
                const int t0 = 2 * g, t1 = 2 * g + 1;
                if (t0 < 9) lo = *reinterpret_cast<const f16x4*>(patch + ((a0 + t0 / 3) * PW + tpx + t0 % 3) * P_LDB + pl * 8);
                if (t1 < 9) hi = *reinterpret_cast<const f16x4*>(patch + ((a0 + t1 / 3) * PW + tpx + t1 % 3) * P_LDB + pl * 8);
                a[mi][pl] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][1], b[ks][0][j], accx[mi][j], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ks][1][j], accx[mi][j], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], b[ks][0][j], acc[mi][j], 0, 0, 0);
    }

    const float inv_s = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nb * BLK_BYTES);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accx[i][j][r], 4.8828125e-4f, acc[i][j][r]) * inv_s;
    vf_halo_epilogue<false>(p, acc, img, img, y0, x0, (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
}

__global__ void absmax27_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    vf_block_max_atomic(m, out);
}

// OIHW [Cout][3][3][3] -> [nblk][ks(3)][plane(2)][half(2)][n(128)][8]; k = (ks*2 + half)*8 + e -> tap = k / 4, channel = k % 4
__global__ void pack_conv_in_x3h_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int Cout, int nb,
                                        unsigned char* __restrict__ tail) {
    const float amax = __uint_as_float(*reinterpret_cast<const unsigned*>(tail + 4));
    const int ex = amax > 0.f ? ilogbf(amax) : 13;
    const float S = ldexpf(1.f, 13 - ex);
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(tail) = ldexpf(1.f, ex - 13);
    const int total = nb * 3 * 2 * BN * 8;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7;
        int t = idx >> 3;
        const int nl = t % BN; t /= BN;
        const int half = t & 1;
        const int ks = (t >> 1) % 3;
        const int nblk = (t >> 1) / 3;
        const int k = (ks * 2 + half) * 8 + e;
        const int tap = k >> 2, c = k & 3;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (tap < 9 && c < 3 && n < Cout) v = w[((size_t)n * 3 + c) * 9 + tap] * S;
        _Float16 h, l;
        split2(v, h, l);
        const size_t base = (((size_t)nblk * 3 + ks) * 2) * (2 * BN * 8) + ((size_t)half * BN + nl) * 8 + e;
        dst[base] = h;
        dst[base + 2 * BN * 8] = l;
    }
}

}  // namespace

extern "C" {

size_t vf_conv_in_x3h_packed_elems(int Cout) {
    if (Cout <= 0 || Cout % BN) return 0;
    return (size_t)(Cout / BN) * 3 * 2 * 2 * BN * 8 + TAIL_BYTES / 2;
}

int vf_conv_in_x3h_pack(const float* w_oihw, void* dst, int Cout, void* stream) {
    if (!w_oihw || !dst || Cout <= 0) return VF_ERR_BAD_ARG;
    if (Cout % BN) return VF_ERR_UNSUPPORTED;
    const int nb = Cout / BN;
    unsigned char* tail = reinterpret_cast<unsigned char*>(dst) + (size_t)nb * BLK_BYTES;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(tail, 0, TAIL_BYTES, s) != hipSuccess) return vf_last_status();
    hipLaunchKernelGGL(absmax27_kernel, dim3(16), dim3(256), 0, s, w_oihw, (long long)Cout * 27, reinterpret_cast<unsigned*>(tail + 4));
    hipLaunchKernelGGL(pack_conv_in_x3h_kernel, dim3(48), dim3(256), 0, s, w_oihw, (_Float16*)dst, Cout, nb, tail);
    return vf_last_status();
}

int vf_conv_in_x3h(const uint8_t* img_u8, const float* img_f32, const void* w_packed, const float* bias, float* out, float* gn_part,
                   int gn_slots, int n_img, int H, int W, int Cout, void* stream) {
    if ((!img_u8 && !img_f32) || !w_packed || !out || n_img <= 0 || H <= 0 || W <= 0 || Cout <= 0) return VF_ERR_BAD_ARG;
    if (H % TH || W % TW || Cout % BN) return VF_ERR_UNSUPPORTED;
    vf_igemm_args a = {};
    a.x = img_f32;
    a.w_packed = reinterpret_cast<const float*>(w_packed);
    a.bias = bias;
    a.out = out;
    a.mode = VF_MODE_CONV3_S1;
    a.M = n_img * H * W; a.Cin = 3; a.Cout = Cout;
    a.Hin = a.Hout = H; a.Win = a.Wout = W;
    a.lda = 3; a.ldc = Cout; a.ldr = Cout;
    a.batch = 1;
    a.gn_part = gn_part;
    a.gn_slots = gn_slots;
    if (int st = vf_halo_gn_check(a)) return st;
    const long long blocks = (long long)n_img * (H / TH) * (W / TW) * (Cout / BN);
    if (img_u8) hipLaunchKernelGGL(conv_in_x3h_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, img_u8);
    else hipLaunchKernelGGL(conv_in_x3h_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, img_u8);
    return vf_last_status();
}

}  // extern "C"
