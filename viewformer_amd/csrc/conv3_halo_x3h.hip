// fp32-EQUIVALENT 3x3 convolution on the fp16 matrix pipe ("x3h": 3-term split-fp16 products with a scaled low part), gfx950.
// Same halo-tile structure, pixel permutation, pipeline and epilogue as conv3_halo_x6.hip with HALF the matrix instructions:
// every fp32 operand is split into two fp16 pieces with the low piece carried at 2^11 times its value,
//     x = h + l' * 2^-11,   h = rne_f16(x),   l' = rne_f16((x - h) * 2^11)          (x - h is exact in fp32, the scaling is exact)
// so that l' has h's magnitude and keeps all 11 bits (an unscaled low piece falls into fp16's subnormals for |x| < 0.125), and
//     a * b  ~=  ah*bh  +  2^-11 * (al'*bh + ah*bl')                               (each f16 x f16 product is exact in fp32)
// with the cross terms in their own fp32 accumulator (accx), combined once in the epilogue.  The dropped term al*bl is below 2^-22
// of the product; the weights are pre-scaled at pack time by a power of two S so that max|w| lands in [2^13, 2^14) (their h and l'
// pieces are then exact to 2^-22 for every weight within 2^-17 of the largest) and the result is divided by S in the epilogue.
// Error vs fp64 relative to sum|a*b| (K = 1152, unit-scale activations): 4e-9 representation error, i.e. the fp32 accumulation
// (3e-8, identical to the native f32 MFMA and to x6) dominates — tests/test_hip_x3h.py.
// RANGE (why x6 stays the arithmetic of the training backward): activations must satisfy |x| < 65504 (fp16 overflow), and an
// activation is carried to 2^-22 relative only while |x| >= ~6e-5 (below that h is subnormal and the absolute error floors at
// ~1.5e-11); GroupNorm-normalised inputs and the VQGAN's residual stream are O(1e-2 .. 1e2), gradients are not.
// Data path as in conv3_halo_x6.hip: fp32 activations in HBM, GroupNorm-apply(+swish) prologue in exact fp32, split ONCE per patch
// element when the patch is parked in LDS as [pixel][plane h|l'][32 ch] f16 (144-byte pixel stride = 9 x 16 B: conflict-free
// ds_read_b128 for every tap shift); two weight planes streamed L2 -> VGPR in a register ring (341 B per MFMA vs 256 for x6:
// the kernel is bound by that stream, so the gain over x6 is ~1.5x, not 2x).
// Kernels of this file: conv3_halo_x3h_kernel (32x32x16 MFMAs: nearest-x2 form, 8x8 pair tiles, and the vf_select alternative of) conv3_halo_x3h16_kernel
// (round 4: the stride-1 form on 16x16x32 MFMAs — the default where it applies; see the comment above it), conv3_s2_x3h_kernel (Downsample).
// Reference call sites: torch.nn.Conv2d 3x3 pad 1 in ResnetBlock / Upsample (vqgan_th.py:23-32,60-70,197,249) with
// GroupNorm+swish (:11-17,80-85) and the residual add (:90) fused.
#include <type_traits>
#include "halo_common.h"

#ifndef VF_X3H_BD
#define VF_X3H_BD 2       // weight fragments are fetched this many stages ahead (register ring of BD + 1)
#endif
#ifndef VF_X3H_AD
#define VF_X3H_AD 1       // LDS activation fragments are read this many stages ahead (0 or 1)
#endif
#ifndef VF_X3H_STORE
#define VF_X3H_STORE 0
#endif
#ifndef VF_X3H_SB
#define VF_X3H_SB 1       // sched_barrier(0) at every stage boundary (pins the prefetch distance)
#endif
#ifndef VF_X3H_S2_TALL
#define VF_X3H_S2_TALL 1  // the stride-2 kernel's wave tile (see conv3_s2_x3h_kernel)
#endif
#ifndef VF_X3H_TALL
#define VF_X3H_TALL 1     // wave tile = all 128 pixels x 32 channels (4 x 1 MFMA tiles) instead of 64 pixels x 64 channels (2 x 2): a stage
#endif                    // then needs 2 weight fragments through the L1 -> VGPR return path instead of 4 (and 8 activation fragments
                          // from LDS instead of 4).  PMC of the 2 x 2 form (profiles/r2_conv_x3h_l1path_pmc.txt): TD (the vector-memory
                          // data-return unit) 92 % busy, TA 71 %, matrix pipe 54 % — the return path was the bound, LDS had 4x headroom.
#ifndef VF_X3H_SKIP_LAST
#define VF_X3H_SKIP_LAST 1  // the last chunk of a tile has no successor to stage: branch around the patch loads and the transform + split slots (the
#endif                      // round-1..3 form re-staged the last chunk into the idle buffer: a fifth staging per four at 128 channels)
#ifndef VF_X3H_PRECISE_SWISH
#define VF_X3H_PRECISE_SWISH 0
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

#ifndef VF_X3H_XBUF
#define VF_X3H_XBUF 1       // patch loads of the stride-1 kernel through a per-image buffer resource
#endif
#ifndef VF_X3H_WFIRST
#define VF_X3H_WFIRST 1
#endif
#ifndef VF_X3H_S2_WFIRST
#define VF_X3H_S2_WFIRST 1
#endif
#ifndef VF_X3H_S2_WBUF
#define VF_X3H_S2_WBUF 0    // the stride-2 kernel measured slower with buffer loads (240 vs 245 TF, 277 vs 296)
#endif
#ifndef VF_X3H_WBUF
#define VF_X3H_WBUF 1       // weight fragments by buffer_load_dwordx4 (SGPR resource + scalar stage offset + 32-bit lane offset) instead of 64-bit lane addresses
#endif
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 wbuf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

constexpr int CK = 32;
constexpr int P_LDB = 144;          // bytes per patch pixel in LDS: 2 planes x 64 B + 16 B pad
constexpr int TH = 8, TW = 16;
constexpr int BN = 128;
constexpr int PLANE_BYTES = 2 * BN * 16;      // one (k-step, plane): [half(2)][n(128)][8 f16] = 4 KB
constexpr int KS_BYTES = 2 * PLANE_BYTES;     // one k-step of 16 channels: 2 planes
constexpr int TAP_BYTES = 2 * KS_BYTES;       // one (chunk, tap, n-block) weight tile: 16 KB
constexpr int TAIL_BYTES = 16;                // behind the planes: float 1/S, uint32 max|w| bits (pack scratch)

template <bool UP2, bool PAIR = false>
struct Geo {
    static constexpr int PH = UP2 ? (TH / 2 + 2) : (TH + 2);
    static constexpr int PW = PAIR ? 20 : (UP2 ? (TW / 2 + 2) : (TW + 2));
    static constexpr int NPIX = PH * PW;
    static constexpr int SLOTS = (NPIX * 8 + 255) / 256;
    static constexpr int BUF = (NPIX + 1) * P_LDB;             // bytes, +1 dummy pixel
};

__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * 2048.f);
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void conv3_halo_x3h_kernel(vf_igemm_args p) {
    using G = Geo<UP2, PAIR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool TALL = VF_X3H_TALL != 0;
    constexpr int MI = TALL ? 4 : 2, NJ = TALL ? 1 : 2;           // MFMA tiles per wave: pixels (tile rows / 2) x channels / 32
    const int wave_m = TALL ? 0 : wave >> 1, wave_n = TALL ? wave : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();                        // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = bid % nb; bid /= nb;
    int tx = 0, ty = 0, img;
    if (PAIR) { img = bid * 2; }
    else { tx = bid % tilesX; bid /= tilesX; ty = bid % tilesY; img = bid / tilesY; }
    const int n_img_total = p.M / (p.Hout * p.Wout);
    const int img1 = PAIR ? min(img + 1, n_img_total - 1) : img;   // 2nd image of the pair (= the 1st when n_img is odd: same values rewritten)
    const int pair_pix = (img1 - img) * p.Hin * p.Win;            // pixel offset of the 2nd image
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = UP2 ? (y0 / 2 - 1) : (y0 - 1);
    const int sx0 = UP2 ? (x0 / 2 - 1) : (x0 - 1);

    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / CK;
    const int last_stage = nchunks * 9 - 1;

    const int c4 = tid & 7;
    int s_off[G::SLOTS];
    unsigned ok_mask = 0, sel_mask = 0;            // bit q: slot q is inside the image / belongs to the pair's second image (two registers
                                                   // instead of 2 x SLOTS: the pair form spilled 10)
    // LDS slot of staging slot q: pixel (tid >> 3) + 32 q — only the LAST slot can run past the patch (-> the dummy pixel), the others are
    // one base register + a compile-time offset
    const int lds0 = (tid >> 3) * P_LDB + c4 * 8;
    int lds_last = 0;
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
        const int pix = (tid >> 3) + 32 * q;
        const int pixc = pix < G::NPIX ? pix : G::NPIX;
        const int pr = pixc / G::PW, pc0 = pixc - pr * G::PW;
        const int sel = PAIR ? (pc0 >= 10) : 0;
        const int pc = pc0 - 10 * sel;
        sel_mask |= (unsigned)sel << q;
        const int sy = sy0 + pr, sx = sx0 + pc;
        const bool ok = pix < G::NPIX && sy >= 0 && sy < p.Hin && sx >= 0 && sx < p.Win;
        ok_mask |= (unsigned)ok << q;
        s_off[q] = ok ? (sel * pair_pix + sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        if (q == G::SLOTS - 1) lds_last = pixc * P_LDB + c4 * 8;
    }

    f32x4 preg[G::SLOTS];
    f32x4 pmean, pscale, pbeta, pmean1, pscale1;
#if VF_X3H_XBUF
    // the image's activations as a buffer resource (an image is < 2 GB; the whole tensor is not): 32-bit lane offsets + a scalar chunk offset
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
#endif
    auto patch_load = [&](int chunk) {
#if VF_X3H_XBUF
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q)
            preg[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(x_rs, (unsigned)s_off[q] * 4u, (unsigned)(chunk * CK * 4), 0));
#else
        const float* xc = X + chunk * CK;
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(xc + s_off[q]);
#endif
        if (PRO) {
            pmean = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pscale = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img * p.Cin + chunk * CK + c4 * 4);
            pbeta = *reinterpret_cast<const f32x4*>(p.pro_beta + chunk * CK + c4 * 4);
            if (PAIR) {
                pmean1 = *reinterpret_cast<const f32x4*>(p.pro_mean + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
                pscale1 = *reinterpret_cast<const f32x4*>(p.pro_scale + (size_t)img1 * p.Cin + chunk * CK + c4 * 4);
            }
        }
    };
    auto patch_store_slot = [&](int buf, int q) {
        unsigned char* dst = smem_h + buf * G::BUF + (q == G::SLOTS - 1 ? lds_last : lds0 + q * 32 * P_LDB);
        f16x4 oh, ol;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = preg[q][e];
            if (PRO) {
                const float mu = (PAIR && ((sel_mask >> q) & 1u)) ? pmean1[e] : pmean[e];
                const float sc = (PAIR && ((sel_mask >> q) & 1u)) ? pscale1[e] : pscale[e];
                t = (t - mu) * sc + pbeta[e];
                if (SWISH) t = VF_X3H_PRECISE_SWISH ? vf_swish(t) : vf_swish_1ulp(t);
            }
            _Float16 h, l;
            split2(((ok_mask >> q) & 1u) ? t : 0.f, h, l);
            oh[e] = h; ol[e] = l;
        }
        *reinterpret_cast<f16x4*>(dst) = oh;
        *reinterpret_cast<f16x4*>(dst + 64) = ol;
    };

    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[MI], a_r[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int a0 = wave_m * 4 + mi * 2 + trow;
        a_r[mi] = a0;
        const int tcol = PAIR ? (tpx >> 3) * 10 + (tpx & 7) : tpx;
        a_base[mi] = (a0 * G::PW + tcol) * P_LDB + half * 16;
    }

    // packed weights [chunk][tap][nblk][ks(2)][plane(2)][half(2)][n(128)][8 f16]; one pipeline stage = one (tap, ks)
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    // software pipeline over stages g = chunk*18 + tap*2 + ks: B two stages ahead in a 3-deep register ring (an L2 hit
    // costs about one stage of MFMA time, so one-ahead left the matrix pipe waiting), A one stage ahead (2-deep)
    constexpr int BD = VF_X3H_BD, RING = BD + 1, AD = VF_X3H_AD;
    static_assert(18 % RING == 0 && (AD == 0 || AD == 1), "ring indices must repeat per chunk");
    f16x8 bring[RING][2][NJ];
    f16x8 aring[2][MI][2];
    const int last_g = nchunks * 18 - 1;
#if VF_X3H_WBUF
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
#endif
    auto b_load = [&](f16x8 (&dst)[2][NJ], int g) {
        g = min(g, last_g);
#if VF_X3H_WBUF
        const unsigned soff = (unsigned)((size_t)(g >> 1) * tap_stride + (g & 1) * KS_BYTES);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[pl][j] = wbuf_load(w_rs, (unsigned)(b_lane + pl * PLANE_BYTES + j * 32 * 16), soff);
#else
        const unsigned char* src = Wb + (size_t)(g >> 1) * tap_stride + (g & 1) * KS_BYTES + b_lane;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[pl][j] = *reinterpret_cast<const f16x8*>(src + pl * PLANE_BYTES + j * 32 * 16);
#endif
    };
    auto a_load = [&](f16x8 (&dst)[MI][2], const unsigned char* patch, int s) {
        const int tap = s >> 1, ks = s & 1;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            int aoff;
            if (UP2) {
                const int pr = (a_r[mi] + dy + 1) >> 1, pc = (tpx + dx + 1) >> 1;
                aoff = (pr * G::PW + pc) * P_LDB + half * 16;
            } else {
                aoff = a_base[mi] + (dy * G::PW + dx) * P_LDB;
            }
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[mi][pl] = *reinterpret_cast<const f16x8*>(patch + aoff + pl * 64 + ks * 32);
        }
    };

    f32x16 acc[MI][NJ], accx[MI][NJ];      // main products ah*bh; cross products (al*2^11)*bh + ah*(bl*2^11)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

#ifdef VF_X3H_STAMPS      // per-wave cycle sums (tools/microbench.py x3h_stamps): stage loop vs the wait at the chunk barrier; the tile's head and epilogue
    unsigned long long st_t[3], st_k[4];
    unsigned st_acc[2] = {0, 0};
#define X3H_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t[i]) :: "memory")
#define X3H_KSTAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k[i]) :: "memory")
#else
#define X3H_STAMP(i)
#define X3H_KSTAMP(i)
#endif
    X3H_KSTAMP(0);
    patch_load(0);
#pragma unroll
    for (int g = 0; g < BD; ++g) b_load(bring[g], g);
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();
    X3H_KSTAMP(1);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        X3H_STAMP(0);
        const unsigned char* patch = smem_h + (chunk & 1) * G::BUF;
        // (vmcnt retires in issue order: stage BD's weight fragments go out BEFORE the patch loads, so that no fragment needed within the
        // next BD stages is queued behind HBM latency — see the stride-2 kernel, where this order is worth 20 %)
#if VF_X3H_WFIRST
        b_load(bring[BD % RING], chunk * 18 + BD);
#endif
#ifdef VF_X3H_X_NOPATCH      // ablation: what do the next chunk's patch loads (HBM latency in front of the in-order vmcnt queue) cost?
        if (chunk == 0)
#endif
        // A/B on one MI355X box, skip vs re-stage: 128 ch @128^2 365 vs 362 TF, @64^2 372 vs 365, stride 2 301 vs 295, 256 ch @32^2 equal; the
        // 8x8 pair form LOSES 2 % (367 vs 374) and the upsampling form spills with the branch: both keep the old order
        const bool more = !VF_X3H_SKIP_LAST || UP2 || PAIR || chunk + 1 < nchunks;
        if (more) patch_load(min(chunk + 1, nchunks - 1));
        if (AD) a_load(aring[0], patch, 0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            if (!VF_X3H_WFIRST || s > 0) b_load(bring[(s + BD) % RING], chunk * 18 + s + BD);
            if (AD == 0) a_load(aring[s & 1], patch, s);
            else if (VF_X3H_SB != 3 && s + 1 < 18) a_load(aring[(s + 1) & 1], patch, s + 1);
            if (VF_X3H_SB == 2) __builtin_amdgcn_sched_barrier(0);      // loads are issued before this stage's MFMAs
            // three partial products (plane 0 = h, 1 = l * 2^11): the two cross terms into accx, the main term into acc
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (VF_X3H_SB == 3 && t == 1) {                         // LDS fragments of the next stage issued mid-stage
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < 18) a_load(aring[(s + 1) & 1], patch, s + 1);
                }
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        if (t == 0) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[s & 1][mi][1], bring[s % RING][0][j], accx[mi][j], 0, 0, 0);
                        else if (t == 1) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[s & 1][mi][0], bring[s % RING][1][j], accx[mi][j], 0, 0, 0);
                        else acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[s & 1][mi][0], bring[s % RING][0][j], acc[mi][j], 0, 0, 0);
                    }
            }
            // the next chunk's patch: one staging slot per odd stage (transform + split in the MFMA shadow)
            if (VF_X3H_SB == 1 || VF_X3H_SB == 3) __builtin_amdgcn_sched_barrier(0);
            if ((s & 1) && (s >> 1) >= VF_X3H_STORE && (s >> 1) - VF_X3H_STORE < G::SLOTS && more) patch_store_slot((chunk + 1) & 1, (s >> 1) - VF_X3H_STORE);
            if (VF_X3H_SB == 2) __builtin_amdgcn_sched_barrier(0);
        }
        X3H_STAMP(1);
        __syncthreads();
#ifdef VF_X3H_STAMPS
        X3H_STAMP(2);
        st_acc[0] += (unsigned)(st_t[1] - st_t[0]);
        st_acc[1] += (unsigned)(st_t[2] - st_t[1]);
#endif
    }

    X3H_KSTAMP(2);
    // out = (acc + accx * 2^-11) / S with S the power-of-two weight scale stored behind the packed planes
    const float inv_s = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nchunks * 9 * nb * TAP_BYTES);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accx[i][j][r], 4.8828125e-4f, acc[i][j][r]) * inv_s;
    vf_halo_epilogue_t<PAIR, MI, NJ>(p, acc, img, img1, y0, x0, PAIR ? 0 : (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);
#ifdef VF_X3H_STAMPS      // behind the GroupNorm partials of the launch (the caller sizes gn_part for it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the epilogue's stores have been accepted)
    X3H_KSTAMP(3);
    if (p.gn_part && lane == 0) {
        unsigned* o = reinterpret_cast<unsigned*>(p.gn_part + (size_t)n_img_total * p.gn_slots * 64) + ((size_t)blockIdx.x * 4 + wave) * 4;
        o[0] = st_acc[0];
        o[1] = st_acc[1];
        o[2] = (unsigned)(st_k[1] - st_k[0]);
        o[3] = (unsigned)(st_k[3] - st_k[2]);
    }
#endif
}


// ---- Downsample (pad right/bottom + 3x3 stride 2, vqgan_th.py:45-49), the x3h form of conv3_s2_x6_kernel ---------------------------
// One 8x16 OUTPUT tile needs a 17x33 input patch; it is staged 16 channels at a time (112-byte pixel stride) in ONE LDS
// buffer (44 KB), stored by column parity
// ([row][even columns | odd columns]) so that the stride-2 reads of a tap are again 16 consecutive LDS pixels.
// Weights: the same packing as the stride-1 kernel (a 16-channel chunk is one k-step of a packed 32-channel chunk).
constexpr int S2_PH = 2 * TH + 1, S2_PW = 2 * TW + 1;        // 17 x 33
constexpr int S2_NPIX = S2_PH * S2_PW;                       // 561
constexpr int S2_LDB = 80;                                   // 2 planes x 32 B + 16 B pad = 5 x 16 B
constexpr int S2_SLOTS = (S2_NPIX * 4 + 255) / 256;          // float4 staging slots per thread (9)
constexpr int S2_BUF = (S2_NPIX + 1) * S2_LDB;

// (Round 6, VERDICT r5 item 4, measured and removed — profiles/r6_conv_ab.txt: the stride-1 kernel's raw-patch staging ported here — the next chunk's raw
// fp32 patch HBM -> LDS by DMA into a landing area behind the patch buffer (80.8 KB, two workgroups per CU still fit), issued at tap 4 behind that tap's
// weight loads, each thread reading its own 16 bytes back when it parks the chunk; 36 VGPRs freed (220 instead of 256).  Bit-identical and 0.1 - 0.9 %
// SLOWER at all three Downsample shapes (taps 2 / 4 / 6 alike): since round 2's "240 TF, waits on its patch loads" the weights-first issue order and the
// skipped last staging had already brought the kernel to 299 - 341 TF.)
// P8: the 16x16 -> 8x8 Downsample (the encoder's last): the 8x16 output tile is TWO images side by side (img, img1), each with its own
// 17x17 input patch; a patch row is [even cols of A (9) | even cols of B (9) | odd cols of A (8) | odd cols of B (8)] = 34 pixels.
template <bool P8>
__global__ __launch_bounds__(256, 2) void conv3_s2_x3h_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [S2_BUF]
    constexpr int PW = P8 ? 34 : S2_PW;
    constexpr int NPIX = S2_PH * PW;
    constexpr int SLOTS = (NPIX * 4 + 255) / 256;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave tile: S2 TALL = all 128 output pixels x 32 channels (4 x 1 MFMA tiles): every weight fragment streamed L2 -> VGPR feeds 4
    // pixel tiles and no two waves load the same fragment — the 2 x 2 form moved 147 KB of weights + 36 KB of patch per 16-channel
    // chunk through the CU's L1 path for 14 MFLOP (77 FLOP/B: a 30 % matrix-pipe ceiling at ~16 B/clk/CU; it measured 28 %)
    constexpr bool TALL = VF_X3H_S2_TALL != 0;
    constexpr int MI = TALL ? 4 : 2, NJ = TALL ? 1 : 2;
    const int wave_m = TALL ? 0 : wave >> 1, wave_n = TALL ? wave : wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / BN;
    const int tilesX = P8 ? 1 : p.Wout / TW, tilesY = P8 ? 1 : p.Hout / TH;
    int bid = (int)vf_xcd_bid();                        // XCD-contiguous logical workgroup id (vf_common.h)
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = P8 ? (bid / tilesY) * 2 : bid / tilesY;
    const int n_img_total = p.M / (p.Hout * p.Wout);
    const int img1 = P8 ? min(img + 1, n_img_total - 1) : img;    // (odd image count: the last image twice, same values rewritten)
    const int pair_pix = (img1 - img) * p.Hin * p.Win;
    const int y0 = ty * TH, x0 = tx * TW;
    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / 16;

    // staging slots: thread -> (patch pixel, float4 column of the 16-channel chunk)
    const int c4 = tid & 3;
    int s_off[SLOTS], s_lds[SLOTS];
    bool s_ok[SLOTS];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int pix = (tid >> 2) + 64 * q;
        const int pixc = pix < NPIX ? pix : 0;
        const int pr = pixc / PW, pc0 = pixc - pr * PW;
        const int sel = P8 ? (pc0 >= 17) : 0;           // which image of the pair
        const int pc = pc0 - 17 * sel;
        const int sy = 2 * y0 + pr, sx = 2 * x0 + pc;
        const bool ok = pix < NPIX && sy < p.Hin && sx < p.Win;               // right / bottom zero padding
        s_ok[q] = ok;
        s_off[q] = ok ? (sel * pair_pix + sy * p.Win + sx) * p.Cin + c4 * 4 : c4 * 4;
        int slot;
        if (P8) slot = pr * PW + ((pc & 1) ? 18 + sel * 8 + (pc >> 1) : sel * 9 + (pc >> 1));
        else slot = pr * PW + (pc & 1) * (TW + 1) + (pc >> 1);                // parity-major row
        s_lds[q] = (pix < NPIX ? slot : NPIX) * S2_LDB + c4 * 8;
    }
    f32x4 preg[SLOTS];
    auto patch_load = [&](int chunk) {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) preg[q] = *reinterpret_cast<const f32x4*>(X + chunk * 16 + s_off[q]);
    };
    auto patch_park = [&]() {
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            f16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 h, l;
                split2(s_ok[q] ? preg[q][e] : 0.f, h, l);
                oh[e] = h; ol[e] = l;
            }
            unsigned char* dst = smem_h + s_lds[q];
            *reinterpret_cast<f16x4*>(dst) = oh;
            *reinterpret_cast<f16x4*>(dst + 32) = ol;
        }
    };

    // fragment bases: even input columns (taps dx = 0, 2) and odd ones (dx = 1) — in the pair form the two sections of a patch row
    // are offset by the image, so they get separate bases; otherwise the odd section simply starts TW + 1 pixels later
    const int trow = vf_perm_row(l31), tpx = vf_perm_px(l31);
    int a_base[MI], a_base_o[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int row2 = 2 * (wave_m * 4 + mi * 2 + trow) * PW;
        if (P8) {
            a_base[mi] = (row2 + (tpx >> 3) * 9 + (tpx & 7)) * S2_LDB + half * 16;
            a_base_o[mi] = (row2 + 18 + (tpx >> 3) * 8 + (tpx & 7)) * S2_LDB + half * 16;
        } else {
            a_base[mi] = (row2 + tpx) * S2_LDB + half * 16;
            a_base_o[mi] = a_base[mi] + (TW + 1) * S2_LDB;
        }
    }

    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = (half * BN + wave_n * (32 * NJ) + l31) * 16;
    const int last_g = nchunks * 9 - 1;
    constexpr int S2_BR = 3;                         // weight-fragment ring depth: must divide the 9 taps (the slot is indexed by the tap)
    f16x8 bring[S2_BR][2][NJ];
    f16x8 aring[2][MI][2];
#if VF_X3H_S2_WBUF
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
#endif
    auto b_load = [&](f16x8 (&dst)[2][NJ], int g) {            // g = chunk16 * 9 + tap
        g = min(g, last_g);
        const int c = g / 9, tap = g - c * 9;
#if VF_X3H_S2_WBUF
        const unsigned soff = (unsigned)((size_t)((c >> 1) * 9 + tap) * tap_stride + (c & 1) * KS_BYTES);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[pl][j] = wbuf_load(w_rs, (unsigned)(b_lane + pl * PLANE_BYTES + j * 32 * 16), soff);
#else
        const unsigned char* src = Wb + (size_t)((c >> 1) * 9 + tap) * tap_stride + (c & 1) * KS_BYTES + b_lane;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int j = 0; j < NJ; ++j) dst[pl][j] = *reinterpret_cast<const f16x8*>(src + pl * PLANE_BYTES + j * 32 * 16);
#endif
    };
    auto a_load = [&](f16x8 (&dst)[MI][2], int tap) {
        const int dy = tap / 3, dx = tap % 3;
        const int off = (dy * PW + (dx >> 1)) * S2_LDB;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) dst[mi][pl] = *reinterpret_cast<const f16x8*>(smem_h + ((dx & 1) ? a_base_o[mi] : a_base[mi]) + off + pl * 32);
    };

    f32x16 acc[MI][NJ], accx[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    patch_load(0);
    b_load(bring[0], 0);
    if (S2_BR > 2) b_load(bring[1], 1);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        __syncthreads();                                        // every wave is done reading the previous chunk
        patch_park();
        // vmcnt retires loads in issue order: a weight fragment issued AFTER the next chunk's patch loads (HBM latency) cannot be consumed
        // before they have landed.  So tap 2's fragments go out first — the patch then has until tap 3 instead of tap 2 (without the
        // patch loads at all the kernel runs 354 instead of 240 TF: that wait is its bound).
#if VF_X3H_S2_WFIRST
        b_load(bring[(S2_BR - 1) % S2_BR], chunk * 9 + S2_BR - 1);
#endif
#ifdef VF_X3H_X_NOPATCH
        if (chunk == 0)
#endif
        if (!VF_X3H_SKIP_LAST || chunk + 1 < nchunks) patch_load(min(chunk + 1, nchunks - 1));
        __syncthreads();
        a_load(aring[0], 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (!VF_X3H_S2_WFIRST || t > 0) b_load(bring[(t + S2_BR - 1) % S2_BR], chunk * 9 + t + S2_BR - 1);
            if (t + 1 < 9) a_load(aring[(t + 1) & 1], t + 1);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[t & 1][mi][1], bring[t % S2_BR][0][j], accx[mi][j], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[t & 1][mi][0], bring[t % S2_BR][1][j], accx[mi][j], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[mi][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aring[t & 1][mi][0], bring[t % S2_BR][0][j], acc[mi][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const float inv_s = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)(p.Cin / CK) * 9 * nb * TAP_BYTES);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_fmaf(accx[i][j][r], 4.8828125e-4f, acc[i][j][r]) * inv_s;
    vf_halo_epilogue_t<P8, MI, NJ, false>(p, acc, img, img1, y0, x0, P8 ? 0 : (ty * tilesX + tx) * 2, nblk, wave_m, wave_n, half, l31);   // (no room for 64 preloaded values)
}



// ---------------------------------------------------------------------------------------------------------------------------------------------
// The same convolution on v_mfma_f32_16x16x32_f16 (round 4).  Why: the kernel above runs at the package power limit (profiles/
// r4_power_ceiling_probe.txt): a loop of nothing but 32x32x16 MFMAs on full-entropy operands sustains 1.72 - 1.80 PF, the same loop of 16x16x32
// MFMAs 2.05 PF (+13.6 % on one box) — the deeper-K shape reads and writes HALF the accumulator registers per flop (4 per 16 K flop against 16 per
// 32 K), and on a kernel whose tile costs a fixed energy that is throughput.  Same workgroup tile (8 x 16 pixels x 128 channels), same LDS patch
// ([pixel][plane h | l'][32 ch], 144-byte stride), same packed weights (a 16x16x32 B fragment — lane: channel lane & 15, k group lane >> 4 — is
// (k-step, half) = (group >> 1, group & 1) of the [ks][plane][half][n][8] layout), same wave tile (all 128 pixels x 32 channels).  A wave's 8 x 2
// accumulator tiles are 16 pixels (one tile row) x 16 channels; one MFMA consumes a whole 32-channel chunk of a tap, so a tap is ONE k-step:
//   per tap: 4 weight fragments (h, l' x 2 channel tiles; double-buffered one tap ahead), per tile row 2 patch fragments (h, l'; ring of 3, two
//   rows ahead) feeding 6 MFMAs ordered so that the two products into the cross accumulator are four instructions apart.
// The accumulation ORDER differs from the 32x32x16 kernel's (32 products per instruction instead of 16, cross terms interleaved differently), so
// the results differ in the last bits: both are fp32-equivalent to the same bound (tests/test_hip_x3h.py) and both reproduce every one of the
// 20 480 reference-recorded tokens (tests/test_hip_parity_scale.py).  Stride 1, no pair tiles, an even number of 32-channel chunks.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifndef X3H16_LB
#define X3H16_LB 2
#endif
#ifndef X3H16_PKXFORM
#define X3H16_PKXFORM 0         // 1 = the staging transform on packed fp32 instructions (build.py variant 'x3h16_pk_xform'): same bits, a third fewer vector
                                // instructions per staged value — and 2.5-4 % SLOWER (profiles/r6_conv_ab.txt): measured in round 6, not shipped
#endif
#ifndef VF_X3H16_ABL
#define VF_X3H16_ABL 0          // ablation bits (WRONG results; tools/variants.sh builds only): 1 weight fragments pinned to tap 0, 2 patch fragments read once per
#endif                          // chunk, 4 no epilogue, 8 no MFMAs, 16 no next-chunk staging (DMA + transform), 32 staging without the transform

template <bool PRO, bool SWISH>
__global__ __launch_bounds__(256, X3H16_LB) void conv3_halo_x3h16_kernel(vf_igemm_args p) {
    using G = Geo<false, false>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];   // [2][BUF]
    constexpr int RT = 8, CT = 2;                      // accumulator tiles per wave: tile rows x 16-channel tiles

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lm = lane & 15, lg = lane >> 4;

    const int nb = p.Cout / BN;
    const int tilesX = p.Wout / TW, tilesY = p.Hout / TH;
    int bid = (int)vf_xcd_bid();
    const int nblk = bid % nb; bid /= nb;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const int img = bid / tilesY;
    const int y0 = ty * TH, x0 = tx * TW;
    const int sy0 = y0 - 1, sx0 = x0 - 1;
    const float* __restrict__ X = p.x + (size_t)img * p.Hin * p.Win * p.Cin;
    const int nchunks = p.Cin / CK;

    // ---- patch staging: as in the kernel above (8 threads per pixel, 4 channels each; transform + split once per element)
    const int c4 = tid & 7;
    unsigned ok_mask = 0;
    // LDS bank conflicts (PMC: 46 % of the LDS-active cycles with the plain mapping).  The hardware serves a ds_read_b128 in 16-lane groups
    // {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} (+32): every group holds each MFMA row (lane & 15) once, half of them from k group g, half from
    // g + 1.  With the 9-slot (144-byte) pixel stride the 16 rows land in 16 distinct 16-byte slots iff the two k groups of a hardware group are an
    // EVEN number of slots apart and the rows of its two halves sit on pixels of opposite parity: channel octet g is parked at slot sigma(g) =
    // {0, 2, 1, 3} of its plane, and MFMA row r is pixel pi(r) = {0,2,4,6, 1,3,5,7,9,11,13,15, 8,10,12,14} of the tile row.
    const int c4s = (((c4 >> 1) & 1) * 2 + (c4 >> 2)) * 16 + (c4 & 1) * 8;      // octet c4 >> 1 -> slot sigma; 8 bytes per thread
    const int lds0 = (tid >> 3) * P_LDB + c4s;
    int lds_last = 0;
    // source offset of staging slot q (pixel (tid >> 3) + 32 q of the 10 x 18 patch): recomputed at every chunk's loads instead of held in six
    // registers (pix / 18 == pix * 3641 >> 16 for pix < 192)
    int pix0 = tid >> 3;
    auto slot_src = [&](int q, bool& ok) {
        const int pix = pix0 + 32 * q;
        const int pixc = min(pix, (int)G::NPIX);
        const int pr = (pixc * 3641) >> 16, pc = pixc - pr * G::PW;
        const int sy = sy0 + pr, sx = sx0 + pc;
        ok = (pix < G::NPIX) & ((unsigned)sy < (unsigned)p.Hin) & ((unsigned)sx < (unsigned)p.Win);      // (bitwise: no branches around six DMA issues)
        const int off = (sy * p.Win + sx) * p.Cin;
        return (ok ? off : 0) + c4 * 4;
    };
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) {
        bool ok;
        (void)slot_src(q, ok);
        ok_mask |= (unsigned)ok << q;
        if (q == G::SLOTS - 1) { const int pix = (tid >> 3) + 32 * q; lds_last = (pix < G::NPIX ? pix : G::NPIX) * P_LDB + c4s; }
    }
    // The next chunk's RAW patch values do not wait in registers (6 x 4 per thread for a whole chunk: with 128 accumulator registers and the
    // fragment rings that spilled, and a spill right behind a load waits for HBM): they travel HBM -> LDS by DMA (buffer_load ... lds, 16 bytes per
    // lane, slot q of wave w at raw_l + q * 4096 + w * 1024) and each thread reads its own 16 bytes back when the slot's turn comes.
    // [SLOTS][256 threads][16 B] behind the two patch buffers; the last KB (slot 5 of wave 3: pixels 184 .. 191 of a 180-pixel patch) is not
    // allocated — with it two workgroups of a 512-channel layer (6 KB of GroupNorm-apply table) would not fit a CU's 160 KB
    unsigned char* __restrict__ raw_l = smem_h + 2 * G::BUF;
    const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, 0x7fffffff, 0x00020000);
    // the image's GroupNorm-apply parameters for ALL input channels, once, behind the two patch buffers: [mean | scale | beta][Cin] (the kernel
    // above keeps a chunk's twelve values per thread in registers; here those registers go to the deeper fragment rings)
    float* __restrict__ pro_l = reinterpret_cast<float*>(smem_h + 2 * G::BUF + G::SLOTS * 4096 - 1024);
    if (PRO) {
        for (int i = tid; i < p.Cin; i += 256) {
            pro_l[i] = p.pro_mean[(size_t)img * p.Cin + i];
            pro_l[p.Cin + i] = p.pro_scale[(size_t)img * p.Cin + i];
            pro_l[2 * p.Cin + i] = p.pro_beta[i];
        }
    }
    int pro_chunk = 0;                                             // the chunk whose patch is in preg
    auto patch_load = [&](int chunk) {
        asm volatile("" : "+v"(pix0));                            // (not loop-invariant as far as the compiler knows: no hoisting into registers)
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) {
            bool ok;
            const int off = slot_src(q, ok);
            if (q == G::SLOTS - 1 && wave == 3) continue;         // (no pixel of the patch; its landing KB is the table's)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (__attribute__((address_space(3))) void*)(raw_l + q * 4096 + wave * 1024), 16, (unsigned)off * 4u,
                                                     (unsigned)(chunk * CK * 4), 0, 0);
        }
        pro_chunk = chunk;
    };
    auto patch_store_slot = [&](int buf, int q) {
        unsigned char* dst = smem_h + buf * G::BUF + (q == G::SLOTS - 1 ? lds_last : lds0 + q * 32 * P_LDB);
        f32x4 pmean, pscale, pbeta;
        if (PRO) {
            const float* pl = pro_l + pro_chunk * CK + c4 * 4;
            pmean = *reinterpret_cast<const f32x4*>(pl);
            pscale = *reinterpret_cast<const f32x4*>(pl + p.Cin);
            pbeta = *reinterpret_cast<const f32x4*>(pl + 2 * p.Cin);
        }
        f32x4 raw = {0.f, 0.f, 0.f, 0.f};                        // (slot 5 of wave 3 has no landing area: pixels 184 .. 191 do not exist)
        if (!(q == G::SLOTS - 1 && wave == 3)) raw = *reinterpret_cast<const f32x4*>(raw_l + q * 4096 + tid * 16);
        f16x4 oh, ol;
#if X3H16_PKXFORM && !VF_X3H_PRECISE_SWISH
        // the GroupNorm-apply + swish + split of two values at a time on packed fp32 instructions (vf_common.h: vf_swish_1ulp_pk): the same
        // operations in the same order as the scalar form below — same bits, a third fewer vector instructions per staged value (A/B only: slower)
        const bool okq = ((ok_mask >> q) & 1u) != 0u;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            vf_f32x2 t = {raw[2 * e2], raw[2 * e2 + 1]};
            if (PRO && !(VF_X3H16_ABL & 32)) {
                const vf_f32x2 m2 = {pmean[2 * e2], pmean[2 * e2 + 1]}, s2 = {pscale[2 * e2], pscale[2 * e2 + 1]}, b2 = {pbeta[2 * e2], pbeta[2 * e2 + 1]};
                t = (t - m2) * s2 + b2;
                if (SWISH) t = vf_swish_1ulp_pk(t);
            }
            if (!okq) t = (vf_f32x2){0.f, 0.f};
            const _Float16 h0 = (_Float16)t.x, h1 = (_Float16)t.y;
            const vf_f32x2 hb = {(float)h0, (float)h1}, k2048 = {2048.f, 2048.f};
            const vf_f32x2 lo = (t - hb) * k2048;
            oh[2 * e2] = h0; oh[2 * e2 + 1] = h1;
            ol[2 * e2] = (_Float16)lo.x; ol[2 * e2 + 1] = (_Float16)lo.y;
        }
#else
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = raw[e];
            if (PRO && !(VF_X3H16_ABL & 32)) {
                t = (t - pmean[e]) * pscale[e] + pbeta[e];
                if (SWISH) t = VF_X3H_PRECISE_SWISH ? vf_swish(t) : vf_swish_1ulp(t);
            }
            _Float16 h, l;
            split2(((ok_mask >> q) & 1u) ? t : 0.f, h, l);
            oh[e] = h; ol[e] = l;
        }
#endif
        *reinterpret_cast<f16x4*>(dst) = oh;
        *reinterpret_cast<f16x4*>(dst + 64) = ol;
    };

    // ---- fragments
    const int lpix = lm < 4 ? 2 * lm : lm < 12 ? 2 * lm - 7 : 2 * lm - 16;       // pi(lm): the tile-row pixel behind MFMA row lm
    const int a_lane = lpix * P_LDB + (((lg & 1) * 2) + (lg >> 1)) * 16;         // channels 8 lg .. 8 lg + 7 of the chunk at slot sigma(lg)
    const unsigned char* __restrict__ Wb = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * TAP_BYTES;
    const size_t tap_stride = (size_t)nb * TAP_BYTES;
    const int b_lane = ((lg >> 1) * 4 + (lg & 1)) * (BN * 16) + (wave * 32 + lm) * 16;
    const int last_tap = nchunks * 9 - 1;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(Wb), 0, 0x7fffffff, 0x00020000);
    f16x8 bring[2][4];                                              // [tap parity][h ct0, h ct1, l' ct0, l' ct1]
#ifndef X3H16_AR
#define X3H16_AR 3
#endif
    constexpr int AR = X3H16_AR;                                    // patch-fragment ring: AR - 1 tile rows ahead (AR divides 72)
    f16x8 aring[AR][2];                                             // [tile-row step % AR][h, l']
    auto b_load = [&](f16x8 (&dst)[4], int gtap) {
        const unsigned soff = (VF_X3H16_ABL & 1) ? 0u : (unsigned)((size_t)min(gtap, last_tap) * tap_stride);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) dst[pl * 2 + ct] = wbuf_load(w_rs, (unsigned)(b_lane + pl * PLANE_BYTES + ct * 256), soff);
    };
    auto a_load = [&](f16x8 (&dst)[2], const unsigned char* patch, int step) {       // step = tap * 8 + tile row
        const int tap = step >> 3, rt = step & 7;
        const int off = ((rt + tap / 3) * G::PW + tap % 3) * P_LDB;
        dst[0] = *reinterpret_cast<const f16x8*>(patch + a_lane + off);
        dst[1] = *reinterpret_cast<const f16x8*>(patch + a_lane + off + 64);
    };

    f32x4v acc[RT][CT], accx[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; accx[i][j][r] = 0.f; }

    patch_load(0);
    b_load(bring[0], 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the first patch has landed (each thread reads back only its own lanes' bytes)
    if (PRO) __syncthreads();                                      // the parameter table
#pragma unroll
    for (int q = 0; q < G::SLOTS; ++q) patch_store_slot(0, q);
    __syncthreads();

    // (Round 6, VERDICT r5 item 5b, measured and removed — profiles/r6_conv_ab.txt: the residual tile of a ResnetBlock's conv2 pulled towards the CU by
    // LDS-DMA during the tile's LAST chunk, whose landing area is idle, so that the epilogue's 64 cold loads per lane hit L2: 0.5 - 0.8 % SLOWER
    // at every residual shape, bit-identical.  The tile's time is its energy: hiding a latency moves nothing, 64 KB more LDS writes cost.)
    // one 32-channel chunk; P0 = chunk & 1 (nine taps: the weight ring's parity flips per chunk, so the chunk loop is unrolled by two)
    auto chunk_body = [&](int chunk, auto parity) {
        constexpr int P0 = decltype(parity)::value;
        const unsigned char* patch = smem_h + P0 * G::BUF;
        const bool more = chunk + 1 < nchunks && !(VF_X3H16_ABL & 16);
        // (vmcnt retires in issue order: the next tap's weights go out BEFORE the patch loads)
        b_load(bring[(P0 + 1) & 1], chunk * 9 + 1);
        // The DMAs sit behind `if (more)`; where that branch rejoins, hipcc's s_waitcnt pass takes the STRICTER path's pending count — the one
        // without the six DMAs — so every wait it then computes for a weight fragment is 6 too strict and would stall tap 0 / tap 1 on the first
        // DMAs (HBM latency).  Six counted one-dword loads of an L2-resident word, issued on BOTH paths, sit between the weights and the DMAs in
        // the queue: the slack the pass lacks is spent on them.  Their values go to an empty asm at tap 2, by which time the in-order queue has
        // retired them anyway.
        unsigned pad[G::SLOTS];
#pragma unroll
        for (int q = 0; q < G::SLOTS; ++q) pad[q] = __builtin_amdgcn_raw_buffer_load_b32(w_rs, (unsigned)(q * 256), 0u, 0);     // (distinct, not adjacent: six instructions)
        if (more) patch_load(chunk + 1);
#pragma unroll
        for (int i = 0; i < AR - 1; ++i) a_load(aring[i], patch, i);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t > 0) b_load(bring[(P0 + t + 1) & 1], chunk * 9 + t + 1);
            if (t == 2) {
#pragma unroll
                for (int q = 0; q < G::SLOTS; ++q) asm volatile("" :: "v"(pad[q]));
            }
            const f16x8 (&B)[4] = bring[(P0 + t) & 1];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const int step = t * 8 + rt;
                if (step + AR - 1 < 72 && !((VF_X3H16_ABL & 2) && step + AR - 1 >= AR)) a_load(aring[(step + AR - 1) % AR], patch, step + AR - 1);
                const f16x8 ah = aring[step % AR][0], al = aring[step % AR][1];
#if VF_X3H16_ABL & 8
                accx[rt][0][0] += (float)al[0] * (float)B[0][0]; accx[rt][1][0] += (float)al[1] * (float)B[1][0];
                acc[rt][0][0] += (float)ah[0] * (float)B[0][1]; acc[rt][1][0] += (float)ah[1] * (float)B[1][1];
                accx[rt][0][1] += (float)ah[2] * (float)B[2][0]; accx[rt][1][1] += (float)ah[3] * (float)B[3][0];
#else
                accx[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[0], accx[rt][0], 0, 0, 0);
                accx[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, B[1], accx[rt][1], 0, 0, 0);
                acc[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[0], acc[rt][0], 0, 0, 0);
                acc[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[1], acc[rt][1], 0, 0, 0);
                accx[rt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[2], accx[rt][0], 0, 0, 0);
                accx[rt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, B[3], accx[rt][1], 0, 0, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t >= 1 && t <= G::SLOTS && more) {                                            // the next chunk's patch, one slot per tap
                // slot 0 at the end of tap 1: behind its DMA the queue holds 5 more DMAs (4 in wave 3) and tap 2's 4 weight fragments; from tap 2
                // on the wait for that tap's weights (issued behind all the DMAs, retired in order) has covered every slot
                if (t == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                patch_store_slot((P0 + 1) & 1, t - 1);
            }
        }
        __syncthreads();
    };
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
        chunk_body(chunk, std::integral_constant<int, 0>{});
        chunk_body(chunk + 1, std::integral_constant<int, 1>{});
    }

    if ((VF_X3H16_ABL & 4) && acc[0][0][0] + accx[7][1][3] != 12345.678f) return;
    // ---- epilogue: out = (acc + accx * 2^-11) / S + bias (+ residual); GroupNorm partials of the stored values (fp64 sums, rounded once)
    const float inv_s = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nchunks * 9 * nb * TAP_BYTES);
    // (buffer resources per image + a lane offset + a scalar (tile row, pixel) offset: 64-bit lane addresses for the 32 pixels of a lane cost 64
    // registers that the chunk loop's rings need)
    const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)img * p.Hout * p.Wout * p.ldc, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res + (size_t)img * p.Hout * p.Wout * p.ldr : p.out), 0, 0x7fffffff, 0x00020000);
    const bool Res = p.res != nullptr;
    const bool stats = p.gn_part != nullptr;
    const int cg = p.Cout >> 5;
    const int tile_slot = (ty * tilesX + tx) * 2;
    // a lane's tile elements: channel lm of the 16-channel tile, four pixels of tile row rt
    // accumulator element r of a lane = MFMA row 4 lg + r = pixel pi(4 lg + r) = pbase(lg) + 2 r of the tile row
    const int pbase = lg == 0 ? 0 : lg == 1 ? 1 : lg == 2 ? 9 : 8;
    auto spix = [&](int rt, int r) { return (y0 + rt) * p.Wout + x0 + 2 * r; };        // wave-uniform part of the pixel index (the lane adds pbase)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[rt][ct][r] = __builtin_fmaf(accx[rt][ct][r], 4.8828125e-4f, acc[rt][ct][r]) * inv_s;
    __builtin_amdgcn_sched_barrier(0);                         // (the cross accumulators are dead before the residual values arrive)
    float rr[CT][RT][4];
    if (Res) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int n = nblk * BN + wave * 32 + ct * 16 + lm;
            const unsigned rv = (unsigned)(pbase * p.ldr + n) * 4u;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rr[ct][rt][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rs, rv, (unsigned)(spix(rt, r) * p.ldr) * 4u, 0));
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int n = nblk * BN + wave * 32 + ct * 16 + lm;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const unsigned ov = (unsigned)(pbase * p.ldc + n) * 4u;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {                       // one statistics slot = 4 tile rows
            // a lane's 16 values of the slot are summed in fp32 (fixed order; 16 terms: ~1e-6 of the lane's partial, the accuracy of an fp32
            // GroupNorm) and only the lane partials go through fp64: the all-fp64 form cost 2.5 - 3.3 % of the launch in cvt / add / fma _f64
            float s32 = 0.f, q32 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rt = sl * 4 + k;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[rt][ct][r] + bias;
                    if (Res) v += rr[ct][rt][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), o_rs, ov, (unsigned)(spix(rt, r) * p.ldc) * 4u, 0);
                    s32 += v;
                    q32 = __builtin_fmaf(v, v, q32);
                }
            }
            if (stats) {
                vf_gn_acc_t s = (vf_gn_acc_t)s32, q = (vf_gn_acc_t)q32;
                for (int o = 1; o < cg; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }     // the group's channels (cg <= 16 lanes)
                s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);                                      // the four pixel groups
                s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
                if (lg == 0 && (lm & (cg - 1)) == 0) {
                    float* dst = p.gn_part + ((((size_t)img * p.gn_slots) + tile_slot + sl) * 32 + n / cg) * 2;
                    dst[0] = (float)s;
                    dst[1] = (float)q;
                }
            }
        }
    }
}

// max |w| over the tensor as the bits of a non-negative float (monotone as unsigned)
__global__ void absmax_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
    vf_block_max_atomic(m, out);
}

// OIHW fp32 -> two fragment-packed f16 planes of w * S, S = 2^(13 - floor(log2 max|w|)); thread 0 leaves 1/S in the tail
__global__ void pack_conv_x3h_kernel(const float* __restrict__ w, _Float16* __restrict__ dst, int Cin, int Cout, int nb, int nchunks,
                                     unsigned char* __restrict__ tail) {
    const float amax = __uint_as_float(*reinterpret_cast<const unsigned*>(tail + 4));
    const int ex = amax > 0.f ? ilogbf(amax) : 13;
    const float S = ldexpf(1.f, 13 - ex);
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(tail) = ldexpf(1.f, ex - 13);
    const long long total = (long long)nchunks * 9 * nb * CK * BN;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7);
        long long t = idx >> 3;
        const int nl = (int)(t % BN); t /= BN;
        const int half = (int)(t & 1);
        const int ks = (int)((t >> 1) & 1);
        t >>= 2;
        const int nblk = (int)(t % nb); t /= nb;
        const int tap = (int)(t % 9);
        const int chunk = (int)(t / 9);
        const int c = chunk * CK + ks * 16 + half * 8 + e;
        const int n = nblk * BN + nl;
        float v = 0.f;
        if (c < Cin && n < Cout) v = w[((size_t)n * Cin + c) * 9 + tap] * S;
        _Float16 h, l;
        split2(v, h, l);
        const size_t base = ((((size_t)(chunk * 9 + tap) * nb + nblk) * 2 + ks) * 2) * (2 * BN * 8) + ((size_t)half * BN + nl) * 8 + e;
        dst[base] = h;
        dst[base + 2 * BN * 8] = l;
    }
}

template <bool UP2, bool PRO, bool SWISH, bool PAIR>
int launch_halo(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<UP2, PAIR>;
#ifndef VF_X3H_LDS_PAD
#define VF_X3H_LDS_PAD 0    // probe: extra LDS bytes per workgroup (> 28 KB: one workgroup per CU, i.e. one wave per SIMD — profiles/r4_power_ceiling_probe.txt)
#endif
    const size_t smem = (size_t)2 * G::BUF + VF_X3H_LDS_PAD;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = PAIR ? (long long)((n_img + 1) / 2) * (a.Cout / BN)
                                  : (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
#if VF_X3H_LDS_PAD
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_x3h_kernel<UP2, PRO, SWISH, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
#endif
    hipLaunchKernelGGL((conv3_halo_x3h_kernel<UP2, PRO, SWISH, PAIR>), dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

// LDS of the 16x16x32 kernel: two patch buffers, the raw-patch landing area, the GroupNorm-apply table; two workgroups must fit a CU (160 KB)
static inline size_t vf_x3h16_lds_bytes(int pro_cin) { return (size_t)2 * Geo<false, false>::BUF + Geo<false, false>::SLOTS * 4096 - 1024 + (size_t)3 * pro_cin * 4; }

template <bool PRO, bool SWISH>
int launch_halo16(const vf_igemm_args& a, hipStream_t stream) {
    using G = Geo<false, false>;
    const int n_img = a.M / (a.Hout * a.Wout);
    const long long blocks = (long long)n_img * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
    const size_t smem = vf_x3h16_lds_bytes(PRO ? a.Cin : 0);
    static unsigned long long attr_devs = 0;          // per instantiation; bit d: raised on device d (> 64 KB of dynamic LDS)
    if (vf_attr_needed(&attr_devs)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_x3h16_kernel<PRO, SWISH>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    hipLaunchKernelGGL((conv3_halo_x3h16_kernel<PRO, SWISH>), dim3((unsigned)blocks), dim3(256), smem, stream, a);
    return vf_last_status();
}

template <bool UP2, bool PAIR>
int dispatch_pro(const vf_igemm_args& a, hipStream_t s) {
    // stride 1, whole 8 x 16 tiles, an even number of 32-channel chunks, GroupNorm groups of <= 16 channels: the 16x16x32 kernel
    // (vf_select(VF_SEL_CONV_X3H_K32, 0): the 32x32x16 kernel above — fp32-equivalent to the same bound, different last bits)
    if (!UP2 && !PAIR && (a.Cin / CK) % 2 == 0 && (!a.gn_part || a.Cout / 32 <= 16) && vf_x3h16_lds_bytes(a.pro_mean ? a.Cin : 0) <= 80 * 1024 &&
        vf_selected(VF_SEL_CONV_X3H_K32)) {
        if (!a.pro_mean) return launch_halo16<false, false>(a, s);
        return a.pro_swish ? launch_halo16<true, true>(a, s) : launch_halo16<true, false>(a, s);
    }
    if (!a.pro_mean) return launch_halo<UP2, false, false, PAIR>(a, s);
    return a.pro_swish ? launch_halo<UP2, true, true, PAIR>(a, s) : launch_halo<UP2, true, false, PAIR>(a, s);
}

}  // namespace

extern "C" {

size_t vf_conv3_x3h_packed_elems(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cin + CK - 1) / CK) * 9 * ((Cout + BN - 1) / BN) * CK * BN * 2 + TAIL_BYTES / 2;
}

int vf_conv3_x3h_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream) {
    if (!w_oihw || !dst || Cin <= 0 || Cout <= 0) return VF_ERR_BAD_ARG;
    const int nb = (Cout + BN - 1) / BN, nchunks = (Cin + CK - 1) / CK;
    const long long total = (long long)nchunks * 9 * nb * CK * BN;
    unsigned char* tail = reinterpret_cast<unsigned char*>(dst) + (size_t)total * 2 * sizeof(_Float16);
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(tail, 0, TAIL_BYTES, s) != hipSuccess) return vf_last_status();
    const long long nw = (long long)Cout * Cin * 9;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)((nw + 2047) / 2048 > 128 ? 128 : (nw + 2047) / 2048)), dim3(256), 0, s, w_oihw, nw,
                       reinterpret_cast<unsigned*>(tail + 4));
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_conv_x3h_kernel, dim3(blocks), dim3(256), 0, s, w_oihw, (_Float16*)dst, Cin, Cout, nb, nchunks, tail);
    return vf_last_status();
}

int vf_conv3_halo_x3h(const vf_igemm_args* args, void* stream) {
    if (!args) return VF_ERR_BAD_ARG;
    const vf_igemm_args& a = *args;
    if (!a.x || !a.w_packed || !a.out || a.M <= 0) return VF_ERR_BAD_ARG;
    if (a.drop_rate != 0.f || a.out_aux) return VF_ERR_UNSUPPORTED;      // fused output dropout / second output: vf_gemm_bf16 only
    if (a.mode == VF_MODE_CONV3_S2PAD) {
        const bool p8 = a.Hout == 8 && a.Wout == 8;                      // two images per tile
        if (a.Cout % BN != 0 || a.Cin % CK != 0 || (!p8 && (a.Hout % TH != 0 || a.Wout % TW != 0)) || a.pro_mean) return VF_ERR_UNSUPPORTED;
        if (a.Hin != a.Hout * 2 || a.Win != a.Wout * 2 || a.M % (a.Hout * a.Wout) != 0) return VF_ERR_BAD_ARG;
        if (a.batch > 1 || a.epilogue != VF_EPI_NONE || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
        if ((long long)a.Hin * a.Win * a.Cin >= (1ll << 31)) return VF_ERR_UNSUPPORTED;
        if (int st = vf_halo_gn_check(a)) return st;
        if (p8) {
            const long long blocks = (long long)((a.M / 64 + 1) / 2) * (a.Cout / BN);
            hipLaunchKernelGGL(conv3_s2_x3h_kernel<true>, dim3((unsigned)blocks), dim3(256), (size_t)(S2_PH * 34 + 1) * S2_LDB, (hipStream_t)stream, a);
            return vf_last_status();
        }
        const long long blocks = (long long)(a.M / (a.Hout * a.Wout)) * (a.Hout / TH) * (a.Wout / TW) * (a.Cout / BN);
        hipLaunchKernelGGL(conv3_s2_x3h_kernel<false>, dim3((unsigned)blocks), dim3(256), (size_t)S2_BUF, (hipStream_t)stream, a);
        return vf_last_status();
    }
    if (a.mode != VF_MODE_CONV3_S1 && a.mode != VF_MODE_CONV3_UP2) return VF_ERR_UNSUPPORTED;
    const bool pair = a.mode == VF_MODE_CONV3_S1 && a.Hout == 8 && a.Wout == 8;     // two 8x8 images per tile
    if (a.Cout % BN != 0 || a.Cin % CK != 0 || (!pair && (a.Hout % TH != 0 || a.Wout % TW != 0))) return VF_ERR_UNSUPPORTED;
    if (a.Hin <= 0 || a.Win <= 0 || a.M % (a.Hout * a.Wout) != 0) return VF_ERR_BAD_ARG;
    if (a.mode == VF_MODE_CONV3_S1 && (a.Hout != a.Hin || a.Wout != a.Win)) return VF_ERR_BAD_ARG;
    if (a.mode == VF_MODE_CONV3_UP2 && (a.Hout != a.Hin * 2 || a.Wout != a.Win * 2)) return VF_ERR_BAD_ARG;
    if (a.batch > 1 || a.epilogue != VF_EPI_NONE || a.ldc < a.Cout || (a.res && a.ldr < a.Cout)) return VF_ERR_BAD_ARG;
    if ((a.pro_mean || a.pro_scale || a.pro_beta) && !(a.pro_mean && a.pro_scale && a.pro_beta)) return VF_ERR_BAD_ARG;
    if ((long long)a.Hin * a.Win * a.Cin >= (1ll << 31)) return VF_ERR_UNSUPPORTED;
    if (int st = vf_halo_gn_check(a)) return st;
    hipStream_t s = (hipStream_t)stream;
    if (pair) return dispatch_pro<false, true>(a, s);
    return (a.mode == VF_MODE_CONV3_UP2) ? dispatch_pro<true, false>(a, s) : dispatch_pro<false, false>(a, s);
}

}  // extern "C"
