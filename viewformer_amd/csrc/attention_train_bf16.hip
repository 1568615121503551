// Backward of the block-causal / streams attention on the bf16 matrix pipe, for the bf16 arm of the training step, gfx950.
//
// Flash-style like attention_bwd_f32.hip — P is re-materialised tile by tile from q, k and the forward's per-query log-sum-exp, masked
// tiles are skipped, no atomics (a query-major dQ kernel and a key-major dK / dV kernel) — but on v_mfma_f32_32x32x16_bf16 with bf16
// q / k / v / dO in HBM (the c_attn GEMM and the c_proj dX GEMM write them as bf16) and every streamed tile moved HBM -> LDS by
// LDS-DMA.  A streamed 64 x 64 tile is read in two ways — its rows as ds_read_b128 A fragments (k = features) and, through
// ds_read_b64_tr_b16, its TRANSPOSE as A fragments (feature rows, k = the streamed rows) — and since the third session of round 6 ONE
// 8 KB image per tile serves both (the "uni" image, dma_uni_piece below: a 4-row x 64-byte block per 256-byte bank line): 16-16.5 KB per
// ring slot, three workgroups per CU.  The layouts it replaced (still here behind -DATB_KV_UNI=0 / -DATB_DQ_UNI=0 for A/B, same results
// bit for bit) are attention_dma.hip's:
//   * "rows" image of a 64 x 64 tile: 128-byte rows, 16-byte chunk index XORed with bits 1..3 of the row -> conflict-free ds_read_b128
//     A fragments (rows of the streamed operand, k = features);
//   * "tr" image: [feature half][row][32 features], read with ds_read_b64_tr_b16 -> the TRANSPOSED A fragment (feature rows, k = the
//     streamed rows) straight from the row-major tile.
// Both kernels keep the owner's operands (dQ kernel: Q and dO rows of its 32 queries; dK/dV kernel: K and V rows of its 32 keys) in
// registers as B fragments and compute the tile products with the owner in the MFMA column = lane, so P / dS sit in registers in the
// B-operand layout of the accumulating MFMA (the k order of those products is the accumulator's row order, which is the order the
// transposed reads deliver: attention_dma.hip's P.V step).
//   dQ kernel, per (wave = 32 queries, visible key tile):  S^T = K.Q^T, dP^T = V.dO^T, dS^T = P^T (dP^T - D) scale, dQ^T += K^T.dS^T   (24 MFMAs)
//   dKV kernel, per (wave = 32 keys, visible query tile):  S = Q.K^T, dP = dO.V^T, dV^T += dO^T.P, dK^T += Q^T.dS                        (32 MFMAs)
// with P = exp(S scale - lse), D = rowsum(dO * O) (vf_attn_bwd_prep_bf16).  P and dS are rounded to bf16 as MFMA operands; sums, the
// exponent and D stay fp32; dQ / dK / dV are written as fp32 rows (transposed through LDS, whole 256-byte rows).
// 64-token views only (every (wave, tile) pair is entirely visible or entirely masked), T a multiple of 64: the trainer takes
// attention_bwd_f32.hip otherwise.
// DROP (round 4): attention dropout (branching_attention.py:15-17) — both kernels re-materialise the forward's mask (vf_common.h: one
// hashed word per four consecutive keys of a query).  With keep in {0, 1} and c = 1 / (1 - rate):  dP gets keep * c before the D term,
// dV accumulates keep * P and is scaled by c once at the end.  The dQ kernel's lanes hold four keys of a query per register group (one
// hash per four elements); the dK / dV kernel's lanes hold four QUERIES of a key, i.e. four groups: one hash per element there.
// Reference: autograd of compute_attention / compute_causal_block_multiend_attention
// (viewformer/models/branching_attention.py:5-18,82-126) inside MIGT.train_step (migt.py:464-505) under mixed_float16.
#include <type_traits>
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64, KT = 64;
constexpr int IMG = KT * DH * 2;             // one 64 x 64 bf16 image: 8 KB
constexpr int OT = 128;                      // owner rows per workgroup: 4 waves x 32
constexpr float LOG2E = 1.4426950408889634f;
#ifndef ATB_ABL
#define ATB_ABL 0           // ablation builds (results WRONG, timing only; build.py 'atb_abl*'): 1 = no "tr" image DMAs, 2 = no tile compute, 4 = no tile DMAs at all,
                            // 9 = 1 + the tr reads aliased onto the rows images: slots of 2 images, ring 3, three workgroups per CU (the timing a unified image would have)
#endif
#ifndef ATB_KV_UNI
#define ATB_KV_UNI 1        // dK / dV kernel: ONE LDS image per streamed operand serves the row reads AND the transposing reads (see dma_uni_piece); slots of two
                            // images, ring of 3, three workgroups per CU.  0 = round 6's earlier form (rows image + tr image per operand, ring of 2; build.py 'dkv_two_images')
#endif
#ifndef ATB_DQ_UNI
#define ATB_DQ_UNI 1        // dQ kernel: the K tile as ONE uni image (row reads for S^T, transposing reads for dQ^T), V as a uni image too; slots of two images
                            // (48 KB ring: three workgroups per CU).  0 = K rows | V rows | K tr (build.py 'dq_three_images')
#endif
#ifndef ATB_HEAVY_FIRST
#define ATB_HEAVY_FIRST 1   // owner blocks dispatched heaviest first (round 6, vf_common.h: vf_attn_block_order); 0 = in index order
#endif
// FOLD (round 6, dQ kernel; ADVICE r5 + VERDICT r5 item 2).  The forward kernel (attention_dma.hip) multiplies q by scale * log2 e and re-rounds
// it to bf16 before its S product; until round 6 the backward re-materialised P from the UN-rounded q — an exponent off by |s| 2^-9, which
// at the score magnitudes of a trained model (MIGT's scores are unscaled) is several per cent of P.  The dQ kernel holds q as its
// register-resident B operand, so it now applies the forward's rounding: S^T = K.q'^T is the forward's product bit for bit (same MFMA, same
// k order), and the per-score arithmetic shrinks with it: a query's -lse log2 e + log2(scale) and -D are constants of the whole kernel (one
// query per lane = all 16 registers of an accumulator), so they are the C operand of each tile's first MFMA (32 loop-invariant registers,
// D != C) and dS = exp2(acc_s) * acc_p — {exp2, mul, 1/2 cvt} per score instead of {fma, exp2, sub, mul, mul, 1/2 cvt}.
// MEASURED (profiles/r6_attention_diet.txt): dq error against fp64 autograd at |s| <= 40: 1.15e-2 instead of 2.7e-2; time of dQ + dK/dV at the
// training shape -3 % (-1 % with dropout) although the dQ kernel issues 55 % fewer vector instructions per score: like the forward, the
// kernel is not bound by its instruction count.  The same fold in the dK / dV kernel (k' = bf16(k log2 e) as the owner operand, the
// lse table as the S accumulators' C operand) was measured and NOT kept: its P then carries a DIFFERENT rounding than the forward's
// (dv error 3.0e-2 instead of 1.5e-2 at |s| <= 40) for no measurable time.

__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p)));
    return __builtin_bit_cast(bf16x4, r);
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p, int gap = 8 * 64) {             // 8 consecutive k of the accumulator order: two 4-row groups
    const bf16x4 v0 = tr_read(p), v1 = tr_read(p + gap);
    bf16x8 a;
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = v0[e]; a[4 + e] = v1[e]; }
    return a;
}

struct Vis {
    int Vc, Sv;
    __device__ __forceinline__ bool operator()(int qv, int kv) const {
        if (Sv > 0) {
            const int qs = qv / Sv, qi = qv - qs * Sv;
            const int ks = kv / Sv, ki = kv - ks * Sv;
            return qs == 0 ? (ks == 0 && ki <= qi) : ((ks == 0 && ki < qi) || kv == qv);
        }
        return kv == qv || min(kv, Vc) < min(qv, Vc);
    }
};
__device__ __forceinline__ Vis make_vis(int twin) { return Vis{twin >= 0 ? twin : 0x3fffffff, twin <= -2 ? -twin : 0}; }
// the same relation as bit masks over views 0 .. nviews - 1 (nviews <= 64), in closed form: the kernels' prologues used to call visible() — two integer divisions
// under the streams mask — for every (owner view, streamed view) pair and once more per tile step; ~120 divisions per workgroup before its first DMA could be issued
__device__ __forceinline__ unsigned long long vis_bits(int lo, int hi) {         // bits lo .. hi - 1
    if (hi <= lo || lo >= 64) return 0ull;
    const unsigned long long upto_hi = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    return upto_hi & ~((1ull << lo) - 1ull);
}
// key views that query view qv sees
__device__ __forceinline__ unsigned long long vis_keys_of(const Vis& V, int qv, int nviews) {
    unsigned long long m;
    if (V.Sv > 0) {
        const int qs = qv / V.Sv, qi = qv - qs * V.Sv;
        m = qs == 0 ? vis_bits(0, qi + 1) : (vis_bits(0, qi) | (1ull << qv));
    } else {
        m = vis_bits(0, min(qv, V.Vc)) | (1ull << qv);
    }
    return m & vis_bits(0, nviews);
}
// query views that see key view kv
__device__ __forceinline__ unsigned long long vis_queries_of(const Vis& V, int kv, int nviews) {
    unsigned long long m = 1ull << kv;
    if (V.Sv > 0) {
        const int ks = kv / V.Sv, ki = kv - ks * V.Sv;
        if (ks == 0) {
            m |= vis_bits(ki, V.Sv);                                                  // the sequence's views from ki on
            for (int s0 = V.Sv; s0 < nviews; s0 += V.Sv) m |= vis_bits(s0 + ki + 1, s0 + V.Sv);      // every branch stream's views above ki
        }
    } else if (kv < V.Vc) {
        m |= vis_bits(kv + 1, nviews);                                                // (an "ending" view kv >= Vc is seen by itself only)
    }
    return m & vis_bits(0, nviews);
}

template <int N>
__device__ __forceinline__ void wait_loads() {                                   // this wave's loads: at most N outstanding; its LDS reads: done
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
}

// a "rows" piece (8 rows x 128 B, lane -> row lane >> 3, LDS chunk lane & 7 = global chunk ^ ((row >> 1) & 7)) and a "tr" piece
// (16 rows x 64 B of one feature half, lane -> row lane >> 2, chunk lane & 3) of a 64-row tile starting at row t * 64 of an operand
// with row stride ld (elements)
__device__ __forceinline__ void dma_rows_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int pi, int lane, int ld, int t) {
    const int r = pi * 8 + (lane >> 3), pc = lane & 7;
    bufds16(rs, img + pi * 1024, (unsigned)(r * ld * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), (unsigned)(t * KT * ld * 2));
}
__device__ __forceinline__ void dma_tr_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int pi, int lane, int ld, int t) {
    const int row = (pi & 3) * 16 + (lane >> 2);
    bufds16(rs, img + pi * 1024, (unsigned)(row * ld * 2 + (pi >> 2) * 64 + (lane & 3) * 16), (unsigned)(t * KT * ld * 2));
}

// "uni" image of a 64 x 64 bf16 tile (8 KB = 512 cells of 16 B; r = tile row, c = 16-byte chunk of its 128-byte row):
//     cell(r, c) = 16 ((r >> 2) 2 + (c >> 2)) + 4 (r & 3) + ((c & 3) ^ m(r >> 2)),     m(j) = (j & 1) | ((j >> 2 & 1) << 1)
// i.e. a block of 4 rows x 64 bytes is one 256-byte bank line, its 16 cells ordered [row & 3][chunk & 3 ^ m].
//   * the transposing read of a 32x32x16 A fragment (ds_read_b64_tr_b16: per 32-lane pass 4 rows x 64 bytes of one feature half) takes exactly one
//     bank line — conflict-free like the contiguous "tr" image;
//   * the row read (ds_read_b128, lane = row, one chunk column per pass) is served in 16-lane passes over rows {0-3, 12-15, 20-27} / {4-11, 16-19,
//     28-31}: r & 3 takes every value four times per pass, with r >> 2 in {0, 3, 5, 6} resp. {1, 2, 4, 7} — m() is distinct on each of the two sets, so
//     the 16 rows land in 16 distinct 16-byte slots — conflict-free like the XOR-swizzled "rows" image.
// The LDS side of the DMA is lane-linear (lane i of piece pi writes cell 64 pi + i), so the permutation sits on the SOURCE address
__device__ __forceinline__ int uni_m(int j) { return (j & 1) | (((j >> 2) & 1) << 1); }
__device__ __forceinline__ void dma_uni_piece(__amdgpu_buffer_rsrc_t rs, unsigned char* img, int pi, int lane, int ld, int t) {
    const int j = 2 * pi + (lane >> 5), d = (lane >> 4) & 1, slot = lane & 15;
    const int r = 4 * j + (slot >> 2), c = 4 * d + ((slot & 3) ^ uni_m(j));
    bufds16(rs, img + pi * 1024, (unsigned)(r * ld * 2 + (c << 4)), (unsigned)(t * KT * ld * 2));
}

// D[b][h][t] = sum_d dO[t][h*64+d] * O[t][h*64+d] (bf16 in, fp32 out); 8 lanes per (row, head), 8 features each
__global__ __launch_bounds__(256) void attn_bwd_prep_bf16_kernel(const __bf16* __restrict__ dout, const __bf16* __restrict__ out,
                                                                 float* __restrict__ D, int B, int H, int T, int lddo, int ldo) {
    const long long i = (blockIdx.x * 256ll + threadIdx.x) >> 3;          // (b, t, h) flat
    const int c8 = threadIdx.x & 7;
    const long long total = (long long)B * T * H;
    float s = 0.f;
    long long bt = 0;
    int h = 0;
    if (i < total) {
        h = (int)(i % H);
        bt = i / H;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(dout + bt * lddo + h * DH + c8 * 8);
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(out + bt * ldo + h * DH + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = __builtin_fmaf((float)a[e], (float)o[e], s);
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (i < total && c8 == 0) {
        const long long b = bt / T, t = bt - b * T;
        D[(b * H + h) * T + t] = s;
    }
}

// the wave's [feature][owner row] accumulators -> fp32 rows of `dst` (32 rows x 64 features), through the wave's 8 KB of LDS
__device__ __forceinline__ void store_transposed(const f32x16 (&acc)[2], unsigned char* Os, float* __restrict__ dst, int ld, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = acc[d][4 * j + e];
            const int c = 8 * d + 2 * j + half;                                   // 16-byte chunk of the row: features 4 c .. 4 c + 3
            *reinterpret_cast<f32x4*>(Os + l31 * 256 + ((c ^ (l31 & 15)) << 4)) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4), cp = lane & 15;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 256 + (cp << 4));
        *reinterpret_cast<f32x4*>(dst + (size_t)row * ld + ((cp ^ (row & 15)) << 2)) = val;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// the same as bf16 rows (the gradient's consumers — the c_attn dX and dW GEMMs of the bf16 arm — round it to bf16 on load anyway):
// 32 rows x 128 B through the wave's LDS, 16-byte chunk c stored at c ^ (row & 7)
__device__ __forceinline__ void store_transposed_bf16(const f32x16 (&acc)[2], unsigned char* Os, __bf16* __restrict__ dst, int ld, int lane) {
    const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[d][4 * j + e];
            const int c = 4 * d + j;                                              // features 8 c .. 8 c + 7; this half: + 4 half .. + 3
            *reinterpret_cast<bf16x4*>(Os + l31 * 128 + ((c ^ (l31 & 7)) << 4) + 8 * half) = o;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + (lane >> 3), cp = lane & 7;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 128 + (cp << 4));
        *reinterpret_cast<f32x4*>(dst + (size_t)row * ld + ((cp ^ (row & 7)) << 3)) = val;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------------- dQ
#ifndef ATB_DQ_RING
#define ATB_DQ_RING 3              // 3 slots = 72 KB: two workgroups per CU, two tiles ahead.  2 slots (48 KB: three per CU, one tile ahead; build.py
                                  // 'dq_ring2') measured the same within the order effect of an alternation (round 6: 162.3 vs 165.6 us for dQ + dK/dV
                                  // as second library, 160.5 vs 160.3 as first; bit-identical): occupancy is not what holds this kernel back
#endif
constexpr bool DQ_UNI = ATB_DQ_UNI != 0 && !(ATB_ABL & 8);
constexpr int DQ_TRK = (DQ_UNI || (ATB_ABL & 8)) ? 0 : 2 * IMG;
constexpr int DQ_SLOT = DQ_TRK + IMG < 2 * IMG ? 2 * IMG : DQ_TRK + IMG, DQ_RING = ATB_DQ_RING, DQ_NL = (ATB_ABL & 4) ? 0 : (DQ_UNI || (ATB_ABL & 1)) ? 4 : 6;      // K rows | V rows | K tr; 6 one-KB pieces per wave and tile

template <bool O16, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_bf16_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                                  const __bf16* __restrict__ v, const __bf16* __restrict__ dout,
                                                                  const float* __restrict__ lse, const float* __restrict__ Dv,
                                                                  void* __restrict__ dq, int H, int T, int ldq, int ldk, int ldv, int lddo,
                                                                  int lddq, float scale, int twin, uint32_t drop_thresh, float drop_scale,
                                                                  uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0, vf_attn_order order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    const int q0 = (int)order.blk[blockIdx.z] * OT;                  // heaviest owner block first (vf_common.h: vf_attn_block_order)
    const int qw0 = q0 + wave * 32;
    const int nviews = T / KT;
    const int qview = qw0 / KT;
    const bool active = qview < nviews;
    const Vis visible = make_vis(twin);

    const __bf16* kb_ = k + b * (size_t)T * ldk + h * DH;
    const __bf16* vb_ = v + b * (size_t)T * ldv + h * DH;
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(kb_), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(vb_), 0, 0x7fffffff, 0x00020000);

    // owner operands: Q and dO rows of this lane's query as B fragments (k = features 16 ks + 8 half .. + 7)
    const int qrow = active ? qw0 + l31 : 0;
    bf16x8 qb[4], dob[4];
    {
        const __bf16* qs = q + (b * (size_t)T + qrow) * ldq + h * DH + 8 * half;
        const __bf16* ds = dout + (b * (size_t)T + qrow) * lddo + h * DH + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qb[ks] = *reinterpret_cast<const bf16x8*>(qs + 16 * ks);
            dob[ks] = *reinterpret_cast<const bf16x8*>(ds + 16 * ks);
        }
    }
    const size_t stat = ((size_t)b * H + h) * T + qrow;
    const float lse2 = lse[stat] * LOG2E, D_q = Dv[stat];
    const float c2 = scale * LOG2E;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) qb[ks][e] = (__bf16)((float)qb[ks][e] * c2);          // the forward's q' (attention_dma.hip: FOLD)
    // loop-invariant C operands (made opaque so that they live in registers instead of being re-splatted per tile)
    f32x16 c_s, c_p;
    {
        const float cs = __builtin_amdgcn_logf(scale) - lse2;                                // v_log_f32 = log2
        const float cp = DROP ? -D_q / drop_scale : -D_q;
#pragma unroll
        for (int r = 0; r < 16; ++r) { c_s[r] = cs; c_p[r] = cp; }
        asm volatile("" : "+v"(c_s), "+v"(c_p));
    }
    // attention dropout: mask plane (scene, head), group q * (T / 4) + (key >> 2); this lane's keys of a tile are + 4 half + ...
    uint32_t drop_key = 0u, drop_q = 0u;
    if constexpr (DROP) {
        drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * H + h));
        drop_q = (uint32_t)qrow * (uint32_t)(T >> 2) + (uint32_t)half;
    }

    // key tiles some wave of this workgroup (two query views) sees
    const int va = q0 / KT, vb2 = va + 1;
#ifdef VF_X_ATB_VISLOOP
    unsigned long long need = 0;
    for (int kt = 0; kt < nviews; ++kt)
        if (visible(va, kt) || (vb2 < nviews && visible(vb2, kt))) need |= 1ull << kt;
    const unsigned long long mine = 0ull;
#else
    const unsigned long long need = vis_keys_of(visible, va, nviews) | (vb2 < nviews ? vis_keys_of(visible, vb2, nviews) : 0ull);
    const unsigned long long mine = active ? vis_keys_of(visible, qview, nviews) : 0ull;
#endif
    const int n = __builtin_popcountll(need);

    auto issue = [&](int seq, int kt) {
        unsigned char* slot = smem + (seq % DQ_RING) * DQ_SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = wave * 2 + j;
            if (ATB_ABL & 4) continue;
            if constexpr (DQ_UNI) {
                dma_uni_piece(k_rs, slot, pi, lane, ldk, kt);
                dma_uni_piece(v_rs, slot + IMG, pi, lane, ldv, kt);
                continue;
            }
            dma_rows_piece(k_rs, slot, pi, lane, ldk, kt);
            dma_rows_piece(v_rs, slot + IMG, pi, lane, ldv, kt);
            if (!(ATB_ABL & 1)) dma_tr_piece(k_rs, slot + DQ_TRK, pi, lane, ldk, kt);
        }
    };
    unsigned long long pend = need;                                  // tiles not issued yet
    auto next_tile = [&]() { const int t = __builtin_ctzll(pend); pend &= pend - 1; return t; };
    int issued = 0;
    for (; issued < DQ_RING - 1 && issued < n; ++issued) issue(issued, next_tile());

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    const unsigned row_off = (unsigned)(l31 * 128);
    const unsigned tr_off = (unsigned)((4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    // uni image (dma_uni_piece): row-read base of row l31, transposing-read bases for the 8-row groups with (r >> 4) & 1 = ks2
    const int uni_ml = uni_m(l31 >> 2);
    const unsigned uni_row = (unsigned)(512 * (l31 >> 2) + 64 * (l31 & 3));
    const int uni_cq = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const unsigned uni_tr[2] = {(unsigned)(512 * half + 64 * ((lane & 15) >> 2) + 16 * (uni_cq ^ half) + 8 * (lane & 1)),
                                (unsigned)(512 * half + 64 * ((lane & 15) >> 2) + 16 * (uni_cq ^ (half | 2)) + 8 * (lane & 1))};

    unsigned long long todo = need;
    for (int i = 0; i < n; ++i) {
        const int kt = __builtin_ctzll(todo);
        todo &= todo - 1;
        if (DQ_RING > 3 && issued - 2 > i) wait_loads<2 * DQ_NL>();               // tile i has landed (at most the next DQ_RING - 2 tiles are in flight)
        else if (issued - 1 > i) wait_loads<DQ_NL>();
        else wait_loads<0>();
        __builtin_amdgcn_s_barrier();                                              // ... for every wave; the slot of tile i - 1 is free
        if (issued < n) { issue(issued, next_tile()); ++issued; }
#ifdef VF_X_ATB_VISLOOP
        if (!active || !visible(qview, kt) || (ATB_ABL & 2)) continue;
#else
        if (!((mine >> kt) & 1ull) || (ATB_ABL & 2)) continue;
#endif
        const unsigned char* slot = smem + (i % DQ_RING) * DQ_SLOT;

        f32x16 st[2], dp[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            st[t2] = c_s;                                            // log2(scale) - lse log2 e
            dp[t2] = c_p;                                            // -D   (DROP: -D / c)
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const unsigned off = DQ_UNI ? uni_row + t2 * 4096 + (ks >> 1) * 256 + ((((unsigned)((ks & 1) * 2 + half)) ^ (unsigned)uni_ml) << 4)
                                            : row_off + t2 * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4);
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(slot + off);
                const bf16x8 c = *reinterpret_cast<const bf16x8*>(slot + IMG + off);
                st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ks], st[t2], 0, 0, 0);      // S^T = K.Q^T
                dp[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, dob[ks], dp[t2], 0, 0, 0);     // dP^T = V.dO^T
            }
        bf16x8 ds[2][2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                uint32_t w[2] = {0u, 0u};                            // registers 8 ks2 + 0..3 / + 4..7: keys 64 kt + 32 t2 + 16 ks2 + 4 half + {0..3} / {8..11}
                if constexpr (DROP) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) w[g] = vf_dropout_word(drop_key, drop_q + (uint32_t)(kt * 16 + t2 * 8 + ks2 * 4 + g * 2));
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = ks2 * 8 + e;
                    const float ps = __builtin_amdgcn_exp2f(st[t2][r]);                    // scale * P
                    float x = dp[t2][r];                                                     // dP - D   (DROP: dP - D / c)
                    if constexpr (DROP) x = vf_dropout_keep(w[e >> 2], e & 3, drop_thresh) ? x * drop_scale : -D_q;      // keep ? c dP - D : -D
                    ds[t2][ks2][e] = (__bf16)(ps * x);
                }
            }
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#ifdef VF_X_TRINTRIN
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const bf16x8 ka = DQ_UNI ? tr_frag(slot + DQ_TRK + uni_tr[ks2] + t2 * 4096 + ks2 * 2048 + d * 256, 1024)
                                             : tr_frag(slot + DQ_TRK + tr_off + d * 4096 + (t2 * 32 + ks2 * 16) * 64);
                    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, ds[t2][ks2], ot[d], 0, 0, 0);   // dQ^T += K^T.dS^T
                }
#else
            {   // (vf_tr_frag2_wait, vf_common.h: through the intrinsic these reads drained the DMA ring — the NEXT tile — before this product)
                bf16x8 ka0, ka1;
                if constexpr (DQ_UNI)
                    vf_tr_frag2_wait(ka0, ka1, vf_lds_addr(slot) + uni_tr[ks2], DQ_TRK + t2 * 4096 + ks2 * 2048, DQ_TRK + t2 * 4096 + ks2 * 2048 + 256, 1024);
                else
                    vf_tr_frag2_wait(ka0, ka1, vf_lds_addr(slot) + tr_off, DQ_TRK + (t2 * 32 + ks2 * 16) * 64, DQ_TRK + 4096 + (t2 * 32 + ks2 * 16) * 64, 8 * 64);
                ot[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka0, ds[t2][ks2], ot[0], 0, 0, 0);      // dQ^T += K^T.dS^T
                ot[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka1, ds[t2][ks2], ot[1], 0, 0, 0);
            }
#endif
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
    if (!active) return;
    if constexpr (O16) store_transposed_bf16(ot, smem + wave * 8192, reinterpret_cast<__bf16*>(dq) + (b * (size_t)T + qw0) * lddq + h * DH, lddq, lane);
    else store_transposed(ot, smem + wave * 8192, reinterpret_cast<float*>(dq) + (b * (size_t)T + qw0) * lddq + h * DH, lddq, lane);
}

// ---------------------------------------------------------------------------------------------------------------- dK, dV
constexpr bool KV_UNI = ATB_KV_UNI != 0 && !(ATB_ABL & 8);
constexpr int KV_QTR = (KV_UNI || (ATB_ABL & 8)) ? 0 : 2 * IMG, KV_DOTR = (KV_UNI || (ATB_ABL & 8)) ? IMG : 3 * IMG, KV_TAB = (KV_UNI || (ATB_ABL & 8)) ? 2 * IMG : 4 * IMG;
constexpr int KV_NL = (ATB_ABL & 4) ? 2 : (KV_UNI || (ATB_ABL & 1)) ? 6 : 10;
constexpr int KV_SLOT = KV_TAB + 512, KV_RING = (KV_UNI || (ATB_ABL & 8)) ? 3 : 2;      // Q rows | dO rows | Q tr | dO tr | lse[64] | D[64]; 10 loads per wave and tile

template <bool O16, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_bf16_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                                   const __bf16* __restrict__ v, const __bf16* __restrict__ dout,
                                                                   const float* __restrict__ lse, const float* __restrict__ Dv,
                                                                   void* __restrict__ dk, void* __restrict__ dv, int H, int T, int ldq, int ldk,
                                                                   int ldv, int lddo, int lddk, int lddv, float scale, int twin,
                                                                   uint32_t drop_thresh, float drop_scale, uint32_t drop_seed, uint32_t drop_site,
                                                                   uint32_t drop_plane0, vf_attn_order order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    const int k0 = (int)order.blk[blockIdx.z] * OT;
    const int kw0 = k0 + wave * 32;
    const int nviews = T / KT;
    const int kview = kw0 / KT;
    const bool active = kview < nviews;
    const Vis visible = make_vis(twin);

    const __bf16* qb_ = q + b * (size_t)T * ldq + h * DH;
    const __bf16* dob_ = dout + b * (size_t)T * lddo + h * DH;
    const float* lse_b = lse + ((size_t)b * H + h) * T;
    const float* D_b = Dv + ((size_t)b * H + h) * T;
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(qb_), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t do_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(dob_), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t l_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(lse_b), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(D_b), 0, 0x7fffffff, 0x00020000);

    // owner operands: K and V rows of this lane's key as B fragments
    const int krow = active ? kw0 + l31 : 0;
    bf16x8 kb[4], vb[4];
    {
        const __bf16* ks_ = k + (b * (size_t)T + krow) * ldk + h * DH + 8 * half;
        const __bf16* vs_ = v + (b * (size_t)T + krow) * ldv + h * DH + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kb[ks] = *reinterpret_cast<const bf16x8*>(ks_ + 16 * ks);
            vb[ks] = *reinterpret_cast<const bf16x8*>(vs_ + 16 * ks);
        }
    }
    const float c2 = scale * LOG2E;
    // attention dropout: this lane's key is element krow & 3 of group (krow >> 2) of every query's row of groups
    uint32_t drop_key = 0u, drop_k = 0u;
    const int drop_j = krow & 3;
    const float drop_cs = drop_scale * scale;
#ifndef VF_X_DKV_ROT3
    const uint32_t drop_rr = vf_dropout_rotr(drop_j);
#endif
    if constexpr (DROP) {
        drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * H + h));
        drop_k = (uint32_t)(krow >> 2) + (uint32_t)(4 * half) * (uint32_t)(T >> 2);
    }

    // query tiles that see a key view of this workgroup
    const int va = k0 / KT, vb2 = va + 1;
#ifdef VF_X_ATB_VISLOOP
    unsigned long long need = 0;
    for (int qt = 0; qt < nviews; ++qt)
        if (visible(qt, va) || (vb2 < nviews && visible(qt, vb2))) need |= 1ull << qt;
    const unsigned long long mine = 0ull;
#else
    const unsigned long long need = vis_queries_of(visible, va, nviews) | (vb2 < nviews ? vis_queries_of(visible, vb2, nviews) : 0ull);
    const unsigned long long mine = active ? vis_queries_of(visible, kview, nviews) : 0ull;
#endif
    const int n = __builtin_popcountll(need);

    auto issue = [&](int seq, int qt) {
        unsigned char* slot = smem + (seq % KV_RING) * KV_SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = wave * 2 + j;
            if (ATB_ABL & 4) continue;
            if constexpr (KV_UNI) {
                dma_uni_piece(q_rs, slot, pi, lane, ldq, qt);
                dma_uni_piece(do_rs, slot + IMG, pi, lane, lddo, qt);
                continue;
            }
            dma_rows_piece(q_rs, slot, pi, lane, ldq, qt);
            dma_rows_piece(do_rs, slot + IMG, pi, lane, lddo, qt);
            if (ATB_ABL & 1) continue;
            dma_tr_piece(q_rs, slot + KV_QTR, pi, lane, ldq, qt);
            dma_tr_piece(do_rs, slot + KV_DOTR, pi, lane, lddo, qt);
        }
        // lse / D of the tile's 64 queries (256 B each): 16 lanes x 16 B; every wave issues them (same bytes) so that all waves count
        // the same number of loads per tile
        if (lane < 16) {
            bufds16(l_rs, slot + KV_TAB, (unsigned)(lane * 16), (unsigned)(qt * KT * 4));
            bufds16(d_rs, slot + KV_TAB + 256, (unsigned)(lane * 16), (unsigned)(qt * KT * 4));
        }
    };
    unsigned long long pend = need;
    auto next_tile = [&]() { const int t = __builtin_ctzll(pend); pend &= pend - 1; return t; };
    int issued = 0;
    for (; issued < KV_RING - 1 && issued < n; ++issued) issue(issued, next_tile());

    f32x16 dvacc[2], dkacc[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dvacc[d][r] = 0.f; dkacc[d][r] = 0.f; }
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    const unsigned row_off = (unsigned)(l31 * 128);
    const unsigned tr_off = (unsigned)((4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    // uni image (dma_uni_piece): this lane's row-read base (row l31 of a 32-row half, chunk column added per k-step) and its two transposing-read bases
    // (rows 4 half + ((lane & 15) >> 2) of an 8-row group whose (r >> 4) & 1 = ks2; 8-byte piece (lane & 3) of the 32-byte column block (lane >> 4) & 1)
    const int uni_ml = uni_m(l31 >> 2);
    const unsigned uni_row = (unsigned)(512 * (l31 >> 2) + 64 * (l31 & 3));
    const int uni_cq = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const unsigned uni_tr[2] = {(unsigned)(512 * half + 64 * ((lane & 15) >> 2) + 16 * (uni_cq ^ half) + 8 * (lane & 1)),
                                (unsigned)(512 * half + 64 * ((lane & 15) >> 2) + 16 * (uni_cq ^ (half | 2)) + 8 * (lane & 1))};

    unsigned long long todo = need;
    for (int i = 0; i < n; ++i) {
        const int qt = __builtin_ctzll(todo);
        todo &= todo - 1;
        if (KV_RING > 2 && issued - 1 > i) wait_loads<KV_NL>(); else wait_loads<0>();   // (ring of 2: nothing else is in flight yet)
        __builtin_amdgcn_s_barrier();
        if (issued < n) { issue(issued, next_tile()); ++issued; }
#ifdef VF_X_ATB_VISLOOP
        if (!active || !visible(qt, kview) || (ATB_ABL & 2)) continue;
#else
        if (!((mine >> qt) & 1ull) || (ATB_ABL & 2)) continue;
#endif
        const unsigned char* slot = smem + (i % KV_RING) * KV_SLOT;

#pragma unroll
        for (int u = 0; u < 2; ++u) {                                              // the tile's two 32-query halves
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const unsigned off = KV_UNI ? uni_row + u * 4096 + (ks >> 1) * 256 + ((((unsigned)((ks & 1) * 2 + half)) ^ (unsigned)uni_ml) << 4)
                                            : row_off + u * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4);
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(slot + off);
                const bf16x8 c = *reinterpret_cast<const bf16x8*>(slot + IMG + off);
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, kb[ks], st, 0, 0, 0);              // S = Q.K^T   [query][key]
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, vb[ks], dp, 0, 0, 0);              // dP = dO.V^T
            }
            // accumulator row r = query 32 u + (r & 3) + 8 (r >> 2) + 4 half: its lse / D from the tile's table
            bf16x8 pf[2], sf[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(slot + KV_TAB + (32 * u + 8 * j + 4 * half) * 4);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(slot + KV_TAB + 256 + (32 * u + 8 * j + 4 * half) * 4);
                // accumulator row r = 4 j + e = query 64 qt + 32 u + 8 j + e + 4 half; its mask word belongs to (query, key >> 2): the SAME word in the
                // four lanes of a quad (keys 4 m .. 4 m + 3, each using its own byte).  Lane c of the quad hashes the word of e = c, a DPP quad
                // broadcast hands it round: 4 hashes per 16 scores instead of 16 (two 8-cycle integer multiplies each) — same words, same masks
                uint32_t w4[4] = {0u, 0u, 0u, 0u};
                if constexpr (DROP) {
                    const int wq = (int)vf_dropout_word(drop_key, drop_k + (uint32_t)(qt * 64 + 32 * u + 8 * j + (lane & 3)) * (uint32_t)(T >> 2));
                    w4[0] = (uint32_t)__builtin_amdgcn_mov_dpp(wq, 0x00, 0xF, 0xF, true);      // quad_perm [0, 0, 0, 0]
                    w4[1] = (uint32_t)__builtin_amdgcn_mov_dpp(wq, 0x55, 0xF, 0xF, true);      // [1, 1, 1, 1]
                    w4[2] = (uint32_t)__builtin_amdgcn_mov_dpp(wq, 0xAA, 0xF, 0xF, true);      // [2, 2, 2, 2]
                    w4[3] = (uint32_t)__builtin_amdgcn_mov_dpp(wq, 0xFF, 0xF, 0xF, true);      // [3, 3, 3, 3]
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * j + e;
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c2, -l4[e] * LOG2E));
                    float dpr = dp[r];
                    bool keep = true;
                    if constexpr (DROP) {
#ifdef VF_X_DKV_HASH_PER_ELEMENT
                        const uint32_t w = vf_dropout_word(drop_key, drop_k + (uint32_t)(qt * 64 + 32 * u + 8 * j + e) * (uint32_t)(T >> 2));
#else
                        const uint32_t w = w4[e];
#endif
#ifdef VF_X_DKV_ROT3
                        keep = vf_dropout_keep(w, drop_j, drop_thresh);
#else
                        keep = vf_dropout_keep_rotr(w, drop_rr, drop_thresh);           // (one v_alignbit_b32 instead of shl / shr / or: same decision)
#endif
#ifdef VF_X_DKV_SEL2
                        dpr = keep ? dpr * drop_scale : 0.f;
#endif
                    }
#ifndef VF_X_DKV_SEL2
                    if constexpr (DROP) {
                        // one select instead of two: dS = scale (c keep P dP - P D) with keep P formed once (it is dV's operand too)
                        const float pk = keep ? p : 0.f;
                        pf[r >> 3][r & 7] = (__bf16)pk;                         // (its 1 / (1 - rate) joins dV at the end)
                        sf[r >> 3][r & 7] = (__bf16)__builtin_fmaf(pk, dpr * drop_cs, -(p * (d4[e] * scale)));
                        continue;
                    }
#endif
                    pf[r >> 3][r & 7] = keep ? (__bf16)p : (__bf16)0.f;          // (its 1 / (1 - rate) joins dV at the end)
                    sf[r >> 3][r & 7] = (__bf16)(p * (dpr - d4[e]) * scale);
                }
            }
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
#ifdef VF_X_TRINTRIN
                    const unsigned off = KV_UNI ? uni_tr[ks2] + u * 4096 + ks2 * 2048 + d * 256 : tr_off + d * 4096 + (u * 32 + ks2 * 16) * 64;
                    const bf16x8 oa = tr_frag(slot + KV_DOTR + off, KV_UNI ? 1024 : 8 * 64);
                    const bf16x8 qa = tr_frag(slot + KV_QTR + off, KV_UNI ? 1024 : 8 * 64);
#else
                    bf16x8 oa, qa;      // (vf_tr_frag2_wait: as intrinsics these reads made hipcc wait for the NEXT tile's DMA — no overlap at all)
                    if constexpr (KV_UNI)
                        vf_tr_frag2_wait(oa, qa, vf_lds_addr(slot) + uni_tr[ks2], KV_DOTR + u * 4096 + ks2 * 2048 + d * 256,
                                         KV_QTR + u * 4096 + ks2 * 2048 + d * 256, 1024);
                    else
                        vf_tr_frag2_wait(oa, qa, vf_lds_addr(slot) + tr_off, KV_DOTR + d * 4096 + (u * 32 + ks2 * 16) * 64,
                                         KV_QTR + d * 4096 + (u * 32 + ks2 * 16) * 64, 8 * 64);
#endif
                    dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(oa, pf[ks2], dvacc[d], 0, 0, 0);   // dV^T += dO^T.P
                    dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, sf[ks2], dkacc[d], 0, 0, 0);   // dK^T += Q^T.dS
                }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!active) return;
    unsigned char* Os = smem + wave * 8192;
    if constexpr (DROP) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) dvacc[d][r] *= drop_scale;
    }
    if constexpr (O16) {
        store_transposed_bf16(dkacc, Os, reinterpret_cast<__bf16*>(dk) + (b * (size_t)T + kw0) * lddk + h * DH, lddk, lane);
        store_transposed_bf16(dvacc, Os, reinterpret_cast<__bf16*>(dv) + (b * (size_t)T + kw0) * lddv + h * DH, lddv, lane);
    } else {
        store_transposed(dkacc, Os, reinterpret_cast<float*>(dk) + (b * (size_t)T + kw0) * lddk + h * DH, lddk, lane);
        store_transposed(dvacc, Os, reinterpret_cast<float*>(dv) + (b * (size_t)T + kw0) * lddv + h * DH, lddv, lane);
    }
}

}  // namespace

extern "C" {

int vf_attn_bwd_prep_bf16(const void* dout, const void* out, float* D, int B, int H, int T, int lddo, int ldo, void* stream) {
    if (!dout || !out || !D || B <= 0 || H <= 0 || T <= 0) return VF_ERR_BAD_ARG;
    if (lddo < H * DH || ldo < H * DH || ((lddo | ldo) & 7)) return VF_ERR_BAD_ARG;
    const long long total = (long long)B * T * H * 8;
    hipLaunchKernelGGL(attn_bwd_prep_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const __bf16*>(dout), reinterpret_cast<const __bf16*>(out), D, B, H, T, lddo, ldo);
    return vf_last_status();
}

int vf_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* D, void* dq, void* dk,
                     void* dv, int out_bf16, int B, int H, int T, int L, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk, int lddv,
                     float scale, int twin_view, float drop_rate, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0, void* stream) {
    // dq == NULL: only the dK / dV launch; dk == dv == NULL: only the dQ launch (the two are independent: a caller may issue them on two streams)
    const bool do_q = dq != nullptr, do_kv = dk != nullptr || dv != nullptr;
    if (!q || !k || !v || !dout || !lse || !D || (!do_q && !do_kv) || (do_kv && (!dk || !dv)) || B <= 0 || H <= 0 || T <= 0 || !(scale > 0.f)) return VF_ERR_BAD_ARG;
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return VF_ERR_BAD_ARG;
    if (!do_q) { dq = dk; lddq = lddk; }                      // (placeholders for the argument checks below; never written)
    if (!do_kv) { dk = dv = dq; lddk = lddv = lddq; }
    if (L != KT || T % KT != 0 || T / KT > 64) return VF_ERR_UNSUPPORTED;                     // 64-token views, at most 64 of them (tile bit masks)
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || lddo < H * DH || lddq < H * DH || lddk < H * DH || lddv < H * DH) return VF_ERR_BAD_ARG;
    if (((ldq | ldk | ldv | lddo) & 7) || ((lddq | lddk | lddv) & (out_bf16 ? 7 : 3))) return VF_ERR_BAD_ARG;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) return VF_ERR_UNSUPPORTED;
    const size_t ldmax = (size_t)(ldq > ldk ? ldq : ldk) > (size_t)(ldv > lddo ? ldv : lddo) ? (size_t)(ldq > ldk ? ldq : ldk) : (size_t)(ldv > lddo ? ldv : lddo);
    if ((size_t)T * ldmax * 2 >= (1ull << 31)) return VF_ERR_UNSUPPORTED;                     // 32-bit buffer offsets per (scene, head)
    hipStream_t s = (hipStream_t)stream;
    const int nblk = (T + OT - 1) / OT;
    if (nblk > 64) return VF_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)H, (unsigned)B, (unsigned)nblk);
    const vf_attn_order ord_q = vf_attn_block_order(T / KT, OT / KT, nblk, twin_view, false, ATB_HEAVY_FIRST != 0);
    const vf_attn_order ord_k = vf_attn_block_order(T / KT, OT / KT, nblk, twin_view, true, ATB_HEAVY_FIRST != 0);
    const __bf16 *q_ = reinterpret_cast<const __bf16*>(q), *k_ = reinterpret_cast<const __bf16*>(k), *v_ = reinterpret_cast<const __bf16*>(v),
                 *do_ = reinterpret_cast<const __bf16*>(dout);
    const uint32_t thr = vf_dropout_thresh(drop_rate);
    const float dsc = 1.0f / (1.0f - drop_rate);
    auto launch = [&](auto o16, auto drop) -> int {
        constexpr bool O16 = decltype(o16)::value, DROP = decltype(drop)::value;
        static unsigned long long attr_devs = 0;                   // (one flag per instantiation pair)
        if (vf_attr_needed(&attr_devs)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_bf16_kernel<O16, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, DQ_RING * DQ_SLOT);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_bf16_kernel<O16, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, KV_RING * KV_SLOT);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr_devs);
        }
        if (do_q) {
            hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<O16, DROP>), grid, dim3(256), (size_t)DQ_RING * DQ_SLOT, s, q_, k_, v_, do_, lse, D, dq, H, T, ldq, ldk,
                               ldv, lddo, lddq, scale, twin_view, thr, dsc, drop_seed, drop_site, drop_plane0, ord_q);
            const int st = vf_last_status();
            if (st) return st;
        }
        if (do_kv)
            hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<O16, DROP>), grid, dim3(256), (size_t)KV_RING * KV_SLOT, s, q_, k_, v_, do_, lse, D, dk, dv, H, T, ldq,
                               ldk, ldv, lddo, lddk, lddv, scale, twin_view, thr, dsc, drop_seed, drop_site, drop_plane0, ord_k);
        return vf_last_status();
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (drop_rate > 0.f) return out_bf16 ? launch(T_{}, T_{}) : launch(F_{}, T_{});
    return out_bf16 ? launch(T_{}, F_{}) : launch(F_{}, F_{});
}

}  // extern "C"
