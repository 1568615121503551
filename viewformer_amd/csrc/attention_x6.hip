// fp32-EQUIVALENT block-causal attention on the bf16 matrix pipe ("x6"), gfx950: the structure of attention_bf16.hip with
// every operand of both contractions (Q, K, V and the probabilities) split exactly into three bf16 pieces and every fp32
// product evaluated as its six partial products of weight >= 2^-24 (see conv3_halo_x6.hip).  Same semantics and masks as
// attention_f32.hip (un-scaled q.k^T, "w*m - 1e4*(1-m)", twin views, streams); the softmax is the same fp32 code.  Per
// 64-key tile a wave issues 96 bf16 MFMAs (3072 matrix-pipe cycles) instead of 128 f32 MFMAs (8192 cycles).
// LDS: K tile [key][plane h|m|l][64 dh] bf16 (400-byte rows), V^T [d][plane][64 keys] bf16 (392-byte rows).
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r1 = x - (float)h;
    m = (__bf16)r1;
    l = (__bf16)(r1 - (float)m);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int DH = 64;
constexpr int QT = 128;     // queries per workgroup
constexpr int KT = 64;      // keys per tile
constexpr int K_LDB = 400;  // bytes per K row in LDS: 3 planes x 128 B + 16 B pad (25 x 16 B: conflict-free ds_read_b128)
constexpr int VT_LDB = 392; // bytes per V^T row: 3 planes x 128 B + 8 B pad (98 = 34 mod 64 banks: 32 rows hit 32 distinct bank pairs)

__global__ __launch_bounds__(256, 2) void attn_blockcausal_x6_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ v, float* __restrict__ out,
                                                                  int T, int L, int ldq, int ldk, int ldv, int ldo,
                                                                  float scale, int skip_masked, int twin) {
    __shared__ __attribute__((aligned(16))) unsigned char Ks[KT * K_LDB];
    __shared__ __attribute__((aligned(16))) unsigned char Vt[DH * VT_LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    // grid (H, B, query tiles): the query tiles of one (scene, head) are gridDim.x * gridDim.y ids apart, i.e. on the SAME XCD
    // whenever H * B % 8 == 0, and share its L2 copy of that head's K / V (consecutive ids go round-robin over the 8 XCDs)
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    // query blocks heaviest first (round 6): under the plain / twin masks a later block sees more key tiles, and a launch that dispatches the light
    // blocks first ends on a partly filled round of its longest workgroups (attention_dma.hip; the streams mask is not monotone: index order there)
    const int q0 = (int)(twin > -2 ? gridDim.z - 1 - blockIdx.z : blockIdx.z) * QT;

    const float* qb_ptr = q + b * (size_t)T * ldq + h * DH;
    const float* kb = k + b * (size_t)T * ldk + h * DH;
    const float* vb = v + b * (size_t)T * ldv + h * DH;
    float* ob = out + b * (size_t)T * ldo + h * DH;

    // ---- Q fragment (B operand): qb[ks][e] = bf16(Q[qrow][16 ks + 8 half + e]) ------------------------
    const int qrow = q0 + wave * 32 + l31;
    const bool qvalid = qrow < T;
    bf16x8 qb[3][4];
    {
        const float* src = qb_ptr + (size_t)(qvalid ? qrow : 0) * ldq + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + 16 * ks);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(e < 4 ? t0[e & 3] : t1[e & 3], h, m, l);
                qb[0][ks][e] = h; qb[1][ks][e] = m; qb[2][ks][e] = l;
            }
        }
    }
    const int qview = (L > 0) ? qrow / L : 0;
    // visibility of key view kv from query view qv.  Plain block-causal: kv <= qv.  With `twin` = Vc >= 0 the
    // views Vc, Vc+1, ... are alternatives of the SAME sequence position (the MASK view and the LOC view of
    // the evaluator's two passes, = the reference's branch streams, branching_attention.py:94-125): each sees
    // the common prefix and itself, never a sibling.
    //   twin <= -2: STREAMS mode with Sv = -twin views per stream: view index = stream*Sv + i.  Stream 0 is the
    //   main block-causal sequence; a branch stream s >= 1 at position i sees main views j < i and its own
    //   (s, i) tile only — compute_causal_block_multiend_attention for every position at once
    //   (branching_attention.py:94-125; used by the multi-context evaluators and the training graph).
    const int Vc = twin >= 0 ? twin : 0x3fffffff;
    const int Sv = twin <= -2 ? -twin : 0;
    auto visible = [&](int qv, int kv) {
        if (Sv > 0) {
            const int qs = qv / Sv, qi = qv - qs * Sv;
            const int ks = kv / Sv, ki = kv - ks * Sv;
            return qs == 0 ? (ks == 0 && ki <= qi) : ((ks == 0 && ki < qi) || kv == qv);
        }
        return kv == qv || min(kv, Vc) < min(qv, Vc);
    };
    const bool uniform_views = L > 0 && (L % KT) == 0;      // a key tile and a wave's 32 queries sit inside one view
    const int qview_w = (L > 0) ? __builtin_amdgcn_readfirstlane((q0 + wave * 32) / L) : 0;

    // number of key tiles the workgroup / this wave must visit
    int kmax = T, kmax_w = T;   // exclusive
    if (L > 0 && skip_masked) {
        const int last_q = min(q0 + QT, T) - 1;
        kmax = min(T, (last_q / L + 1) * L);
        const int last_qw = min(q0 + wave * 32 + 32, T) - 1;
        kmax_w = last_qw < 0 ? 0 : min(T, (last_qw / L + 1) * L);
        if (q0 + wave * 32 >= T) kmax_w = 0;
    }
    const int ntiles = (kmax + KT - 1) / KT;
    const int ntiles_w = __builtin_amdgcn_readfirstlane((kmax_w + KT - 1) / KT);

    // staging map: thread -> float4 column tid&15; K rows (tid>>4) + 16 i; V key PAIRS 2p, 2p+1 with p = (tid>>4) + 16 i
    // (adjacent keys of one feature become one 32-bit write of the transposed tile)
    const int s_col4 = tid & 15;
    const int s_row0 = tid >> 4;
    f32x4 kreg[4], vreg[4];
    auto prefetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kt * KT + s_row0 + 16 * i;
            const int vkey = kt * KT + 2 * (s_row0 + 16 * (i >> 1)) + (i & 1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (key < T) a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * ldk + s_col4 * 4);
            if (vkey < T) c = *reinterpret_cast<const f32x4*>(vb + (size_t)vkey * ldv + s_col4 * 4);
            kreg[i] = a;
            vreg[i] = c;
        }
    };

    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    float m_run = -INFINITY;
    float l_run = 0.f;

    prefetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x4 kh, km, kl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h, m, l;
                split3(kreg[i][e], h, m, l);
                kh[e] = h; km[e] = m; kl[e] = l;
            }
            unsigned char* dst = Ks + (s_row0 + 16 * i) * K_LDB + s_col4 * 8;
            *reinterpret_cast<bf16x4*>(dst) = kh;
            *reinterpret_cast<bf16x4*>(dst + 128) = km;
            *reinterpret_cast<bf16x4*>(dst + 256) = kl;
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int p2 = 2 * (s_row0 + 16 * ip);                   // even key of the pair
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16x2 ph, pm, pl;
                __bf16 h, m, l;
                split3(vreg[2 * ip][e], h, m, l);
                ph[0] = h; pm[0] = m; pl[0] = l;
                split3(vreg[2 * ip + 1][e], h, m, l);
                ph[1] = h; pm[1] = m; pl[1] = l;
                unsigned char* dst = Vt + (s_col4 * 4 + e) * VT_LDB + p2 * 2;
                *reinterpret_cast<bf16x2*>(dst) = ph;
                *reinterpret_cast<bf16x2*>(dst + 128) = pm;
                *reinterpret_cast<bf16x2*>(dst + 256) = pl;
            }
        }
        __syncthreads();
        if (kt + 1 < ntiles) prefetch(kt + 1);
        // every key of this tile is masked for this wave's 32 queries -> contributes exactly 0.0f
        if (kt >= ntiles_w) continue;
        if (skip_masked && uniform_views && !visible(qview_w, (kt * KT) / L)) continue;

        // ---- S^T = K . Q^T ----------------------------------------------------------------
        f32x16 st[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[t2][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                bf16x8 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[pl] = *reinterpret_cast<const bf16x8*>(Ks + (t2 * 32 + l31) * K_LDB + pl * 128 + ks * 32 + half * 16);
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0};      // plane 0 = h, 1 = m, 2 = l; smallest products first
                constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t]], qb[PB[t]][ks], st[t2], 0, 0, 0);
            }
        }

        // ---- mask + online softmax (lane = one query; its 32 keys of this tile) ----------------
        float mx = -INFINITY;
        // a tile that lies inside one view, is visible to the wave (we did not skip it) and has no padding keys
        // needs no per-element masking at all (wave-uniform): the common case for L = 64
        const bool plain = (kt * KT + KT <= T) &&
                           (L == 0 || (uniform_views && visible(qview_w, (kt * KT) / L)));   // (dense mode visits masked tiles too)
        if (plain) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float s = st[t2][r] * scale;
                    st[t2][r] = s;
                    mx = fmaxf(mx, s);
                }
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * KT + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float s = st[t2][r] * scale;
                    if (L > 0 && !visible(qview, key / L)) s = -1e4f;     // w*m - 1e4*(1-m)
                    if (key >= T) s = -INFINITY;                   // padding keys do not exist
                    st[t2][r] = s;
                    mx = fmaxf(mx, s);
                }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        // exp via the hardware exp2 (v_exp_f32, 1 ulp) on a pre-scaled argument: 2 VALU instead of ~15 for the
        // libm expf.  Relative error of a weight <= |x| * 6e-8 (argument rounding), far inside the fp32-class
        // tolerance of the logits; the 32 exps per lane per tile were ~60 % of the kernel's VALU work.
        constexpr float LOG2E = 1.4426950408889634f;
        // (x - m) * log2e, not fma(x, log2e, -m*log2e): a masked-only tile must give alpha == 1.0f EXACTLY so that
        // skipping it is bit-identical to visiting it
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);   // 0 on the first tile (m_run = -inf)
        float psum = 0.f;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f((st[t2][r] - m_new) * LOG2E);
                st[t2][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;

        // ---- O^T += V^T . P^T -------------------------------------------------------------
        // k-step (t2, ks2) covers the lane's accumulator rows r = 8 ks2 .. 8 ks2 + 7 = keys 32 t2 + 16 ks2 + 8 (e>>2) + 4 half + (e&3)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                bf16x8 pb[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 h, m, l;
                    split3(st[t2][ks2 * 8 + e], h, m, l);
                    pb[0][e] = h; pb[1][e] = m; pb[2][e] = l;
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vrow = Vt + (d * 32 + l31) * VT_LDB + (t2 * 32 + 16 * ks2 + 4 * half) * 2;
                    bf16x8 va[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow + pl * 128);
                        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + pl * 128 + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[pl][e] = v0[e]; va[pl][4 + e] = v1[e]; }
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
                    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
                        ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[PA[t]], pb[PB[t]], ot[d], 0, 0, 0);
                }
            }
    }

    // ---- normalise and store: lane = query, regs 4j..4j+3 = 4 consecutive features -------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qvalid) {
        float* orow = ob + (size_t)qrow * ldo + 4 * half;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[d][4 * j + e] / l_tot;
                *reinterpret_cast<f32x4*>(orow + d * 32 + 8 * j) = o;
            }
    }
}

}  // namespace

extern "C" {

int vf_attn_blockcausal_x6(const float* q, const float* k, const float* v, float* out, int B, int H, int T, int L,
                            int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view,
                            void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || T <= 0 || L < 0) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | ldo) & 3) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    hipLaunchKernelGGL(attn_blockcausal_x6_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, out, T, L, ldq, ldk,
                       ldv, ldo, scale, skip_masked, twin_view);
    return vf_last_status();
}

}  // extern "C"
