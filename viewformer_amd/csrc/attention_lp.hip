// Low-precision block-causal attention of the tolerance-bounded transformer arms, gfx950: one kernel, two operand formats.
//   MODE 0  bf16 (v_mfma_f32_32x32x16_bf16)       — Q, K, V and the probabilities rounded to bf16, fp32 sums and softmax
//   MODE 1  fp8  (v_mfma_f32_32x32x16_fp8_fp8)    — the same with OCP e4m3 operands (BASELINE configs[4] "fp8 MFMA attention"); the
//           probabilities are carried at 2^8 times their value so that the tail of the softmax stays above e4m3's subnormals
// Same semantics as attention_f32.hip (un-scaled q.k^T, "w*m - 1e4*(1-m)" block mask incl. twin views and streams, softmax, .v;
// viewformer/models/branching_attention.py:5-18,41-61,82-126) and the same transposed-score trick as the first bf16 kernel
// (attention_bf16.hip): S^T = K.Q^T puts a query in a lane and its keys in that lane's accumulator registers, so the probabilities go
// straight back in as the B operand of O^T += V^T.P^T with V^T read from LDS in the matching key order.
//
// What changed against attention_bf16.hip (7.8 % of the bf16 matrix peak, softmax- and staging-bound):
//   * a wave owns 64 queries = TWO 32-query tiles that share every K / V^T fragment read from LDS (half the LDS reads and half the
//     K/V staging work and barriers per MFMA) and give the scheduler two independent score tiles: the MFMAs of one run beside the
//     softmax VALU of the other.  With L = 64 tokens per view a wave is exactly one view: masks are wave-uniform.
//   * the softmax is 5 VALU per score instead of 8: scale folded into the exp2 argument (one fma), v_max3 for the running maximum,
//     the accumulator rescale skipped while the maximum does not move (wave-uniform test), v_cvt_pk for the operand rounding.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int DH = 64;
constexpr int QW = 64;      // queries per wave (2 MFMA tiles)
constexpr int QT = 256;     // queries per workgroup
constexpr int KT = 64;      // keys per tile
constexpr float P_SCALE_FP8 = 256.0f;

template <int MODE> struct Fmt;
template <> struct Fmt<0> {
    typedef bf16x8 frag;
    static constexpr int EB = 2;                 // bytes per element
    static constexpr int K_LDB = 144;            // K row [key][dh]: 128 B + 16 (conflict-free ds_read_b128)
    static constexpr int VT_LDB = 136;           // V^T row [feature][key]: 128 B + 8
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Fmt<1> {
    typedef long frag;
    static constexpr int EB = 1;
    static constexpr int K_LDB = 72;             // 64 B + 8 (18-word stride: 32 rows hit 32 distinct bank pairs with ds_read_b64)
    static constexpr int VT_LDB = 72;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0); }
};

// 8 consecutive fp32 values -> one operand fragment
template <int MODE>
__device__ __forceinline__ typename Fmt<MODE>::frag pack8(const float (&x)[8]) {
    if constexpr (MODE == 0) {
        bf16x8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = (__bf16)x[e];
        return r;
    } else {
        // (callers clamp to e4m3's finite range: v_cvt_pk_fp8_f32 does not saturate)
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2], x[3], lo, true);
        int hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[4], x[5], 0, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[6], x[7], hi, true);
        return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void attn_lp_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                         float* __restrict__ out, int T, int L, int ldq, int ldk, int ldv, int ldo,
                                                         float scale, int skip_masked, int twin, int out16, int in16) {
    using F = Fmt<MODE>;
    typedef typename F::frag frag;
    __shared__ __attribute__((aligned(16))) unsigned char Ks[KT * F::K_LDB];
    __shared__ __attribute__((aligned(16))) unsigned char Vt[DH * F::VT_LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int l31 = lane & 31;
    // grid (H, B, query tiles): the query tiles of one (scene, head) sit H*B ids apart = on one XCD whenever H*B % 8 == 0
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    // query blocks heaviest first (round 6): under the plain / twin masks a later block sees more key tiles, and a launch that dispatches the light
    // blocks first ends on a partly filled round of its longest workgroups (attention_dma.hip; the streams mask is not monotone: index order there)
    const int q0 = (int)(twin > -2 ? gridDim.z - 1 - blockIdx.z : blockIdx.z) * QT;
    const int qw0 = q0 + wave * QW;                      // this wave's first query

    const float* qf = q + b * (size_t)T * ldq + h * DH;
    const float* kf = k + b * (size_t)T * ldk + h * DH;
    const float* vf = v + b * (size_t)T * ldv + h * DH;
    const __bf16* q16 = reinterpret_cast<const __bf16*>(q) + b * (size_t)T * ldq + h * DH;
    const __bf16* k16 = reinterpret_cast<const __bf16*>(k) + b * (size_t)T * ldk + h * DH;
    const __bf16* v16 = reinterpret_cast<const __bf16*>(v) + b * (size_t)T * ldv + h * DH;

    // ---- Q fragments (B operand of S^T = K.Q^T): qb[u][ks] = Q[qw0 + 32 u + l31][16 ks + 8 half + 0..7]
    frag qb[2][4];
    int qrow[2];
    bool qvalid[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        qrow[u] = qw0 + 32 * u + l31;
        qvalid[u] = qrow[u] < T;
        const size_t ro = (size_t)(qvalid[u] ? qrow[u] : 0) * ldq + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float x[8];
            if (in16) {
                const bf16x8 t = *reinterpret_cast<const bf16x8*>(q16 + ro + 16 * ks);
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (float)t[e];
            } else {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(qf + ro + 16 * ks);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(qf + ro + 16 * ks + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { x[e] = t0[e]; x[4 + e] = t1[e]; }
            }
            if (MODE == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __builtin_amdgcn_fmed3f(x[e], -448.f, 448.f);
            }
            qb[u][ks] = pack8<MODE>(x);
        }
    }
    // visibility of key view kv from query view qv (see attention_f32.hip): plain block-causal kv <= qv; twin = Vc >= 0: views
    // Vc, Vc+1, ... are alternative endings (each sees the prefix and itself); twin <= -2: STREAMS with Sv = -twin views per stream
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
    // a key tile and the wave's 64 queries each sit inside one view <=> masks are wave-uniform
    const bool uniform_views = L > 0 && (L % KT) == 0 && (L % QW) == 0;
    const int qview_w = (L > 0) ? __builtin_amdgcn_readfirstlane(qw0 / L) : 0;
    int qview[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) qview[u] = (L > 0) ? qrow[u] / L : 0;

    // key tiles the workgroup / this wave must visit (exclusive upper bounds)
    int kmax = T, kmax_w = T;
    if (L > 0 && skip_masked) {              // in every mask mode a query sees no view index above its own
        const int last_q = min(q0 + QT, T) - 1;
        kmax = min(T, (last_q / L + 1) * L);
        const int last_qw = min(qw0 + QW, T) - 1;
        kmax_w = min(T, (max(last_qw, 0) / L + 1) * L);
    }
    if (qw0 >= T) kmax_w = 0;
    const int ntiles = (kmax + KT - 1) / KT;
    const int ntiles_w = __builtin_amdgcn_readfirstlane((kmax_w + KT - 1) / KT);

    // staging map: thread -> float4 column tid&15; K rows (tid>>4) + 16 i; V key pairs 2p, 2p+1 with p = (tid>>4) + 16 (i>>1)
    const int s_col4 = tid & 15;
    const int s_row0 = tid >> 4;
    f32x4 kreg[4], vreg[4];
    auto prefetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kt * KT + s_row0 + 16 * i;
            const int vkey = kt * KT + 2 * (s_row0 + 16 * (i >> 1)) + (i & 1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (in16) {
                if (key < T) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(k16 + (size_t)key * ldk + s_col4 * 4); a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
                if (vkey < T) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(v16 + (size_t)vkey * ldv + s_col4 * 4); c = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
            } else {
                if (key < T) a = *reinterpret_cast<const f32x4*>(kf + (size_t)key * ldk + s_col4 * 4);
                if (vkey < T) c = *reinterpret_cast<const f32x4*>(vf + (size_t)vkey * ldv + s_col4 * 4);
            }
            if (MODE == 1) {                                         // e4m3 has no infinity: clamp to its finite range
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[e] = __builtin_amdgcn_fmed3f(a[e], -448.f, 448.f); c[e] = __builtin_amdgcn_fmed3f(c[e], -448.f, 448.f); }
            }
            kreg[i] = a;
            vreg[i] = c;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned char* dst = Ks + (s_row0 + 16 * i) * F::K_LDB + s_col4 * 4 * F::EB;
            if constexpr (MODE == 0) {
                bf16x4 kk;
#pragma unroll
                for (int e = 0; e < 4; ++e) kk[e] = (__bf16)kreg[i][e];
                *reinterpret_cast<bf16x4*>(dst) = kk;
            } else {
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(kreg[i][0], kreg[i][1], 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(kreg[i][2], kreg[i][3], w, true);
                *reinterpret_cast<int*>(dst) = w;
            }
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int p2 = 2 * (s_row0 + 16 * ip);                   // even key of the pair
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned char* dst = Vt + (s_col4 * 4 + e) * F::VT_LDB + p2 * F::EB;
                if constexpr (MODE == 0) {
                    bf16x2 pr;
                    pr[0] = (__bf16)vreg[2 * ip][e];
                    pr[1] = (__bf16)vreg[2 * ip + 1][e];
                    *reinterpret_cast<bf16x2*>(dst) = pr;
                } else {
                    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(vreg[2 * ip][e], vreg[2 * ip + 1][e], 0, false);
                    *reinterpret_cast<unsigned short*>(dst) = (unsigned short)w;
                }
            }
        }
    };

    f32x16 ot[2][2];                                                 // [query tile][feature half]
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[u][d][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = scale * LOG2E;                                  // softmax weight = exp2(score * c2 - max * c2)
    const float pscale = MODE == 1 ? P_SCALE_FP8 : 1.0f;

    prefetch(0);
    for (int kt = 0; kt < ntiles; ++kt) {
#ifdef ATT_X_NOSTAGE
        if (kt == 0) { stage(); __syncthreads(); }
#else
        __syncthreads();                                             // previous tile fully consumed
        stage();
        __syncthreads();
#ifndef ATT_X_NOGLOBAL
        if (kt + 1 < ntiles) prefetch(kt + 1);
#endif
#endif
        if (kt >= ntiles_w) continue;                                // beyond this wave's last visible view
        const bool tile_vis = uniform_views && visible(qview_w, (kt * KT) / L);
        if (skip_masked && uniform_views && !tile_vis) continue;     // every key masked for all 64 queries: contributes exactly 0.0f

        // ---- S^T = K . Q^T: each K fragment (LDS) feeds both query tiles ------------------------------------------------
        f32x16 st[2][2];                                             // [query tile][key half]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[u][t2][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const frag a = *reinterpret_cast<const frag*>(Ks + (t2 * 32 + l31) * F::K_LDB + (ks * 16 + half * 8) * F::EB);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#ifdef ATT_X_NOMFMA
                    st[u][t2][ks] += (float)(*reinterpret_cast<const int*>(&a)) * (float)(*reinterpret_cast<const int*>(&qb[u][ks]));
#else
                    st[u][t2] = F::mfma(a, qb[u][ks], st[u][t2]);
#endif
                }
            }

        // ---- mask + online softmax (lane = one query of each tile; its 32 keys of this key tile) ------------------------
        const bool plain = (kt * KT + KT <= T) && (L == 0 || tile_vis);     // no per-element masking needed (wave-uniform)
        frag pb[2][2][2];                                            // [query tile][key half][k-step]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!plain) {
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kt * KT + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float s = st[u][t2][r];
                        if (L > 0 && !visible(qview[u], key / L)) s = -1e4f / scale;     // w*m - 1e4*(1-m), in units of the raw score
                        if (key >= T) s = -INFINITY;                                      // padding keys do not exist
                        st[u][t2][r] = s;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[u][t2][r]), st[u][t2][r + 1]);   // v_max3_f32
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx * scale);
            // (m_run - m_new) * log2e: a masked-only tile gives alpha == 1.0f EXACTLY, so skipping it is bit-identical to visiting it
            const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * LOG2E);       // 0 on the first tile (m_run = -inf)
            const float mc = m_new * LOG2E;
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    float pv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
#ifdef ATT_X_NOSOFTMAX
                        const float p = st[u][t2][ks2 * 8 + e];
#else
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t2][ks2 * 8 + e], c2, -mc));
#endif
                        psum += p;
                        pv[e] = MODE == 1 ? p * pscale : p;
                    }
                    pb[u][t2][ks2] = pack8<MODE>(pv);
                }
            l_run[u] = l_run[u] * alpha + psum;
            m_run[u] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {        // the maximum moved for some query of the wave: rescale
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[u][d][r] *= alpha;
            }
        }

        // ---- O^T += V^T . P^T: k-step (t2, ks2) = keys 32 t2 + 16 ks2 + 8 (e>>2) + 4 half + (e&3); each V^T fragment feeds both tiles
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vrow = Vt + (d * 32 + l31) * F::VT_LDB + (t2 * 32 + 16 * ks2 + 4 * half) * F::EB;
                    frag va;
                    if constexpr (MODE == 0) {
                        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow);
                        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
                    } else {
                        const unsigned lo = *reinterpret_cast<const unsigned*>(vrow);
                        const unsigned hi = *reinterpret_cast<const unsigned*>(vrow + 8);
                        va = (long)(((unsigned long long)hi << 32) | lo);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#ifdef ATT_X_NOMFMA
                        ot[u][d][t2 * 2 + ks2] += (float)(*reinterpret_cast<const int*>(&va)) * (float)(*reinterpret_cast<const int*>(&pb[u][t2][ks2]));
#else
                        ot[u][d] = F::mfma(va, pb[u][t2][ks2], ot[u][d]);
#endif
                    }
                }
    }

    // ---- normalise and store: lane = query, regs 4j..4j+3 = 4 consecutive features -------------
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float l_tot = (l_run[u] + __shfl_xor(l_run[u], 32, 64)) * pscale;
        if (!qvalid[u]) continue;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[u][d][4 * j + e] / l_tot;
                if (out16) {                                         // bf16 output for a bf16-MFMA consumer (ldo in elements)
                    bf16x4 o4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o4[e] = (__bf16)o[e];
                    __bf16* o16 = reinterpret_cast<__bf16*>(out) + b * (size_t)T * ldo + h * DH + (size_t)qrow[u] * ldo + 4 * half;
                    *reinterpret_cast<bf16x4*>(o16 + d * 32 + 8 * j) = o4;
                } else {
                    float* orow = out + b * (size_t)T * ldo + h * DH + (size_t)qrow[u] * ldo + 4 * half;
                    *reinterpret_cast<f32x4*>(orow + d * 32 + 8 * j) = o;
                }
            }
    }
}

template <int MODE>
int launch_lp(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L, int ldq, int ldk,
              int ldv, int ldo, float scale, int skip_masked, int twin_view, void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || T <= 0 || L < 0 || !(scale > 0.f)) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | ldo) & 3) return VF_ERR_BAD_ARG;
    if (in_bf16 && ((ldq | ldk | ldv) & 7)) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    hipLaunchKernelGGL(attn_lp_kernel<MODE>, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(q),
                       reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(v), reinterpret_cast<float*>(out), T, L, ldq, ldk, ldv,
                       ldo, scale, skip_masked, twin_view, out_bf16, in_bf16);
    return vf_last_status();
}

}  // namespace

int vf_attn_dma_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                       float scale, int twin_view, hipStream_t stream, float* lse_out = nullptr, float drop_rate = 0.f, uint32_t drop_seed = 0,
                       uint32_t drop_site = 0, uint32_t drop_plane0 = 0);        // attention_dma.hip: bf16 in / out, 64-token views, LDS-DMA ring

extern "C" {

int vf_attn_blockcausal_bf16_v2(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L,
                                int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view, void* stream) {
    if (skip_masked && in_bf16 && out_bf16 && q && k && v && out && B > 0 && H > 0 && T > 0 && scale > 0.f && ldq >= H * DH && ldk >= H * DH && ldv >= H * DH &&
        ldo >= H * DH) {
        // the DMA kernel always skips masked tiles (their weights are exactly 0.0f, so the result equals the dense form bit for bit); a caller
        // asking for skip_masked = 0 gets the dense register-staged kernel, so that A/B and parity runs measure what they ask for
        if (vf_selected(VF_SEL_ATTN_DMA)) {                          // (0: keep the register-staged kernel — A/B runs, the parity test)
            const int rc = vf_attn_dma_launch(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale, twin_view, (hipStream_t)stream);
            if (rc != VF_ERR_UNSUPPORTED) return rc;
        }
    }
    return launch_lp<0>(q, k, v, in_bf16, out, out_bf16, B, H, T, L, ldq, ldk, ldv, ldo, scale, skip_masked, twin_view, stream);
}

int vf_attn_blockcausal_fp8(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L,
                            int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view, void* stream) {
    return launch_lp<1>(q, k, v, in_bf16, out, out_bf16, B, H, T, L, ldq, ldk, ldv, ldo, scale, skip_masked, twin_view, stream);
}

}  // extern "C"
