// Block-causal, un-scaled, "-1e4 masked" attention of the MIGT transformer on exact-f32 MFMA (gfx950).
//
// Replaces compute_causal_block_attention + compute_attention
// (viewformer/models/branching_attention.py:41-61,5-18) reached through the single-stream branch of
// compute_causal_block_multiend_attention (:82-92).  Semantics kept bit-for-bit in structure:
//   w = q.k^T (NO 1/sqrt(d));  w = w*m - 1e4*(1-m) with m[i][j] = view(i) >= view(j), view = token / L;
//   softmax over keys;  out = w.v.   The [T][T] score matrix is never materialised.
//
// One 256-thread workgroup = 128 query rows of one (batch, head) sharing each 64-key K/V tile through
// LDS; with L = 64 a wave's 32 queries lie inside one view, so a key tile is either fully visible or
// fully masked for the wave and masked tiles are skipped wave-uniformly.  Each wave owns 32 queries and
// computes the TRANSPOSED score tile S^T[key][query] = K.Q^T (MFMA A = K tile from LDS, B = Q held
// in 32 VGPRs), so that a lane owns one query column: the row max / row sum are in-lane over 32
// values plus ONE cross-half shuffle, and the probabilities sit exactly in the B-operand layout of
// the second MFMA  O^T[d][query] += V^T[d][key] . P^T[key][query]  (32x32x2: lanes 0-31 feed key a,
// lanes 32-63 key a+4 — the C layout of S^T).  No LDS round trip or permute for P.
// Per 64-key tile and wave: 64 + 64 MFMAs (8192 MFMA cycles) against ~200 VALU ops of softmax.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int DH = 64;
constexpr int QT = 128;     // queries per workgroup
constexpr int KT = 64;      // keys per tile
constexpr int K_LD = 68;    // LDS row stride of the K tile (conflict-free ds_read_b128)
constexpr int VT_LD = 68;    // V is parked TRANSPOSED ([feature][key]) so a lane's 4 consecutive keys are one ds_read_b128

__global__ __launch_bounds__(256, 2) void attn_blockcausal_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ v, float* __restrict__ out,
                                                                  int T, int L, int ldq, int ldk, int ldv, int ldo,
                                                                  float scale, int skip_masked, int twin, float* __restrict__ lse,
                                                                  uint32_t drop_thresh, float drop_scale, uint32_t drop_seed,
                                                                  uint32_t drop_site, uint32_t drop_plane0) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * K_LD];
    __shared__ __attribute__((aligned(16))) float Vt[DH * VT_LD];

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

    const float* qb = q + b * (size_t)T * ldq + h * DH;
    const float* kb = k + b * (size_t)T * ldk + h * DH;
    const float* vb = v + b * (size_t)T * ldv + h * DH;
    float* ob = out + b * (size_t)T * ldo + h * DH;

    // ---- Q fragment: qreg[g*4+e] = Q[qrow][8g + 4*half + e] ---------------------------------
    const int qrow = q0 + wave * 32 + l31;
    const bool qvalid = qrow < T;
    const uint32_t drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * gridDim.x + h));       // mask plane = (global scene, head)
    float qreg[32];
    {
        const float* src = qb + (size_t)(qvalid ? qrow : 0) * ldq + 4 * half;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) qreg[g * 4 + e] = t[e];
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

    // staging map: thread -> rows (tid>>4) + 16*i, float4 column tid&15
    const int s_col4 = tid & 15;
    const int s_row0 = tid >> 4;
    f32x4 kreg[4], vreg[4];
    auto prefetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kt * KT + s_row0 + 16 * i;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * ldk + s_col4 * 4);
                c = *reinterpret_cast<const f32x4*>(vb + (size_t)key * ldv + s_col4 * 4);
            }
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
            *reinterpret_cast<f32x4*>(Ks + (s_row0 + 16 * i) * K_LD + s_col4 * 4) = kreg[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) Vt[(s_col4 * 4 + e) * VT_LD + s_row0 + 16 * i] = vreg[i][e];
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
        for (int g = 0; g < 8; ++g) {
            f32x4 a[2];
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
                a[t2] = *reinterpret_cast<const f32x4*>(Ks + (t2 * 32 + l31) * K_LD + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2)
                    st[t2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t2][e], qreg[g * 4 + e], st[t2], 0, 0, 0);
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
                float p = __builtin_amdgcn_exp2f((st[t2][r] - m_new) * LOG2E);
                psum += p;                                   // the softmax normaliser is over the undropped weights
                if (drop_thresh) {                           // attn_dropout (branching_attention.py:15-17): applied to softmax(w)
                    // mask group of (query, keys 4 g .. 4 g + 3): registers r & 3 = 0..3 of a lane are one group (vf_common.h)
                    const int key = kt * KT + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const uint32_t w = vf_dropout_word(drop_key, (uint32_t)qrow * (uint32_t)((T + 3) >> 2) + (uint32_t)(key >> 2));
                    p = vf_dropout_keep(w, key & 3, drop_thresh) ? p * drop_scale : 0.f;
                }
                st[t2][r] = p;
            }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;

        // ---- O^T += V^T . P^T -------------------------------------------------------------
        // lane (feature d = l31 [+32], half): keys t2*32 + 8j + 4*half + {0..3} are contiguous in Vt's row -> one b128
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 vv[2];
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    vv[d] = *reinterpret_cast<const f32x4*>(Vt + (d * 32 + l31) * VT_LD + t2 * 32 + 8 * j + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        ot[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[d][e], st[t2][4 * j + e], ot[d], 0, 0, 0);
            }
    }

    // ---- normalise and store: lane = query, regs 4j..4j+3 = 4 consecutive features -------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    // log-sum-exp of the (scaled, masked) scores of this query: what the backward pass needs to re-materialise P
    if (lse && qvalid && half == 0) lse[((size_t)b * gridDim.x + h) * T + qrow] = m_run + logf(l_tot);
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

int vf_attn_blockcausal_f32(const float* q, const float* k, const float* v, float* out, int B, int H, int T, int L,
                            int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view,
                            void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || T <= 0 || L < 0) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | ldo) & 3) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    hipLaunchKernelGGL(attn_blockcausal_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, out, T, L, ldq, ldk,
                       ldv, ldo, scale, skip_masked, twin_view, (float*)nullptr, 0u, 1.0f, 0u, 0u, 0u);
    return vf_last_status();
}

int vf_attn_blockcausal_lse_f32(const float* q, const float* k, const float* v, float* out, float* lse, int B, int H, int T,
                                int L, int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view,
                                float drop_rate, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0, void* stream) {
    if (!q || !k || !v || !out || !lse || B <= 0 || H <= 0 || T <= 0 || L < 0) return VF_ERR_BAD_ARG;
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return VF_ERR_BAD_ARG;
    const uint32_t thresh = vf_dropout_thresh(drop_rate);
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | ldo) & 3) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    hipLaunchKernelGGL(attn_blockcausal_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, out, T, L, ldq, ldk,
                       ldv, ldo, scale, skip_masked, twin_view, lse, thresh, 1.0f / (1.0f - drop_rate), drop_seed, drop_site, drop_plane0);
    return vf_last_status();
}

}  // extern "C"
