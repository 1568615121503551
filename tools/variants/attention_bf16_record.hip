// RECORD, not part of libvf_hip.so since round 4: the FIRST bf16 attention kernel (round 1/2; 297 us at the bench shape against 253 us for attention_lp.hip and 117-131 us for attention_dma.hip) - was reachable through vf_attn_blockcausal_bf16 / VF_ATTN_BF16_V1=1.
// Kept as the source the measurements in DESIGN.md refer to; builds against the round-3 C-ABI (git show 7e8c4c4:include/vf_hip.h).
// bf16-MFMA sibling of attention_f32.hip for the reduced-precision (tolerance-bounded) transformer arm, gfx950.
// Same semantics (un-scaled q.k^T, "w*m - 1e4*(1-m)" block mask incl. twin views and streams, softmax, .v), same
// workgroup shape (128 queries sharing 64-key K/V tiles through LDS, masked tiles skipped wave-uniformly), same
// transposed-score trick — but both contractions run on v_mfma_f32_32x32x16_bf16: Q, K, V and the probabilities are
// rounded to bf16 (RNE), products are exact, sums and the whole softmax stay fp32.  Per 64-key tile a wave issues 16 MFMAs
// (512 matrix-pipe cycles) instead of 128 f32 MFMAs (8192 cycles); the kernel becomes softmax(VALU)-bound.
//   S^T[key][query] = K . Q^T : A = K tile (LDS, [key][dh] bf16, 144-byte rows), B = Q (16 VGPRs per lane)
//   O^T[d][query]  += V^T . P^T: the C layout of S^T gives a lane the keys {8a + 4*half + b}; feeding P straight back as
//     the B operand fixes the k-index -> key map to kappa(half, e) = 16*ks + 8*(e>>2) + 4*half + (e&3), and the A operand
//     (V^T from LDS, [d][key] bf16, 136-byte rows) is read with the same map: two 8-byte reads per fragment.  No LDS
//     round trip or permute for P.
#include "../../viewformer_amd/csrc/vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int DH = 64;
constexpr int QT = 128;     // queries per workgroup
constexpr int KT = 64;      // keys per tile
constexpr int K_LDB = 144;  // bytes per K row in LDS (128 B of bf16 + 16 B pad: conflict-free ds_read_b128)
constexpr int VT_LDB = 136; // bytes per V^T row ([feature][key] bf16): 34 banks -> 32 rows hit 32 distinct bank pairs

__global__ __launch_bounds__(256, 2) void attn_blockcausal_bf16_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                  const float* __restrict__ v, float* __restrict__ out,
                                                                  int T, int L, int ldq, int ldk, int ldv, int ldo,
                                                                  float scale, int skip_masked, int twin, int out16, int in16) {
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
    const int q0 = blockIdx.z * QT;

    // in16: q, k, v are bf16 in HBM (ld* in elements) — the fused c_attn output written as bf16 by its GEMM; loaded and widened
    // exactly, so the rest of the kernel (which rounds fp32 inputs to bf16) sees the same values
    const float* qb_ptr = q + b * (size_t)T * ldq + h * DH;
    const float* kb = k + b * (size_t)T * ldk + h * DH;
    const float* vb = v + b * (size_t)T * ldv + h * DH;
    const __bf16* q16 = reinterpret_cast<const __bf16*>(q) + b * (size_t)T * ldq + h * DH;
    const __bf16* k16 = reinterpret_cast<const __bf16*>(k) + b * (size_t)T * ldk + h * DH;
    const __bf16* v16 = reinterpret_cast<const __bf16*>(v) + b * (size_t)T * ldv + h * DH;
    float* ob = out + b * (size_t)T * ldo + h * DH;

    // ---- Q fragment (B operand): qb[ks][e] = bf16(Q[qrow][16 ks + 8 half + e]) ------------------------
    const int qrow = q0 + wave * 32 + l31;
    const bool qvalid = qrow < T;
    bf16x8 qb[4];
    {
        const float* src = qb_ptr + (size_t)(qvalid ? qrow : 0) * ldq + 8 * half;
        const __bf16* src16 = q16 + (size_t)(qvalid ? qrow : 0) * ldq + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (in16) { qb[ks] = *reinterpret_cast<const bf16x8*>(src16 + 16 * ks); continue; }
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + 16 * ks);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(src + 16 * ks + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qb[ks][e] = (__bf16)t0[e]; qb[ks][4 + e] = (__bf16)t1[e]; }
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
            if (in16) {
                if (key < T) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(k16 + (size_t)key * ldk + s_col4 * 4); a = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
                if (vkey < T) { const bf16x4 t = *reinterpret_cast<const bf16x4*>(v16 + (size_t)vkey * ldv + s_col4 * 4); c = f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]}; }
            } else {
                if (key < T) a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * ldk + s_col4 * 4);
                if (vkey < T) c = *reinterpret_cast<const f32x4*>(vb + (size_t)vkey * ldv + s_col4 * 4);
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
            bf16x4 kk;
#pragma unroll
            for (int e = 0; e < 4; ++e) kk[e] = (__bf16)kreg[i][e];
            *reinterpret_cast<bf16x4*>(Ks + (s_row0 + 16 * i) * K_LDB + s_col4 * 8) = kk;
        }
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int p2 = 2 * (s_row0 + 16 * ip);                   // even key of the pair
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bf16x2 pr;
                pr[0] = (__bf16)vreg[2 * ip][e];
                pr[1] = (__bf16)vreg[2 * ip + 1][e];
                *reinterpret_cast<bf16x2*>(Vt + (s_col4 * 4 + e) * VT_LDB + p2 * 2) = pr;
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + (t2 * 32 + l31) * K_LDB + ks * 32 + half * 16);
                st[t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ks], st[t2], 0, 0, 0);
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
                bf16x8 pb;
#pragma unroll
                for (int e = 0; e < 8; ++e) pb[e] = (__bf16)st[t2][ks2 * 8 + e];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vrow = Vt + (d * 32 + l31) * VT_LDB + (t2 * 32 + 16 * ks2 + 4 * half) * 2;
                    const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow);
                    const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + 16);
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
                    ot[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb, ot[d], 0, 0, 0);
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
                if (out16) {                           // bf16 output for a bf16-MFMA consumer (ldo in elements)
                    bf16x4 ob;
#pragma unroll
                    for (int e = 0; e < 4; ++e) ob[e] = (__bf16)o[e];
                    __bf16* o16 = reinterpret_cast<__bf16*>(out) + b * (size_t)T * ldo + h * DH + (size_t)qrow * ldo + 4 * half;
                    *reinterpret_cast<bf16x4*>(o16 + d * 32 + 8 * j) = ob;
                } else
                *reinterpret_cast<f32x4*>(orow + d * 32 + 8 * j) = o;
            }
    }
}

}  // namespace

extern "C" {

int vf_attn_blockcausal_bf16(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L,
                            int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view,
                            void* stream) {
    if (!q || !k || !v || !out || B <= 0 || H <= 0 || T <= 0 || L < 0) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | ldo) & 3) return VF_ERR_BAD_ARG;
    if (in_bf16 && ((ldq | ldk | ldv) & 7)) return VF_ERR_BAD_ARG;
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    hipLaunchKernelGGL(attn_blockcausal_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float*>(q),
                       reinterpret_cast<const float*>(k), reinterpret_cast<const float*>(v), reinterpret_cast<float*>(out), T, L, ldq,
                       ldk, ldv, ldo, scale, skip_masked, twin_view, out_bf16, in_bf16);
    return vf_last_status();
}

}  // extern "C"
