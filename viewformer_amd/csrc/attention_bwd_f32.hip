// Backward of the block-causal / streams attention (attention_f32.hip) for the training step, exact-f32 MFMA, gfx950.
// Flash-style: the [T][T] probabilities are never materialised; they are re-computed tile by tile from Q, K and the
// per-query log-sum-exp the forward pass saved, and fully masked tiles are skipped exactly like in the forward kernel
// (with the training graph's 3 streams only ~18 % of the 64x64 tiles are visible).  Deterministic: no atomics —
//   vf_attn_bwd_dq_f32  (query-major, like the forward):  dQ = dS . K
//   vf_attn_bwd_dkv_f32 (key-major):                      dK = dS^T . Q,  dV = P^T . dO
// with  P = exp(S - lse),  dP = dO . V^T,  dS = P * (dP - D),  D = rowsum(dO * O)  (vf_attn_bwd_prep_f32).
// Reference: the autograd of compute_attention / compute_causal_block_multiend_attention
// (viewformer/models/branching_attention.py:5-18,82-126) inside MIGT.train_step (migt.py:464-505).
//
// Both kernels reuse the forward's transposed-score trick: the tile product is computed with the "owner" dimension
// (queries in dq, keys in dkv) in the MFMA column = lane, so that P / dS sit in registers in exactly the B-operand layout of
// the following accumulation MFMA (32x32x2: lanes 0-31 feed row a, lanes 32-63 row a+4), whose A operand is a TRANSPOSED
// tile read from LDS.  No LDS round trip or permute for P or dS.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

constexpr int DH = 64;
constexpr int OT = 128;     // owner rows (queries in dq, keys in dkv) per workgroup: 4 waves x 32
constexpr int TT = 64;      // rows of the streamed tile
constexpr int LD = 68;      // LDS row stride in floats (conflict-free ds_read_b128)
constexpr float LOG2E = 1.4426950408889634f;

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

// D[b][h][t] = sum_d dO[t][h*64+d] * O[t][h*64+d]; 16 lanes per (row, head)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            float* __restrict__ D, int B, int H, int T, int lddo, int ldo) {
    const long long i = (blockIdx.x * 256ll + threadIdx.x) >> 4;          // (b, t, h) flat
    const int c4 = threadIdx.x & 15;
    const long long total = (long long)B * T * H;
    float s = 0.f;
    long long bt = 0;
    int h = 0;
    if (i < total) {
        h = (int)(i % H);
        bt = i / H;
        const f32x4 a = *reinterpret_cast<const f32x4*>(dout + bt * lddo + h * DH + c4 * 4);
        const f32x4 o = *reinterpret_cast<const f32x4*>(out + bt * ldo + h * DH + c4 * 4);
        s = a[0] * o[0] + a[1] * o[1] + a[2] * o[2] + a[3] * o[3];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (i < total && c4 == 0) {
        const long long b = bt / T, t = bt - b * T;
        D[(b * H + h) * T + t] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------- dQ
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ dout,
                                                             const float* __restrict__ lse, const float* __restrict__ Dv,
                                                             float* __restrict__ dq, int T, int L, int ldq, int ldk, int ldv,
                                                             int lddo, int lddq, float scale, int twin, uint32_t drop_thresh,
                                                             float drop_scale, uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0) {
    __shared__ __attribute__((aligned(16))) float Ks[TT * LD];
    __shared__ __attribute__((aligned(16))) float Vs[TT * LD];
    __shared__ __attribute__((aligned(16))) float Kt[DH * LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x, H = gridDim.x;      // grid (H, B, tiles): the tiles of one (scene, head) share an XCD's L2 (attention_f32.hip)
    const size_t b = blockIdx.y;
    const int q0 = blockIdx.z * OT;
    const float* qb = q + b * (size_t)T * ldq + h * DH;
    const float* kb = k + b * (size_t)T * ldk + h * DH;
    const float* vb = v + b * (size_t)T * ldv + h * DH;
    const float* dob = dout + b * (size_t)T * lddo + h * DH;
    float* dqb = dq + b * (size_t)T * lddq + h * DH;

    const int qrow = q0 + wave * 32 + l31;
    const bool qvalid = qrow < T;
    float qreg[32], doreg[32];
    {
        const float* s0 = qb + (size_t)(qvalid ? qrow : 0) * ldq + 4 * half;
        const float* s1 = dob + (size_t)(qvalid ? qrow : 0) * lddo + 4 * half;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(s0 + 8 * g);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(s1 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { qreg[g * 4 + e] = t0[e]; doreg[g * 4 + e] = t1[e]; }
        }
    }
    const size_t stat = ((size_t)b * H + h) * T + (qvalid ? qrow : 0);
    const float lse_q = lse[stat], D_q = Dv[stat];
    const uint32_t drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * H + h));               // mask plane = (global scene, head)
    const Vis visible = make_vis(twin);
    const int qview = (L > 0) ? qrow / L : 0;
    const bool uniform_views = L > 0 && (L % TT) == 0;
    const int qview_w = (L > 0) ? __builtin_amdgcn_readfirstlane((q0 + wave * 32) / L) : 0;
    const int ntiles = (T + TT - 1) / TT;

    const int s_col4 = tid & 15, s_row0 = tid >> 4;
    f32x16 ot[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;

    for (int kt = 0; kt < ntiles; ++kt) {
        // workgroup-uniform skip: no wave of this workgroup sees the tile
        bool any = !uniform_views;
        if (uniform_views) {
            const int kvw = (kt * TT) / L;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (q0 + w * 32 < T) any |= visible((q0 + w * 32) / L, kvw);
        }
        if (!any) continue;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = kt * TT + s_row0 + 16 * i;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (key < T) {
                a = *reinterpret_cast<const f32x4*>(kb + (size_t)key * ldk + s_col4 * 4);
                c = *reinterpret_cast<const f32x4*>(vb + (size_t)key * ldv + s_col4 * 4);
            }
            *reinterpret_cast<f32x4*>(Ks + (s_row0 + 16 * i) * LD + s_col4 * 4) = a;
            *reinterpret_cast<f32x4*>(Vs + (s_row0 + 16 * i) * LD + s_col4 * 4) = c;
#pragma unroll
            for (int e = 0; e < 4; ++e) Kt[(s_col4 * 4 + e) * LD + s_row0 + 16 * i] = a[e];
        }
        __syncthreads();
        if (uniform_views && !visible(qview_w, (kt * TT) / L)) continue;      // wave-uniform

        const bool plain = (kt * TT + TT <= T) && (L == 0 || uniform_views);
        // the two 32-key halves of the tile one after the other: half the live score registers (no spills)
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Ks + (t2 * 32 + l31) * LD + 8 * g + 4 * half);
                const f32x4 c = *reinterpret_cast<const f32x4*>(Vs + (t2 * 32 + l31) * LD + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qreg[g * 4 + e], st, 0, 0, 0);      // S^T = K.Q^T
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c[e], doreg[g * 4 + e], dp, 0, 0, 0);    // dP^T = V.dO^T
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sc = st[r] * scale;
                if (!plain) {
                    const int key = kt * TT + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (L > 0 && !visible(qview, key / L)) sc = -1e4f;
                    if (key >= T) sc = -INFINITY;
                }
                const float p = __builtin_amdgcn_exp2f((sc - lse_q) * LOG2E);
                float dpr = dp[r];
                if (drop_thresh) {                                         // d(dropped P)/dP = mask * scale
                    const int key = kt * TT + t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const uint32_t w = vf_dropout_word(drop_key, (uint32_t)qrow * (uint32_t)((T + 3) >> 2) + (uint32_t)(key >> 2));
                    dpr = vf_dropout_keep(w, key & 3, drop_thresh) ? dpr * drop_scale : 0.f;
                }
                st[r] = p * (dpr - D_q) * scale;                           // dS^T (d/dS of the scaled score)
            }
            // dQ^T[d][query] += K^T[d][key] . dS^T[key][query]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 kk[2];
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    kk[d] = *reinterpret_cast<const f32x4*>(Kt + (d * 32 + l31) * LD + t2 * 32 + 8 * j + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < 2; ++d)
                        ot[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[d][e], st[4 * j + e], ot[d], 0, 0, 0);
            }
        }
    }
    if (qvalid) {
        float* orow = dqb + (size_t)qrow * lddq + 4 * half;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = ot[d][4 * j + e];
                *reinterpret_cast<f32x4*>(orow + d * 32 + 8 * j) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ dout,
                                                              const float* __restrict__ lse, const float* __restrict__ Dv,
                                                              float* __restrict__ dk, float* __restrict__ dv, int T, int L,
                                                              int ldq, int ldk, int ldv, int lddo, int lddk, int lddv,
                                                              float scale, int twin, uint32_t drop_thresh, float drop_scale,
                                                              uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0) {
    extern __shared__ __attribute__((aligned(16))) float smem_a[];
    float* Qs = smem_a;                 // [TT][LD]   queries of the tile, row-major
    float* Os = Qs + TT * LD;           // [TT][LD]   dO rows
    float* Qt = Os + TT * LD;           // [DH][LD]   Q transposed ([feature][query])
    float* Ot = Qt + DH * LD;           // [DH][LD]   dO transposed
    float* Ls = Ot + DH * LD;           // [TT] lse, [TT] D
    float* Ds = Ls + TT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x, H = gridDim.x;      // grid (H, B, tiles)
    const size_t b = blockIdx.y;
    const int k0 = blockIdx.z * OT;
    const float* qb = q + b * (size_t)T * ldq + h * DH;
    const float* kb = k + b * (size_t)T * ldk + h * DH;
    const float* vb = v + b * (size_t)T * ldv + h * DH;
    const float* dob = dout + b * (size_t)T * lddo + h * DH;
    const float* lseb = lse + ((size_t)b * H + h) * T;
    const float* Db = Dv + ((size_t)b * H + h) * T;

    const int krow = k0 + wave * 32 + l31;              // this lane's key
    const bool kvalid = krow < T;
    const uint32_t drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * H + h));               // mask plane = (global scene, head)
    float kreg[32], vreg[32];
    {
        const float* s0 = kb + (size_t)(kvalid ? krow : 0) * ldk + 4 * half;
        const float* s1 = vb + (size_t)(kvalid ? krow : 0) * ldv + 4 * half;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(s0 + 8 * g);
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(s1 + 8 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) { kreg[g * 4 + e] = t0[e]; vreg[g * 4 + e] = t1[e]; }
        }
    }
    const Vis visible = make_vis(twin);
    const int kview = (L > 0) ? krow / L : 0;
    const bool uniform_views = L > 0 && (L % TT) == 0;
    const int kview_w = (L > 0) ? __builtin_amdgcn_readfirstlane((k0 + wave * 32) / L) : 0;
    const int ntiles = (T + TT - 1) / TT;
    const int s_col4 = tid & 15, s_row0 = tid >> 4;

    f32x16 okt[2], ovt[2];
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { okt[d][r] = 0.f; ovt[d][r] = 0.f; }

    for (int qt = 0; qt < ntiles; ++qt) {
        bool any = !uniform_views;
        if (uniform_views) {
            const int qvw = (qt * TT) / L;
#pragma unroll
            for (int w = 0; w < 4; ++w)
                if (k0 + w * 32 < T) any |= visible(qvw, (k0 + w * 32) / L);
        }
        if (!any) continue;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = qt * TT + s_row0 + 16 * i;
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {0.f, 0.f, 0.f, 0.f};
            if (row < T) {
                a = *reinterpret_cast<const f32x4*>(qb + (size_t)row * ldq + s_col4 * 4);
                c = *reinterpret_cast<const f32x4*>(dob + (size_t)row * lddo + s_col4 * 4);
            }
            *reinterpret_cast<f32x4*>(Qs + (s_row0 + 16 * i) * LD + s_col4 * 4) = a;
            *reinterpret_cast<f32x4*>(Os + (s_row0 + 16 * i) * LD + s_col4 * 4) = c;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                Qt[(s_col4 * 4 + e) * LD + s_row0 + 16 * i] = a[e];
                Ot[(s_col4 * 4 + e) * LD + s_row0 + 16 * i] = c[e];
            }
        }
        if (tid < TT) {
            const int row = qt * TT + tid;
            Ls[tid] = row < T ? lseb[row] : INFINITY;       // padding queries: p = exp2(-inf) = 0
            Ds[tid] = row < T ? Db[row] : 0.f;
        }
        __syncthreads();
        if (uniform_views && !visible((qt * TT) / L, kview_w)) continue;      // wave-uniform

        const bool plain = kvalid && (L == 0 || uniform_views);      // (kvalid is per lane: folded into the element test)
        // the two 32-query halves of the tile one after the other (register budget: a real loop, not unrolled)
#pragma unroll 1
        for (int t2 = 0; t2 < 2; ++t2) {
            // S[query][key] = Q.K^T and dP[query][key] = dO.V^T (lane = key column)
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(Qs + (t2 * 32 + l31) * LD + 8 * g + 4 * half);
                const f32x4 c = *reinterpret_cast<const f32x4*>(Os + (t2 * 32 + l31) * LD + 8 * g + 4 * half);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    st = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], kreg[g * 4 + e], st, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(c[e], vreg[g * 4 + e], dp, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = t2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;     // query row inside the tile
                float sc = st[r] * scale;
                if (!plain) {
                    const int query = qt * TT + ql;
                    if (L > 0 && !visible(query / L, kview)) sc = -1e4f;
                    if (!kvalid) sc = -INFINITY;
                }
                const float p = __builtin_amdgcn_exp2f((sc - Ls[ql]) * LOG2E);
                float pd = p, dpr = dp[r];
                if (drop_thresh) {
                    const uint32_t w = vf_dropout_word(drop_key, (uint32_t)(qt * TT + ql) * (uint32_t)((T + 3) >> 2) + (uint32_t)(krow >> 2));
                    const bool keep = vf_dropout_keep(w, krow & 3, drop_thresh);
                    pd = keep ? p * drop_scale : 0.f;                      // the forward multiplied V by the dropped P
                    dpr = keep ? dpr * drop_scale : 0.f;
                }
                st[r] = pd;
                dp[r] = p * (dpr - Ds[ql]) * scale;                        // dS
            }
            // dV^T[d][key] += dO^T[d][query] . P[query][key];  dK^T[d][key] += Q^T[d][query] . dS[query][key]
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 oo[2], qq[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    oo[d] = *reinterpret_cast<const f32x4*>(Ot + (d * 32 + l31) * LD + t2 * 32 + 8 * j + 4 * half);
                    qq[d] = *reinterpret_cast<const f32x4*>(Qt + (d * 32 + l31) * LD + t2 * 32 + 8 * j + 4 * half);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        ovt[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(oo[d][e], st[4 * j + e], ovt[d], 0, 0, 0);
                        okt[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qq[d][e], dp[4 * j + e], okt[d], 0, 0, 0);
                    }
            }
        }
    }
    if (kvalid) {
        float* krow_o = dk + b * (size_t)T * lddk + h * DH + (size_t)krow * lddk + 4 * half;
        float* vrow_o = dv + b * (size_t)T * lddv + h * DH + (size_t)krow * lddv + 4 * half;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o1[e] = okt[d][4 * j + e]; o2[e] = ovt[d][4 * j + e]; }
                *reinterpret_cast<f32x4*>(krow_o + d * 32 + 8 * j) = o1;
                *reinterpret_cast<f32x4*>(vrow_o + d * 32 + 8 * j) = o2;
            }
    }
}

}  // namespace

extern "C" {

int vf_attn_bwd_prep_f32(const float* dout, const float* out, float* D, int B, int H, int T, int lddo, int ldo, void* stream) {
    if (!dout || !out || !D || B <= 0 || H <= 0 || T <= 0 || lddo < H * DH || ldo < H * DH || ((lddo | ldo) & 3)) return VF_ERR_BAD_ARG;
    const long long total = (long long)B * T * H * 16;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, out,
                       D, B, H, T, lddo, ldo);
    return vf_last_status();
}

int vf_attn_bwd_f32(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* D,
                    float* dq, float* dk, float* dv, int B, int H, int T, int L, int ldq, int ldk, int ldv, int lddo, int lddq,
                    int lddk, int lddv, float scale, int twin_view, float drop_rate, uint32_t drop_seed,
                    uint32_t drop_site, uint32_t drop_plane0, void* stream) {
    if (!q || !k || !v || !dout || !lse || !D || !dq || !dk || !dv || B <= 0 || H <= 0 || T <= 0 || L < 0) return VF_ERR_BAD_ARG;
    const int w = H * DH;
    if (ldq < w || ldk < w || ldv < w || lddo < w || lddq < w || lddk < w || lddv < w) return VF_ERR_BAD_ARG;
    if ((ldq | ldk | ldv | lddo | lddq | lddk | lddv) & 3) return VF_ERR_BAD_ARG;
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return VF_ERR_BAD_ARG;
    const uint32_t thresh = vf_dropout_thresh(drop_rate);
    const float dscale = 1.0f / (1.0f - drop_rate);
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + OT - 1) / OT));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), 0, s, q, k, v, dout, lse, D, dq, T, L, ldq, ldk, ldv, lddo, lddq,
                       scale, twin_view, thresh, dscale, drop_seed, drop_site, drop_plane0);
    int st = vf_last_status();
    if (st) return st;
    const size_t smem = (size_t)(2 * TT * LD + 2 * DH * LD + 2 * TT) * sizeof(float);
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), smem, s, q, k, v, dout, lse, D, dk, dv, T, L, ldq, ldk, ldv, lddo,
                       lddk, lddv, scale, twin_view, thresh, dscale, drop_seed, drop_site, drop_plane0);
    return vf_last_status();
}

}  // extern "C"
