// Single-head spatial self-attention of the VQGAN AttnBlock, fused and exact fp32, gfx950.
//
// Replaces AttnBlock.forward's core (viewformer/models/vqgan_th.py:124-141): w = q^T k * C^-0.5, softmax over the keys, h = v w^T —
// per image, HW tokens of C channels, (HW, C) in {(256, 256), (64, 512)} for the 128-px model — given the fused q|k|v projection
// [n*HW][3C] (q at column 0, k at C, v at 2C).  The [HW][HW] score matrix stays on chip: the first version ran
// pack(K) -> batched igemm -> softmax -> pack(V) -> batched igemm with the scores and two re-packed operands through HBM (5 launches,
// 4 % of the inference step).
//
// One workgroup = 64 queries of one image against ALL its keys (HW <= 256 keys = at most 8 key tiles of 32, spread over the 4 waves):
//   1. S^T = K . Q^T on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain).  Transposed, so that a query is a lane and its
//      keys are that lane's accumulator registers.  Both operands come straight from the qkv rows as float4 along the channel axis.
//   2. softmax over ALL keys at once (no online rescaling): per-wave partial max / sum in registers, combined across the 4 waves through
//      LDS; precise expf, then the normalised probabilities P^T[key][query] are parked in LDS (<= 64 KiB).
//   3. O^T = V^T . P^T: every wave takes C/4 output channels; V rows from global (128-byte rows per half-wave), P^T from LDS; the MFMA's
//      two k slots are the keys kappa and kappa + 4 of the accumulator layout, so P needs no permutation.
// Arithmetic: the same as the batched f32 igemm path it replaces up to summation order (scores: k ascending inside 8-channel groups;
// outputs: keys in accumulator-row order), all fp32.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

template <int HW, int C>
__global__ __launch_bounds__(256, 2) void attn_spatial_kernel(const float* __restrict__ qkv, float* __restrict__ out, long long ld,
                                                              long long ldo, float scale) {
    constexpr int NKT = HW / 32;                  // key tiles
    constexpr int TPW = NKT * 2 / 4;              // (key tile, query tile) pairs per wave: 4 (HW = 256) or 1 (HW = 64)
    constexpr int CT = C / 4 / 32;                // output-channel tiles per wave
    constexpr int P_LD = 64;                      // P^T row: 64 queries
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Pt = smem_f;                           // [HW][P_LD]
    float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(smem_f + HW * P_LD);      // [max | sum][wave][query]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long img = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const float* __restrict__ base = qkv + img * HW * ld;

    // ---- 1. scores: acc[i] = S^T tile (32 keys x 32 queries) for this wave's pair i = (kt, u)
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NK = TPW >= 2 ? TPW / 2 : 1;    // distinct key tiles of the wave
    constexpr int NU = TPW >= 2 ? 2 : 1;          // distinct query tiles of the wave
    const int kt0 = TPW >= 2 ? wave * NK : (wave >> 1);
    const int u0 = TPW >= 2 ? 0 : (wave & 1);
    const float* krow[NK];
    const float* qrow[NU];
#pragma unroll
    for (int a = 0; a < NK; ++a) krow[a] = base + (size_t)((kt0 + a) * 32 + l31) * ld + C + 4 * half;
#pragma unroll
    for (int b = 0; b < NU; ++b) qrow[b] = base + (size_t)(q0 + (u0 + b) * 32 + l31) * ld + 4 * half;
#pragma unroll 4
    for (int g = 0; g < C / 8; ++g) {
        f32x4 ka[NK], qb[NU];
#pragma unroll
        for (int a = 0; a < NK; ++a) ka[a] = *reinterpret_cast<const f32x4*>(krow[a] + 8 * g);
#pragma unroll
        for (int b = 0; b < NU; ++b) qb[b] = *reinterpret_cast<const f32x4*>(qrow[b] + 8 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < NK; ++a)
#pragma unroll
                for (int b = 0; b < NU; ++b)
                    acc[a * NU + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(ka[a][e], qb[b][e], acc[a * NU + b], 0, 0, 0);
    }

    // ---- 2. softmax over all keys.  Lane (l31, half) of pair (a, b): query (u0 + b) * 32 + l31, keys (kt0 + a) * 32 + rowmap(r, half)
    // (statistics are kept per query tile b of THIS wave; a wave that owns no key tile of query tile u contributes -inf / 0 for it)
    float mxl[NU];
#pragma unroll
    for (int b = 0; b < NU; ++b) mxl[b] = -INFINITY;
#pragma unroll
    for (int a = 0; a < NK; ++a)
#pragma unroll
        for (int b = 0; b < NU; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = acc[a * NU + b][r] * scale;
                acc[a * NU + b][r] = s;
                mxl[b] = fmaxf(mxl[b], s);
            }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float v = -INFINITY;
#pragma unroll
        for (int b = 0; b < NU; ++b) v = (u0 + b == u) ? mxl[b] : v;
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        if (half == 0) red[0][wave][u * 32 + l31] = v;
    }
    __syncthreads();
    float ml[NU], suml[NU];
#pragma unroll
    for (int b = 0; b < NU; ++b) {
        const int qi = (u0 + b) * 32 + l31;
        ml[b] = fmaxf(fmaxf(red[0][0][qi], red[0][1][qi]), fmaxf(red[0][2][qi], red[0][3][qi]));
        suml[b] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < NK; ++a)
#pragma unroll
        for (int b = 0; b < NU; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(acc[a * NU + b][r] - ml[b]);
                acc[a * NU + b][r] = p;
                suml[b] += p;
            }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < NU; ++b) v = (u0 + b == u) ? suml[b] : v;
        v += __shfl_xor(v, 32, 64);
        if (half == 0) red[1][wave][u * 32 + l31] = v;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NU; ++b) {
        const int qi = (u0 + b) * 32 + l31;
        const float l = (red[1][0][qi] + red[1][1][qi]) + (red[1][2][qi] + red[1][3][qi]);
#pragma unroll
        for (int a = 0; a < NK; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = (kt0 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                Pt[key * P_LD + qi] = acc[a * NU + b][r] / l;
            }
    }
    __syncthreads();

    // ---- 3. O^T[channel][query] = sum_key V[key][channel] * P^T[key][query]; this wave's channels: [wave * C/4, +C/4)
    f32x16 o[CT][2];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[ct][u][r] = 0.f;
    const int c0 = wave * (C / 4);
    const float* __restrict__ vbase = base + 2 * C + c0 + l31;
#pragma unroll 2
    for (int kk = 0; kk < NKT; ++kk) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;      // the MFMA's k slot `half` carries this key
            float va[CT], pb[2];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) va[ct] = vbase[(size_t)key * ld + ct * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u) pb[u] = Pt[key * P_LD + u * 32 + l31];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int u = 0; u < 2; ++u) o[ct][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[ct], pb[u], o[ct][u], 0, 0, 0);
        }
    }
    // lane = query; accumulator rows 4j .. 4j+3 = 4 consecutive channels (+ 4 half, + 8 j)
    float* __restrict__ obase = out + (img * HW + q0) * ldo + c0 + 4 * half;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = o[ct][u][4 * j + e];
                *reinterpret_cast<f32x4*>(obase + (size_t)(u * 32 + l31) * ldo + ct * 32 + 8 * j) = v;
            }
}


// ---- the same kernel on the fp16 matrix pipe in "x3h" arithmetic (conv3_halo_x3h.hip: every fp32 operand = h + l' * 2^-11 with two
// fp16 pieces, three exact products, the cross terms in their own accumulator): fp32-equivalent results at 3/16 of the f32 MFMA's
// matrix time.  Used when the encoder runs conv_arith = 'x3h' (its q, k, v are GroupNorm-scale O(1) values, the probabilities lie in
// [0, 1]: a probability below fp16's normal range keeps an ABSOLUTE error of 1.5e-11).  Differences to the f32 form above:
//   1. scores: lane = (key | query row, 8 consecutive channels of a 16-channel k-step): two float4 loads, split in registers;
//   2. the normalised probabilities are parked already split and TRANSPOSED — Ph / Pl [query][key] f16, 528-byte rows (conflict-free
//      8-byte writes and 16-byte reads) — so a P fragment of step 3 is one ds_read_b128 per plane;
//   3. O^T = V^T . P^T in k-steps of 16 keys: a lane gathers its channel's 8 keys from global (as the f32 form gathers 2) and splits them.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3h_split8(const float (&x)[8], h16x8& h, h16x8& l) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)x[e];
        h[e] = hh;
        l[e] = (_Float16)((x[e] - (float)hh) * 2048.f);
    }
}

template <int HW, int C>
__global__ __launch_bounds__(256, 2) void attn_spatial_x3h_kernel(const float* __restrict__ qkv, float* __restrict__ out, long long ld,
                                                                  long long ldo, float scale) {
    constexpr int NKT = HW / 32;                  // key tiles
    constexpr int TPW = NKT * 2 / 4;              // (key tile, query tile) pairs per wave: 4 (HW = 256) or 1 (HW = 64)
    constexpr int CT = C / 4 / 32;                // output-channel tiles per wave
    constexpr int P_LDB = HW * 2 + 16;            // bytes per query row of one probability plane
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    unsigned char* Ph = reinterpret_cast<unsigned char*>(smem_f);                       // [64 queries][P_LDB]
    unsigned char* Pl = Ph + 64 * P_LDB;
    float (*red)[4][64] = reinterpret_cast<float (*)[4][64]>(Pl + 64 * P_LDB);         // [max | sum][wave][query]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long img = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const float* __restrict__ base = qkv + img * HW * ld;

    // ---- 1. scores (acc: h.h products, accx: the two cross products at 2^11)
    f32x16 acc[TPW], accx[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    constexpr int NK = TPW >= 2 ? TPW / 2 : 1;    // distinct key tiles of the wave
    constexpr int NU = TPW >= 2 ? 2 : 1;          // distinct query tiles of the wave
    const int kt0 = TPW >= 2 ? wave * NK : (wave >> 1);
    const int u0 = TPW >= 2 ? 0 : (wave & 1);
    const float* krow[NK];
    const float* qrow[NU];
#pragma unroll
    for (int a = 0; a < NK; ++a) krow[a] = base + (size_t)((kt0 + a) * 32 + l31) * ld + C + 8 * half;
#pragma unroll
    for (int b = 0; b < NU; ++b) qrow[b] = base + (size_t)(q0 + (u0 + b) * 32 + l31) * ld + 8 * half;
    // (raw rows one k-step ahead in registers: the loads of step ks + 1 fly while step ks is split and multiplied)
    f32x4 kraw[NK][2], qraw[NU][2];
    auto fetch1 = [&](int ks) {
#pragma unroll
        for (int a = 0; a < NK; ++a) { kraw[a][0] = *reinterpret_cast<const f32x4*>(krow[a] + 16 * ks); kraw[a][1] = *reinterpret_cast<const f32x4*>(krow[a] + 16 * ks + 4); }
#pragma unroll
        for (int b = 0; b < NU; ++b) { qraw[b][0] = *reinterpret_cast<const f32x4*>(qrow[b] + 16 * ks); qraw[b][1] = *reinterpret_cast<const f32x4*>(qrow[b] + 16 * ks + 4); }
    };
    fetch1(0);
#pragma unroll 2
    for (int ks = 0; ks < C / 16; ++ks) {
        h16x8 kh[NK], kl[NK], qh[NU], ql[NU];
#pragma unroll
        for (int a = 0; a < NK; ++a) {
            const float x[8] = {kraw[a][0][0], kraw[a][0][1], kraw[a][0][2], kraw[a][0][3], kraw[a][1][0], kraw[a][1][1], kraw[a][1][2], kraw[a][1][3]};
            x3h_split8(x, kh[a], kl[a]);
        }
#pragma unroll
        for (int b = 0; b < NU; ++b) {
            const float x[8] = {qraw[b][0][0], qraw[b][0][1], qraw[b][0][2], qraw[b][0][3], qraw[b][1][0], qraw[b][1][1], qraw[b][1][2], qraw[b][1][3]};
            x3h_split8(x, qh[b], ql[b]);
        }
        fetch1(ks + 1 < C / 16 ? ks + 1 : ks);
#pragma unroll
        for (int a = 0; a < NK; ++a)
#pragma unroll
            for (int b = 0; b < NU; ++b) {
                accx[a * NU + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[a], qh[b], accx[a * NU + b], 0, 0, 0);
                accx[a * NU + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[a], ql[b], accx[a * NU + b], 0, 0, 0);
                acc[a * NU + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[a], qh[b], acc[a * NU + b], 0, 0, 0);
            }
    }

    // ---- 2. softmax over all keys (as the f32 form)
    float mxl[NU];
#pragma unroll
    for (int b = 0; b < NU; ++b) mxl[b] = -INFINITY;
#pragma unroll
    for (int a = 0; a < NK; ++a)
#pragma unroll
        for (int b = 0; b < NU; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float s = __builtin_fmaf(accx[a * NU + b][r], 4.8828125e-4f, acc[a * NU + b][r]) * scale;
                acc[a * NU + b][r] = s;
                mxl[b] = fmaxf(mxl[b], s);
            }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float v = -INFINITY;
#pragma unroll
        for (int b = 0; b < NU; ++b) v = (u0 + b == u) ? mxl[b] : v;
        v = fmaxf(v, __shfl_xor(v, 32, 64));
        if (half == 0) red[0][wave][u * 32 + l31] = v;
    }
    __syncthreads();
    float ml[NU], suml[NU];
#pragma unroll
    for (int b = 0; b < NU; ++b) {
        const int qi = (u0 + b) * 32 + l31;
        ml[b] = fmaxf(fmaxf(red[0][0][qi], red[0][1][qi]), fmaxf(red[0][2][qi], red[0][3][qi]));
        suml[b] = 0.f;
    }
#pragma unroll
    for (int a = 0; a < NK; ++a)
#pragma unroll
        for (int b = 0; b < NU; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(acc[a * NU + b][r] - ml[b]);
                acc[a * NU + b][r] = p;
                suml[b] += p;
            }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float v = 0.f;
#pragma unroll
        for (int b = 0; b < NU; ++b) v = (u0 + b == u) ? suml[b] : v;
        v += __shfl_xor(v, 32, 64);
        if (half == 0) red[1][wave][u * 32 + l31] = v;
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NU; ++b) {
        const int qi = (u0 + b) * 32 + l31;
        const float l = (red[1][0][qi] + red[1][1][qi]) + (red[1][2][qi] + red[1][3][qi]);
#pragma unroll
        for (int a = 0; a < NK; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                   // accumulator rows 4j .. 4j+3 = 4 consecutive keys
                h16x4 ph, pl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float p = acc[a * NU + b][4 * j + e] / l;
                    const _Float16 hh = (_Float16)p;
                    ph[e] = hh;
                    pl[e] = (_Float16)((p - (float)hh) * 2048.f);
                }
                const int key = (kt0 + a) * 32 + 8 * j + 4 * half;
                *reinterpret_cast<h16x4*>(Ph + qi * P_LDB + key * 2) = ph;
                *reinterpret_cast<h16x4*>(Pl + qi * P_LDB + key * 2) = pl;
            }
    }
    __syncthreads();

    // ---- 3. O^T[channel][query] = sum_key V[key][channel] * P^T[key][query]; this wave's channels: [wave * C/4, +C/4); k-steps of 16 keys;
    // at most two channel tiles at a time (two accumulator sets each: 128 registers), C = 512 walks the keys twice
    constexpr int CTP = CT > 2 ? 2 : CT;
    const int c0 = wave * (C / 4);
#pragma unroll 1
    for (int cp = 0; cp < CT; cp += CTP) {
        f32x16 o[CTP][2], ox[CTP][2];
#pragma unroll
        for (int ct = 0; ct < CTP; ++ct)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[ct][u][r] = 0.f; ox[ct][u][r] = 0.f; }
        const float* __restrict__ vbase = base + 2 * C + c0 + cp * 32 + l31 + (size_t)(8 * half) * ld;
        float vraw[CTP][8];                                                  // the gathered V values of the next k-step
        auto fetch3 = [&](int kk) {
#pragma unroll
            for (int ct = 0; ct < CTP; ++ct)
#pragma unroll
                for (int e = 0; e < 8; ++e) vraw[ct][e] = vbase[(size_t)(kk * 16 + e) * ld + ct * 32];
        };
        fetch3(0);
#pragma unroll 2
        for (int kk = 0; kk < HW / 16; ++kk) {
            h16x8 vh[CTP], vl[CTP], pbh[2], pbl[2];
#pragma unroll
            for (int ct = 0; ct < CTP; ++ct) x3h_split8(vraw[ct], vh[ct], vl[ct]);
            fetch3(kk + 1 < HW / 16 ? kk + 1 : kk);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                pbh[u] = *reinterpret_cast<const h16x8*>(Ph + (u * 32 + l31) * P_LDB + (kk * 16 + 8 * half) * 2);
                pbl[u] = *reinterpret_cast<const h16x8*>(Pl + (u * 32 + l31) * P_LDB + (kk * 16 + 8 * half) * 2);
            }
#pragma unroll
            for (int ct = 0; ct < CTP; ++ct)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    ox[ct][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[ct], pbh[u], ox[ct][u], 0, 0, 0);
                    ox[ct][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[ct], pbl[u], ox[ct][u], 0, 0, 0);
                    o[ct][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[ct], pbh[u], o[ct][u], 0, 0, 0);
                }
        }
        float* __restrict__ obase = out + (img * HW + q0) * ldo + c0 + cp * 32 + 4 * half;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int ct = 0; ct < CTP; ++ct)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(ox[ct][u][4 * j + e], 4.8828125e-4f, o[ct][u][4 * j + e]);
                    *reinterpret_cast<f32x4*>(obase + (size_t)(u * 32 + l31) * ldo + ct * 32 + 8 * j) = v;
                }
    }
}

}  // namespace

extern "C" {

int vf_attn_spatial_f32(const float* qkv, float* out, int n_img, int HW, int C, int64_t ld, int64_t ldo, float scale, void* stream) {
    if (n_img == 0) return VF_OK;
    if (!qkv || !out || n_img < 0 || ld < 3 * (int64_t)C || ldo < C || (ld & 3) || (ldo & 3)) return VF_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = ((size_t)HW * 64 + 2 * 4 * 64) * sizeof(float);
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_spatial_kernel<256, 256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((256 * 64 + 512) * sizeof(float)));
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    if (HW == 256 && C == 256)
        hipLaunchKernelGGL((attn_spatial_kernel<256, 256>), dim3(4, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else if (HW == 64 && C == 512)
        hipLaunchKernelGGL((attn_spatial_kernel<64, 512>), dim3(1, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else if (HW == 64 && C == 256)
        hipLaunchKernelGGL((attn_spatial_kernel<64, 256>), dim3(1, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else
        return VF_ERR_UNSUPPORTED;
    return vf_last_status();
}

int vf_attn_spatial_x3h(const float* qkv, float* out, int n_img, int HW, int C, int64_t ld, int64_t ldo, float scale, void* stream) {
    if (n_img == 0) return VF_OK;
    if (!qkv || !out || n_img < 0 || ld < 3 * (int64_t)C || ldo < C || (ld & 3) || (ldo & 3)) return VF_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const size_t smem = (size_t)2 * 64 * (HW * 2 + 16) + 2 * 4 * 64 * sizeof(float);
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_spatial_x3h_kernel<256, 256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 64 * (256 * 2 + 16) + 2048));
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    if (HW == 256 && C == 256)
        hipLaunchKernelGGL((attn_spatial_x3h_kernel<256, 256>), dim3(4, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else if (HW == 64 && C == 512)
        hipLaunchKernelGGL((attn_spatial_x3h_kernel<64, 512>), dim3(1, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else if (HW == 64 && C == 256)
        hipLaunchKernelGGL((attn_spatial_x3h_kernel<64, 256>), dim3(1, (unsigned)n_img), dim3(256), smem, s, qkv, out, (long long)ld, (long long)ldo, scale);
    else
        return VF_ERR_UNSUPPORTED;
    return vf_last_status();
}

}  // extern "C"
