// ROUND-3 RECORD, not part of libvf_hip.so: the LDS-DMA attention file as it stood at the end of round 3, with the three measured-and-slower
// forms (8-wave K/V-once 177 us, resident 188 us, software-pipelined ring 273 us against the ring kernel's 131 us) behind their environment
// switches (VF_ATTN_DMA8 / VF_ATTN_RES / VF_ATTN_PIPE).  Build a side-by-side library with
//     bash tools/variants.sh attention_dma r3records:"" --source tools/variants/attention_dma_r3_records.hip
// and run with VF_HIP_LIB=viewformer_amd/variants/libvf_r3records.so (no attention dropout, round-3 C-ABI of vf_attn_blockcausal_bf16_lse).
// Block-causal attention of the bf16 transformer arm for bf16 q / k / v in HBM (the c_attn GEMM's bf16 output) and 64-token views, gfx950.
//
// Same semantics and arithmetic as attention_lp.hip MODE 0 (S^T = K.Q^T on v_mfma_f32_32x32x16_bf16, fp32 online softmax with the
// scale folded into exp2, probabilities rounded to bf16, O^T += V^T.P^T; viewformer/models/branching_attention.py:5-18,41-61,82-126),
// but the operands never pass through VGPRs on their way in.  attention_lp.hip at the bench shape (128 scenes x 12 heads x 512
// tokens) spends its time waiting for memory, not computing: removing its MFMAs or its softmax changes nothing (239 -> 224 us),
// removing the K / V tile loads gives 155 us (ablation builds, tools/variants.sh).  Per CU the deliverable load bandwidth is set by
// the bytes in flight (L2 latency x ~16 B/clk), so this kernel keeps THREE key tiles in flight per workgroup and halves the bytes:
//   * a workgroup = 4 waves = 4 consecutive query views (a wave = one view = 64 queries as two 32-query MFMA tiles);
//   * K / V tiles (one key view: 64 keys x 64 features, 8 KB each in bf16) arrive by global_load_lds_dwordx4 into a 4-slot ring
//     (64 KB -> 2 workgroups per CU), issued three tiles ahead behind counted vmcnt waits and one raw s_barrier per tile;
//   * K image: 128-byte rows, 16-byte chunk index XORed with bits 1..3 of the row (on the global source address — the LDS side of the
//     DMA is lane-linear) -> conflict-free ds_read_b128 A fragments;
//   * V image: [feature half][key][32 features], read with ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 features]
//     block and every lane receives 4 consecutive keys of ITS feature — the V^T A-fragment of the P.V MFMA straight from a
//     row-major V tile (no transposing write pass, no padded image);
//   * Q (64 rows x 128 B per wave) also comes by DMA into ring slots 2-3 before they are needed for tiles, then lives in registers;
//   * with 64-token views every (query wave, key tile) pair is entirely visible or entirely masked: no per-element mask, masked
//     tiles are skipped (their weights are exactly 0.0f);
//   * O is normalised, rounded to bf16, transposed through the wave's now idle ring slot and stored as whole 128-byte rows.
#include "../../viewformer_amd/csrc/vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64, KT = 64, QT = 256;
constexpr int K_BYTES = KT * DH * 2;         // 8192
constexpr int TILE_BYTES = 2 * K_BYTES;      // K image then V image

#ifndef ADMA_BUFFER
#define ADMA_BUFFER 1       // buffer_load ... lds with an SGPR resource per (scene, head) and 32-bit lane offsets instead of 64-bit lane addresses
#endif
__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p)));
    return __builtin_bit_cast(bf16x4, r);
}

// PIPE (round 3): the wave's S^T MFMAs of tile t are issued BEFORE the softmax + P.V of tile t - 1 (one more score set in registers, a fifth
// ring slot because tile t - 1's V image stays live one iteration longer): the two records below located the kernel's time in dependency
// stalls inside a wave — S MFMAs -> softmax -> P.V MFMAs with two waves per SIMD — so the matrix pipe now works on the next tile's scores
// while the vector ALU does this tile's exponentials.  Per-query arithmetic and order are unchanged: bit-identical to PIPE = false.
// MEASURED: 273 us against 130 us — st_new + st_old + O^T + Q + P are 256 registers before any address arithmetic, the compiler spills 65,
// and (the round-2 rule) spilled registers in the inner loop cost more than what they buy.  Kept as the third attention record of the
// round; at two waves per SIMD this kernel has no registers left to pipeline with.
template <int RINGN, bool PIPE>
__global__ __launch_bounds__(256, 2) void attn_dma_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                          const __bf16* __restrict__ v, __bf16* __restrict__ out, int H, int T, int ldq, int ldk,
                                                          int ldv, int ldo, float scale, int twin, float* __restrict__ lse_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // RINGN x (K image | V image)
    constexpr int RING = RINGN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // grid (H, B, query blocks): all first blocks (4 key tiles), then all second blocks (8) — measured faster than interleaving the two
    // kinds on a CU (145 us) although the second kind re-reads tiles 0-3 from HBM; ADMA_HEAVY_FIRST flips the order
#ifdef ADMA_HEAVY_FIRST
    const int qblk = (T + QT - 1) / QT - 1 - (int)blockIdx.z;
#else
    const int qblk = (int)blockIdx.z;
#endif
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    const int q0 = qblk * QT;
    const int qw0 = q0 + wave * 64;
    const int nviews = T / KT;
    const int qview = qw0 / KT;                            // this wave's view (>= nviews: the wave only helps moving tiles)
    const bool active = qview < nviews;

    const unsigned char* qb8 = reinterpret_cast<const unsigned char*>(q + b * (size_t)T * ldq + h * DH);
    const unsigned char* kb8 = reinterpret_cast<const unsigned char*>(k + b * (size_t)T * ldk + h * DH);
    const unsigned char* vb8 = reinterpret_cast<const unsigned char*>(v + b * (size_t)T * ldv + h * DH);
#if ADMA_BUFFER
    // one resource per operand, based at this (scene, head): lane offsets stay below T * ld * 2 bytes
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(qb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(kb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(vb8), 0, 0x7fffffff, 0x00020000);
#endif

    // visibility of key view kv from query view qv (attention_f32.hip): plain block-causal kv <= qv; twin = Vc >= 0: views Vc, Vc+1, ...
    // are alternative endings (each sees the prefix and itself); twin <= -2: STREAMS with Sv = -twin views per stream
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
    // in every mask mode a query sees no view index above its own: the workgroup walks key tiles 0 .. its last view — those of them that
    // at least one of its four query views sees (round 3: under the training step's 3-stream mask a workgroup of stream 1 / 2 walked every
    // tile below it, 142 tile steps per (scene, head) for the 78 some wave needs).  `need` = bit kv set <=> some wave sees key view kv, from the
    // closed forms of visible(); the issue and the consume pointer pop its bits in ascending order, ring slots go by SEQUENCE index.
    const int nwalk = min(nviews, q0 / KT + QT / KT);
    const bool dense = nwalk > 64;                          // (more than 64 key views: walk them all, as before)
    unsigned long long need = 0;
    if (!dense) {
        for (int w = 0; w < QT / KT; ++w) {
            const int qv = q0 / KT + w;
            if (qv >= nviews) break;
            int lim;                                        // this view sees key views [0, lim) and itself
            if (Sv > 0) {
                lim = qv % Sv;                              // stream 0: views 0 .. qi (qi is the view itself); streams >= 1: stream-0 views below qi
            } else {
                lim = min(qv, Vc);
            }
            need |= (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull)) | (1ull << qv);
        }
        need &= nwalk >= 64 ? ~0ull : ((1ull << nwalk) - 1ull);
    }
    const int ntiles = dense ? nwalk : __builtin_popcountll(need);            // tile STEPS of this workgroup
    unsigned long long rem_issue = need, rem_use = need;
    auto pop = [&](unsigned long long& rem, int seq) {
        if (dense) return seq;
        const int t = __builtin_ctzll(rem);
        rem &= rem - 1ull;
        return t;
    };

    // ---- DMA.  Every 1 KB piece = 64 lanes x 16 B, lane-linear in LDS.
    // K piece (8 rows x 128 B): lane -> row (lane >> 3), LDS chunk c' = lane & 7 holds global chunk c' ^ ((row >> 1) & 7).
    // V piece (16 keys x 64 B of one feature half): lane -> key (lane >> 2), 16-byte chunk lane & 3.
    const int pr = lane >> 3, pc = lane & 7;
    auto issue_tile = [&](int seq) {
        const int t = pop(rem_issue, seq);                           // (calls come in ascending sequence order)
        unsigned char* dst = smem + (seq % RING) * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = wave * 2 + j;
            const int r = pi * 8 + pr;
#if ADMA_BUFFER
            bufds16(k_rs, dst + pi * 1024, (unsigned)(r * ldk * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), (unsigned)(t * KT * ldk * 2));
#else
            glds16(kb8 + ((size_t)(t * KT + r) * ldk) * 2 + ((pc ^ ((r >> 1) & 7)) << 4), dst + pi * 1024);
#endif
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pi = wave * 2 + j;
            const int key = (pi & 3) * 16 + (lane >> 2);
#if ADMA_BUFFER
            bufds16(v_rs, dst + K_BYTES + pi * 1024, (unsigned)(key * ldv * 2 + (pi >> 2) * 64 + (lane & 3) * 16), (unsigned)(t * KT * ldv * 2));
#else
            glds16(vb8 + ((size_t)(t * KT + key) * ldv) * 2 + (pi >> 2) * 64 + (lane & 3) * 16, dst + K_BYTES + pi * 1024);
#endif
        }
    };
    // Q: the wave's 64 rows -> its private 8 KB of ring slots 2-3 (same swizzled row image as K)
    unsigned char* Qs = smem + 2 * TILE_BYTES + wave * K_BYTES;
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) {
        const int r = pi * 8 + pr;
        const int row = min(qw0 + r, T - 1);
#if ADMA_BUFFER
        bufds16(q_rs, Qs + pi * 1024, (unsigned)(row * ldq * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), 0u);
#else
        glds16(qb8 + ((size_t)row * ldq) * 2 + ((pc ^ ((r >> 1) & 7)) << 4), Qs + pi * 1024);
#endif
    }
    issue_tile(0);
    if (ntiles > 1) issue_tile(1);
    if (ntiles > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    // Q fragments (B operand of S^T = K.Q^T): qb[u][ks] = Q[32 u + l31][16 ks + 8 half + 0..7]
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    bf16x8 qb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qb[u][ks] = *reinterpret_cast<const bf16x8*>(Qs + (u * 32 + l31) * 128 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));

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

    // fragment addresses inside a tile
    const unsigned k_off = (unsigned)(l31 * 128);                    // + t2 * 4096 + (((ks * 2 + half) ^ swz) << 4)
    // V^T fragment of (t2, ks2, d): 16-lane group g = lane >> 4: keys 32 t2 + 16 ks2 + 4 half + (i >> 2) (+ 8), features
    // 32 d + 16 (g & 1) + 4 (i & 3) .. + 3 with i = lane & 15  ->  lane receives keys .. + 0..3 of feature 32 d + l31
    const unsigned v_off = (unsigned)(K_BYTES + (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);

    // ---- S^T = K . Q^T of one tile: each K fragment feeds both query tiles
    auto scores = [&](const unsigned char* tile, f32x16 (&st)[2][2]) {
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + k_off + t2 * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[u][ks], st[u][t2], 0, 0, 0);
            }

    };
    // ---- online softmax + O^T += V^T . P^T of one tile
    auto softmax_pv = [&](f32x16 (&st)[2][2], const unsigned char* tile) {
        // ---- online softmax (lane = one query of each tile; its 32 keys of this key tile per half-wave)
        bf16x8 pb[2][2][2];                                          // [query tile][key half][k-step]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[u][t2][r]), st[u][t2][r + 1]);   // v_max3_f32
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx * scale);
            const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * LOG2E);       // 0 on the first tile (m_run = -inf)
            const float mc = m_new * LOG2E;
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
#ifdef ADMA_X_NOSM
                        const float p = st[u][t2][ks2 * 8 + e];
#else
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t2][ks2 * 8 + e], c2, -mc));
#endif
                        psum += p;
                        pk[e] = (__bf16)p;
                    }
                    pb[u][t2][ks2] = pk;
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

        // ---- O^T += V^T . P^T: k-step (t2, ks2) = keys 32 t2 + 16 ks2 + 8 (e >> 2) + 4 half + (e & 3); each V^T fragment feeds both tiles
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vp = tile + v_off + d * 4096 + (t2 * 32 + ks2 * 16) * 64;
                    const bf16x4 v0 = tr_read(vp);
                    const bf16x4 v1 = tr_read(vp + 8 * 64);
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
#pragma unroll
                    for (int u = 0; u < 2; ++u) ot[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[u][t2][ks2], ot[u][d], 0, 0, 0);
                }
    };
    f32x16 st_old[2][2];
    bool have_old = false;
    int tile_old = 0;

    for (int kt = 0; kt < ntiles; ++kt) {
        // this wave's pieces of tile kt have landed (counted: up to two later tiles stay in flight), its LDS reads of tile kt - 1 (and
        // of Q) are done; the barrier extends both to the workgroup, which frees the slot of tile kt - 1 (kt = 0: the Q slots)
        const int last_issued = min(ntiles - 1, kt == 0 ? 1 : kt + 2);
        if (last_issued - kt >= 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else if (last_issued - kt == 1) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef ADMA_X_NODMA
        if (kt == 0) { issue_tile(ntiles > 2 ? 2 : 0); issue_tile(ntiles > 3 ? 3 : 0); }
        else if (kt + 3 < ntiles) { asm volatile("s_nop 0"); }
#else
        if (kt == 0) {
            if (ntiles > 2) issue_tile(2);
            if (ntiles > 3) issue_tile(3);
        } else if (kt + 3 < ntiles) {
            issue_tile(kt + 3);
        }
#endif
#ifdef ADMA_X_NOCOMPUTE
        continue;
#endif
        const int tcur = pop(rem_use, kt);                           // the key view in ring slot kt % RING
        const bool vis = active && visible(qview, tcur);             // (masked for all 64 queries: the tile contributes exactly 0.0f)
        if constexpr (!PIPE) {
            if (!vis) continue;
            f32x16 st[2][2];                                         // [query tile][key half]
            scores(smem + (kt % RING) * TILE_BYTES, st);
            softmax_pv(st, smem + (kt % RING) * TILE_BYTES);
        } else {
            f32x16 st_new[2][2];
            if (vis) scores(smem + (kt % RING) * TILE_BYTES, st_new);          // 16 MFMAs in flight ...
            __builtin_amdgcn_sched_barrier(0);
            if (have_old) softmax_pv(st_old, smem + (tile_old % RING) * TILE_BYTES);   // ... under the previous visible tile's softmax + P.V
            if (vis) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t2 = 0; t2 < 2; ++t2) st_old[u][t2] = st_new[u][t2];
                tile_old = kt;
            }
            have_old = vis;
        }
    }
    if constexpr (PIPE) {
        if (have_old) softmax_pv(st_old, smem + (tile_old % RING) * TILE_BYTES);
    }

    // ---- normalise, round, transpose through the wave's slice of the (now idle) ring, store whole rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
    if (!active) return;
    unsigned char* Os = smem + wave * K_BYTES;                       // [64 queries][128 B], chunk c stored at c ^ ((row >> 1) & 7)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const int row = u * 32 + l31;
        // the training step's flash backward re-materialises P = exp(S scale - lse) from this per-query log-sum-exp
        if (lse_out && half == 0) lse_out[((size_t)b * H + h) * T + qw0 + row] = m_run[u] + logf(l_tot);
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (__bf16)(ot[u][d][4 * j + e] / l_tot);      // (a division, as attention_lp.hip: same bits)
                *reinterpret_cast<bf16x4*>(Os + row * 128 + ((((unsigned)(d * 4 + j)) ^ swz) << 4) + 8 * half) = o4;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __bf16* __restrict__ ob = out + (b * (size_t)T + qw0) * ldo + h * DH;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + pr;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 128 + pc * 16);
        const int c = pc ^ ((row >> 1) & 7);
        *reinterpret_cast<f32x4*>(ob + (size_t)row * ldo + c * 8) = val;
    }
}


// ---- 8-wave form: one workgroup = 8 consecutive query views = ALL of a 6-context-view scene (T = 512 in the fused twin pass) -------------
// The 4-wave kernel above covers a (scene, head) with two workgroups and the second one re-reads key tiles 0-3 (FETCH_SIZE 393 MB for 302 MB
// of q / k / v, L2 hit rate 18 %: profiles/r2_new_kernels_pmc.txt).  Here every K / V tile of a (scene, head) is fetched ONCE per 8 query
// views: 8 waves (wave = view), an 8-slot ring (128 KB: one workgroup per CU, the same 8 waves per CU as two 4-wave workgroups), the
// waves' Q rows parked in slots 4-7 until they are in registers, tiles 0-3 in flight from the first instruction and 4-7 right behind the
// Q barrier — with <= 8 key views (the bench shape) no slot is ever recycled.  Per-wave arithmetic (tile order, MFMA sequence, softmax)
// is the 4-wave kernel's: results are bit-identical.  Each wave moves ONE 1 KB piece of K and one of V per tile.
// MEASURED (round 3, bench shape 128 scenes x 12 heads x 512 tokens): 176.9 us against the 4-wave kernel's 131.9 us.  The traffic goes
// down as intended, the time goes up: with one barrier per key tile the workgroup runs in lockstep, and under the block-causal mask wave w
// only has work for tiles <= w — 36 of the 64 (wave, tile) slots of a workgroup are busy (56 %), where two co-resident 4-wave workgroups
// (62 % and 81 % busy, not synchronised with each other) fill each other's gaps.  The tile step is bound by the waves' own softmax + MFMA
// work (~3.7 us), not by the DMA.  Kept opt-in as the record of the experiment; what would help is equal work per wave (32 queries of view
// w and 32 of view 7 - w per wave, no per-tile barrier once the ring is resident), not fewer bytes.
constexpr int NW8 = 8, RING8 = 8, QT8 = NW8 * KT;

template <int N>
__device__ __forceinline__ void wait_vmcnt_lgkm0() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
}

__global__ __launch_bounds__(512, 1) void attn_dma8_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                           const __bf16* __restrict__ v, __bf16* __restrict__ out, int H, int T, int ldq, int ldk,
                                                           int ldv, int ldo, float scale, int twin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // RING8 x (K image | V image)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int qblk = (int)blockIdx.z;
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    const int q0 = qblk * QT8;
    const int qw0 = q0 + wave * 64;
    const int nviews = T / KT;
    const int qview = qw0 / KT;
    const bool active = qview < nviews;

    const unsigned char* qb8 = reinterpret_cast<const unsigned char*>(q + b * (size_t)T * ldq + h * DH);
    const unsigned char* kb8 = reinterpret_cast<const unsigned char*>(k + b * (size_t)T * ldk + h * DH);
    const unsigned char* vb8 = reinterpret_cast<const unsigned char*>(v + b * (size_t)T * ldv + h * DH);
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(qb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(kb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(vb8), 0, 0x7fffffff, 0x00020000);

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
    const int ntiles = min(nviews, q0 / KT + NW8);

    const int pr = lane >> 3, pc = lane & 7;
    auto issue_tile = [&](int t) {                                   // this wave's two 1 KB pieces of tile t: K rows 8 wave .. + 7, V piece `wave`
        unsigned char* dst = smem + (t % RING8) * TILE_BYTES;
        const int r = wave * 8 + pr;
        bufds16(k_rs, dst + wave * 1024, (unsigned)(r * ldk * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), (unsigned)(t * KT * ldk * 2));
        const int key = (wave & 3) * 16 + (lane >> 2);
        bufds16(v_rs, dst + K_BYTES + wave * 1024, (unsigned)(key * ldv * 2 + (wave >> 2) * 64 + (lane & 3) * 16), (unsigned)(t * KT * ldv * 2));
    };
    // Q: the wave's 64 rows -> its private 8 KB of slots 4-7 (same swizzled row image as K)
    unsigned char* Qs = smem + 4 * TILE_BYTES + wave * K_BYTES;
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) {
        const int r = pi * 8 + pr;
        const int row = min(qw0 + r, T - 1);
        bufds16(q_rs, Qs + pi * 1024, (unsigned)(row * ldq * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), 0u);
    }
    const int first = min(ntiles, 4);
    for (int t = 0; t < first; ++t) issue_tile(t);
    // Q has landed once at most the 2 * first tile loads issued behind it are outstanding (vmcnt retires in issue order)
    if (first == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (first == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (first == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    bf16x8 qb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            qb[u][ks] = *reinterpret_cast<const bf16x8*>(Qs + (u * 32 + l31) * 128 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave holds its Q: slots 4-7 are free
    const int issued0 = min(ntiles, RING8);
    for (int t = first; t < issued0; ++t) issue_tile(t);

    f32x16 ot[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[u][d][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = scale * LOG2E;
    const unsigned k_off = (unsigned)(l31 * 128);
    const unsigned v_off = (unsigned)(K_BYTES + (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);

    for (int kt = 0; kt < ntiles; ++kt) {
        // tiles issued so far: 0 .. issued0 - 1 before the loop, then tile j + RING8 - 1 at iteration j >= 1: this wave's two pieces of tile
        // kt have landed once at most 2 loads per LATER tile are outstanding
        const int last_issued = min(ntiles - 1, kt == 0 ? issued0 - 1 : max(issued0 - 1, kt + RING8 - 2));
        switch (last_issued - kt) {
            case 0: wait_vmcnt_lgkm0<0>(); break;
            case 1: wait_vmcnt_lgkm0<2>(); break;
            case 2: wait_vmcnt_lgkm0<4>(); break;
            case 3: wait_vmcnt_lgkm0<6>(); break;
            case 4: wait_vmcnt_lgkm0<8>(); break;
            case 5: wait_vmcnt_lgkm0<10>(); break;
            case 6: wait_vmcnt_lgkm0<12>(); break;
            default: wait_vmcnt_lgkm0<14>(); break;
        }
        __builtin_amdgcn_s_barrier();                                // tile kt is complete; every wave is done with tile kt - 1
        if (kt >= 1 && kt + RING8 - 1 < ntiles) issue_tile(kt + RING8 - 1);      // into the slot of tile kt - 1 (more than 8 key views only)
        if (!active || !visible(qview, kt)) continue;
        const unsigned char* tile = smem + (kt % RING8) * TILE_BYTES;

        f32x16 st[2][2];
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + k_off + t2 * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
#pragma unroll
                for (int u = 0; u < 2; ++u) st[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[u][ks], st[u][t2], 0, 0, 0);
            }

        bf16x8 pb[2][2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[u][t2][r]), st[u][t2][r + 1]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx * scale);
            const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * LOG2E);
            const float mc = m_new * LOG2E;
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t2][ks2 * 8 + e], c2, -mc));
                        psum += p;
                        pk[e] = (__bf16)p;
                    }
                    pb[u][t2][ks2] = pk;
                }
            l_run[u] = l_run[u] * alpha + psum;
            m_run[u] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[u][d][r] *= alpha;
            }
        }

#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vp = tile + v_off + d * 4096 + (t2 * 32 + ks2 * 16) * 64;
                    const bf16x4 v0 = tr_read(vp);
                    const bf16x4 v1 = tr_read(vp + 8 * 64);
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
#pragma unroll
                    for (int u = 0; u < 2; ++u) ot[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[u][t2][ks2], ot[u][d], 0, 0, 0);
                }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
    if (!active) return;
    unsigned char* Os = smem + wave * K_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        const int row = u * 32 + l31;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (__bf16)(ot[u][d][4 * j + e] / l_tot);
                *reinterpret_cast<bf16x4*>(Os + row * 128 + ((((unsigned)(d * 4 + j)) ^ swz) << 4) + 8 * half) = o4;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __bf16* __restrict__ ob = out + (b * (size_t)T + qw0) * ldo + h * DH;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 8 + pr;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 128 + pc * 16);
        const int c = pc ^ ((row >> 1) & 7);
        *reinterpret_cast<f32x4*>(ob + (size_t)row * ldo + c * 8) = val;
    }
}


// ---- resident form: K / V of a whole (scene, head) in LDS, equal work per wave, ONE barrier ----------------------------------------
// What the 8-wave record above showed is that the bytes were not the problem, the lockstep was: under the block-causal mask view w has
// w + 1 key tiles of work, so a workgroup that walks the tiles together waits for its last view at every step.  Here (<= 8 views, i.e.
// T <= 512: the evaluator's 6-context-view scenes incl. the fused twin pass) the 8 waves of a workgroup first move the (scene, head)'s
// K / V into LDS — all 8 tiles, 128 KB, every byte fetched once — behind a single vmcnt(0) + barrier, and then never synchronise again:
// wave (p, hq) owns queries 32 hq .. 32 hq + 31 of view p AND of view nviews - 1 - p, so every wave has the same number of
// (32-query, 64-key) units (9 of 36 per pair at 8 views) and walks its tiles at its own pace.  K fragments are shared by the two views
// of a wave where both see the tile.  Per-query arithmetic (tile order, MFMA sequence, fp32 online softmax) is the 4-wave kernel's, so
// results are bit-identical to it.  Q comes straight from global memory as B fragments; O leaves through 2 KB of LDS per wave as
// whole 128-byte rows, 16 rows at a time.
// MEASURED (round 3, bench shape): 188.5 us against the ring kernel's 130.7 us, bit-identical outputs.  Equal work and no barriers did not
// help, because splitting a view's 64 queries over two waves halves what a K / V^T fragment feeds: a wave's second view sees most of its
// tiles alone, so there one LDS fragment read serves one MFMA instead of two, and the (load everything, then compute) order leaves the
// memory pipe idle while the single workgroup of a CU computes (6 x ~5 us of exposed loads per CU).  Together with the 8-wave record
// above this says where the ring kernel's time is: ~3300 SIMD cycles per (64-query, 64-key) unit against 1024 of MFMA and ~1600 of
// softmax VALU work — dependency stalls inside a wave (S MFMAs -> softmax -> P.V MFMAs, two waves per SIMD), not bytes and not balance.
// The form left to try keeps the ring kernel and issues tile t + 1's S MFMAs before tile t's softmax (one more score set in registers).
constexpr int RES_TILES = 8, RES_OS = 2048;

__global__ __launch_bounds__(512, 1) void attn_res_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ k,
                                                          const __bf16* __restrict__ v, __bf16* __restrict__ out, int H, int T, int ldq, int ldk,
                                                          int ldv, int ldo, float scale, int twin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // RES_TILES x (K image | V image), then 8 x RES_OS of O staging

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    const int nviews = T / KT;                                       // <= 8 (launcher)

    const unsigned char* kb8 = reinterpret_cast<const unsigned char*>(k + b * (size_t)T * ldk + h * DH);
    const unsigned char* vb8 = reinterpret_cast<const unsigned char*>(v + b * (size_t)T * ldv + h * DH);
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(kb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(vb8), 0, 0x7fffffff, 0x00020000);

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

    // this wave's two query blocks: view p (few key tiles) and view nviews - 1 - p (many)
    const int p = wave & 3, hq = wave >> 2;
    const int vw[2] = {p, nviews - 1 - p};
    const bool act[2] = {p < (nviews + 1) / 2, p < nviews / 2};      // (odd view count: the middle view is block 0 of its wave only)

    // Q fragments straight from global memory (B operand of S^T = K.Q^T): Q[64 view + 32 hq + l31][16 ks + 8 half + 0..7]
    bf16x8 qb[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = act[u] ? vw[u] * KT + hq * 32 + l31 : 0;
        const __bf16* qs = q + (b * (size_t)T + row) * ldq + h * DH + 8 * half;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qb[u][ks] = *reinterpret_cast<const bf16x8*>(qs + 16 * ks);
    }
    // K / V: every wave moves one 1 KB piece of K and one of V of every tile (as the 8-wave kernel), all in flight at once
    const int pr = lane >> 3, pc = lane & 7;
    for (int t = 0; t < nviews; ++t) {
        unsigned char* dst = smem + t * TILE_BYTES;
        const int r = wave * 8 + pr;
        bufds16(k_rs, dst + wave * 1024, (unsigned)(r * ldk * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), (unsigned)(t * KT * ldk * 2));
        const int key = (wave & 3) * 16 + (lane >> 2);
        bufds16(v_rs, dst + K_BYTES + wave * 1024, (unsigned)(key * ldv * 2 + (wave >> 2) * 64 + (lane & 3) * 16), (unsigned)(t * KT * ldv * 2));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // the (scene, head)'s K / V are resident: no synchronisation from here on

    f32x16 ot[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[u][d][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY};
    float l_run[2] = {0.f, 0.f};
    constexpr float LOG2E = 1.4426950408889634f;
    const float c2 = scale * LOG2E;
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    const unsigned k_off = (unsigned)(l31 * 128);
    const unsigned v_off = (unsigned)(K_BYTES + (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);

    for (int kt = 0; kt < nviews; ++kt) {
        const bool on[2] = {act[0] && visible(vw[0], kt), act[1] && visible(vw[1], kt)};     // wave-uniform
        if (!on[0] && !on[1]) continue;
        const unsigned char* tile = smem + kt * TILE_BYTES;

        f32x16 st[2][2];
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
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + k_off + t2 * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (on[u]) st[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[u][ks], st[u][t2], 0, 0, 0);
            }

        bf16x8 pb[2][2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (!on[u]) continue;
            float mx = -INFINITY;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[u][t2][r]), st[u][t2][r + 1]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[u], mx * scale);
            const float alpha = __builtin_amdgcn_exp2f((m_run[u] - m_new) * LOG2E);
            const float mc = m_new * LOG2E;
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2) {
                    bf16x8 pk;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t2][ks2 * 8 + e], c2, -mc));
                        psum += pe;
                        pk[e] = (__bf16)pe;
                    }
                    pb[u][t2][ks2] = pk;
                }
            l_run[u] = l_run[u] * alpha + psum;
            m_run[u] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[u][d][r] *= alpha;
            }
        }

#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const unsigned char* vp = tile + v_off + d * 4096 + (t2 * 32 + ks2 * 16) * 64;
                    const bf16x4 v0 = tr_read(vp);
                    const bf16x4 v1 = tr_read(vp + 8 * 64);
                    bf16x8 va;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (on[u]) ot[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[u][t2][ks2], ot[u][d], 0, 0, 0);
                }
    }

    // ---- normalise, round, 16 rows at a time through the wave's 2 KB, store whole 128-byte rows
    unsigned char* Os = smem + RES_TILES * TILE_BYTES + wave * RES_OS;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (!act[u]) continue;
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);
        __bf16* __restrict__ ob = out + (b * (size_t)T + vw[u] * KT + hq * 32) * ldo + h * DH;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            if ((l31 >> 4) == part) {
                const int row = l31 & 15;
                const unsigned sw = (unsigned)((l31 >> 1) & 7);
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x4 o4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o4[e] = (__bf16)(ot[u][d][4 * j + e] / l_tot);
                        *reinterpret_cast<bf16x4*>(Os + row * 128 + ((((unsigned)(d * 4 + j)) ^ sw) << 4) + 8 * half) = o4;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int row = it * 8 + pr;                             // 0 .. 15 within the part
                const int qrow = part * 16 + row;
                const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 128 + pc * 16);
                const int c = pc ^ ((qrow >> 1) & 7);
                *reinterpret_cast<f32x4*>(ob + (size_t)qrow * ldo + c * 8) = val;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

}  // namespace

// Launcher used by vf_attn_blockcausal_bf16_v2 (attention_lp.hip).  VF_ERR_UNSUPPORTED when the call does not qualify (the caller then
// takes the register-staged kernel): bf16 q / k / v / out, 64-token views, T a multiple of 64, 16-byte aligned rows.
int vf_attn_dma_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                       float scale, int twin_view, hipStream_t stream, float* lse_out) {
    if (L != KT || T % KT != 0 || ((ldq | ldk | ldv | ldo) & 7)) return VF_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return VF_ERR_UNSUPPORTED;
    if ((size_t)T * (size_t)(ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv)) * 2 >= (1ull << 31)) return VF_ERR_UNSUPPORTED;   // 32-bit offsets per (scene, head)
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 5 * TILE_BYTES);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    // <= 8 views: the resident form (K / V of the (scene, head) in LDS, equal work per wave, one barrier) is OPT-IN (VF_ATTN_RES=1): measured
    // SLOWER at the bench shape, 188.5 vs 130.7 us per launch inside the step (gpurun_out r3h) — see the note above attn_res_kernel
    const char* er = getenv("VF_ATTN_RES");
    if (T <= RES_TILES * KT && T > KT && !lse_out && er && er[0] == '1') {
        static unsigned long long attr_res_devs = 0;
        constexpr int RES_SMEM = RES_TILES * TILE_BYTES + 8 * RES_OS;
        if (vf_attr_needed(&attr_res_devs)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_res_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RES_SMEM);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr_res_devs);
        }
        hipLaunchKernelGGL(attn_res_kernel, dim3((unsigned)H, (unsigned)B), dim3(512), (size_t)RES_SMEM, stream, reinterpret_cast<const __bf16*>(q),
                           reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<__bf16*>(out), H, T, ldq, ldk,
                           ldv, ldo, scale, twin_view);
        return vf_last_status();
    }
    // the 8-wave form (every K / V tile fetched once per 8 query views) is OPT-IN (VF_ATTN_DMA8=1): measured SLOWER at the bench shape — 176.9
    // vs 131.9 us per launch inside the step (gpurun_out r3c) — see the note above attn_dma8_kernel
    const char* e8 = getenv("VF_ATTN_DMA8");
    if (T > QT && e8 && e8[0] == '1' && !lse_out) {
        static unsigned long long attr8_devs = 0;
        if (vf_attr_needed(&attr8_devs)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, RING8 * TILE_BYTES);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr8_devs);
        }
        dim3 grid8((unsigned)H, (unsigned)B, (unsigned)((T + QT8 - 1) / QT8));
        hipLaunchKernelGGL(attn_dma8_kernel, grid8, dim3(512), (size_t)RING8 * TILE_BYTES, stream, reinterpret_cast<const __bf16*>(q),
                           reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<__bf16*>(out), H, T, ldq, ldk,
                           ldv, ldo, scale, twin_view);
        return vf_last_status();
    }
    dim3 grid((unsigned)H, (unsigned)B, (unsigned)((T + QT - 1) / QT));
    // the software-pipelined form is OPT-IN (VF_ATTN_PIPE=1): measured 273 us against 130 us at the bench shape (gpurun_out r3k) — the second
    // score set takes the kernel to 256 VGPRs with 65 spilled (two waves per SIMD leave no more), and the spills cost more than the stalls
    const char* ep = getenv("VF_ATTN_PIPE");
    if (!(ep && ep[0] == '1'))
        hipLaunchKernelGGL((attn_dma_kernel<4, false>), grid, dim3(256), (size_t)4 * TILE_BYTES, stream, reinterpret_cast<const __bf16*>(q),
                           reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<__bf16*>(out), H, T, ldq, ldk, ldv,
                           ldo, scale, twin_view, lse_out);
    else
        hipLaunchKernelGGL((attn_dma_kernel<5, true>), grid, dim3(256), (size_t)5 * TILE_BYTES, stream, reinterpret_cast<const __bf16*>(q),
                           reinterpret_cast<const __bf16*>(k), reinterpret_cast<const __bf16*>(v), reinterpret_cast<__bf16*>(out), H, T, ldq, ldk, ldv,
                           ldo, scale, twin_view, lse_out);
    return vf_last_status();
}

// forward of the bf16 training arm: the same kernel, also writing the per-query log-sum-exp [B][H][T] (fp32) the flash backward
// (attention_train_bf16.hip) needs.  VF_ERR_UNSUPPORTED for shapes the DMA kernel does not take (the trainer then uses the f32 kernels).
extern "C" int vf_attn_blockcausal_bf16_lse(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int T, int L,
                                            int ldq, int ldk, int ldv, int ldo, float scale, int twin_view, void* stream) {
    if (!q || !k || !v || !out || !lse || B <= 0 || H <= 0 || T <= 0 || !(scale > 0.f)) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    return vf_attn_dma_launch(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale, twin_view, (hipStream_t)stream, lse);
}
