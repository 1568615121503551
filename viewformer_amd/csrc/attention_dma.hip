// Block-causal attention of the bf16 transformer arm for bf16 q / k / v in HBM (the c_attn GEMM's bf16 output) and 64-token views, gfx950.
//
// Semantics of attention_lp.hip MODE 0 (S^T = K.Q^T on v_mfma_f32_32x32x16_bf16, fp32 online softmax, probabilities rounded to bf16,
// O^T += V^T.P^T; viewformer/models/branching_attention.py:5-18,41-61,82-126), but the operands never pass through VGPRs on their way in.
// attention_lp.hip at the bench shape (128 scenes x 12 heads x 512 tokens) spends its time waiting for memory, not computing: removing its
// MFMAs or its softmax changes nothing (239 -> 224 us), removing the K / V tile loads gives 155 us (ablation builds, tools/variants.sh).
// Per CU the deliverable load bandwidth is set by the bytes in flight (L2 latency x ~16 B/clk), so this kernel keeps THREE key tiles in
// flight per workgroup and halves the bytes:
//   * a workgroup = 4 consecutive query views = 256 queries, as 4 waves x 64 queries (U = 2: a wave = one view = two 32-query MFMA tiles
//     that share every K / V^T fragment) or as 8 waves x 32 queries (U = 1, see below);
//   * K / V tiles (one key view: 64 keys x 64 features, 8 KB each in bf16) arrive by buffer_load ... lds (LDS-DMA) into a 4-slot ring
//     (64 KB -> 2 workgroups per CU), issued three tiles ahead behind counted vmcnt waits and one raw s_barrier per tile;
//   * K image: 128-byte rows, 16-byte chunk index XORed with bits 1..3 of the row (on the global source address — the LDS side of the
//     DMA is lane-linear) -> conflict-free ds_read_b128 A fragments;
//   * V image: [feature half][key][32 features], read with ds_read_b64_tr_b16: a 16-lane group fetches a [4 keys][16 features]
//     block and every lane receives 4 consecutive keys of ITS feature — the V^T A-fragment of the P.V MFMA straight from a
//     row-major V tile (no transposing write pass, no padded image);
//   * Q also comes by DMA into ring slots 2-3 before they are needed for tiles, then lives in registers;
//   * with 64-token views every (query wave, key tile) pair is entirely visible or entirely masked: no per-element mask, masked
//     tiles are skipped (their weights are exactly 0.0f);
//   * O is normalised, rounded to bf16, transposed through the wave's now idle ring slot and stored as whole 128-byte rows.
//
// Three restructurings of the 4-wave form were built and measured in round 3 — an 8-wave workgroup of 64-query waves that reads every
// K / V tile once (177 us), K / V resident in LDS with equal work per wave (188 us), the ring software-pipelined by one tile (273 us: 65
// spilled registers) — all bit-identical to it and all slower than its 131 us at the bench shape; they live in
// tools/variants/attention_dma_r3_records.hip, outside the product library.
//
// FOLD (round 4).  The round-2 PMC pass (profiles/r2_new_kernels_pmc.txt) counts 17.6 vector instructions per MFMA — 560 per (64-query,
// 64-key) unit of 32 MFMAs — so the per-score work was cut from {fma, exp2, add, 1/2 cvt, 1/2 max} to {exp2, add, 1/2 cvt, 1/2 max}:
//   * Q is multiplied by scale * log2(e) once per wave on its way into registers (re-rounded to bf16: one more 2^-9 rounding of q, inside
//     the bf16 arm's stated tolerance), so the MFMA produces the exponent of 2 directly;
//   * the score accumulators START at -m_ref, the query's reference maximum so far, instead of 0: the subtraction of the maximum is the
//     MFMA's C operand, p = exp2(accumulator) with no arithmetic in between;
//   * m_ref is only moved when a tile's largest score exceeds it by more than THR = 8 (p <= 2^8: harmless in fp32 / bf16, whose
//     exponent range is fp32's) — the exact softmax, evaluated against a stale reference; the O^T rescale and the extra subtraction
//     happen only for the tiles (and only in the waves) where some query's reference moves;
//   * the output normalisation is one reciprocal per query and 64 multiplications (64 IEEE divisions were a tile's worth of VALU).
//   Per-query arithmetic depends on that query's own visible tiles only (the wave-uniform tests merely skip work that would multiply by 1 or
//   subtract 0), so the fused twin pass stays bit-identical to two passes and the two wave shapes below are bit-identical to each other.
//   No longer bit-identical to attention_lp.hip (which keeps the fma form for fp32 inputs and ragged views): tests compare the two within
//   the arm's tolerance.  MEASURED (gpurun_out r4d, bench shape): 23.8 M instead of 30.2 M vector instructions per launch, 109 us
//   instead of 118 us — the instruction count was NOT the bound: the wave-parked share (SQ_WAIT_ANY, 38 % of the wave cycles) did not move.
//
// U = 1 (round 4, the "32-query wave tile" of VERDICT r3): what the counters say is that waves are PARKED — a wave's tile step is a serial
// chain S MFMAs -> softmax -> P.V MFMAs, the workgroup advances one tile per barrier at the pace of its busiest wave, and with 237 registers
// only two waves share a SIMD.  With one 32-query tile per wave the chain per tile step is half as long, the same 256 queries occupy 8
// waves of <= 128 registers (4 waves per SIMD from the same two workgroups per CU), and the wave -> view map (w < 4 ? w : 7 - w, second
// half of the view for w >= 4) puts a light and a heavy view on every SIMD pair of a workgroup.  The price: every K / V^T fragment read
// from LDS feeds one MFMA instead of two.
//
// DROP (round 4): attention dropout of the training step (branching_attention.py:15-17) on the probabilities — the mask of vf_common.h,
// one hashed word per four consecutive keys of a query = registers 4 g .. 4 g + 3 of a lane's score tile; the softmax normaliser and the
// log-sum-exp are over the UNdropped weights, the dropped ones enter P.V, 1 / (1 - rate) is folded into the output normalisation.
#include <type_traits>
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64, KT = 64, QT = 256;
constexpr int K_BYTES = KT * DH * 2;         // 8192
constexpr int TILE_BYTES = 2 * K_BYTES;      // K image then V image
#ifndef ADMA_RING
#define ADMA_RING 4         // 4 slots = 64 KB: two workgroups per CU, three tiles in flight.  3 slots = 48 KB: THREE workgroups per CU, two tiles in flight, and the
                            // first tile steps one ahead only (Q lands in slots 1-2, so tiles 1 and 2 are issued behind the first barrier) — build.py 'adma_ring3'
#endif
constexpr int RING = ADMA_RING;
static_assert(RING == 3 || RING == 4, "ring depth");
constexpr float LOG2E = 1.4426950408889634f;
#ifndef ADMA_PSWAP
#define ADMA_PSWAP 1
#endif
#ifndef ADMA_HEAVY_FIRST
#define ADMA_HEAVY_FIRST 1  // query blocks dispatched heaviest first (round 6); 0 = in index order (round 5)
#endif
#ifndef ADMA_REGROUP
#define ADMA_REGROUP 1      // streams mask: query views grouped so that a workgroup's views share their key tiles (vf_common.h: vf_attn_query_groups); 0 = four
                            // consecutive views per workgroup (build.py 'adma_consecutive')
#endif
// (round 5 A/B, removed with the view groups: ADMA_PAIRGRID — the query blocks of a (scene, head) adjacent on one XCD: fewer HBM re-reads, 20 % slower)
constexpr float ADMA_THR = 8.0f;             // a query's reference maximum moves when a tile exceeds it by more than this (exponent-of-2 units)

__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ bf16x4 tr_read(const unsigned char* p) {
    const s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const_cast<unsigned char*>(p)));
    return __builtin_bit_cast(bf16x4, r);
}
// this wave's loads: at most N outstanding; its LDS reads: done
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0() {
    static_assert(N == 0 || N == 2 || N == 4 || N == 8, "vmcnt immediates of this kernel");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N == 2 || N == 4 || N == 8, "vmcnt immediates of this kernel");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// the two half-waves of a lane pair (lane, lane ^ 32) hold the two key halves of one query: their maximum / sum through
// v_permlane32_swap (one VALU instruction, gfx950) instead of ds_bpermute (an LDS round trip in the middle of the softmax's serial chain
// plus six address instructions).  Both operands = x: r[0] = x of the low half-wave in all 64 lanes, r[1] = x of the high one
__device__ __forceinline__ float xhalf_max(float x) {
#if ADMA_PSWAP
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
#else
    return fmaxf(x, __shfl_xor(x, 32, 64));
#endif
}
__device__ __forceinline__ float xhalf_sum(float x) {
#if ADMA_PSWAP
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);          // (low + high in every lane: the same sum the shuffle form makes in both half-waves)
#else
    return x + __shfl_xor(x, 32, 64);
#endif
}

#ifdef ADMA_STAMPS       // phase timeline (tools/microbench.py attn_stamps): per wave, cycles summed over its tile steps
#define ADMA_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt[i]) :: "memory")
#else
#define ADMA_STAMP(i)
#endif

// U = 32-query MFMA tiles per wave: 2 -> 4 waves of 64 queries, 1 -> 8 waves of 32 queries; both cover QT = 256 queries = 4 views
template <bool DROP, int U>
__global__ __launch_bounds__(U == 2 ? 256 : 512, U == 2 ? 2 : 4) void attn_dma_kernel(
    const __bf16* __restrict__ q, const __bf16* __restrict__ k, const __bf16* __restrict__ v, __bf16* __restrict__ out, int H, int T, int ldq,
    int ldk, int ldv, int ldo, float scale, int twin, float* __restrict__ lse_out, uint32_t drop_thresh, float drop_scale, uint32_t drop_seed,
    uint32_t drop_site, uint32_t drop_plane0, vf_attn_groups groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // RING x (K image | V image)
    constexpr int NW = 8 / U;                  // waves per workgroup
    constexpr int PW = 8 / NW;                 // 1 KB pieces of K (and of V) a wave moves per tile
    constexpr int LPT = 2 * PW;                // loads per wave and tile
    constexpr int QB = 32 * U * 128;           // bytes of a wave's Q rows (and of its O staging)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // grid (H, B, query groups): all groups of one kind back to back — measured faster than interleaving two kinds on a CU (145 us) although the
    // later kind re-reads key tiles from HBM — and, since round 6, the HEAVIEST kind first (the launch no longer ends with a partly filled round of
    // its longest workgroups)
    // a workgroup = one GROUP of up to four query views (vf_common.h: vf_attn_query_groups — four consecutive views, or under the streams mask views that
    // share their key tiles), groups in dispatch order heaviest first
    const int grp = (int)blockIdx.z;
    const int h = blockIdx.x;
    const size_t b = blockIdx.y;
    // U = 2: wave w is view w of the group.  U = 1: waves w and 7 - w share view min(w, 7 - w) (first / second 32 queries): with waves going
    // to SIMDs round-robin, every SIMD of the CU then hosts an early (few visible tiles) and a late view of the group
    const int wview = U == 2 ? wave : (wave < 4 ? wave : 7 - wave);
    const int nviews = T / KT;
    const int qview = groups.view[grp][wview];             // this wave's view (0xFF / >= nviews: the wave only helps moving tiles)
    const bool active = qview < nviews;
    const int qw0 = (active ? qview : 0) * 64 + (U == 2 ? 0 : (wave >> 2) * 32);

    const unsigned char* qb8 = reinterpret_cast<const unsigned char*>(q + b * (size_t)T * ldq + h * DH);
    const unsigned char* kb8 = reinterpret_cast<const unsigned char*>(k + b * (size_t)T * ldk + h * DH);
    const unsigned char* vb8 = reinterpret_cast<const unsigned char*>(v + b * (size_t)T * ldv + h * DH);
    // one resource per operand, based at this (scene, head): lane offsets stay below T * ld * 2 bytes
    const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(qb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t k_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(kb8), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(vb8), 0, 0x7fffffff, 0x00020000);

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
    int vmax = 0;
    for (int w = 0; w < QT / KT; ++w) { const int qv = groups.view[grp][w]; if (qv < nviews) vmax = max(vmax, qv); }
    const int nwalk = min(nviews, vmax + 1);
    const bool dense = nwalk > 64;                          // (more than 64 key views: walk them all)
    unsigned long long need = 0;
    if (!dense) {
        for (int w = 0; w < QT / KT; ++w) {
            const int qv = groups.view[grp][w];
            if (qv >= nviews) continue;
            // this view sees key views [0, lim) and itself.  Streams: stream 0 sees views 0 .. qi (qi is the view itself); streams >= 1 the
            // stream-0 views below qi
            const int lim = Sv > 0 ? qv % Sv : min(qv, Vc);
            need |= (lim >= 64 ? ~0ull : ((1ull << lim) - 1ull)) | (1ull << qv);
        }
        need &= nwalk >= 64 ? ~0ull : ((1ull << nwalk) - 1ull);
    }
    const int ntiles = dense ? nwalk : __builtin_popcountll(need);            // tile STEPS of this workgroup
    // this wave's own visible key views as a bit mask (same closed form): the per-tile test is a shift, not visible()'s integer divisions —
    // the stamps of the first round-4 build showed ~600 cycles per tile step between the barrier and the first MFMA
    unsigned long long mine = 0;
    if (!dense && active) {
        const int lim = Sv > 0 ? qview % Sv : min(qview, Vc);
        mine = ((lim >= 64 ? ~0ull : ((1ull << lim) - 1ull)) | (1ull << qview)) & need;
    }
    unsigned long long rem_issue = need, rem_use = need;
    auto pop = [&](unsigned long long& rem, int seq) {
        if (dense) return seq;
        const int t = __builtin_ctzll(rem);
        rem &= rem - 1ull;
        return t;
    };

#ifdef ADMA_STAMPS
    unsigned long long tt[8];
    unsigned acc_t[6] = {0, 0, 0, 0, 0, 0};
    ADMA_STAMP(7);
#endif
    // ---- DMA.  Every 1 KB piece = 64 lanes x 16 B, lane-linear in LDS.
    // K piece (8 rows x 128 B): lane -> row (lane >> 3), LDS chunk c' = lane & 7 holds global chunk c' ^ ((row >> 1) & 7).
    // V piece (16 keys x 64 B of one feature half): lane -> key (lane >> 2), 16-byte chunk lane & 3.
    const int pr = lane >> 3, pc = lane & 7;
    auto issue_tile = [&](int seq) {
        const int t = pop(rem_issue, seq);                           // (calls come in ascending sequence order)
        unsigned char* dst = smem + (seq % RING) * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int pi = wave * PW + j;
            const int r = pi * 8 + pr;
            bufds16(k_rs, dst + pi * 1024, (unsigned)(r * ldk * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), (unsigned)(t * KT * ldk * 2));
        }
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int pi = wave * PW + j;
            const int key = (pi & 3) * 16 + (lane >> 2);
            bufds16(v_rs, dst + K_BYTES + pi * 1024, (unsigned)(key * ldv * 2 + (pi >> 2) * 64 + (lane & 3) * 16), (unsigned)(t * KT * ldv * 2));
        }
    };
    // Q: the wave's 32 U rows -> its private slice of ring slots 2-3 (same swizzled row image as K)
    unsigned char* Qs = smem + (RING - 2) * TILE_BYTES + wave * QB;
#pragma unroll
    for (int pi = 0; pi < 4 * U; ++pi) {
        const int r = pi * 8 + pr;
        const int row = min(qw0 + r, T - 1);
        bufds16(q_rs, Qs + pi * 1024, (unsigned)(row * ldq * 2 + ((pc ^ ((r >> 1) & 7)) << 4)), 0u);
    }
    issue_tile(0);
    if (RING == 4 && ntiles > 1) issue_tile(1);
    if (RING == 4 && ntiles > 1) wait_vm<2 * LPT>(); else wait_vm<LPT>();         // Q has landed (vmcnt retires in issue order)
    // Q fragments (B operand of S^T = K.Q^T): qb[u][ks] = Q[32 u + l31][16 ks + 8 half + 0..7] * scale * log2 e, re-rounded to bf16 (FOLD)
    const unsigned swz = (unsigned)((l31 >> 1) & 7);
    const float c2 = scale * LOG2E;
    bf16x8 qb[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 raw = *reinterpret_cast<const bf16x8*>(Qs + (u * 32 + l31) * 128 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
#pragma unroll
            for (int e = 0; e < 8; ++e) qb[u][ks][e] = (__bf16)((float)raw[e] * c2);
        }

    f32x16 ot[U][2];                                                 // [query tile][feature half]
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[u][d][r] = 0.f;
    float m_ref[U], l_run[U];                                        // reference maximum (exponent-of-2 units; valid once `seen`), running sum
#pragma unroll
    for (int u = 0; u < U; ++u) { m_ref[u] = 0.f; l_run[u] = 0.f; }
    bool seen = false;                                               // wave-uniform: a visible tile has been processed
    // attention dropout: mask plane (scene, head), group q * (T / 4) + (key >> 2); this lane's keys of a tile are + 4 half + ...
    uint32_t drop_key = 0u, drop_q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) drop_q[u] = 0u;
    if constexpr (DROP) {
        drop_key = vf_dropout_key(drop_seed, drop_site, drop_plane0 + (uint32_t)(b * H + h));
#pragma unroll
        for (int u = 0; u < U; ++u) drop_q[u] = (uint32_t)(qw0 + u * 32 + l31) * (uint32_t)(T >> 2) + (uint32_t)half;
    }

    // fragment addresses inside a tile
    const unsigned k_off = (unsigned)(l31 * 128);                    // + t2 * 4096 + (((ks * 2 + half) ^ swz) << 4)
    // V^T fragment of (t2, ks2, d): 16-lane group g = lane >> 4: keys 32 t2 + 16 ks2 + 4 half + (i >> 2) (+ 8), features
    // 32 d + 16 (g & 1) + 4 (i & 3) .. + 3 with i = lane & 15  ->  lane receives keys .. + 0..3 of feature 32 d + l31
    const unsigned v_off = (unsigned)(K_BYTES + (4 * half + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);

    // ---- S^T = K . Q^T of one tile (each K fragment feeds the wave's U query tiles), accumulated from minus the reference maximum
    auto scores = [&](const unsigned char* tile, f32x16 (&st)[U][2]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float c0 = seen ? -m_ref[u] : 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[u][t2][r] = c0;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + k_off + t2 * 4096 + ((((unsigned)(ks * 2 + half)) ^ swz) << 4));
#pragma unroll
                for (int u = 0; u < U; ++u) st[u][t2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[u][ks], st[u][t2], 0, 0, 0);
            }
            // U = 1 lives in 128 registers (four waves per SIMD): K fragments two at a time — all eight in flight would push the Q fragments
            // into scratch, and a scratch reload's vmcnt(0) would drain the LDS-DMA ring every tile
            if constexpr (U == 1) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the probabilities of k-step (t2, ks2) of query tile u as the B fragment of the P.V MFMA; SUB: the reference maximum moves by dlt
    auto prob_frag = [&](auto sub, const f32x16 (&stu)[2], int u, int t2, int ks2, int tcur, float dlt, float& psum) {
        bf16x8 pk;
        // DROP: registers 8 ks2 + 0..3 and + 4..7 are keys 64 tcur + 32 t2 + 16 ks2 + 4 half + {0..3} and {8..11}: two mask groups
        uint32_t w[2] = {0u, 0u};
        if constexpr (DROP) {
#pragma unroll
            for (int g = 0; g < 2; ++g) w[g] = vf_dropout_word(drop_key, drop_q[u] + (uint32_t)(tcur * 16 + t2 * 8 + ks2 * 4 + g * 2));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = stu[t2][ks2 * 8 + e];
#ifdef ADMA_X_NOSM
            const float p = x;
#else
            const float p = __builtin_amdgcn_exp2f(decltype(sub)::value ? x - dlt : x);
#endif
            psum += p;                                               // the normaliser is over the undropped weights
            if constexpr (DROP) pk[e] = (__bf16)(vf_dropout_keep(w[e >> 2], e & 3, drop_thresh) ? p : 0.f);
            else pk[e] = (__bf16)p;
        }
        return pk;
    };
    auto vt_frag = [&](const unsigned char* tile, int t2, int ks2, int d) {
        const unsigned char* vp = tile + v_off + d * 4096 + (t2 * 32 + ks2 * 16) * 64;
        const bf16x4 v0 = tr_read(vp);
        const bf16x4 v1 = tr_read(vp + 8 * 64);
        bf16x8 va;
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = v0[e]; va[4 + e] = v1[e]; }
        return va;
    };
    // the tile's largest score of this lane's query (both half-waves), relative to m_ref once `seen`
    auto tile_max = [&](const f32x16 (&stu)[2]) {
        float mx = -INFINITY;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, stu[t2][r]), stu[t2][r + 1]);   // v_max3_f32
        return xhalf_max(mx);
    };
    // ---- online softmax + O^T += V^T . P^T of one tile (lane = one query of each of the wave's tiles; its 32 keys of the key tile per half-wave)
    auto softmax_pv = [&](f32x16 (&st)[U][2], const unsigned char* tile, int tcur) {
        // all probabilities first, then the P.V MFMAs: every V^T fragment feeds the wave's U query tiles.  (For U = 1 a form that fed each
        // k-step's probabilities straight into its two MFMAs was tried to shorten live ranges: the allocator spilled MORE — 20 registers
        // against 6 — because it then kept the V^T fragments of all four steps in flight.)
        bf16x8 pb[U][2][2];                                          // [query tile][key half][k-step]
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float mx = tile_max(st[u]);
            // dlt: what this query's reference moves by — everything on its first tile, the excess over the threshold later
            const float dlt = !seen ? mx : (mx > ADMA_THR ? mx : 0.f);
            const bool moved = !seen || __builtin_amdgcn_ballot_w64(dlt != 0.f) != 0ull;      // wave-uniform
            float psum = 0.f;
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
                for (int ks2 = 0; ks2 < 2; ++ks2)
                    pb[u][t2][ks2] = moved ? prob_frag(std::true_type{}, st[u], u, t2, ks2, tcur, dlt, psum)
                                           : prob_frag(std::false_type{}, st[u], u, t2, ks2, tcur, dlt, psum);
            if (!seen) {
                l_run[u] = psum;
                m_ref[u] = dlt;
            } else if (moved) {
                const float alpha = __builtin_amdgcn_exp2f(-dlt);    // 1 for the queries of the wave whose reference stays
                l_run[u] = l_run[u] * alpha + psum;
                m_ref[u] += dlt;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ot[u][d][r] *= alpha;
            } else {
                l_run[u] += psum;
            }
        }
        seen = true;
        ADMA_STAMP(4);
        // k-step (t2, ks2) = keys 32 t2 + 16 ks2 + 8 (e >> 2) + 4 half + (e & 3)
#ifdef VF_X_TRINTRIN
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const bf16x8 va = vt_frag(tile, t2, ks2, d);
#pragma unroll
                    for (int u = 0; u < U; ++u) ot[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pb[u][t2][ks2], ot[u][d], 0, 0, 0);
                }
                if constexpr (U == 1) __builtin_amdgcn_sched_barrier(0);
            }
#else
        // the V^T fragments through vf_tr_frag2_wait (vf_common.h): as compiler intrinsics these reads made hipcc drain vmcnt — the whole
        // DMA ring, incl. the tile issued at the top of this step — in front of every tile's P.V
        const unsigned vaddr = vf_lds_addr(tile) + v_off;
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
            for (int ks2 = 0; ks2 < 2; ++ks2) {
                bf16x8 va0, va1;                                     // feature halves d = 0, 1
                vf_tr_frag2_wait(va0, va1, vaddr, (t2 * 32 + ks2 * 16) * 64, 4096 + (t2 * 32 + ks2 * 16) * 64, 8 * 64);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    ot[u][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va0, pb[u][t2][ks2], ot[u][0], 0, 0, 0);
                    ot[u][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va1, pb[u][t2][ks2], ot[u][1], 0, 0, 0);
                }
                if constexpr (U == 1) __builtin_amdgcn_sched_barrier(0);
            }
#endif
    };

    // (Round 5 measured a SKEWED form of this loop and removed it again: waves 4-7 — the second wave of every SIMD — ran half a step behind
    // waves 0-3 (after barrier kt: softmax + P.V of tile kt - 1 from scores carried across the barrier, then the S MFMAs of tile kt), so
    // that one wave of a SIMD multiplies while the other exponentiates; ring two tiles ahead instead of three.  Bit-identical outputs,
    // 128 registers, and 2.3 % SLOWER in an in-process alternation (112.2 vs 109.7 us at the bench shape, 208 vs 204 us at S = 21):
    // profiles/r5_attention_ab.txt, commit "attention: skewed half-step schedule" in the history.)
    ADMA_STAMP(6);
    for (int kt = 0; kt < ntiles; ++kt) {
        // this wave's pieces of tile kt have landed (counted: up to two later tiles stay in flight), its LDS reads of tile kt - 1 (and
        // of Q) are done; the barrier extends both to the workgroup, which frees the slot of tile kt - 1 (kt = 0: the Q slots)
        ADMA_STAMP(0);
        const int last_issued = RING == 4 ? min(ntiles - 1, kt == 0 ? 1 : kt + 2) : min(ntiles - 1, kt == 0 ? 0 : kt + 1);
        if (last_issued - kt >= 2) wait_vm_lgkm0<2 * LPT>();
        else if (last_issued - kt == 1) wait_vm_lgkm0<LPT>();
        else wait_vm_lgkm0<0>();
        ADMA_STAMP(1);
        __builtin_amdgcn_s_barrier();
        ADMA_STAMP(2);
#ifdef ADMA_STAMPS
        acc_t[0] += (unsigned)(tt[1] - tt[0]);                       // wait for the tile's DMA (and the wave's LDS reads)
        acc_t[1] += (unsigned)(tt[2] - tt[1]);                       // wait at the barrier
#endif
#ifdef ADMA_X_NODMA
        if (kt == 0) { issue_tile(ntiles > 2 ? 2 : 0); issue_tile(ntiles > 3 ? 3 : 0); }
        else if (kt + 3 < ntiles) { asm volatile("s_nop 0"); }
#else
        if (kt == 0) {
            if (ntiles > RING - 2) issue_tile(RING - 2);
            if (ntiles > RING - 1) issue_tile(RING - 1);
        } else if (kt + RING - 1 < ntiles) {
            issue_tile(kt + RING - 1);
        }
#endif
#ifdef ADMA_X_NOCOMPUTE
        continue;
#endif
        const int tcur = pop(rem_use, kt);                           // the key view in ring slot kt % RING
        // (a tile masked for all of the wave's queries contributes exactly 0.0f)
        const bool vis = dense ? (active && visible(qview, tcur)) : ((mine >> tcur) & 1ull) != 0ull;
        if (!vis) continue;
        f32x16 st[U][2];                                             // [query tile][key half]
        ADMA_STAMP(3);
        scores(smem + (kt % RING) * TILE_BYTES, st);
        softmax_pv(st, smem + (kt % RING) * TILE_BYTES, tcur);
#ifdef ADMA_STAMPS
        ADMA_STAMP(5);
        acc_t[2] += (unsigned)(tt[3] - tt[2]);                       // DMA issue + bookkeeping
        acc_t[3] += (unsigned)(tt[4] - tt[3]);                       // S MFMAs issued + softmax done (the MFMAs' results consumed)
        acc_t[4] += (unsigned)(tt[5] - tt[4]);                       // V^T reads + P.V MFMAs issued
        acc_t[5] += 1u;                                              // visible tile steps of this wave
#endif
    }

    // ---- normalise, round, transpose through the wave's slice of the (now idle) ring, store whole rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // every wave is done with the ring
#ifdef ADMA_STAMPS
    {
        unsigned long long te;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te) :: "memory");
        if (lse_out && lane == 0) {                                  // (the stamp build writes no log-sum-exp: the pointer carries the stamp buffer)
            const size_t wg = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            unsigned* o = reinterpret_cast<unsigned*>(lse_out) + (wg * NW + wave) * 8;
            for (int i = 0; i < 6; ++i) o[i] = acc_t[i];
            o[6] = (unsigned)(tt[6] - tt[7]);                        // kernel entry -> first tile step (Q in registers)
            o[7] = (unsigned)(te - tt[6]);                           // the tile loop incl. the final barrier
        }
    }
#endif
    if (!active) return;
    unsigned char* Os = smem + wave * QB;                            // [32 U queries][128 B], chunk c stored at c ^ ((row >> 1) & 7)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float l_tot = xhalf_sum(l_run[u]);
        const float inv_l = (DROP ? drop_scale : 1.0f) / l_tot;
        const int row = u * 32 + l31;
#ifndef ADMA_STAMPS
        // the training step's flash backward re-materialises P = exp(S scale - lse) from this per-query log-sum-exp (natural-log units)
        if (lse_out && half == 0) lse_out[((size_t)b * H + h) * T + qw0 + row] = __builtin_fmaf(m_ref[u], 0.69314718055994531f, logf(l_tot));      // (spelled as ONE fma: left to the compiler, the contraction depended on the code around it — 1-ulp differences between two builds of this file)
#endif
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o4;
#pragma unroll
                for (int e = 0; e < 4; ++e) o4[e] = (__bf16)(ot[u][d][4 * j + e] * inv_l);
                *reinterpret_cast<bf16x4*>(Os + row * 128 + ((((unsigned)(d * 4 + j)) ^ swz) << 4) + 8 * half) = o4;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __bf16* __restrict__ ob = out + (b * (size_t)T + qw0) * ldo + h * DH;
#pragma unroll
    for (int it = 0; it < 4 * U; ++it) {
        const int row = it * 8 + pr;
        const f32x4 val = *reinterpret_cast<const f32x4*>(Os + row * 128 + pc * 16);
        const int c = pc ^ ((row >> 1) & 7);
        *reinterpret_cast<f32x4*>(ob + (size_t)row * ldo + c * 8) = val;
    }
}

}  // namespace

// Launcher used by vf_attn_blockcausal_bf16_v2 (attention_lp.hip).  VF_ERR_UNSUPPORTED when the call does not qualify (the caller then
// takes the register-staged kernel): bf16 q / k / v / out, 64-token views, T a multiple of 64, 16-byte aligned rows.
int vf_attn_dma_launch(const void* q, const void* k, const void* v, void* out, int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                       float scale, int twin_view, hipStream_t stream, float* lse_out, float drop_rate, uint32_t drop_seed, uint32_t drop_site,
                       uint32_t drop_plane0) {
    if (L != KT || T % KT != 0 || ((ldq | ldk | ldv | ldo) & 7)) return VF_ERR_UNSUPPORTED;
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) return VF_ERR_UNSUPPORTED;
    if ((size_t)T * (size_t)(ldq > ldk ? (ldq > ldv ? ldq : ldv) : (ldk > ldv ? ldk : ldv)) * 2 >= (1ull << 31)) return VF_ERR_UNSUPPORTED;   // 32-bit offsets per (scene, head)
    if (!(drop_rate >= 0.f && drop_rate < 1.f)) return VF_ERR_BAD_ARG;
    if (drop_rate > 0.f && (unsigned long long)T * (unsigned long long)(T >> 2) >= (1ull << 32)) return VF_ERR_UNSUPPORTED;                   // 32-bit mask groups per plane
    const __bf16 *q_ = reinterpret_cast<const __bf16*>(q), *k_ = reinterpret_cast<const __bf16*>(k), *v_ = reinterpret_cast<const __bf16*>(v);
    __bf16* o_ = reinterpret_cast<__bf16*>(out);
    const vf_attn_groups groups = vf_attn_query_groups(T / KT, twin_view, ADMA_REGROUP != 0, ADMA_HEAVY_FIRST != 0);
    const dim3 grid((unsigned)H, (unsigned)B, (unsigned)groups.n);
    const uint32_t thr = vf_dropout_thresh(drop_rate);
    const float dsc = 1.0f / (1.0f - drop_rate);
    const int nq = (T + QT - 1) / QT;
    if (nq > 64) return VF_ERR_UNSUPPORTED;

    auto launch = [&](auto drop, auto u) -> int {
        constexpr bool DROP = decltype(drop)::value;
        constexpr int U = decltype(u)::value;
        static unsigned long long attr_devs = 0;                   // (one flag per instantiation; bit d: raised on device d)
        if (vf_attr_needed(&attr_devs)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_dma_kernel<DROP, U>), hipFuncAttributeMaxDynamicSharedMemorySize, RING * TILE_BYTES);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr_devs);
        }
        hipLaunchKernelGGL((attn_dma_kernel<DROP, U>), grid, dim3(U == 2 ? 256 : 512), (size_t)RING * TILE_BYTES, stream, q_, k_, v_, o_, H, T, ldq, ldk, ldv, ldo,
                           scale, twin_view, lse_out, thr, dsc, drop_seed, drop_site, drop_plane0, groups);
        return vf_last_status();
    };
    using D1 = std::true_type;
    using D0 = std::false_type;
    using U2 = std::integral_constant<int, 2>;
    using U1 = std::integral_constant<int, 1>;
    // the two wave shapes are bit-identical (tests/test_hip_bf16.py); VF_SEL_ATTN_Q32 picks 8 waves x 32 queries (1, the default) or 4 x 64 (0)
    const bool q32 = vf_selected(VF_SEL_ATTN_Q32) != 0;
    if (drop_rate > 0.f) return q32 ? launch(D1{}, U1{}) : launch(D1{}, U2{});
    return q32 ? launch(D0{}, U1{}) : launch(D0{}, U2{});
}

// forward of the bf16 training arm: the same kernel, also writing the per-query log-sum-exp [B][H][T] (fp32) the flash backward
// (attention_train_bf16.hip) needs, with the training step's attention dropout (drop_rate 0 = off).  VF_ERR_UNSUPPORTED for shapes the
// DMA kernel does not take (the trainer then uses the f32 kernels).
extern "C" int vf_attn_blockcausal_bf16_lse(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int T, int L,
                                            int ldq, int ldk, int ldv, int ldo, float scale, int twin_view, float drop_rate,
                                            uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0, void* stream) {
    if (!q || !k || !v || !out || !lse || B <= 0 || H <= 0 || T <= 0 || !(scale > 0.f)) return VF_ERR_BAD_ARG;
    if (ldq < H * DH || ldk < H * DH || ldv < H * DH || ldo < H * DH) return VF_ERR_BAD_ARG;
    return vf_attn_dma_launch(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale, twin_view, (hipStream_t)stream, lse, drop_rate, drop_seed, drop_site,
                              drop_plane0);
}
