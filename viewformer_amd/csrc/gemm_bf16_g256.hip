// bf16 dense GEMM for the transformer's token matrix (M = scenes x views x 64 tokens, tens of thousands of rows), gfx950:
// 256 x 256 workgroup tile, both operands through LDS by LDS-DMA.
//
//   out[m][n] = epi( sum_k A[m][k] * W[k][n] + bias[n] ) (+ res[m][n])       A bf16 in HBM (LayerNorm / GELU / attention outputs),
//                                                                             W packed bf16 (vf_gemm_bf16_pack), fp32 sums
//
// Why a second kernel (gemm_bf16.hip keeps the 128 x 128 one for ragged / small shapes): at 128 x 128 x 64 per stage a workgroup moves
// 32-48 KB through the L1 -> VGPR/LDS path for 2 MFLOP, i.e. the 64 B/clk/CU of that path are needed in full at the matrix pipe's
// peak rate, and the kernel sat at 25-31 % of the bf16 peak on the four dense layers of a block (tools/microbench.py gemm_tf).  A
// 256 x 256 tile halves the operand bytes per flop:
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 = 4 x 2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers), 32 MFMAs per
//     64-deep stage per wave against 24 ds_read_b128;
//   * a stage = 32 KB of A (256 rows x 128 B) + 32 KB of W, double buffered = 128 KB of the CU's 160 KB LDS, one workgroup per CU;
//   * both arrive by global_load_lds_dwordx4 (1 KB per wave instruction, no VGPR round trip, no ds_write pass): W's packed layout
//     [ks][half][n][16 B] already is the fragment order (a lane's B fragment = one conflict-free ds_read_b128); A rows are 128 B with
//     the 16-byte chunk index XORed by bits 1..3 of the row — applied on the GLOBAL source address, the LDS image stays lane-linear
//     as the DMA requires — so that the 16 rows a ds_read_b128 lane group touches hit 16 distinct 16-byte bank slots;
//   * stage s + 1 is in flight while stage s is multiplied: one "s_waitcnt vmcnt(0) lgkmcnt(0)" + raw s_barrier per stage.
// Same arithmetic as gemm_bf16.hip (same MFMA, same k order, bias / GELU / residual in fp32): results are bit-identical to it.
#include "vf_common.h"
#include "epilogue.h"
#include "../../include/vf_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

#ifndef G256_PANEL
#define G256_PANEL 4          // column tiles per panel (0: all column tiles of a row block together)
#endif
// (Measured and removed, in the history with their numbers: the A tile through VGPRs (round 2), a persistent one-workgroup-per-CU form that
// issues the next tile's first stage before its epilogue — a wash, round 3 —, row groups inside an XCD — more FETCH, 4-5 % slower, round 5,
// profiles/r5_gemm_g256_tile_order.txt —, a 16x16x32-MFMA feasibility probe.)
constexpr int GM = 256, GN = 256, GK = 64;
constexpr int GA_BYTES = GM * GK * 2;        // 32768
constexpr int GB_BYTES = GN * GK * 2;        // 32768 = two 128-column packed blocks
constexpr int GSTAGE = GA_BYTES + GB_BYTES;  // 65536

// LDS-DMA through an SGPR buffer resource + 32-bit lane offsets (buffer_load_dwordx4 ... lds) instead of global_load_lds_dwordx4's 64-bit lane
// addresses: the address path is part of what a piece costs — mlp.c_proj 325 -> 304 us, c_attn 261 -> 244 us (round 2)
__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}

// fp32 output: bias, optional GELU, optional residual.  One code path with wave-uniform flags (eight template instantiations of the
// unrolled 8-tile store made the compiler hoist every tile's addresses and spill 350 registers around the 128 accumulators); a tile is
// still handled as ONE block of 16 back-to-back loads / stores (epilogue.h).
template <bool DROP, int IC>
__device__ __forceinline__ void g256_store_f32(const vf_igemm_args& p, const f32x16 (&acc)[IC][2], int m_tile0, int n_tile0, int wrow0,
                                               int wave_n, int half, int l31, bool full, bool gelu) {
    const long long ldc = p.ldc, ldr = p.ldr;
    const bool has_res = p.res != nullptr;
    const bool dual = p.epilogue == VF_EPI_GELU_DUAL;
    const int odd = l31 & 1;
    const uint32_t drop_thresh = vf_dropout_thresh(p.drop_rate), drop_key = vf_dropout_key(p.drop_seed, p.drop_site, 0u);
    const float drop_scale = 1.0f / (1.0f - p.drop_rate);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_tile0 + wave_n * 64 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < IC; ++i) {
            int m0 = m_tile0 + wrow0 + i * 32 + 4 * half;
            // (DROP instantiation: the tile's row index passes through an opaque asm, so the eight tiles' addresses and mask words are
            // formed tile by tile — hoisted together beside the 128 accumulators they cost 43 spilled registers)
            if constexpr (DROP) asm volatile("" : "+v"(m0) :: "memory");
            const int rows_left = p.M - m0;
            float* o = p.out + (size_t)(rows_left > 0 ? m0 : 0) * ldc + n;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bias;
            if (gelu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = vf_gelu_erf_fast(v[r]);
            }
            if constexpr (DROP) {      // (its own kernel instantiation: as a wave-uniform branch of the common epilogue it cost 47 spilled registers)
                // residual / MLP dropout of the training step (migt.py:216,72) on the layer's output, before the residual joins: registers
                // 4 g .. 4 g + 3 of a lane are rows m0 + 8 g .. + 3 of column n = ONE mask group (vf_common.h): one hash per four values
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint32_t w = vf_dropout_word(drop_key, (uint32_t)(((m0 + p.drop_row0) >> 2) + 2 * g) * (uint32_t)p.Cout + (uint32_t)n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = vf_dropout_keep(w, e, drop_thresh) ? v[4 * g + e] * drop_scale : 0.f;
                }
            }
            if (has_res) {
                const float* rs = p.res + (size_t)(rows_left > 0 ? m0 : 0) * ldr + n;
                float rr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    rr[r] = rs[(long long)(full || row < rows_left ? row : 0) * ldr];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rr[r];
            }
            if (full) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(long long)((r & 3) + 8 * (r >> 2)) * ldc] = v[r];
            } else if (rows_left > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2);
                    if (row < rows_left) o[(long long)row * ldc] = v[r];
                }
            }
            if (dual) {
                // VF_EPI_GELU_DUAL: v is the pre-activation just stored; its GELU goes out as bf16 — lane pairs swap half of their rows so
                // every lane stores two adjacent columns of one row (g256_store_bf16's scheme)
                __builtin_amdgcn_sched_barrier(0);
                __bf16* o16 = reinterpret_cast<__bf16*>(p.out_aux) + (size_t)(rows_left > 0 ? m0 : 0) * ldc + (n - odd);
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = vf_gelu_erf_fast(v[r]);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float give = odd ? v[r] : v[r + 1];
                    const float got = vf_lane_xor1(give);
                    const int row = ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                    bf16x2_t h;
                    h[0] = (__bf16)(odd ? got : v[r]);
                    h[1] = (__bf16)(odd ? v[r + 1] : got);
                    if (full || row < rows_left) *reinterpret_cast<bf16x2_t*>(o16 + (long long)row * ldc) = h;
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // one tile at a time
        }
    }
}

// bf16 output (bias + optional GELU, no residual): neighbouring lanes hold neighbouring columns, so each lane pair swaps half of its
// rows and every lane stores two adjacent columns of one row (4 bytes) — same scheme as gemm_bf16_direct_kernel
// EPI: 0 = bias, 1 = bias + GELU (forward), 2 = VF_EPI_GELU_BWD: (acc + bias) * gelu'(u), u = p.res[m][n] the saved fp32 pre-activation —
// gemm_bf16_direct_kernel's expression (vf_gelu_grad_fast, explicitly rounded): the same bits as that kernel and as the stand-alone pass
template <int EPI, bool FULL, int IC>
__device__ __forceinline__ void g256_store_bf16(const vf_igemm_args& p, const f32x16 (&acc)[IC][2], int m_tile0, int n_tile0, int wrow0,
                                                int wave_n, int half, int l31) {
    __bf16* __restrict__ O = reinterpret_cast<__bf16*>(p.out);
    const int odd = l31 & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_tile0 + wave_n * 64 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < IC; ++i) {
            const int m0 = m_tile0 + wrow0 + i * 32 + 4 * half;
            float t[16];
            if (EPI == 2 || EPI == 4) {
                // (EPI 4, round 6: `res` holds the SAVED DERIVATIVE gelu'(u) as bf16 — written by the forward's EPI 5 — and the epilogue multiplies by it)
                // u through an SGPR buffer resource based at the tile's first row (32-bit lane offsets, no 64-bit lane addresses) that ends
                // with the matrix's last row: rows of a ragged last tile beyond M read as zero instead of faulting.  All of a 32 x 32 block's
                // loads are issued first; the gelu' evaluations then run four at a time, fenced (sixteen interleaved beside the 128
                // accumulator registers spill; fencing the LOADS per four exposed their latency 32 times per tile: 196 us in the step).
                // reserved0 bit 2: u was saved as bf16 — [M][ldr] 2-byte elements behind the same pointer
                const bool u16 = p.reserved0 & 4;
                const unsigned esz = u16 ? 2u : 4u;
                const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(p.res) + (size_t)m_tile0 * p.ldr * esz), 0,
                    (int)min((long long)0x7fffffff, (long long)(p.M - m_tile0) * p.ldr * esz), 0x00020000);
                float uu[16];
                if (u16) {
                    // bf16 u: lane pairs share dwords — the even lane fetches (n, n+1) of row r, the odd lane (n-1, n) of row r+1, and one
                    // exchange hands each lane its own column of both rows (2-byte lane loads ran at a third of the dword rate)
                    const unsigned voff = ((unsigned)(m0 - m_tile0) * (unsigned)p.ldr + (unsigned)(n - odd)) * 2u;
                    unsigned w[8];
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int row = ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                        w[r >> 1] = __builtin_amdgcn_raw_buffer_load_b32(u_rs, voff + (unsigned)row * (unsigned)p.ldr * 2u, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const unsigned mine = w[r >> 1];
                        const unsigned give = odd ? (mine << 16) : (mine & 0xffff0000u);      // as fp32 bits: odd gives its low half, even its high half
                        const unsigned got = vf_lane_xor1(give);
                        uu[r] = __builtin_bit_cast(float, odd ? got : (mine << 16));
                        uu[r + 1] = __builtin_bit_cast(float, odd ? (mine & 0xffff0000u) : got);
                    }
                } else {
                    const unsigned voff = ((unsigned)(m0 - m_tile0) * (unsigned)p.ldr + (unsigned)n) * 4u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned soff = (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldr * 4u;
                        uu[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(u_rs, voff, soff, 0));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        t[4 * q + e] = __fmul_rn(__fadd_rn(acc[i][j][4 * q + e], bias), EPI == 4 ? uu[4 * q + e] : vf_gelu_grad_fast(uu[4 * q + e]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    t[r] = acc[i][j][r] + bias;
                    if (EPI == 1) t[r] = vf_gelu_erf_fast(t[r]);
                }
            }
            // EPI 3 (VF_EPI_GELU_DUAL with a bf16 `out`): first the pre-activation as bf16 to `out`, then — from the fp32 value, not the
            // rounded one — its GELU as bf16 to `out_aux`.  EPI 5 (round 6, reserved0 bit 3): `out` receives gelu'(u) instead of u — what the
            // backward needs u for — from the same erf / exp evaluation as the GELU (vf_gelu_and_grad_fast)
            float gq[EPI == 5 ? 16 : 1];
            if (EPI == 5) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { float f, g; vf_gelu_and_grad_fast(t[r], f, g); t[r] = f; gq[r] = g; }
            }
#pragma unroll
            for (int pass = 0; pass < ((EPI == 3 || EPI == 5) ? 2 : 1); ++pass) {
                __bf16* __restrict__ dst = pass == 0 ? O : reinterpret_cast<__bf16*>(p.out_aux);
                if (EPI == 3 && pass == 1) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = vf_gelu_erf_fast(t[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float lo = (EPI == 5 && pass == 0) ? gq[r] : t[r], hi = (EPI == 5 && pass == 0) ? gq[r + 1] : t[r + 1];
                    const float give = odd ? lo : hi;
                    const float got = vf_lane_xor1(give);
                    const int m = m0 + ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                    bf16x2_t v;
                    v[0] = (__bf16)(odd ? got : lo);
                    v[1] = (__bf16)(odd ? hi : got);
                    if (FULL || m < p.M) *reinterpret_cast<bf16x2_t*>(dst + (size_t)m * p.ldc + (n - odd)) = v;
                }
            }
        }
    }
}

// One output tile: R = 2 IC row blocks of 32 rows (IC = 4: the 256 x 256 tile; IC = 3 / 2: the 192- / 128-row TAIL tiles below) x 256
// columns.  The two wave_m groups take IC blocks each, so every SIMD keeps its two waves whatever the height; the k order of every output
// element is the full tile's (K is never split): a tail tile's results are bit-identical to the full tile's.
template <bool O16, bool DROP, int IC>
__device__ __forceinline__ void g256_tile(const vf_igemm_args& p, unsigned char* smem_b, int m_tile0, int nblk, int tid) {
    constexpr int TROWS = 2 * IC * 32;                     // rows of this tile
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int wrow0 = wave_m * (IC * 32);                  // this wave's first row inside the tile
    const int n_tile0 = nblk * GN;
    const int nstages = p.Cin / GK;

    // ---- LDS-DMA sources.  A: wave w moves rows [32 w, 32 w + 32) of the tile (waves past the tile's height move none), 8 rows (1 KB) per
    // instruction; lane -> row (lane >> 3), LDS chunk c' = lane & 7, global chunk c = c' ^ ((row >> 1) & 7).  W: the 32 KB of the stage are
    // contiguous, wave w moves 4 KB.
    const bool a_mover = wave * 32 < TROWS;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_packed), 0, 0x7fffffff, 0x00020000);
    unsigned avoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 32 + q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int m = m_tile0 + r;
        m = m < p.M ? m : p.M - 1;
        avoff[q] = (unsigned)(((size_t)m * p.lda) * 2 + c * 16);
    }
    const unsigned wvoff = (unsigned)((size_t)nblk * GB_BYTES + wave * 4096 + lane * 16);
    const size_t w_stage_stride = (size_t)(p.Cout / 128) * (GK * 128 * 2);
    auto issue = [&](int s) {
        unsigned char* dst = smem_b + (s & 1) * GSTAGE;
        if (IC == 4 || a_mover) {
#pragma unroll
            for (int q = 0; q < 4; ++q) bufds16(a_rsrc, dst + (wave * 32 + q * 8) * 128, avoff[q], (unsigned)(s * (GK * 2)));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) bufds16(w_rsrc, dst + GA_BYTES + wave * 4096 + q * 1024, wvoff + q * 1024, (unsigned)((size_t)s * w_stage_stride));
    };

    // ---- fragment addresses: A row = wrow0 + i * 32 + l31, chunk (ks * 2 + half) ^ ((l31 >> 1) & 7)   (wrow0 is a multiple of 32: the
    // swizzle bits of the row are l31's)
    const unsigned a_row_off = (unsigned)((wrow0 + l31) * 128);
    const unsigned a_swz = (unsigned)((l31 >> 1) & 7);
    unsigned a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_row_off + ((((unsigned)(ks * 2 + half)) ^ a_swz) << 4);
    // W image: [block (2)][ks][half][n (128)][16 B]; this wave's columns: block wave_n >> 1, n = (wave_n & 1) * 64 + j * 32 + l31
    const unsigned b_off = (unsigned)(GA_BYTES + (wave_n >> 1) * (GK * 128 * 2) + ((half * 128) + (wave_n & 1) * 64 + l31) * 16);

    f32x16 acc[IC][2];
#pragma unroll
    for (int i = 0; i < IC; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef G256_STAMPS       // phase timeline (tools/microbench.py g256_stamps): per wave, cycles summed over the stages
    unsigned long long tt[6];
    unsigned acc_t[6] = {0, 0, 0, 0, 0, 0};
#define G256_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt[i]) :: "memory")
    G256_STAMP(5);
#else
#define G256_STAMP(i)
#endif
    // (A first-round stagger — four groups of CUs 4 000-16 000 cycles apart, so that the 256 epilogues of a round do not reach the memory system
    // together — was measured in round 3 and removed: 388 -> 397-400 us for c_fc at M = 65 536, the tail it adds outweighs what it spreads.)
    issue(0);
    for (int s = 0; s < nstages; ++s) {
        // stage s has landed (this wave's pieces: vmcnt; everyone's: the barrier); every wave has finished reading stage s - 1
        // (lgkmcnt: the compiler may leave the last ds_reads in flight up to their MFMA), whose buffer the next DMA overwrites
        G256_STAMP(3);                                     // MFMAs issued (fragments all consumed)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        G256_STAMP(4);                                     // own DMA pieces landed
        __builtin_amdgcn_s_barrier();
        G256_STAMP(0);                                     // barrier passed
#ifdef G256_STAMPS
        acc_t[3] += (unsigned)(tt[4] - tt[3]);             // wait for the DMA
        acc_t[4] += (unsigned)(tt[0] - tt[4]);             // wait at the barrier
        if (s > 0) acc_t[2] += (unsigned)(tt[3] - tt[2]);  // MFMA phase of the previous stage
#endif
        if (s + 1 < nstages) issue(s + 1);
        G256_STAMP(1);                                     // DMA issued
        const unsigned char* buf = smem_b + (s & 1) * GSTAGE;
        // fragments of k-step ks + 1 are read while the 2 IC MFMAs of k-step ks run (two register sets)
        bf16x8 a[2][IC], b[2][2];
        auto frags = [&](int ks, bf16x8 (&af)[IC], bf16x8 (&bf)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(buf + b_off + (ks * 2 * 128 + j * 32) * 16);
#pragma unroll
            for (int i = 0; i < IC; ++i) af[i] = *reinterpret_cast<const bf16x8*>(buf + a_off[ks] + i * (32 * 128));
        };
        frags(0, a[0], b[0]);
#ifdef G256_STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        G256_STAMP(2);                                     // first fragments in registers
        acc_t[0] += (unsigned)(tt[1] - tt[0]);
        acc_t[1] += (unsigned)(tt[2] - tt[1]);
#endif
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);            // keep the reads ahead of the MFMAs (the scheduler sinks them to their use)
#pragma unroll
            for (int i = 0; i < IC; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#ifdef G256_STAMPS
    G256_STAMP(3);
    acc_t[2] += (unsigned)(tt[3] - tt[2]);
    if (p.pro_beta && lane == 0) {                         // (the GEMM has no prologue: the pointer carries the stamp buffer)
        unsigned* o = reinterpret_cast<unsigned*>(const_cast<float*>(p.pro_beta)) + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = acc_t[i];
        o[5] = (unsigned)(tt[3] - tt[5]);                  // kernel entry -> end of the main loop
    }
#endif
    const bool full = m_tile0 + TROWS <= p.M;
    const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
    if (O16) {
        const bool gbwd = p.epilogue == VF_EPI_GELU_BWD;
        const bool dual16 = p.epilogue == VF_EPI_GELU_DUAL;
        const bool deriv = p.reserved0 & 8;                // (round 6) DUAL: `out` receives gelu'(u); GELU_BWD: `res` holds it
        if (dual16 && deriv) {
            if (full) g256_store_bf16<5, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else g256_store_bf16<5, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
        } else if (gbwd && deriv) {
            if (full) g256_store_bf16<4, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else g256_store_bf16<4, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
        } else if (dual16) {
            if (full) g256_store_bf16<3, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else g256_store_bf16<3, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
        } else if (full) {
            if (gbwd) g256_store_bf16<2, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else if (gelu) g256_store_bf16<1, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else g256_store_bf16<0, true, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
        } else {
            if (gbwd) g256_store_bf16<2, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else if (gelu) g256_store_bf16<1, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
            else g256_store_bf16<0, false, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31);
        }
    } else {
        g256_store_f32<DROP, IC>(p, acc, m_tile0, n_tile0, wrow0, wave_n, half, l31, full, gelu);
    }
#ifdef G256_STAMPS
    {   // epilogue: [6] = bias / convert / store instructions issued, [7] = the stores drained (vmcnt 0)
        unsigned long long te0, te1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te0) :: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te1) :: "memory");
        if (p.pro_beta && lane == 0) {
            unsigned* o = reinterpret_cast<unsigned*>(const_cast<float*>(p.pro_beta)) + ((size_t)blockIdx.x * 8 + wave) * 8;
            o[6] = (unsigned)(te0 - tt[3]);
            o[7] = (unsigned)(te1 - te0);
        }
    }
#endif
}

// column panels of sn column tiles over mt row tiles (the round-2 order): logical id -> (row tile, column tile).  The ~32 workgroups an XCD
// runs at a time are then ~8 row blocks x sn column tiles, whose weight panel (sn x 393 KB at K = 768) stays in the XCD's 4 MB L2 for the
// whole pass over the rows — with all nb column tiles in flight (12 for c_fc: 4.7 MB of weights) the weights thrash it (L2 hit rate 69 %)
__device__ __forceinline__ void g256_panel_order(unsigned lbid, int mt, int nb, int& mtile, int& nblk) {
    constexpr int PANEL = G256_PANEL;
    const int sn = PANEL <= 0 || nb <= PANEL ? nb : nb % PANEL == 0 ? PANEL : (PANEL >= 3 && nb % 3 == 0) ? 3 : (PANEL >= 2 && nb % 2 == 0) ? 2 : 1;
    const int per_panel = mt * sn;
    const int panel = (int)(lbid / (unsigned)per_panel);
    const int in_panel = (int)(lbid - (unsigned)panel * per_panel);
    nblk = panel * sn + in_panel % sn;
    mtile = in_panel / sn;
}

// TAIL tiles (round 6).  One workgroup per CU (128 KB of LDS): a launch of T tiles costs ceil(T / 256) tile times, and the
// transformer's shapes sit just past a half round (training, M = 19 200: c_fc 900 tiles = 3.52 rounds, c_attn 675 = 2.64; inference,
// M = 57 344: c_fc 2 688 = 10.5, c_proj 672 = 2.625).  With f_rows > 0 the launch is F = f_rows x nb FULL tiles — whole rounds of them —
// followed by H = h_rows x nb tail tiles of 2 tail_ic x 32 rows that cover the remaining rows in ONE shorter round (a 192-row tile costs
// ~3/4 of a full one: 3 of 4 row blocks per wave, 56 of 64 KB per stage).  Each XCD runs its full tiles first, then its tail tiles (ids are
// XCD-contiguous within each kind: a kind's panel order and L2 reuse are those of a plain launch).  f_rows = 0: the plain grid.
// TAIL = false: the plain grid and nothing else in the kernel (the tail form's three tile bodies in ONE kernel cost the plain path 66 spilled
// SGPRs and ~1 % — measured against round 5's kernel, profiles/r6_gemm_tail_ab.txt — so the two grids are two instantiations)
template <bool O16, bool DROP = false, bool TAIL = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_g256_kernel(vf_igemm_args p, int f_rows, int h_rows, int tail_ic) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][GSTAGE]: A image, then W image
    const int tid = threadIdx.x;
    const int nb = p.Cout / GN;
    int mtile, nblk;
    if constexpr (!TAIL) {
        // XCD-contiguous logical workgroup id (vf_common.h), walked in column panels
        g256_panel_order(vf_xcd_bid(), (p.M + GM - 1) / GM, nb, mtile, nblk);
        g256_tile<O16, DROP, 4>(p, smem_b, mtile * GM, nblk, tid);
    } else {
        const unsigned F = (unsigned)(f_rows * nb), tot = gridDim.x;
        const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        const unsigned nF = F / 8u + (x < (F & 7u) ? 1u : 0u);                          // this XCD's full tiles, then its tail tiles
        const unsigned baseF = x * (F / 8u) + (x < (F & 7u) ? x : (F & 7u));
        if (i < nF) {
            g256_panel_order(baseF + i, f_rows, nb, mtile, nblk);
            g256_tile<O16, DROP, 4>(p, smem_b, mtile * GM, nblk, tid);
            return;
        }
        const unsigned baseT = x * (tot / 8u) + (x < (tot & 7u) ? x : (tot & 7u));
        g256_panel_order(baseT - baseF + (i - nF), h_rows, nb, mtile, nblk);
        const int m_tile0 = f_rows * GM + mtile * (tail_ic * 64);
        if (tail_ic == 3) g256_tile<O16, DROP, 3>(p, smem_b, m_tile0, nblk, tid);
        else g256_tile<O16, DROP, 2>(p, smem_b, m_tile0, nblk, tid);
    }
}

}  // namespace

// The tail policy of a launch (host; also what tests ask): full row tiles per column and tail tiles per column / their height.
// Applied when the launch is more than one round of the 256 CUs, its last round is between a few tiles and 3/4 full, and the rows left over after
// whole rounds of full tiles fit ONE round of 192-row (or 128-row) tail tiles.
struct G256Tail { int f_rows, h_rows, ic; };
static G256Tail g256_tail_policy(int M, int nb) {
    const int mt = (M + GM - 1) / GM;
    const long long ntiles = (long long)mt * nb;
    constexpr int CUS = 256;
    G256Tail none{0, 0, 0};
    if (ntiles <= CUS || !vf_selected(VF_SEL_GEMM_TAIL)) return none;
    const int last = (int)(ntiles % CUS);
    if (last == 0 || last > (CUS * 3) / 4 - 8 || last < 16) return none;       // (nothing to gain past ~3/4 of a round; a handful of tiles: leave it)
    const int f_rows = (int)((ntiles / CUS) * CUS / nb);                       // whole rounds of full tiles (F = f_rows x nb <= rounds x 256)
    if (f_rows <= 0 || f_rows >= mt) return none;
    const int rem_blocks = (M - f_rows * GM + 31) / 32;                        // 32-row blocks left per column
    for (int ic = 2; ic <= 3; ++ic) {                                          // the shortest tail tile that fits one round
        const int h_rows = (rem_blocks + 2 * ic - 1) / (2 * ic);
        if ((long long)h_rows * nb <= CUS) return G256Tail{f_rows, h_rows, ic};
    }
    return none;
}

// Launcher used by vf_gemm_bf16 (gemm_bf16.hip).  Returns VF_ERR_UNSUPPORTED when the shape does not qualify (the caller then takes the
// 128 x 128 kernel): bf16 A, Cout % 256 == 0, Cin % 64 == 0, no batch, at least one full row tile.
int vf_gemm_bf16_g256_launch(const vf_igemm_args& a, hipStream_t stream) {
    const bool a16 = a.reserved0 & 1, o16 = a.reserved0 & 2;
    if (!a16 || a.batch > 1 || a.Cout % GN != 0 || a.Cin % GK != 0 || a.M < GM || (a.lda & 7)) return VF_ERR_UNSUPPORTED;
    if (o16 && ((a.res && a.epilogue != VF_EPI_GELU_BWD) || (a.ldc & 1))) return VF_ERR_UNSUPPORTED;      // (GELU_BWD: `res` carries the pre-activation)
    if (a.epilogue == VF_EPI_GELU_BWD && !(o16 && a.res)) return VF_ERR_UNSUPPORTED;
    if (a.epilogue == VF_EPI_GELU_DUAL && (a.res || !a.out_aux || (a.ldc & 1))) return VF_ERR_UNSUPPORTED;      // (out fp32 or bf16)
    if ((a.reserved0 & 4) && a.epilogue != VF_EPI_GELU_BWD) return VF_ERR_BAD_ARG;                             // (bit 2: a bf16 u for GELU_BWD)
    if (a.reserved0 & 8) {                                                                                     // (bit 3, round 6: the saved-derivative forms)
        if (a.epilogue == VF_EPI_GELU_BWD ? !(a.reserved0 & 4) : !(a.epilogue == VF_EPI_GELU_DUAL && o16)) return VF_ERR_BAD_ARG;
    }
    if (a.res && a.epilogue == VF_EPI_GELU_ERF) return VF_ERR_UNSUPPORTED;      // (no layer has both; the 128-tile kernel contracts gelu * + res)
    // fused output dropout (drop_rate > 0): the fp32-output path with no epilogue function; mask group indices are 32-bit here
    if (a.drop_rate != 0.f && (!(a.drop_rate > 0.f && a.drop_rate < 1.f) || o16 || a.epilogue != VF_EPI_NONE || a.drop_row0 < 0 || (a.drop_row0 & 3) ||
                               (((unsigned long long)a.M + (unsigned long long)a.drop_row0 + 3) / 4) * (unsigned long long)a.Cout >= (1ull << 32)))
        return VF_ERR_UNSUPPORTED;
    if ((size_t)a.M * a.lda * 2 >= (1ull << 31) || (size_t)a.Cin * a.Cout * 2 >= (1ull << 31)) return VF_ERR_UNSUPPORTED;   // 32-bit buffer offsets
    const int mt = (a.M + GM - 1) / GM, nb = a.Cout / GN;
    const G256Tail t = g256_tail_policy(a.M, nb);
    const dim3 g((unsigned)(t.f_rows ? (t.f_rows + t.h_rows) * nb : mt * nb));
    auto launch = [&](auto kernel, unsigned long long* devs) -> int {
        if (vf_attr_needed(devs)) {                // (per instantiation and device: > 64 KB of dynamic LDS)
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(devs);
        }
        hipLaunchKernelGGL(kernel, g, dim3(512), (size_t)2 * GSTAGE, stream, a, t.f_rows, t.h_rows, t.ic);
        return vf_last_status();
    };
    static unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0;      // bit d: raised on device d
    const bool drop = a.drop_rate > 0.f;
    if (t.f_rows) {
        if (o16) return launch(gemm_bf16_g256_kernel<true, false, true>, &d3);
        return drop ? launch(gemm_bf16_g256_kernel<false, true, true>, &d5) : launch(gemm_bf16_g256_kernel<false, false, true>, &d4);
    }
    if (o16) return launch(gemm_bf16_g256_kernel<true, false, false>, &d0);
    return drop ? launch(gemm_bf16_g256_kernel<false, true, false>, &d2) : launch(gemm_bf16_g256_kernel<false, false, false>, &d1);
}
