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

#ifndef G256_A_VIA_REGS
#define G256_A_VIA_REGS 0
#endif
#ifndef G256_PANEL
#define G256_PANEL 4          // column tiles per panel (0: all column tiles of a row block together)
#endif
#ifndef G256_ROWGROUP
#define G256_ROWGROUP 0       // row blocks per group (0: column panels over all row blocks, the round-2 order); see the kernel
#endif
constexpr int GM = 256, GN = 256, GK = 64;
constexpr int GA_BYTES = GM * GK * 2;        // 32768
constexpr int GB_BYTES = GN * GK * 2;        // 32768 = two 128-column packed blocks
constexpr int GSTAGE = GA_BYTES + GB_BYTES;  // 65536

#ifndef G256_AUX_A
#define G256_AUX_A 0        // cache-policy bits of the A pieces' DMA (2 = nt)
#endif
#ifndef G256_AUX_W
#define G256_AUX_W 0
#endif
#ifndef G256_PERSIST
#define G256_PERSIST 0       // workgroups of the persistent form (256 = one per CU).  A/B at M = 65 536: c_attn 241 vs 247 us, c_fc 391 vs 381,
                             // mlp.c_proj 312 vs 308 — a wash: the hidden first-stage flight is paid back by the lost overlap of tile tails -> off
#endif
#ifndef G256_BUFFER
#define G256_BUFFER 1       // buffer_load_dwordx4 ... lds (SGPR resource + 32-bit lane offsets) instead of global_load_lds_dwordx4 (64-bit lane
                            // addresses): the address path is part of what a piece costs — mlp.c_proj 325 -> 304 us, c_attn 261 -> 244 us
#endif
__device__ __forceinline__ void bufds16(__amdgpu_buffer_rsrc_t r, void* l, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)l, 16, voff, soff, 0, 0);
}
template <int AUX = 0>
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

// fp32 output: bias, optional GELU, optional residual.  One code path with wave-uniform flags (eight template instantiations of the
// unrolled 8-tile store made the compiler hoist every tile's addresses and spill 350 registers around the 128 accumulators); a tile is
// still handled as ONE block of 16 back-to-back loads / stores (epilogue.h).
template <bool DROP>
__device__ __forceinline__ void g256_store_f32(const vf_igemm_args& p, const f32x16 (&acc)[4][2], int m_tile0, int n_tile0, int wave_m,
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
        for (int i = 0; i < 4; ++i) {
            int m0 = m_tile0 + wave_m * 128 + i * 32 + 4 * half;
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
                    const float got = __shfl_xor(give, 1, 64);
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
template <int EPI, bool FULL>
__device__ __forceinline__ void g256_store_bf16(const vf_igemm_args& p, const f32x16 (&acc)[4][2], int m_tile0, int n_tile0, int wave_m,
                                                int wave_n, int half, int l31) {
    __bf16* __restrict__ O = reinterpret_cast<__bf16*>(p.out);
    const int odd = l31 & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n_tile0 + wave_n * 64 + j * 32 + l31;
        const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m0 = m_tile0 + wave_m * 128 + i * 32 + 4 * half;
            float t[16];
            if (EPI == 2) {
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
                        const unsigned got = (unsigned)__shfl_xor((int)give, 1, 64);
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
                        t[4 * q + e] = __fmul_rn(__fadd_rn(acc[i][j][4 * q + e], bias), vf_gelu_grad_fast(uu[4 * q + e]));
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
            // rounded one — its GELU as bf16 to `out_aux`
#pragma unroll
            for (int pass = 0; pass < (EPI == 3 ? 2 : 1); ++pass) {
                __bf16* __restrict__ dst = pass == 0 ? O : reinterpret_cast<__bf16*>(p.out_aux);
                if (pass == 1) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = vf_gelu_erf_fast(t[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float give = odd ? t[r] : t[r + 1];
                    const float got = __shfl_xor(give, 1, 64);
                    const int m = m0 + ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                    bf16x2_t v;
                    v[0] = (__bf16)(odd ? got : t[r]);
                    v[1] = (__bf16)(odd ? t[r + 1] : got);
                    if (FULL || m < p.M) *reinterpret_cast<bf16x2_t*>(dst + (size_t)m * p.ldc + (n - odd)) = v;
                }
            }
        }
    }
}

template <bool O16, bool DROP = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_g256_kernel(vf_igemm_args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][GSTAGE]: A image, then W image

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;

    const int nb = p.Cout / GN;
    const int mt = (p.M + GM - 1) / GM;
    // XCD-contiguous logical workgroup id (vf_common.h), walked in column PANELS of sn tiles: the ~32 workgroups an XCD runs at a time
    // are then ~8 row blocks x sn column tiles, whose weight panel (sn x 393 KB at K = 768) stays in the XCD's 4 MB L2 for the whole
    // pass over the rows — with all nb column tiles in flight (12 for c_fc: 4.7 MB of weights) the weights thrash it (L2 hit rate 69 %)
    const unsigned lbid = vf_xcd_bid();
    constexpr int PANEL = G256_PANEL;
    const int sn = PANEL <= 0 || nb <= PANEL ? nb : nb % PANEL == 0 ? PANEL : (PANEL >= 3 && nb % 3 == 0) ? 3 : (PANEL >= 2 && nb % 2 == 0) ? 2 : 1;
#if G256_ROWGROUP > 0
    // ROW GROUPS (round 5): with column panels walked over ALL row blocks (below), a 256-row A panel is fetched once per
    // column panel — 3 x for c_fc / c_attn — and the panels of one row block run on different XCDs at different times: FETCH_SIZE 434 MB per
    // c_fc launch for 105 MB of operands (profiles/r4_bench_mixed_pmc_traffic.txt).  Here the ids are walked in groups of G256_ROWGROUP row
    // blocks; inside a group, column panel after column panel: the ~32 workgroups an XCD runs (one per CU) are RG row blocks x sn column
    // tiles, and the group's next panel follows on the SAME XCD right behind — the group's A rows (RG x 393 KB at K = 768) are still in its L2
    constexpr int RG = G256_ROWGROUP;
    const int per_group = RG * nb;
    const int group = (int)(lbid / (unsigned)per_group);
    const int in_group = (int)(lbid - (unsigned)group * per_group);
    const int rows_g = min(RG, mt - group * RG);                     // (the last group may be short)
    const int per_panel = rows_g * sn;
    const int panel = in_group / per_panel;
    const int in_panel = in_group - panel * per_panel;
    const int nblk = panel * sn + in_panel % sn;
    const int mtile = group * RG + in_panel / sn;
#else
    const int per_panel = mt * sn;
    const int panel = (int)(lbid / (unsigned)per_panel);
    const int in_panel = (int)(lbid - (unsigned)panel * per_panel);
    const int nblk = panel * sn + in_panel % sn;
    const int mtile = in_panel / sn;
#endif
    const int m_tile0 = mtile * GM, n_tile0 = nblk * GN;
    const int nstages = p.Cin / GK;

    // ---- LDS-DMA sources.  A: wave w moves rows [32 w, 32 w + 32) of the tile, 8 rows (1 KB) per instruction; lane -> row (lane >> 3),
    // LDS chunk c' = lane & 7, global chunk c = c' ^ ((row >> 1) & 7).  W: the 32 KB of the stage are contiguous, wave w moves 4 KB.
    const unsigned char* asrc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = wave * 32 + q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int m = m_tile0 + r;
        m = m < p.M ? m : p.M - 1;
        asrc[q] = reinterpret_cast<const unsigned char*>(p.x) + ((size_t)m * p.lda) * 2 + c * 16;
    }
#if G256_BUFFER
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_packed), 0, 0x7fffffff, 0x00020000);
    unsigned avoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) avoff[q] = (unsigned)(asrc[q] - reinterpret_cast<const unsigned char*>(p.x));
    const unsigned wvoff = (unsigned)((size_t)nblk * GB_BYTES + wave * 4096 + lane * 16);
#endif
    const size_t w_stage_stride = (size_t)(p.Cout / 128) * (GK * 128 * 2);
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(p.w_packed) + (size_t)nblk * GB_BYTES + wave * 4096 + lane * 16;
#if G256_A_VIA_REGS      // experiment: the A tile through VGPRs (global_load_dwordx4 + ds_write_b128 into the same swizzled image), W by DMA
    f32x4 areg[4];
    auto a_fetch = [&](int s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) areg[q] = *reinterpret_cast<const f32x4*>(asrc[q] + (size_t)s * (GK * 2));
    };
    auto a_park = [&](int s) {
        unsigned char* dst = smem_b + (s & 1) * GSTAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(dst + (wave * 32 + q * 8) * 128 + lane * 16) = areg[q];
    };
#endif
    auto issue = [&](int s) {
        unsigned char* dst = smem_b + (s & 1) * GSTAGE;
#if G256_BUFFER
#pragma unroll
        for (int q = 0; q < 4; ++q) bufds16(a_rsrc, dst + (wave * 32 + q * 8) * 128, avoff[q], (unsigned)(s * (GK * 2)));
#pragma unroll
        for (int q = 0; q < 4; ++q) bufds16(w_rsrc, dst + GA_BYTES + wave * 4096 + q * 1024, wvoff + q * 1024, (unsigned)((size_t)s * w_stage_stride));
#else
#if !G256_A_VIA_REGS
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16<G256_AUX_A>(asrc[q] + (size_t)s * (GK * 2), dst + (wave * 32 + q * 8) * 128);
#endif
        const unsigned char* ws = wsrc + (size_t)s * w_stage_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16<G256_AUX_W>(ws + q * 1024, dst + GA_BYTES + wave * 4096 + q * 1024);
#endif
    };

    // ---- fragment addresses: A row = wave_m * 128 + i * 32 + l31, chunk (ks * 2 + half) ^ ((l31 >> 1) & 7)
    const unsigned a_row_off = (unsigned)((wave_m * 128 + l31) * 128);
    const unsigned a_swz = (unsigned)((l31 >> 1) & 7);
    unsigned a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_row_off + ((((unsigned)(ks * 2 + half)) ^ a_swz) << 4);
    // W image: [block (2)][ks][half][n (128)][16 B]; this wave's columns: block wave_n >> 1, n = (wave_n & 1) * 64 + j * 32 + l31
    const unsigned b_off = (unsigned)(GA_BYTES + (wave_n >> 1) * (GK * 128 * 2) + ((half * 128) + (wave_n & 1) * 64 + l31) * 16);

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
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
#if G256_A_VIA_REGS
    a_fetch(0);
    a_park(0);
#endif
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
#if G256_A_VIA_REGS
        if (s + 1 < nstages) a_fetch(s + 1);
#endif
        G256_STAMP(1);                                     // DMA issued
        const unsigned char* buf = smem_b + (s & 1) * GSTAGE;
        // fragments of k-step ks + 1 are read while the 8 MFMAs of k-step ks run (two register sets)
        bf16x8 a[2][4], b[2][2];
        auto frags = [&](int ks, bf16x8 (&af)[4], bf16x8 (&bf)[2]) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(buf + b_off + (ks * 2 * 128 + j * 32) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(buf + a_off[ks] + i * (32 * 128));
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
#ifdef G256_X_K32PROBE      // feasibility probe (WRONG results): the same operand traffic, each 32x32x16 MFMA replaced by two 16x16x32 MFMAs (same flops) on
            // quarter accumulators — what would the deeper-K shape buy this kernel under the package power limit?
            typedef float f32x4p __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int qd = (ks & 1) * 2 + h2;
                        f32x4p c4 = {acc[i][j][qd * 4], acc[i][j][qd * 4 + 1], acc[i][j][qd * 4 + 2], acc[i][j][qd * 4 + 3]};
                        c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks & 1][i], b[ks & 1][j], c4, 0, 0, 0);
                        acc[i][j][qd * 4] = c4[0]; acc[i][j][qd * 4 + 1] = c4[1]; acc[i][j][qd * 4 + 2] = c4[2]; acc[i][j][qd * 4 + 3] = c4[3];
                    }
#else
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#if G256_A_VIA_REGS
        if (s + 1 < nstages) a_park(s + 1);                // the idle buffer: last read in stage s - 1
#endif
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
    const bool full = m_tile0 + GM <= p.M;
    const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
    if (O16) {
        const bool gbwd = p.epilogue == VF_EPI_GELU_BWD;
        const bool dual16 = p.epilogue == VF_EPI_GELU_DUAL;
        if (dual16) {
            if (full) g256_store_bf16<3, true>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
            else g256_store_bf16<3, false>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
        } else if (full) {
            if (gbwd) g256_store_bf16<2, true>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
            else if (gelu) g256_store_bf16<1, true>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
            else g256_store_bf16<0, true>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
        } else {
            if (gbwd) g256_store_bf16<2, false>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
            else if (gelu) g256_store_bf16<1, false>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
            else g256_store_bf16<0, false>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31);
        }
    } else {
        g256_store_f32<DROP>(p, acc, m_tile0, n_tile0, wave_m, wave_n, half, l31, full, gelu);
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

#if G256_PERSIST
// ---- persistent form: one workgroup per CU walks the tiles (tile = blockIdx.x + k * gridDim.x, same XCD / panel order) and issues the
// NEXT tile's first stage before its epilogue: the DMA's flight (~3500 cycles, a whole stage-time for which a fresh workgroup sits idle)
// passes under the bias / GELU / store work of the tile that just finished.  Buffer-resource DMA only; needs an even stage count (the
// last stage then reads buffer 1 and buffer 0 is free for the prefetch).
template <bool O16>
__global__ __launch_bounds__(512, 1) void gemm_bf16_g256p_kernel(vf_igemm_args p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];   // [2][GSTAGE]: A image, then W image
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    const int nb = p.Cout / GN;
    const int mt = (p.M + GM - 1) / GM;
    constexpr int PANEL = G256_PANEL;
    const int sn = PANEL <= 0 || nb <= PANEL ? nb : nb % PANEL == 0 ? PANEL : (PANEL >= 3 && nb % 3 == 0) ? 3 : (PANEL >= 2 && nb % 2 == 0) ? 2 : 1;
    const int per_panel = mt * sn;
    const int nstages = p.Cin / GK;
    const size_t w_stage_stride = (size_t)(p.Cout / 128) * (GK * 128 * 2);
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_packed), 0, 0x7fffffff, 0x00020000);

    // tile id -> (row block, column tile): XCD-contiguous ids (vf_xcd_bid's map for an arbitrary id), then column panels
    auto tile_of = [&](unsigned t, int& m_tile0, int& n_tile0) {
        const unsigned n = (unsigned)ntiles, q = n >> 3, r = n & 7u, x = t & 7u, i = t >> 3;
        const unsigned lbid = x * q + (x < r ? x : r) + i;
        const int panel = (int)(lbid / (unsigned)per_panel);
        const int in_panel = (int)(lbid - (unsigned)panel * per_panel);
        n_tile0 = (panel * sn + in_panel % sn) * GN;
        m_tile0 = (in_panel / sn) * GM;
    };
    unsigned avoff[4], wvoff = 0;
    auto sources = [&](int m_tile0, int n_tile0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 32 + q * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int m = m_tile0 + r;
            m = m < p.M ? m : p.M - 1;
            avoff[q] = (unsigned)(((size_t)m * p.lda) * 2 + c * 16);
        }
        wvoff = (unsigned)((size_t)(n_tile0 / GN) * GB_BYTES + wave * 4096 + lane * 16);
    };
    auto issue = [&](int s) {
        unsigned char* dst = smem_b + (s & 1) * GSTAGE;
#pragma unroll
        for (int q = 0; q < 4; ++q) bufds16(a_rsrc, dst + (wave * 32 + q * 8) * 128, avoff[q], (unsigned)(s * (GK * 2)));
#pragma unroll
        for (int q = 0; q < 4; ++q) bufds16(w_rsrc, dst + GA_BYTES + wave * 4096 + q * 1024, wvoff + q * 1024, (unsigned)((size_t)s * w_stage_stride));
    };
    const unsigned a_row_off = (unsigned)((wave_m * 128 + l31) * 128);
    const unsigned a_swz = (unsigned)((l31 >> 1) & 7);
    unsigned a_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) a_off[ks] = a_row_off + ((((unsigned)(ks * 2 + half)) ^ a_swz) << 4);
    const unsigned b_off = (unsigned)(GA_BYTES + (wave_n >> 1) * (GK * 128 * 2) + ((half * 128) + (wave_n & 1) * 64 + l31) * 16);

    int m_tile0, n_tile0;
    tile_of(blockIdx.x, m_tile0, n_tile0);
    sources(m_tile0, n_tile0);
    issue(0);
    for (unsigned t = blockIdx.x; t < (unsigned)ntiles; t += gridDim.x) {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int s = 0; s < nstages; ++s) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 1 < nstages) issue(s + 1);
            const unsigned char* buf = smem_b + (s & 1) * GSTAGE;
            bf16x8 a[2][4], b[2][2];
            auto frags = [&](int ks, bf16x8 (&af)[4], bf16x8 (&bf)[2]) {
#pragma unroll
                for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(buf + b_off + (ks * 2 * 128 + j * 32) * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(buf + a_off[ks] + i * (32 * 128));
            };
            frags(0, a[0], b[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks < 3) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the next tile's first stage goes out now (buffer 0: last read in stage nstages - 2, which every wave has left)
        const int em = m_tile0, en = n_tile0;
        const unsigned tn = t + gridDim.x;
        if (tn < (unsigned)ntiles) {
            tile_of(tn, m_tile0, n_tile0);
            sources(m_tile0, n_tile0);
            issue(0);
        }
        const bool full = em + GM <= p.M;
        const bool gelu = p.epilogue == VF_EPI_GELU_ERF;
        if (O16) {
            const bool gbwd = p.epilogue == VF_EPI_GELU_BWD;
            const bool dual16 = p.epilogue == VF_EPI_GELU_DUAL;
            if (dual16) {
                if (full) g256_store_bf16<3, true>(p, acc, em, en, wave_m, wave_n, half, l31);
                else g256_store_bf16<3, false>(p, acc, em, en, wave_m, wave_n, half, l31);
            } else if (full) {
                if (gbwd) g256_store_bf16<2, true>(p, acc, em, en, wave_m, wave_n, half, l31);
                else if (gelu) g256_store_bf16<1, true>(p, acc, em, en, wave_m, wave_n, half, l31);
                else g256_store_bf16<0, true>(p, acc, em, en, wave_m, wave_n, half, l31);
            } else {
                if (gbwd) g256_store_bf16<2, false>(p, acc, em, en, wave_m, wave_n, half, l31);
                else if (gelu) g256_store_bf16<1, false>(p, acc, em, en, wave_m, wave_n, half, l31);
                else g256_store_bf16<0, false>(p, acc, em, en, wave_m, wave_n, half, l31);
            }
        } else {
            g256_store_f32<false>(p, acc, em, en, wave_m, wave_n, half, l31, full, gelu);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#endif  // G256_PERSIST

}  // namespace

// Launcher used by vf_gemm_bf16 (gemm_bf16.hip).  Returns VF_ERR_UNSUPPORTED when the shape does not qualify (the caller then takes the
// 128 x 128 kernel): bf16 A, Cout % 256 == 0, Cin % 64 == 0, no batch, at least one full row tile.
int vf_gemm_bf16_g256_launch(const vf_igemm_args& a, hipStream_t stream) {
    const bool a16 = a.reserved0 & 1, o16 = a.reserved0 & 2;
    if (!a16 || a.batch > 1 || a.Cout % GN != 0 || a.Cin % GK != 0 || a.M < GM || (a.lda & 7)) return VF_ERR_UNSUPPORTED;
    if (o16 && ((a.res && a.epilogue != VF_EPI_GELU_BWD) || (a.ldc & 1))) return VF_ERR_UNSUPPORTED;      // (GELU_BWD: `res` carries the pre-activation)
    if (a.epilogue == VF_EPI_GELU_BWD && !(o16 && a.res)) return VF_ERR_UNSUPPORTED;
    if (a.epilogue == VF_EPI_GELU_DUAL && (a.res || !a.out_aux || (a.ldc & 1))) return VF_ERR_UNSUPPORTED;      // (out fp32 or bf16)
    if ((a.reserved0 & 4) && a.epilogue != VF_EPI_GELU_BWD) return VF_ERR_BAD_ARG;                             // (bit 2: a bf16 u for GELU_BWD)
    if (a.res && a.epilogue == VF_EPI_GELU_ERF) return VF_ERR_UNSUPPORTED;
    // fused output dropout (drop_rate > 0): the fp32-output path with no epilogue function; mask group indices are 32-bit here
    if (a.drop_rate != 0.f && (!(a.drop_rate > 0.f && a.drop_rate < 1.f) || o16 || a.epilogue != VF_EPI_NONE || a.drop_row0 < 0 || (a.drop_row0 & 3) ||
                               (((unsigned long long)a.M + (unsigned long long)a.drop_row0 + 3) / 4) * (unsigned long long)a.Cout >= (1ull << 32)))
        return VF_ERR_UNSUPPORTED;
    if (G256_BUFFER && ((size_t)a.M * a.lda * 2 >= (1ull << 31) || (size_t)a.Cin * a.Cout * 2 >= (1ull << 31))) return VF_ERR_UNSUPPORTED;   // 32-bit buffer offsets     // (no layer has both; the 128-tile kernel contracts gelu * + res)
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_g256_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_g256_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_g256_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const int mt = (a.M + GM - 1) / GM, nb = a.Cout / GN;
    const dim3 g((unsigned)(mt * nb));
#if G256_PERSIST
    // persistent form (developer build -DG256_PERSIST=256): that many workgroups (one per CU; a multiple of the 8 XCDs) walk the tiles
    if (G256_BUFFER && (a.Cin / GK) % 2 == 0 && mt * nb > G256_PERSIST && a.drop_rate == 0.f) {
        static unsigned long long attr_p_devs = 0;
        if (vf_attr_needed(&attr_p_devs)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_g256p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
            if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_g256p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GSTAGE);
            if (e != hipSuccess) return (int)e;
            vf_attr_done(&attr_p_devs);
        }
        const dim3 gp((unsigned)G256_PERSIST);
        if (o16) hipLaunchKernelGGL((gemm_bf16_g256p_kernel<true>), gp, dim3(512), (size_t)2 * GSTAGE, stream, a, mt * nb);
        else hipLaunchKernelGGL((gemm_bf16_g256p_kernel<false>), gp, dim3(512), (size_t)2 * GSTAGE, stream, a, mt * nb);
        return vf_last_status();
    }
#endif
    if (o16) hipLaunchKernelGGL((gemm_bf16_g256_kernel<true>), g, dim3(512), (size_t)2 * GSTAGE, stream, a);
    else if (a.drop_rate > 0.f) hipLaunchKernelGGL((gemm_bf16_g256_kernel<false, true>), g, dim3(512), (size_t)2 * GSTAGE, stream, a);
    else hipLaunchKernelGGL((gemm_bf16_g256_kernel<false>), g, dim3(512), (size_t)2 * GSTAGE, stream, a);
    return vf_last_status();
}
