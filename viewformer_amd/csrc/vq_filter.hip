// Codebook L2-argmin as a 16-bit candidate filter + exact fp32 re-rank, gfx950.
//
// Replaces QuantizeEMA.forward (eval branch), viewformer/models/utils_th.py:32-44 — the same contract as vq_argmin.hip
// (dist = (zz - 2 z.E) + ee in fp32, first arg-min) and THE SAME RESULT BIT FOR BIT, at a fraction of the matrix time:
//
//   1. filter   s~[m][k] = sum_d f16(z[m][d]) * f16(E[d][k]) - ee[k]/2   on v_mfma_f32_32x32x16_f16 (16x the f32 MFMA rate).
//               Minimising dist == maximising s = z.e_k - ee_k/2.  A wave keeps its 32 rows of z as 16 fp16 A fragments in
//               registers for the whole kernel and streams the codebook, pre-packed fragment-major fp16, through a 3-deep LDS
//               ring filled by LDS-DMA (global_load_lds_dwordx4, counted vmcnt across a raw s_barrier).  The accumulator is
//               initialised with -ee[k]/2, so the epilogue per element is: splice the 10-bit code into the low mantissa bits
//               of the score (one v_and_or_b32) and update a per-lane top-3 of such keys (v_med3, v_med3, v_max) — 4 VALU.
//   2. bound    |s~ - s| <= eps(row), PROVEN below from the fp16 rounding of both operands (Cauchy-Schwarz over the row),
//               fp32 accumulation, the key's 10 spliced bits and the fp32 rounding of the exact evaluation itself.  The exact
//               arg-min k* therefore has s~[k*] >= max_k s~[k] - 2 eps: every code inside that window is a candidate.
//   3. re-rank  candidates (1.2 per row on encoder outputs) are re-evaluated EXACTLY: a scalar fmaf chain in the k-order of
//               vq_argmin.hip's v_mfma_f32_32x32x2_f32 pipeline (the MFMA is bitwise a k-ordered fmaf chain), zz in that
//               kernel's summation tree, dist with the reference's association; min distance, ties -> lowest index.  A row
//               with a single candidate needs no re-rank at all (nothing else can win).  A lane whose third-best key is still
//               inside the window may have dropped a candidate: all 32 codes of that lane's column join the re-rank (2 rows in
//               57 344 on random inputs).  Rows with |z| beyond fp16's range or non-finite values are scanned exactly over the
//               whole codebook.
//
// eps(row), in units of s (half a distance): with u = 2^-11 (fp16 unit roundoff), N = |z| |e_k| >= sum_d |z_d e_dk|:
//   operand rounding   |z e - f16(z) f16(e)| <= (2u + u^2) |z||e| + 2^-25 (|z| + |e|)(1 + u) + 2^-50 per term (2^-25: fp16 subnormal
//                      spacing / 2), summed: (2^-10 + 2^-22) N + 2^-25 (1 + u) 16 (|z| + |e_k|) + 2^-42
//   fp32 accumulation  of 256 exact products + the init term, any order:  gamma_257 (N + ee/2) <= 2^-15.9 (N + ee/2)
//   key splice         10 low mantissa bits replaced: <= 2^-13 |s~| <= 2^-13 (N + ee/2)(1 + 2^-9)
//   exact evaluation   dot chain gamma_256 N <= 2^-16 N, two fp32 additions 2^-24 (zz + 2N + |dist|)/... (in s units: / 2)
// => eps <= N (2^-10 + 2^-13 + 2^-15 + 2^-23) + ee_max (2^-14 + 2^-16) + 2^-24 zz + 5e-7 (|z| + e_max) + 1e-12,
// evaluated with |z| from an fp32 sum of squares inflated by 1e-5, e_max / ee_max over the codebook, and inflated by 1 % again.
#include "vf_common.h"
#include "../../include/vf_hip.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int D = 256;                 // embed_dim the kernel is built for (16 k-steps of the 32x32x16 MFMA)
constexpr int KS = D / 16;
constexpr int TN = 32;                 // codes per tile
constexpr int TILE_BYTES = KS * 64 * 16;      // 16 KiB: [k-step][lane = half*32 + n][8 f16]
constexpr int BM = 128;                // rows per workgroup (4 waves x 32)
#ifndef VQF_NT
#define VQF_NT 2                       // code tiles per step: independent accumulator chains per wave (one dependent chain leaves the
#endif                                 // matrix pipe idle for the MFMA's result latency)
#ifndef VQF_RING
#define VQF_RING 2                     // LDS ring depth in steps (2: the next step lands while this one is multiplied)
#endif
constexpr int NT = VQF_NT;
constexpr int RING = VQF_RING;
constexpr int STEP_BYTES = NT * TILE_BYTES;
constexpr int PAIR_CAP = 192;          // (row, code) pairs per wave queued for the exact re-rank
constexpr int MAX_KC = 1024;           // 10 code bits in the key

struct Blob {                          // device blob written by vf_vq_filter_pack
    const unsigned char* tiles;        // [Kc/32][16 KiB] fp16 fragment-major
    const float* Et;                   // [Kc][256] fp32, row = one code
    const float* ee;                   // [Kc] = vf_colsumsq_f32 (the exact kernel's e_sq)
    const float* hee;                  // [Kc] = -ee/2
    const float* consts;               // [0] = max_k |e_k| (inflated), [1] = max_k ee_k, [2] = +inf
};

__host__ __device__ inline size_t blob_tiles_bytes(int Kc) { return (size_t)(Kc / TN) * TILE_BYTES; }
__host__ __device__ inline size_t blob_bytes(int Kc) {
    return blob_tiles_bytes(Kc) + (size_t)Kc * D * 4 + (size_t)Kc * 4 * 2 + 64;
}
__host__ __device__ inline Blob blob_view(const void* p, int Kc) {
    const unsigned char* b = reinterpret_cast<const unsigned char*>(p);
    Blob v;
    v.tiles = b;
    v.Et = reinterpret_cast<const float*>(b + blob_tiles_bytes(Kc));
    v.ee = v.Et + (size_t)Kc * D;
    v.hee = v.ee + Kc;
    v.consts = v.hee + Kc;
    return v;
}

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// sum of squares of one float4 with a FIXED association (explicit fma: no contraction freedom); shared with vq_argmin.hip's
// definition of zz through vf_vq_sq4 in vf_common.h
// exact distance of (row, code) in vq_argmin.hip's arithmetic: dot = fmaf chain in the MFMA's k order, zz in its summation tree
__device__ __forceinline__ float exact_dist(const float* __restrict__ zr, const float* __restrict__ er, float ee) {
    float acc = 0.f;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int c = 0; c < D / 32; ++c) {
        f32x4 zv[8], ev[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            zv[j] = *reinterpret_cast<const f32x4*>(zr + c * 32 + j * 4);
            ev[j] = *reinterpret_cast<const f32x4*>(er + c * 32 + j * 4);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = __builtin_fmaf(zv[2 * g][e], ev[2 * g][e], acc);               // k = 32c + 8g + e       (half 0 of the 32x32x2 MFMA)
                acc = __builtin_fmaf(zv[2 * g + 1][e], ev[2 * g + 1][e], acc);       // k = 32c + 8g + 4 + e   (half 1)
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] += vf_vq_sq4(zv[j]);
    }
    const float zz = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    const float t = __builtin_fmaf(-2.0f, acc, zz);                                  // zz - 2*dot (2*dot is exact)
    return t + ee;
}

__device__ __forceinline__ unsigned long long dist_key(float d, unsigned code);
// exact arg-min of one row over the whole codebook by one wave (rows the filter cannot bound: rare, kept out of line)
__device__ __attribute__((noinline)) int scan_row(const float* __restrict__ zr, const float* __restrict__ Et, const float* __restrict__ ee,
                                                  int Kc, int lane);

__device__ __forceinline__ unsigned long long dist_key(float d, unsigned code) {
    unsigned u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);                                 // order-preserving float -> uint
    return ((unsigned long long)u << 32) | code;                                    // min: smallest distance, then lowest index
}

__device__ __attribute__((noinline)) int scan_row(const float* __restrict__ zr, const float* __restrict__ Et, const float* __restrict__ ee,
                                                  int Kc, int lane) {
    unsigned long long best = ~0ull;
    for (int c = lane; c < Kc; c += 64) {
        const float d = exact_dist(zr, Et + (size_t)c * D, ee[c]);
        const unsigned long long key = d == d ? dist_key(d, (unsigned)c) : ~0ull;       // NaN distances never win
        best = key < best ? key : best;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long ob = __shfl_xor(best, o, 64);
        best = ob < best ? ob : best;
    }
    return best == ~0ull ? 0 : (int)(best & 0xFFFFFFFFull);
}

// a lane whose THIRD key is inside the window may have dropped a candidate: every code of its column (code = 32 t + lane) joins the
// re-rank (2 rows in 57 344 on random inputs: out of line)
__device__ __attribute__((noinline)) void append_column(unsigned* pairs, unsigned* cnt, unsigned rl, unsigned col, unsigned ntiles) {
    for (unsigned tt = 0; tt < ntiles; ++tt) {
        const unsigned pos = atomicAdd(cnt, 1u);
        if (pos < (unsigned)PAIR_CAP) pairs[pos] = (rl << 16) | (tt * 32u + col);
    }
}

// LDS map of one workgroup (2 workgroups per CU: <= 80 KiB each).  R0 is reused by the three phases that never overlap:
// staging of z (4 waves x 32 rows x 528 B), the codebook ring (RING x STEP_BYTES), the re-rank's row staging (4 x 16.5 KiB).
constexpr int A_LDB = 528;                                   // bytes per staged fp16 row (512 + 16: conflict-free ds_read_b128)
constexpr int WAVE_R0 = 32 * A_LDB;                          // 16 896 B per wave
constexpr int R0_BYTES = (4 * WAVE_R0 > RING * STEP_BYTES) ? 4 * WAVE_R0 : RING * STEP_BYTES;
constexpr int PAIR_STRIDE = 2064;                            // staged (z row, code row) of one re-rank pair: 2 x 1024 B + 16 (bank spread)
constexpr int PAIRS_PER_PASS = 8;                             // 16 512 B of the wave's R0 slice
constexpr int OFF_HEE = R0_BYTES;                            // float [MAX_KC]
constexpr int OFF_ROWMAX = OFF_HEE + MAX_KC * 4;             // u32 [BM]   order-preserving image of the row's best key
constexpr int OFF_ROWCNT = OFF_ROWMAX + BM * 4;              // u32 [BM]   candidates in the window (+64 per lane that may have dropped one)
constexpr int OFF_ROWIDX = OFF_ROWCNT + BM * 4;              // i32 [BM]   result
constexpr int OFF_EPS = OFF_ROWIDX + BM * 4;                 // f32 [BM]   window half-width; < 0 marks a row for the exact scan
constexpr int OFF_ROWBEST = OFF_EPS + BM * 4;                // u64 [BM]
constexpr int OFF_PAIRS = OFF_ROWBEST + BM * 8;              // u32 [4][PAIR_CAP]
constexpr int OFF_WCNT = OFF_PAIRS + 4 * PAIR_CAP * 4;       // u32 [4][4]
constexpr int OFF_SCAN = OFF_WCNT + 64;                      // u32 [4][32]
constexpr int SMEM_BYTES = OFF_SCAN + 4 * 32 * 4;

__global__ __launch_bounds__(256, 2) void vq_filter_kernel(const float* __restrict__ z, const void* __restrict__ blob_p, long long M,
                                                           int Kc, long long* __restrict__ idx_out, unsigned* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* hee_s = reinterpret_cast<float*>(smem + OFF_HEE);
    int* rowidx = reinterpret_cast<int*>(smem + OFF_ROWIDX);
    float* eps_s = reinterpret_cast<float*>(smem + OFF_EPS);
    unsigned long long* rowbest = reinterpret_cast<unsigned long long*>(smem + OFF_ROWBEST);
    unsigned* pairs = reinterpret_cast<unsigned*>(smem + OFF_PAIRS);
    unsigned* wcnt = reinterpret_cast<unsigned*>(smem + OFF_WCNT);
    unsigned* scanrows = reinterpret_cast<unsigned*>(smem + OFF_SCAN);

#ifdef VQF_STAMPS
    unsigned long long stamp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define VQF_STAMP(i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i]) :: "memory")
#else
#define VQF_STAMP(i)
#endif
    VQF_STAMP(0);
#ifdef VQF_X_STAGGER      // experiment (round 4, tools/variants.sh): every second workgroup starts VQF_X_STAGGER cycles late, so that the two
    if (blockIdx.x & 1) {  // workgroups of a CU are in different phases (z read / matrix loop / classification / re-rank) at any time
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        do {
            __builtin_amdgcn_s_sleep(32);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        } while (t1 - t0 < (unsigned long long)(VQF_X_STAGGER));
    }
#endif
    const Blob B = blob_view(blob_p, Kc);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const long long m0 = (long long)blockIdx.x * BM + wave * 32;
    const int ntiles = Kc / TN;
    const int wr = wave * 32;                                                       // this wave's first row in the per-row LDS arrays

    // ---- z -> fp16 A fragments.  Coalesced: one wave-instruction reads one whole row (64 lanes x 16 B); the rows pass through LDS
    // as fp16 and come back fragment-shaped (row l31, k = 16 ks + 8 half + [0, 8)) for the whole kernel.
    unsigned char* my_r0 = smem + wave * WAVE_R0;
    {
        f32x4 v[32];                                                                // all 32 rows in flight: one HBM round trip
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            long long row = m0 + i;
            row = row < M ? row : M - 1;
            v[i] = *reinterpret_cast<const f32x4*>(z + (size_t)row * D + lane * 4);
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (_Float16)v[i][e];
            *reinterpret_cast<f16x4*>(my_r0 + i * A_LDB + lane * 8) = h;
        }
    }
    for (int i = tid; i < MAX_KC; i += 256) hee_s[i] = i < Kc ? B.hee[i] : -3.0e38f;
    if (tid < 16) wcnt[tid] = 0;
    if (tid < BM) { rowbest[tid] = ~0ull; rowidx[tid] = 0; }
    const float e_max = B.consts[0], ee_max = B.consts[1], pinf = B.consts[2];
    __syncthreads();
    f16x8 a[KS];
    float ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        a[ks] = *reinterpret_cast<const f16x8*>(my_r0 + l31 * A_LDB + ks * 32 + half * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float x = (float)a[ks][e]; ss = __builtin_fmaf(x, x, ss); }
    }
    ss += __shfl_xor(ss, 32, 64);
    // window half-width eps of this lane's row (header).  ss is the sum of squares of the fp16-ROUNDED row: |z| <= |f16(z)| (1 + 2^-11)
    // + 16 * 2^-25; a value beyond fp16's range rounds to inf and a NaN stays one, so !(ss < 3e38) catches every row the filter cannot
    // bound: those are scanned exactly (eps < 0 marks them).
    {
        const float zn = sqrtf(ss) * 1.0005f + 5.0e-7f;
        const float zz_up = zn * zn;
        float eps_row = (zn * e_max) * 1.1300e-3f + ee_max * 7.7e-5f + zz_up * 6.0e-8f + (zn + e_max) * 5.0e-7f + 1e-12f;
        eps_row *= 1.01f;
        if (!(ss < 3.0e38f) || !(e_max < 6.5e4f)) eps_row = -1.0f;      // (e_max: see consts_kernel — a codebook outside fp16's range)
        if (half == 0) eps_s[wr + l31] = eps_row;
    }
    __syncthreads();                                                                // every wave is done with R0: the ring may start
    VQF_STAMP(1);

    // ---- codebook ring: a step = NT tiles of 32 codes; each wave moves a quarter (4 x 1 KiB) of every 16 KiB tile
    const int nsteps = ntiles / NT;                 // (Kc % (32 NT) == 0 is checked by the launcher)
    auto issue_step = [&](int st) {
        const unsigned char* src = B.tiles + (size_t)st * STEP_BYTES + wave * 4096 + lane * 16;
        unsigned char* dst = ring + (st % RING) * STEP_BYTES + wave * 4096;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16(src + j * TILE_BYTES + q * 1024, dst + j * TILE_BYTES + q * 1024);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 // nothing of this wave in flight: the ring owns vmcnt now
    issue_step(0);
    if (RING > 2 && nsteps > 1) issue_step(1);

    float k1[16], k2[16], k3[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) k1[r] = k2[r] = k3[r] = -INFINITY;

    // software pipeline: the MFMAs of step t run beside the key updates (VALU) of step t-1, which read the previous accumulators
    auto update_keys = [&](const f32x16& acc, unsigned code) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float k = __uint_as_float((__float_as_uint(acc[r]) & 0xFFFFFC00u) | code);
            // v_med3_f32 throughout — med3(k, k1, +inf) = max(k, k1) with +inf read from memory: with a literal the compiler folds it
            // to maxnum, whose IEEE lowering adds a canonicalising v_max per operand (2 more VALU per element)
            k3[r] = __builtin_amdgcn_fmed3f(k, k2[r], k3[r]);
            k2[r] = __builtin_amdgcn_fmed3f(k, k1[r], k2[r]);
            k1[r] = __builtin_amdgcn_fmed3f(k, k1[r], pinf);
        }
    };
    auto step = [&](int t, f32x16 (&acc)[NT], const f32x16 (&prev)[NT]) {
        // lgkmcnt(0) too: the compiler leaves the last ds_reads of step t-1 in flight across the barrier (it only needs them at their
        // MFMA), and another wave's LDS-DMA for step t+RING-1 — issued right after ITS barrier — targets the buffer they read
        if (RING > 2) {
            if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * NT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();                                               // step t visible; everyone is done with step t-1
        if (t + RING - 1 < nsteps) issue_step(t + RING - 1);
        const unsigned char* bsrc = ring + (t % RING) * STEP_BYTES + lane * 16;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float c0 = hee_s[(t * NT + j) * TN + l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = c0;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const f16x8 b = *reinterpret_cast<const f16x8*>(bsrc + j * TILE_BYTES + ks * 1024);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], b, acc[j], 0, 0, 0);
            }
#pragma unroll
        for (int j = 0; j < NT; ++j) update_keys(prev[j], (unsigned)(((t > 0 ? t - 1 : 0) * NT + j) * TN + l31));
    };
    f32x16 accA[NT], accB[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[j][r] = -3.0e38f;    // "step -1": finite (a spliced -inf would be a NaN), below every real score
    for (int t = 0; t < nsteps; t += 2) {           // ping-pong: no accumulator copies; nsteps is even (launcher: Kc % (64 NT) == 0)
        step(t, accA, accB);
        step(t + 1, accB, accA);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) update_keys(accB[j], (unsigned)(((nsteps - 1) * NT + j) * TN + l31));
    VQF_STAMP(2);
    __syncthreads();                                                                // ring is free: R0 becomes the re-rank's staging area
    VQF_STAMP(6);

    // ---- candidates per row, through LDS atomics (a wave only touches its own 32 rows: wave-level ordering is enough).
    // Accumulator row r of a lane = row (r&3) + 8 (r>>2) + 4 half of the wave tile, column = l31.
    auto wave_sync = [&]() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier(); };      // lgkmcnt(0)
    unsigned* my_pairs = pairs + wave * PAIR_CAP;
    unsigned* my_scan = scanrows + wave * 32;
    // Row maxima by DPP (a 32-way same-address LDS atomic costs ~500 cycles per instruction, a ds_bpermute chain ~1000 per row —
    // measured; this phase is latency, not work): shr 1/2/4/8 inside each row of 16 lanes, row_bcast:15 into rows 1 and 3, then lanes
    // 31 / 63 hold the maxima of the two half-waves = the two rows accumulator register r covers.
#define VQF_DPP_MAX(v, ctrl, rmask)                                                                                              \
    v = __builtin_amdgcn_fmed3f(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, rmask, 0xF, \
                                                                              false)), pinf)    /* max without a canonicalising v_max */
    unsigned n_single = 0, n_amb = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;                           // row within the wave tile (uniform per half)
        float m = k1[r];
        VQF_DPP_MAX(m, 0x111, 0xF);      // row_shr:1
        VQF_DPP_MAX(m, 0x112, 0xF);      // row_shr:2
        VQF_DPP_MAX(m, 0x114, 0xF);      // row_shr:4
        VQF_DPP_MAX(m, 0x118, 0xF);      // row_shr:8
        VQF_DPP_MAX(m, 0x142, 0xA);      // row_bcast:15 -> rows 1, 3
        const float m_lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 31));
        const float m_hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
        m = half ? m_hi : m_lo;
        const float eps = eps_s[wr + rl];
        const bool scan = !(eps >= 0.f);
        const float thr = m - 2.0f * eps - fabsf(m) * 2.5e-4f;                      // (2^-12 |m|: the spliced bits of both keys compared)
        const bool c1 = k1[r] >= thr, c2 = k2[r] >= thr, c3 = k3[r] >= thr;
        const unsigned long long b1 = __ballot(c1), b2 = __ballot(c2), b3 = __ballot(c3);
        const unsigned h1 = half ? (unsigned)(b1 >> 32) : (unsigned)b1, h2 = half ? (unsigned)(b2 >> 32) : (unsigned)b2;
        const unsigned h3 = half ? (unsigned)(b3 >> 32) : (unsigned)b3;
        const int n = __popc(h1) + __popc(h2);
        auto append = [&](unsigned code) {
            const unsigned pos = atomicAdd(&wcnt[wave * 4 + 0], 1u);
            if (pos < PAIR_CAP) my_pairs[pos] = ((unsigned)rl << 16) | code;
        };
        if (scan) {                                                                 // the filter cannot bound this row: exact scan
            if (l31 == 0) { const unsigned pos = atomicAdd(&wcnt[wave * 4 + 1], 1u); my_scan[pos] = (unsigned)rl; }
        } else if (n == 1 && h3 == 0) {                                             // one code inside the window: nothing else can win
            if (c1) rowidx[wr + rl] = (int)(__float_as_uint(k1[r]) & 1023u);
            n_single += (l31 == 0);
        } else {
            if (c1 && !c3) append(__float_as_uint(k1[r]) & 1023u);
            if (c2 && !c3) append(__float_as_uint(k2[r]) & 1023u);
            if (c3) append_column(my_pairs, &wcnt[wave * 4 + 0], (unsigned)rl, (unsigned)l31, (unsigned)ntiles);
            n_amb += (l31 == 0);
        }
    }
    n_single += __shfl_xor(n_single, 32, 64);
    n_amb += __shfl_xor(n_amb, 32, 64);
    wave_sync();
    const unsigned npairs_raw = wcnt[wave * 4 + 0], nscan = wcnt[wave * 4 + 1];
    const bool pair_overflow = npairs_raw > PAIR_CAP;                               // (needs > 6 candidates per row on average)
    const unsigned npairs = pair_overflow ? 0u : npairs_raw;
    VQF_STAMP(3);

    // ---- exact re-rank of the queued (row, code) pairs.  The wave stages PAIRS_PER_PASS pairs' rows (z row, code row: coalesced
    // 1 KiB reads) into its slice of R0; lane p then runs pair p's fmaf chain out of LDS.
    for (unsigned p0 = 0; p0 < npairs; p0 += PAIRS_PER_PASS) {
        const unsigned np = min((unsigned)PAIRS_PER_PASS, npairs - p0);
        for (unsigned p = 0; p < np; ++p) {
            const unsigned e = my_pairs[p0 + p];
            long long row = m0 + (e >> 16);
            row = row < M ? row : M - 1;
            const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (size_t)row * D + lane * 4);
            const f32x4 ev = *reinterpret_cast<const f32x4*>(B.Et + (size_t)(e & 0xFFFFu) * D + lane * 4);
            *reinterpret_cast<f32x4*>(my_r0 + p * PAIR_STRIDE + lane * 16) = zv;
            *reinterpret_cast<f32x4*>(my_r0 + p * PAIR_STRIDE + 1024 + lane * 16) = ev;
        }
        wave_sync();
        if (p0 == 0) { VQF_STAMP(9); }
        if ((unsigned)lane < np) {
            const unsigned e = my_pairs[p0 + lane], rl = e >> 16, code = e & 0xFFFFu;
            const float d = exact_dist(reinterpret_cast<const float*>(my_r0 + lane * PAIR_STRIDE),
                                       reinterpret_cast<const float*>(my_r0 + lane * PAIR_STRIDE + 1024), B.ee[code]);
            atomicMin(&rowbest[wr + rl], dist_key(d, code));
        }
        wave_sync();
    }
    if (!pair_overflow && lane < 32) {
        const unsigned long long best = rowbest[wr + lane];
        if (best != ~0ull) rowidx[wr + lane] = (int)(best & 0xFFFFFFFFull);
    }
    VQF_STAMP(4);
    // ---- exact scan over every code for the rows the filter could not certify (and for all rows after a queue overflow)
    const unsigned nfull = pair_overflow ? 32u : nscan;
    for (unsigned s = 0; s < nfull; ++s) {
        const unsigned rl = pair_overflow ? s : my_scan[s];
        const long long row = m0 + rl;
        if (row >= M) continue;
        const int best = scan_row(z + (size_t)row * D, B.Et, B.ee, Kc, lane);
        if (lane == 0) rowidx[wr + rl] = best;
    }
    wave_sync();
    if (lane < 32 && m0 + lane < M) idx_out[m0 + lane] = (long long)rowidx[wr + lane];   // one coalesced 256-byte store per wave
#ifdef VQF_STAMPS
    VQF_STAMP(5);
    if (stats && tid == 0 && blockIdx.x < 1024) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(stats + 4) + (size_t)blockIdx.x * 10;
        for (int i = 0; i < 10; ++i) o[i] = stamp[i];
    }
#endif
    if (stats) {
        const unsigned s1 = n_single, s2 = n_amb;
        if (lane == 0) {
            atomicAdd(&stats[0], s1);            // rows certified by the filter alone
            atomicAdd(&stats[1], s2);            // rows re-ranked
            atomicAdd(&stats[2], npairs);        // exact distance evaluations in the re-rank
            atomicAdd(&stats[3], nfull);         // rows scanned exactly over the whole codebook
        }
    }
}

// ---- packing --------------------------------------------------------------------------------------------------------
__global__ void pack_tiles_kernel(const float* __restrict__ E, unsigned char* __restrict__ tiles, float* __restrict__ Et, int Kc) {
    // one thread per (code, k-step, half): 8 consecutive d of one code -> one 16-byte fragment; also the fp32 transpose
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kc * KS * 2) return;
    const int code = i % Kc, kh = i / Kc, ks = kh >> 1, half = kh & 1;
    f16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = E[(size_t)(ks * 16 + half * 8 + e) * Kc + code];
        v[e] = (_Float16)x;
        Et[(size_t)code * D + ks * 16 + half * 8 + e] = x;
    }
    const int t = code / TN, n = code % TN;
    *reinterpret_cast<f16x8*>(tiles + (size_t)t * TILE_BYTES + ks * 1024 + (half * 32 + n) * 16) = v;
}

__global__ void consts_kernel(const float* __restrict__ ee, float* __restrict__ hee, float* __restrict__ consts, int Kc) {
    __shared__ float red[256];
    float m = 0.f;
    for (int k = threadIdx.x; k < Kc; k += 256) {
        const float v = ee[k];
        hee[k] = -0.5f * v;
        m = fmaxf(m, (v < 3.0e38f) ? v : INFINITY);       // a NaN / inf squared norm (fmaxf would drop a NaN) poisons the maximum
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // e_max >= max_k |e_k| (ee is an fp32 sum of squares: relative error < 2^-15).  A codebook the fp16 tiles cannot carry — an
        // entry beyond fp16's range (|e_k| >= 65504 for some k implies e_max >= 65504) or a non-finite one — leaves e_max = inf here:
        // the kernel then certifies nothing and every row takes the exact scan (same indices as vq_argmin, at its speed)
        consts[0] = sqrtf(red[0]) * 1.00001f;
        consts[1] = red[0] * 1.00004f;             // ee_max
        consts[2] = INFINITY;                      // (see update_keys)
    }
}

}  // namespace

extern "C" {

size_t vf_vq_filter_packed_bytes(int Dd, int Kc) {
    if (Dd != D || Kc <= 0 || Kc % (2 * TN * NT) != 0 || Kc > MAX_KC) return 0;
    return blob_bytes(Kc);
}

int vf_vq_filter_pack(const float* E, void* dst, int Dd, int Kc, void* stream) {
    if (!E || !dst || Dd <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (Dd != D || Kc % (2 * TN * NT) != 0 || Kc > MAX_KC) return VF_ERR_UNSUPPORTED;
    const Blob B = blob_view(dst, Kc);
    hipStream_t s = (hipStream_t)stream;
    const int n = Kc * KS * 2;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, s, E, const_cast<unsigned char*>(B.tiles),
                       const_cast<float*>(B.Et), Kc);
    int rc = vf_colsumsq_f32(E, const_cast<float*>(B.ee), Dd, Kc, stream);     // the exact kernel's e_sq, bit for bit
    if (rc != VF_OK) return rc;
    hipLaunchKernelGGL(consts_kernel, dim3(1), dim3(256), 0, s, B.ee, const_cast<float*>(B.hee), const_cast<float*>(B.consts), Kc);
    return vf_last_status();
}

int vf_vq_argmin_filtered_f32(const float* z, const void* packed, int64_t M, int Dd, int Kc, int64_t* idx, uint32_t* stats4,
                              void* stream) {
    if (M == 0) return VF_OK;
    if (!z || !packed || !idx || M < 0 || Dd <= 0 || Kc <= 0) return VF_ERR_BAD_ARG;
    if (Dd != D || Kc % (2 * TN * NT) != 0 || Kc > MAX_KC) return VF_ERR_UNSUPPORTED;
#ifdef VQF_ONE_PER_CU
    const size_t smem = 100 * 1024;        // experiment: one workgroup per CU
#else
    const size_t smem = SMEM_BYTES;
#endif
    static unsigned long long attr_devs = 0;      // bit d: raised on device d (the attribute is per device)
    if (vf_attr_needed(&attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vq_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)smem);
        if (e != hipSuccess) return (int)e;
        vf_attr_done(&attr_devs);
    }
    const unsigned grid = (unsigned)((M + BM - 1) / BM);
    hipLaunchKernelGGL(vq_filter_kernel, dim3(grid), dim3(256), smem, (hipStream_t)stream, z, packed, (long long)M, Kc,
                       reinterpret_cast<long long*>(idx), stats4);
    return vf_last_status();
}

}  // extern "C"
