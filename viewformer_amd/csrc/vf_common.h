// Shared definitions for the gfx950 kernels behind libvf_hip.so (C-ABI in include/vf_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VF_OK 0
#define VF_ERR_BAD_ARG (-1)
#define VF_ERR_UNSUPPORTED (-2)

#define VF_WAVE 64

static inline int vf_last_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VF_OK : (int)e;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: a launcher remembers per device (bit = device
// ordinal) that it has raised its kernel's limit, so a second GPU driven from the same process gets the limit too
static inline bool vf_attr_needed(unsigned long long* mask) {
    int d = 0;
    (void)hipGetDevice(&d);
    return !((__atomic_load_n(mask, __ATOMIC_RELAXED) >> (d & 63)) & 1ull);
}
static inline void vf_attr_done(unsigned long long* mask) {
    int d = 0;
    (void)hipGetDevice(&d);
    __atomic_fetch_or(mask, 1ull << (d & 63), __ATOMIC_RELAXED);
}

__device__ __forceinline__ float vf_swish(float v) {
    // x * sigmoid(x), precise expf + IEEE division (no fast-math: fp32 parity with the oracle)
    return v / (1.0f + expf(-v));
}

__device__ __forceinline__ float vf_swish_1ulp(float v) {
    // x * sigmoid(x) to ~1.5 ulp without the library expf / IEEE-division sequences (25 -> 11 VALU per element):
    // exp(-v) = exp2(t_hi) * (1 + t_lo ln2) with t = -v log2(e) carried as hi + lo (the rounding of t would otherwise
    // cost |t| 2^-24 relative), v_exp_f32 (1 ulp); 1/(1+e) = v_rcp_f32 + one Newton step.
    const float LH = -1.4426950408889634f, LL = -1.9259629911266175e-8f;      // -log2(e) = LH + LL
    float th = v * LH;
    const float tl = __builtin_fmaf(v, LH, -th) + v * LL;
    th = fminf(th, 126.0f);                                                    // keeps 1 + e finite (v < -87: result ~ -0)
    const float e0 = __builtin_amdgcn_exp2f(th);
    const float d = 1.0f + __builtin_fmaf(e0 * tl, 0.6931471805599453f, e0);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return v * r;
}

// the same function on two values with packed fp32 arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two lanes of work per issue slot;
// v_exp_f32 / v_rcp_f32 / v_min_f32 stay scalar).  Operation for operation the scalar sequence, so the results are its results bit for bit.
typedef float vf_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vf_f32x2 vf_swish_1ulp_pk(vf_f32x2 v) {
    const vf_f32x2 LH = {-1.4426950408889634f, -1.4426950408889634f}, LL = {-1.9259629911266175e-8f, -1.9259629911266175e-8f};
    const vf_f32x2 LN2 = {0.6931471805599453f, 0.6931471805599453f}, ONE = {1.0f, 1.0f};
    vf_f32x2 th = v * LH;
    const vf_f32x2 tl = __builtin_elementwise_fma(v, LH, -th) + v * LL;
    th.x = fminf(th.x, 126.0f);
    th.y = fminf(th.y, 126.0f);
    const vf_f32x2 e0 = {__builtin_amdgcn_exp2f(th.x), __builtin_amdgcn_exp2f(th.y)};
    const vf_f32x2 d = ONE + __builtin_elementwise_fma(e0 * tl, LN2, e0);
    vf_f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    r = __builtin_elementwise_fma(__builtin_elementwise_fma(-d, r, ONE), r, r);
    return v * r;
}

__device__ __forceinline__ float vf_gelu_erf(float v) {
    // tf.nn.gelu(approximate=False): 0.5 x (1 + erf(x / sqrt 2))
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

// d/dx of the exact-erf GELU, with explicitly rounded operations (no contraction freedom): the stand-alone backward kernels
// (train_ops.hip) and the GELU-backward epilogue of vf_gemm_bf16 (VF_EPI_GELU_BWD) must produce the same bits
__device__ __forceinline__ float vf_gelu_grad(float x) {
    const float cdf = __fmul_rn(0.5f, __fadd_rn(1.0f, erff(__fmul_rn(x, 0.70710678118654752440f))));
    const float pdf = __fmul_rn(0.39894228040143267794f, expf(__fmul_rn(-0.5f, __fmul_rn(x, x))));
    return __fmaf_rn(x, pdf, cdf);
}

// The same derivative for the bf16 training arm, whose consumers round it (times the incoming gradient) to 8 mantissa bits: erf from
// Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) and the Gaussian from the SAME hardware exp2 (exp(-x^2/2) is the erf formula's
// exponential), ~18 VALU instead of ~60 for erff + expf — as an epilogue of the dX GEMM the library forms cost 60 % of the K loop.
// Explicitly rounded, so the stand-alone bf16-output kernel and the GEMM epilogue agree bit for bit.
__device__ __forceinline__ float vf_gelu_grad_fast(float x) {
    const float z = __fmul_rn(fabsf(x), 0.70710678118654752440f);
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, z, 1.0f));
    float p = __fmaf_rn(1.061405429f, t, -1.453152027f);
    p = __fmaf_rn(p, t, 1.421413741f);
    p = __fmaf_rn(p, t, -0.284496736f);
    p = __fmaf_rn(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(__fmul_rn(-1.4426950408889634f, __fmul_rn(z, z)));      // exp(-z^2) = exp(-x^2 / 2)
    const float erf_abs = __fmaf_rn(-__fmul_rn(p, t), e, 1.0f);
    const float cdf = __fmul_rn(0.5f, __fadd_rn(1.0f, copysignf(erf_abs, x)));
    return __fmaf_rn(x, __fmul_rn(0.39894228040143267794f, e), cdf);
}

__device__ __forceinline__ float vf_gelu_erf_fast(float v) {
    // erf-GELU with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute) on the hardware exp2 / rcp: ~16 VALU instead
    // of ~45 for erff.  Only for the bf16 tolerance arm, whose consumers round this value to 8 mantissa bits anyway (the erff
    // epilogue was the bound of the c_fc GEMM there: 64 outputs per thread against 192 MFMAs per wave).
    const float z = fabsf(v) * 0.70710678118654752440f;
    float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = __builtin_fmaf(-p * t, e, 1.0f);
    return 0.5f * v * (1.0f + copysignf(erf_abs, v));
}

// GELU and its derivative from ONE evaluation of the erf pieces (round 6: the c_fc epilogue can save gelu'(u) instead of u, so that the backward's
// epilogue multiplies by a loaded value instead of evaluating erf + exp per element again).  f is vf_gelu_erf_fast(v) operation for operation (same bits);
// g = Phi(v) + v phi(v) = 0.5 (1 + erf) + v (1 / sqrt(2 pi)) exp(-v^2 / 2) reuses its (1 + erf) and its exponential: three more instructions.
__device__ __forceinline__ void vf_gelu_and_grad_fast(float v, float& f, float& g) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    p = __builtin_fmaf(p, t, 1.421413741f);
    p = __builtin_fmaf(p, t, -0.284496736f);
    p = __builtin_fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erf_abs = __builtin_fmaf(-p * t, e, 1.0f);
    const float one_plus = 1.0f + copysignf(erf_abs, v);
    f = 0.5f * v * one_plus;
    g = __builtin_fmaf(v, 0.39894228040143267794f * e, 0.5f * one_plus);
}

// the value of the neighbouring lane (lane ^ 1) on the vector ALU's DPP path (quad_perm [1, 0, 3, 2]) — __shfl_xor(v, 1) is a ds_bpermute_b32: a trip
// through the LDS pipeline with its address register and lgkmcnt wait; the bf16 epilogues exchange 64 values per thread and tile this way
__device__ __forceinline__ int vf_lane_xor1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned vf_lane_xor1(unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ float vf_lane_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }

__device__ __forceinline__ float vf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// wave sum on the vector ALU's data-parallel primitives instead of six ds_bpermute round trips (each a trip through the LDS pipeline, ~100+
// cycles of latency that a one-row-per-wave kernel cannot hide): prefix adds inside each row of 16 lanes (row_shr 1 / 2 / 4 / 8), row 0 -> 1 and
// row 2 -> 3 (row_bcast:15), rows 0-1 -> 2-3 (row_bcast:31); lane 63 then holds the total, broadcast through an SGPR.  A FIXED order, different
// from vf_wave_sum's butterfly: use one or the other consistently where bits matter.
__device__ __forceinline__ float vf_wave_sum_dpp(float v) {
#define VF_DPP_ADD(ctrl, rmask) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xF, false))
    VF_DPP_ADD(0x111, 0xF);      // row_shr:1
    VF_DPP_ADD(0x112, 0xF);      // row_shr:2
    VF_DPP_ADD(0x114, 0xF);      // row_shr:4
    VF_DPP_ADD(0x118, 0xF);      // row_shr:8   -> lane 15 of every row: the row's sum
    VF_DPP_ADD(0x142, 0xA);      // row_bcast:15 -> rows 1, 3
    VF_DPP_ADD(0x143, 0xC);      // row_bcast:31 -> rows 2, 3
#undef VF_DPP_ADD
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float vf_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide max of a non-negative value into *out (bits of a non-negative float order like unsigned): ONE atomic per workgroup
// (the first version issued one per wavefront from 1024 workgroups — 4096 serialised atomics on one address, 50 us per weight
// tensor, 4 % of a training step that re-packs every weight)
__device__ __forceinline__ void vf_block_max_atomic(float m, unsigned* out) {
    __shared__ float vf_bm_red[4];
    m = vf_wave_max(m);
    if ((threadIdx.x & 63) == 0) vf_bm_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(vf_bm_red[0], vf_bm_red[1]), fmaxf(vf_bm_red[2], vf_bm_red[3]))));
}

// Counter-based dropout masks (training step): pure functions of (seed, site, element), so that the forward and backward passes recompute
// the same mask and nothing is stored; viewformer_amd/_hash.py and oracle/train_oracle.py restate them in numpy for the tests.
//
// vf_dropout_hash: the strong (3-multiply) 32-bit mix of (seed, site, 64-bit index) — host-side per-scene draws (random pose multiplier)
// and the per-plane KEY of the mask words below.
__host__ __device__ __forceinline__ uint32_t vf_dropout_hash(uint32_t seed, uint32_t site, uint64_t idx) {
    uint32_t h = seed ^ (site * 0x9E3779B9u);
    h ^= (uint32_t)idx;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h += (uint32_t)(idx >> 32) * 0xC2B2AE35u + 0x27D4EB2Fu;
    h ^= h >> 16;
    h *= 0x165667B1u;
    h ^= h >> 15;
    h *= 0xD3A2646Cu;
    h ^= h >> 16;
    return h;
}
// The masks themselves (round 4).  Elements come in GROUPS of four that share one 32-bit word:
//     word(g) = lowbias32(lo32(g) ^ key),  key = vf_dropout_hash(seed, site, hi32(g))           g = 64-bit group index
//     keep(element j of group g) = rotl32(word, 8 j) >= thresh,  thresh = floor(rate * 2^32)
// A rotation of a uniform word is uniform, so every element is kept with probability exactly 1 - thresh / 2^32; the four decisions of a
// group are decided by four different bytes of the word except in the 2^-8 boundary band (tests/test_host_logic.py checks rates, pair
// correlations and the joint distribution of a group).  One hash (two v_mul_lo_u32, 8 VALU cycles each) serves the four elements a
// lane holds: per element that is a rotate, a compare and a select.  Group index / position of an element:
//     [M][N] activations (embedding, residual, MLP sites):  g = (m >> 2) * N + n,  j = m & 3   — four consecutive ROWS of a column, which
//         is what a lane of a GEMM epilogue holds (accumulator rows (r & 3) + 8 (r >> 2) + 4 half of column lane & 31);
//     attention weights (b, h, q, k), T tokens:  g = ((b H + h) << 32) | (q * ceil(T / 4) + (k >> 2)),  j = k & 3   — four consecutive
//         KEYS of a query: what a lane of the S^T = K.Q^T accumulator holds (forward and dQ kernels; the dK / dV kernel's lanes hold four
//         queries of a key and pay one hash per element).
__host__ __device__ __forceinline__ uint32_t vf_lowbias32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t vf_dropout_key(uint32_t seed, uint32_t site, uint32_t hi) { return vf_dropout_hash(seed, site, (uint64_t)hi); }
__host__ __device__ __forceinline__ uint32_t vf_dropout_word(uint32_t key, uint32_t lo) { return vf_lowbias32(lo ^ key); }
__host__ __device__ __forceinline__ bool vf_dropout_keep(uint32_t word, int j, uint32_t thresh) {
    const uint32_t s = 8u * ((uint32_t)j & 3u);
    return ((word << s) | (word >> ((32u - s) & 31u))) >= thresh;
}
// the same decision for a lane whose position j is a RUN-TIME value (the dK / dV kernel: j = key & 3 of the lane's key): rotl(word, 8 j) as ONE
// v_alignbit_b32 by rr = vf_dropout_rotr(j), computed once per lane — hipcc does not fuse the shl / shr / or of the form above for a variable shift
__device__ __forceinline__ uint32_t vf_dropout_rotr(int j) { return (32u - 8u * ((uint32_t)j & 3u)) & 31u; }
__device__ __forceinline__ bool vf_dropout_keep_rotr(uint32_t word, uint32_t rr, uint32_t thresh) { return __builtin_amdgcn_alignbit(word, word, rr) >= thresh; }
// one element of an [M][N] activation (generic, 64-bit safe; kernels that hold a whole group hash once and call vf_dropout_keep four times)
__host__ __device__ __forceinline__ bool vf_dropout_keep_elem(uint32_t seed, uint32_t site, uint64_t m, uint32_t n, uint32_t N, uint32_t thresh) {
    const uint64_t g = (m >> 2) * (uint64_t)N + n;
    return vf_dropout_keep(vf_dropout_word(vf_dropout_key(seed, site, (uint32_t)(g >> 32)), (uint32_t)g), (int)(m & 3), thresh);
}
__host__ __device__ __forceinline__ uint32_t vf_dropout_thresh(float rate) { return (uint32_t)((double)rate * 4294967296.0); }

// XCD-aware workgroup order.  The dispatcher hands consecutive workgroup ids round-robin to the 8 XCDs, each with its own L2, so
// the workgroups that share an operand tile (the column blocks of one GEMM row tile, the output-channel blocks and halo
// neighbours of one conv tile) would land in 8 different L2s and the tile would be fetched from HBM 8 times (measured on the bf16
// GEMM 65536x768x3072: FETCH_SIZE 1.65 GB for 0.25 GB of operands).  Remap so that every XCD works on one contiguous range of
// logical ids: XCD x owns the physical ids {x + 8 i}; it gets the logical range starting at x*q + min(x, r), q = n / 8, r = n % 8.
#ifndef VF_XCD_SWIZZLE
#define VF_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ unsigned vf_xcd_bid() {
#if VF_XCD_SWIZZLE
    const unsigned n = gridDim.x, b = blockIdx.x;
    const unsigned q = n >> 3, r = n & 7u, x = b & 7u, i = b >> 3;
    return x * q + (x < r ? x : r) + i;
#else
    return blockIdx.x;
#endif
}

// sum of squares of four consecutive z values with a FIXED association and explicit fmas (no contraction freedom): the unit of the
// codebook lookup's zz = sum z^2, shared by the exact kernel (vq_argmin.hip) and the re-rank of the filtered one (vq_filter.hip) so
// that both produce the same distance bit for bit
__device__ __forceinline__ float vf_vq_sq4(const f32x4 v) {
    return __builtin_fmaf(v[0], v[0], v[1] * v[1]) + __builtin_fmaf(v[2], v[2], v[3] * v[3]);
}

// ---- build-flag registry (vf_build_flags / vf_build_flag_name, include/vf_hip.h) -------------------------------------------------
// Developer switches that change RESULTS (ablations: a phase of a kernel compiled out) or make a kernel write debug data (cycle
// stamps) live in the product sources behind -D macros.  Every translation unit built with one of them registers its name at load
// time; a product build registers nothing, vf_build_flags() returns 0, and tests/test_abi.py asserts that of the shipped .so.
extern "C" int vf_register_build_flag(const char* name);
#define VF_REG_FLAG(m) namespace { const int vf_flagreg_##m = vf_register_build_flag(#m); }
#ifdef ADMA_X_NOCOMPUTE
VF_REG_FLAG(ADMA_X_NOCOMPUTE)
#endif
#ifdef ADMA_X_NODMA
VF_REG_FLAG(ADMA_X_NODMA)
#endif
#ifdef ADMA_X_NOSM
VF_REG_FLAG(ADMA_X_NOSM)
#endif
#ifdef ADMA_STAMPS
VF_REG_FLAG(ADMA_STAMPS)
#endif
#ifdef ATT_X_NOMFMA
VF_REG_FLAG(ATT_X_NOMFMA)
#endif
#ifdef ATT_X_NOSTAGE
VF_REG_FLAG(ATT_X_NOSTAGE)
#endif
#ifdef ATT_X_NOSOFTMAX
VF_REG_FLAG(ATT_X_NOSOFTMAX)
#endif
#ifdef ATT_X_NOGLOBAL
VF_REG_FLAG(ATT_X_NOGLOBAL)
#endif
#ifdef G256_A_VIA_REGS
VF_REG_FLAG(G256_A_VIA_REGS)
#endif
#ifdef G256_STAMPS
VF_REG_FLAG(G256_STAMPS)
#endif
#if defined(G256_PERSIST) && G256_PERSIST
VF_REG_FLAG(G256_PERSIST)
#endif
#ifdef VQF_STAMPS
VF_REG_FLAG(VQF_STAMPS)
#endif
#ifdef VQF_X_NORERANK
VF_REG_FLAG(VQF_X_NORERANK)
#endif
#ifdef VQF_X_STAGGER
VF_REG_FLAG(VQF_X_STAGGER)
#endif
#if defined(VF_W2_ABL) && VF_W2_ABL
VF_REG_FLAG(VF_W2_ABL)
#endif
#if defined(VF_X3H_LDS_PAD) && VF_X3H_LDS_PAD
VF_REG_FLAG(VF_X3H_LDS_PAD)
#endif
#if defined(VF_X3H16_ABL) && VF_X3H16_ABL
VF_REG_FLAG(VF_X3H16_ABL)
#endif
#ifdef G256_X_K32PROBE
VF_REG_FLAG(G256_X_K32PROBE)
#endif
#ifdef VF_X3H_STAMPS
VF_REG_FLAG(VF_X3H_STAMPS)
#endif
#ifdef VF_X3H_X_NOPATCH
VF_REG_FLAG(VF_X3H_X_NOPATCH)
#endif
#ifdef VF_X6_CLOCKPROBE
VF_REG_FLAG(VF_X6_CLOCKPROBE)
#endif
#ifdef VF_X_DKV_HASH_PER_ELEMENT      // A/B: the dK / dV kernel's dropout words hashed by every lane (round 5's form) instead of once per lane quad
VF_REG_FLAG(VF_X_DKV_HASH_PER_ELEMENT)
#endif
#ifdef VF_X_ATB_VISLOOP   // A/B: the backward attention kernels' tile lists from loops over visible()
VF_REG_FLAG(VF_X_ATB_VISLOOP)
#endif
#ifdef VF_X_DKV_SEL2      // A/B: the dK / dV kernel's dropout with two selects per score (dP and P) instead of one
VF_REG_FLAG(VF_X_DKV_SEL2)
#endif
#ifdef VF_X_DKV_ROT3      // A/B: the dK / dV kernel's mask rotation as shl / shr / or (before the third session of round 6)
VF_REG_FLAG(VF_X_DKV_ROT3)
#endif
#if defined(ATB_ABL) && ATB_ABL      // ablation builds of the backward attention kernels (results wrong, timing only)
VF_REG_FLAG(ATB_ABL)
#endif
#if defined(ATB_KV_UNI) && !ATB_KV_UNI      // A/B: rows image + tr image per streamed operand in the dK / dV kernel
VF_REG_FLAG(ATB_KV_UNI)
#endif
#if defined(ATB_DQ_UNI) && !ATB_DQ_UNI      // A/B: K rows | V rows | K tr per slot in the dQ kernel
VF_REG_FLAG(ATB_DQ_UNI)
#endif
#if defined(ADMA_REGROUP) && !ADMA_REGROUP    // A/B: four consecutive query views per workgroup under the streams mask
VF_REG_FLAG(ADMA_REGROUP)
#endif
#if defined(ADMA_RING) && ADMA_RING != 4    // A/B: the forward DMA-ring attention with three slots
VF_REG_FLAG(ADMA_RING)
#endif
#ifdef VF_X_TRINTRIN      // A/B: the transposing LDS reads through the compiler intrinsic again (hipcc then drains vmcnt in front of them)
VF_REG_FLAG(VF_X_TRINTRIN)
#endif

// shared host helper (defined in igemm_f32.hip): pack [taps][K][N] into the fragment-major B layout
int vf_pack_b_impl(const float* src, float* dst, int K, int N, int taps, long long sk, long long sn, long long st,
                   int BN, int batch, long long src_bstride, hipStream_t stream);

// ---- heaviest-first order of an attention launch's owner blocks (round 6) -------------------------------------------------------------------
// Under the block-causal / twin / streams masks the owner blocks of a (scene, head) differ in work by up to 10 x (a query block walks 4 .. 21 key
// tiles at configs[2], a key block of the training step's stream 0 is seen by 28 query tiles, most others by 2), and the grid's last dimension is
// dispatched in index order: with the light blocks first the launch ends with a partly filled round of its HEAVIEST workgroups (configs[2]: 100
// step-units on 512 workgroup slots against 88 heaviest-first, 84 ideal; the training forward 23 vs 20, its dQ 29 vs 25).  The launchers
// therefore hand the kernels a permutation: blockIdx.z = r runs the block with the r-th largest number of visible tiles (ties: lower index first).
// Every workgroup computes what it computed before — results are bit-identical; only the dispatch order changes.
struct vf_attn_order { unsigned char blk[64]; };
static inline bool vf_attn_visible(int qv, int kv, int twin) {           // the kernels' visible(): plain / twin (Vc = twin >= 0) / streams (Sv = -twin >= 2)
    if (twin <= -2) {
        const int Sv = -twin, qs = qv / Sv, qi = qv - qs * Sv, ks = kv / Sv, ki = kv - ks * Sv;
        return qs == 0 ? (ks == 0 && ki <= qi) : ((ks == 0 && ki < qi) || kv == qv);
    }
    const int Vc = twin >= 0 ? twin : 0x3fffffff;
    return kv == qv || (kv < Vc ? kv : Vc) < (qv < Vc ? qv : Vc);
}
// nblocks owner blocks of `vpb` views each over nviews views; by_key: the owners are KEY views (weight = query tiles that see one of them),
// otherwise QUERY views (weight = key tiles one of them sees).  heaviest_first = false: the identity
static inline vf_attn_order vf_attn_block_order(int nviews, int vpb, int nblocks, int twin, bool by_key, bool heaviest_first) {
    vf_attn_order o;
    int w[64];
    if (nblocks > 64) nblocks = 64;
    for (int b = 0; b < 64; ++b) o.blk[b] = (unsigned char)b;
    if (!heaviest_first) return o;
    for (int b = 0; b < nblocks; ++b) {
        int cnt = 0;
        for (int t = 0; t < nviews; ++t) {
            bool any = false;
            for (int v = b * vpb; v < (b + 1) * vpb && v < nviews && !any; ++v) any = by_key ? vf_attn_visible(t, v, twin) : vf_attn_visible(v, t, twin);
            cnt += any;
        }
        w[b] = cnt;
    }
    for (int i = 1; i < nblocks; ++i) {                                  // stable insertion sort, descending weight
        const unsigned char bi = o.blk[i];
        int j = i - 1;
        while (j >= 0 && w[o.blk[j]] < w[bi]) { o.blk[j + 1] = o.blk[j]; --j; }
        o.blk[j + 1] = bi;
    }
    return o;
}

// Query-view GROUPS of the forward DMA-ring kernel (attention_dma.hip; third session of round 6).  A workgroup of that kernel serves up to four query
// views and walks the UNION of the key views they see, one barrier per key tile, so what a launch costs is the sum of the groups' union sizes.  Four
// CONSECUTIVE views are the right group under the causal and twin masks; under the training step's STREAMS mask (Sv views per stream, stream 0 = the
// sequence, streams >= 1 = branch views that see the sequence's views below their own index plus themselves) they are not: branch view i of every
// stream sees the same sequence views, so {stream 1 view i, stream 2 view i, stream 1 view i - 1, stream 2 view i - 1} shares i sequence tiles where four
// consecutive views of one stream share i - 2 and walk i + 2.  Grouping (regroup = true, streams mask): the sequence's views four by four; a last
// partial group is topped up with the highest branch views; the branch views by descending index (all streams of an index together) four by four.
// 3 x 10 views: 62 tile steps per (scene, head) instead of 75.  Every wave still walks ITS keys in ascending order: results are bit-identical.
// Groups are dispatched heaviest first (as the blocks of vf_attn_block_order).  view = 0xFF: no view (the wave only helps moving tiles).
struct vf_attn_groups { unsigned char view[64][4]; int n; };
static inline vf_attn_groups vf_attn_query_groups(int nviews, int twin, bool regroup, bool heaviest_first) {
    vf_attn_groups g;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) g.view[i][j] = 0xFF;
    g.n = 0;
    unsigned char list[256];
    int nl = 0;
    const int Sv = twin <= -2 ? -twin : 0;
    if (regroup && Sv > 0 && nviews % Sv == 0 && nviews / Sv >= 2 && nviews <= 255) {
        const int NS = nviews / Sv;
        for (int v = 0; v < Sv; ++v) list[nl++] = (unsigned char)v;                       // the sequence's views, ascending
        const int pad = (4 - Sv % 4) % 4;                                                 // branch views that complete its last group
        int taken = 0;
        // branch views by descending index, the streams of an index together
        unsigned char br[256];
        int nb = 0;
        for (int i = Sv - 1; i >= 0; --i)
            for (int s_ = 1; s_ < NS; ++s_) br[nb++] = (unsigned char)(s_ * Sv + i);
        for (; taken < pad && taken < nb; ++taken) list[nl++] = br[taken];
        while (nl % 4) list[nl++] = 0xFF;
        for (int i = taken; i < nb; ++i) list[nl++] = br[i];
    } else {
        for (int v = 0; v < nviews && v < 256; ++v) list[nl++] = (unsigned char)v;
    }
    while (nl % 4) list[nl++] = 0xFF;
    g.n = nl / 4 > 64 ? 64 : nl / 4;
    int w[64];
    for (int b = 0; b < g.n; ++b) {
        int cnt = 0;
        for (int t = 0; t < nviews; ++t) {
            bool any = false;
            for (int j = 0; j < 4 && !any; ++j) { const int v = list[4 * b + j]; any = v != 0xFF && vf_attn_visible(v, t, twin); }
            cnt += any;
        }
        w[b] = cnt;
    }
    int ord[64];
    for (int b = 0; b < g.n; ++b) ord[b] = b;
    if (heaviest_first)
        for (int i = 1; i < g.n; ++i) {                                                   // stable insertion sort, descending weight
            const int bi = ord[i];
            int j = i - 1;
            while (j >= 0 && w[ord[j]] < w[bi]) { ord[j + 1] = ord[j]; --j; }
            ord[j + 1] = bi;
        }
    for (int b = 0; b < g.n; ++b) for (int j = 0; j < 4; ++j) g.view[b][j] = list[4 * ord[b] + j];
    return g;
}

// ---- ds_read_b64_tr_b16 behind inline asm (round 5) ---------------------------------------------------------------------------------
// hipcc's waitcnt pass cannot prove that an LDS read issued through the transposing-read INTRINSIC (__builtin_amdgcn_ds_read_tr16_b64)
// is independent of a pending LDS-DMA (buffer_load ... lds): in front of the first such read after a DMA issue it inserts
// `s_waitcnt vmcnt(0)`.  Inside a DMA ring that turns "three tiles in flight behind counted waits" into "wait for the tile issued at the
// top of THIS step before the step's P.V / dK / dW products" — every tile step then costs at least one HBM latency, which is what the
// round-4 stamps showed as a constant ~2 200 cycles per visible tile step (tools/isa_vmcnt_audit.py lists such waits; plain ds_read_b128
// reads are not affected).  The kernels order these reads themselves (counted vmcnt + s_barrier at the top of every ring step), so the
// reads are issued as inline asm, which the pass does not look into: the asm block carries its own `s_waitcnt lgkmcnt(0)`, i.e. its
// outputs are valid when it ends (it also waits for any LDS read the compiler still has in flight: harmless).  Same instructions, same
// data, same arithmetic: bit-identical results.  `off*` must fold to immediates (unrolled loop indices do under -O3).
// INVARIANTS the callers keep, because the compiler no longer orders these reads against the ring:
//   * a ring slot is read only behind the issuing waves' COUNTED `s_waitcnt vmcnt(N)` for it plus the `s_barrier` at the top of the ring step; N counts
//     LDS-DMA pieces only — no other vector-memory operation may be added inside a ring loop without re-deriving every N of that loop
//     (tools/isa_vmcnt_audit.py lists the loop's VMEM instructions and waits; tests/test_isa_audit.py runs it);
//   * offsets are DS immediates: offA / offB (+ gap) must stay below 65 536 — the assembler REJECTS a larger `offset:` (hipcc fails the build, probed in
//     round 6), so the bound needs no static_assert of its own; today's maximum is ~32 KB;
//   * results are compared with a build that issues the same reads through the intrinsic (-DVF_X_TRINTRIN: compiler-ordered, slower), output for output,
//     under memory noise: tests/test_hip_ring_stress.py (viewformer_amd.build.build_variant('trintrin')).
typedef __bf16 vf_bf16x8 __attribute__((ext_vector_type(8)));
typedef short vf_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned vf_lds_addr(const void* p) {          // byte address inside the workgroup's LDS allocation
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ vf_bf16x8 vf_tr_join(vf_s16x4 lo, vf_s16x4 hi) {
    return __builtin_bit_cast(vf_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// two MFMA operands (at offA and offB), each = two transposing reads `gap` bytes apart (the two 4-row groups of a 32x32x16 operand's 8 consecutive k),
// in one block: four reads in flight, one wait
__device__ __forceinline__ void vf_tr_frag2_wait(vf_bf16x8& a, vf_bf16x8& b, unsigned addr, int offA, int offB, int gap) {
    vf_s16x4 a0, a1, b0, b1;
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\tds_read_b64_tr_b16 %2, %4 offset:%7\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(addr), "i"(offA), "i"(offA + gap), "i"(offB), "i"(offB + gap) : "memory");
    a = vf_tr_join(a0, a1);
    b = vf_tr_join(b0, b1);
}
