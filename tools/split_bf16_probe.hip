// Probe (gfx950): (1) bf16 MFMA ceiling, (2) accuracy of fp32 products emulated by split-bf16 MFMAs
// (x = h + m + l, each a bf16; 6 or 3 partial products per fp32 product) against the native f32 MFMA and fp64.
// hipcc --offload-arch=gfx950 -O3 tools/split_bf16_probe.hip -o /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void bf16_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)(a + threadIdx.x + e); bv[e] = (__bf16)(b + threadIdx.x * e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run_peak(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    bf16_loop<NACC><<<blocks, 256>>>(out, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    bf16_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 16;
    printf("%-34s blocks=%5d  %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}

// one wave: C[32x32] = A[32xK] * B[Kx32], A row-major [32][K], B stored [32 cols][K] (k contiguous)
// mode 0: native f32 MFMA (32x32x2), 1: bf16 x6 (small terms first), 2: bf16 x3 (hh, hm, mh), 3: bf16 x1
__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x; float r = x - (float)h; m = (__bf16)r; r = r - (float)m; l = (__bf16)r;
}
__global__ void numerics(const float* A, const float* B, float* C, int K, int mode) {
    int lane = threadIdx.x, rc = lane & 31, kg = lane >> 5;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[rc * K + k + kg], B[rc * K + k + kg], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 16) {
            bf16x8 ah, am, al, bh, bm, bl;
            for (int e = 0; e < 8; ++e) {
                __bf16 h, m, l;
                split3(A[rc * K + k + kg * 8 + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
                split3(B[rc * K + k + kg * 8 + e], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
            }
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            }
            if (mode == 1 || mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r >> 2) * 8 + kg * 4 + (r & 3)) * 32 + rc] = acc[r];
}
// mode 4: x6 with the small terms in their own accumulator, added once at the end
__global__ void numerics_2acc(const float* A, const float* B, float* C, int K) {
    int lane = threadIdx.x, rc = lane & 31, kg = lane >> 5;
    f32x16 acc, lo; for (int r = 0; r < 16; ++r) { acc[r] = 0.f; lo[r] = 0.f; }
    for (int k = 0; k < K; k += 16) {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int e = 0; e < 8; ++e) {
            __bf16 h, m, l;
            split3(A[rc * K + k + kg * 8 + e], h, m, l); ah[e] = h; am[e] = m; al[e] = l;
            split3(B[rc * K + k + kg * 8 + e], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l;
        }
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, lo, 0, 0, 0);
        lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, lo, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r >> 2) * 8 + kg * 4 + (r & 3)) * 32 + rc] = acc[r] + lo[r];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }
static double nrand() { return sqrt(-2 * log(urand())) * cos(6.283185307179586 * urand()); }

// same loop with operands that change every iteration (random bit patterns from a register LCG, exponent kept sane): does
// data toggling lower the sustainable rate (power)?  2 VALU per 64 MFMAs: issue overhead is negligible
template <int NACC>
__global__ __launch_bounds__(256) void bf16_loop_toggle(float* out, int iters, unsigned seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 a, b;
    unsigned st = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int e = 0; e < 4; ++e) { st = st * 1664525u + 1013904223u; a[e] = st; st = st * 1664525u + 1013904223u; b[e] = st; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            // new mantissa/sign bits, exponents pinned near 1.0 (0x3F80 pattern in both bf16 halves)
            st = st * 1664525u + 1013904223u;
            a[u & 3] = (a[u & 3] ^ st) & 0x807F807Fu | 0x3F003F00u;
            b[(u + 1) & 3] = (b[(u + 1) & 3] ^ (st >> 3)) & 0x807F807Fu | 0x3F003F00u;
            bf16x8 av = __builtin_bit_cast(bf16x8, a), bv = __builtin_bit_cast(bf16x8, b);
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the x6 kernels' operand traffic without anything else: per 24 MFMAs, 6 LDS b128 fragments (A planes) and/or 6 global b128
// fragments (B planes, L2-resident 24 KB window), prefetched one stage ahead; 4 accumulators; 2 workgroups per CU
template <bool USE_LDS, bool USE_GLB, int PHASE = 0>
__global__ __launch_bounds__(256, 2) void feed_loop(float* out, const unsigned char* __restrict__ wts, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[36 * 1024];
    for (int i = threadIdx.x; i < 36 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = (i * 2654435761u) & 0x807F807Fu | 0x3F003F00u;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    bf16x8 a[2][2][3], b[2][3][2];
    for (int x = 0; x < 2; ++x) for (int m = 0; m < 2; ++m) for (int pl = 0; pl < 3; ++pl)
        for (int e = 0; e < 8; ++e) { a[x][m][pl][e] = (__bf16)(1.0f + 0.01f * e); }
    for (int x = 0; x < 2; ++x) for (int pl = 0; pl < 3; ++pl) for (int j = 0; j < 2; ++j)
        for (int e = 0; e < 8; ++e) { b[x][pl][j][e] = (__bf16)(1.0f - 0.01f * e); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int nx = s ^ 1;
            if (USE_GLB) {
                // PHASE 1: every workgroup walks the same 16-stage window with its own phase (are simultaneous reads of the SAME
                // lines by all CUs the problem?); PHASE 2: the two waves that share fragments (wave_m pair) read the same lines, as in the kernels
                const int ph = PHASE ? (blockIdx.x * 5) : 0;
                const int wsel = PHASE == 2 ? ((threadIdx.x >> 6) & 1) : 0;
                const unsigned char* src = wts + ((it * 2 + s + ph) % 16) * 12288 + wsel * 2048 + lane * 16;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[nx][pl][j] = *reinterpret_cast<const bf16x8*>(src + pl * 4096 + j * 1024);
            }
            if (USE_LDS) {
                const unsigned char* src = lds + (((it * 2 + s) * 7) % 16) * 208 + (lane & 31) * 208 + (lane >> 5) * 16;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) a[nx][m][pl] = *reinterpret_cast<const bf16x8*>(src + m * 32 * 208 + pl * 64);
            }
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][m][PA[t]], b[s][PB[t]][j], acc[m * 2 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <bool USE_LDS, bool USE_GLB, int PHASE = 0>
void run_feed(const char* name, int iters) {
    const int blocks = 512;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned char* w; hipMalloc(&w, 16 * 12288 + 4096);
    std::vector<unsigned> hw((16 * 12288 + 4096) / 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((unsigned)i * 2246822519u) & 0x807F807Fu | 0x3F003F00u;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * 2 * 24 * 2.0 * 32 * 32 * 16;
    printf("%-58s", name);
    for (int l = 0; l < 6; ++l) {
        hipEventRecord(e0);
        feed_loop<USE_LDS, USE_GLB, PHASE><<<blocks, 256>>>(out, w, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %.0f", fl / ms / 1e9);
    }
    printf(" TF\n");
    hipFree(out); hipFree(w);
}

// the same skeleton with the weight fragments fetched BD stages ahead in a (BD+1)-deep register ring: how much prefetch
// distance does the L2 -> VGPR path need?  WAVES_PER_SIMD 2: launch_bounds(256, 2); 1: (256, 1) with 48-MFMA stages (4 x 2 tiles)
template <int BD, int MI>
__global__ __launch_bounds__(256, (MI == 2 ? 2 : 1)) void ring_loop(float* out, const unsigned char* __restrict__ wts, int iters) {
    constexpr int RING = BD + 1;
    f32x16 acc[MI][2];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int lane = threadIdx.x & 63;
    bf16x8 a[MI][3], b[RING][3][2];
    for (int m = 0; m < MI; ++m) for (int pl = 0; pl < 3; ++pl) for (int e = 0; e < 8; ++e) a[m][pl][e] = (__bf16)(1.0f + 0.01f * e);
    auto b_load = [&](bf16x8 (&dst)[3][2], int g) {
        const unsigned char* src = wts + (g % 16) * 12288 + lane * 16;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < 2; ++j) dst[pl][j] = *reinterpret_cast<const bf16x8*>(src + pl * 4096 + j * 1024);
    };
#pragma unroll
    for (int g = 0; g < BD; ++g) b_load(b[g], g);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            b_load(b[(s + BD) % RING], it * RING + s + BD);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int m = 0; m < MI; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[t]], b[s][PB[t]][j], acc[m][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float sum = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int BD, int MI>
void run_ring(int iters) {
    const int blocks = MI == 2 ? 512 : 256;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned char* w; hipMalloc(&w, 16 * 12288 + 4096);
    std::vector<unsigned> hw((16 * 12288 + 4096) / 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((unsigned)i * 2246822519u) & 0x807F807Fu | 0x3F003F00u;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * (BD + 1) * (MI * 2 * 6) * 2.0 * 32 * 32 * 16;
    printf("  weights %d stage(s) ahead, %d x 2 tiles per wave, %d wave(s)/SIMD:", BD, MI, MI == 2 ? 2 : 1);
    for (int l = 0; l < 5; ++l) {
        hipEventRecord(e0);
        ring_loop<BD, MI><<<blocks, 256>>>(out, w, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %.0f", fl / ms / 1e9);
    }
    printf(" TF\n");
    hipFree(out); hipFree(w);
}

// candidate structure: weight fragments staged through LDS by direct-to-LDS loads (global_load_lds_dwordx4, no VGPR round trip),
// 12 KB per stage shared by the 4 waves of the workgroup, 3-slot ring filled two stages ahead, ONE barrier per stage; every wave
// then reads 6 A + 6 B fragments (b128) from LDS per 24 MFMAs.  2 workgroups per CU.
__global__ __launch_bounds__(256, 2) void ldsb_loop(float* out, const unsigned char* __restrict__ wts, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[36 * 1024 + 3 * 12288];
    unsigned char* bring = lds + 36 * 1024;
    for (int i = threadIdx.x; i < 36 * 1024 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = (i * 2654435761u) & 0x807F807Fu | 0x3F003F00u;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto fill = [&](int g) {          // this wave's quarter (3 KB = 3 wave-wide 1 KB moves) of stage g's 12 KB
        const unsigned char* src = wts + (g % 16) * 12288 + wave * 3072 + lane * 16;
        unsigned char* dst = bring + (g % 3) * 12288 + wave * 3072;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((const void*)(src + k * 1024), (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, 0, 0);
    };
    fill(0); fill(1);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int wave_n = wave & 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int g = it * 3 + s;
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");            // this wave's part of stage g has landed (stage g+1's 3 moves may fly)
            __syncthreads();
            bf16x8 a[2][3], b[3][2];
            const unsigned char* asrc = lds + ((g * 7) % 16) * 208 + (lane & 31) * 208 + (lane >> 5) * 16;
            const unsigned char* bsrc = bring + s * 12288 + ((lane >> 5) * 128 + wave_n * 64 + (lane & 31)) * 16;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[m][pl] = *reinterpret_cast<const bf16x8*>(asrc + m * 32 * 208 + pl * 64);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const bf16x8*>(bsrc + pl * 4096 + j * 512);
            fill(g + 2);
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
            constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[t]], b[PB[t]][j], acc[m * 2 + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

void run_ldsb(int iters) {
    const int blocks = 512;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned char* w; hipMalloc(&w, 16 * 12288 + 4096);
    std::vector<unsigned> hw((16 * 12288 + 4096) / 4);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = ((unsigned)i * 2246822519u) & 0x807F807Fu | 0x3F003F00u;
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(ldsb_loop), hipFuncAttributeMaxDynamicSharedMemorySize, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * 3 * 24 * 2.0 * 32 * 32 * 16;
    printf("  weights through LDS (direct-to-LDS loads, barrier per stage), A and B fragments from LDS:");
    for (int l = 0; l < 5; ++l) {
        hipEventRecord(e0);
        ldsb_loop<<<blocks, 256>>>(out, w, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %.0f", fl / ms / 1e9);
    }
    printf(" TF  (%s)\n", hipGetErrorString(hipGetLastError()));
    hipFree(out); hipFree(w);
}

template <int NACC>
void run_sustained_toggle(const char* name, int blocks, int iters, int launches) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 16;
    printf("%s, %d launches:", name, launches);
    for (int l = 0; l < launches; ++l) {
        hipEventRecord(e0);
        bf16_loop_toggle<NACC><<<blocks, 256>>>(out, iters, 12345u + l);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (l % 4 == 0) printf(" %.0f", fl / ms / 1e9);
    }
    printf(" TF\n");
    hipFree(out);
}

// sustained rate: back-to-back launches for ~2 s (is the short-run ceiling power/clock limited when held?)
template <int NACC>
void run_sustained(const char* name, int blocks, int iters, int launches) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double fl = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 16;
    printf("%s, %d launches:", name, launches);
    for (int l = 0; l < launches; ++l) {
        hipEventRecord(e0);
        bf16_loop<NACC><<<blocks, 256>>>(out, iters, 1.f, 2.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (l % 4 == 0) printf(" %.0f", fl / ms / 1e9);
    }
    printf(" TF\n");
    hipFree(out);
}

int main(int argc, char** argv) {
    if (argc > 1) {           // sustained mode
        run_sustained<4>("sustained bf16 32x32x16 4acc 2 waves/SIMD (~50 ms per launch)", 512, 24000, 40);
        run_sustained<2>("sustained bf16 32x32x16 2acc 2 waves/SIMD (~50 ms per launch)", 512, 48000, 40);
        run_sustained_toggle<4>("sustained, operands re-randomised every 4 MFMAs, 4acc 2 waves/SIMD", 512, 24000, 40);
        run_sustained_toggle<4>("sustained, operands re-randomised every 4 MFMAs, 4acc 1 wave/SIMD", 256, 48000, 40);
        run_feed<false, false>("x6 stage skeleton, operands in registers only", 20000);
        run_feed<true, false>("x6 stage skeleton, A planes from LDS (6 b128 / 24 MFMA)", 20000);
        run_feed<false, true>("x6 stage skeleton, B planes from L2 (6 b128 / 24 MFMA)", 20000);
        run_feed<true, true>("x6 stage skeleton, A from LDS and B from L2", 20000);
        run_ring<1, 2>(20000); run_ring<2, 2>(14000); run_ring<3, 2>(10000); run_ring<5, 2>(7000);
        run_ring<1, 4>(10000); run_ring<2, 4>(7000); run_ring<3, 4>(5000); run_ring<5, 4>(3500);
        run_ldsb(14000);
        run_feed<false, true, 1>("  B from L2, per-workgroup phase offsets", 20000);
        run_feed<false, true, 2>("  B from L2, phase offsets + 2 distinct fragment sets per WG", 20000);
        return 0;
    }
    run_peak<4>("bf16 32x32x16 4acc 1 wave/SIMD", 256, 4000);
    run_peak<4>("bf16 32x32x16 4acc 2 waves/SIMD", 512, 4000);
    run_peak<8>("bf16 32x32x16 8acc 1 wave/SIMD", 256, 2000);
    run_peak<2>("bf16 32x32x16 2acc 2 waves/SIMD", 512, 4000);
    for (int variant = 0; variant < 3; ++variant) {
        const int K = variant == 2 ? 4608 : 1152;
        std::vector<float> A(32 * K), B(32 * K);
        srand(7 + variant);
        for (int i = 0; i < 32 * K; ++i) {
            // variant 0: zero-mean operands (cancellation); 1: swish-like positive activations times zero-mean weights
            double a = nrand(), b = nrand() * 0.03;
            if (variant >= 1) a = a / (1 + exp(-a)) + 0.3;
            A[i] = (float)a; B[i] = (float)b;
        }
        std::vector<double> ref(1024), mag(1024);
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double s = 0, m = 0;
            for (int k = 0; k < K; ++k) { double p = (double)A[r * K + k] * (double)B[c * K + k]; s += p; m += fabs(p); }
            ref[r * 32 + c] = s; mag[r * 32 + c] = m;
        }
        float *dA, *dB, *dC; hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 32 * K * 4, hipMemcpyHostToDevice);
        const char* names[5] = {"f32 MFMA 32x32x2", "bf16 x6 (one accumulator)", "bf16 x3", "bf16 x1", "bf16 x6 (lo/hi accumulators)"};
        printf("variant %d  K=%d   error relative to sum|a*b| (max, rms):\n", variant, K);
        for (int mode = 0; mode < 5; ++mode) {
            if (mode < 4) numerics<<<1, 64>>>(dA, dB, dC, K, mode); else numerics_2acc<<<1, 64>>>(dA, dB, dC, K);
            std::vector<float> C(1024);
            hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
            double mx = 0, ss = 0;
            for (int i = 0; i < 1024; ++i) { double e = fabs(C[i] - ref[i]) / mag[i]; mx = fmax(mx, e); ss += e * e; }
            printf("  %-30s max %.3e  rms %.3e\n", names[mode], mx, sqrt(ss / 1024));
        }
        // also the host fp32 fmaf chain in k order (what a scalar CPU loop gives)
        double mx = 0, ss = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s = fmaf(A[r * K + k], B[c * K + k], s);
            double e = fabs(s - ref[r * 32 + c]) / mag[r * 32 + c]; mx = fmax(mx, e); ss += e * e;
        }
        printf("  %-30s max %.3e  rms %.3e\n", "host fmaf chain", mx, sqrt(ss / 1024));
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    return 0;
}
