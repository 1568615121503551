// 16-bit MFMA ceiling probe under the package power limit (gfx950): what does a loop of nothing but v_mfma_f32_32x32x16_f16 sustain with
// all-zero operands and with full-entropy operands?  hipcc --offload-arch=gfx950 -O3 tools/mfma16_peak.hip -o tools/bin/mfma16_peak
// Each wave keeps NACC independent accumulators (NACC x 16 registers) and 4 + 4 operand fragments in registers; no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int NACC, bool BF16>
__global__ __launch_bounds__(256, 2) void loop16(float* out, unsigned long long* clk, int iters, int mode) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 4; ++e) {
            // mode 0: zeros; 1: random sign + mantissa, exponent of O(1) values (f16 0x3c00 region); 2: small constant
            unsigned ra = mix(threadIdx.x * 977u + q * 31u + e * 7u + 1u), rb = mix(threadIdx.x * 1361u + q * 17u + e * 3u + 5u);
            unsigned ha = BF16 ? ((ra & 0x807f807fu) | 0x3f003f00u) : ((ra & 0x83ff83ffu) | 0x38003800u);
            unsigned hb = BF16 ? ((rb & 0x807f807fu) | 0x3c003c00u) : ((rb & 0x83ff83ffu) | 0x2c002c00u);
            a[q][e] = mode == 0 ? 0u : mode == 1 ? ha : (BF16 ? 0x3f803f80u : 0x3c003c00u);
            b[q][e] = mode == 0 ? 0u : mode == 1 ? hb : (BF16 ? 0x3c003c00u : 0x2c002c00u);
        }
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u + i) & 3]), __builtin_bit_cast(bf16x8, b[(u * 3 + i) & 3]), acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + i) & 3]), __builtin_bit_cast(f16x8, b[(u * 3 + i) & 3]), acc[i], 0, 0, 0);
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(a[q]), "+v"(b[q]));
        if (mode == 1 && (it & 255) == 255) {             // keep the accumulators bounded (a sign flip of the operands)
#pragma unroll
            for (int q = 0; q < 4; ++q) for (int e = 0; e < 4; ++e) a[q][e] ^= 0x80008000u;
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// the 16x16x32 shape (4 accumulator registers per tile, half the flops per instruction) and operand-reuse orders of the 32x32x16 shape:
// reuse 0: A and B both change every instruction; 1: B fixed for runs of 4 (the x3h kernel's order: one weight fragment, four pixel fragments);
// 2: A fixed for runs of 4
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int REUSE, bool SMALL>
__global__ __launch_bounds__(256, 2) void loop16b(float* out, int iters) {
    f32x16 acc[4];
    f32x4 acc4[8];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    u32x4 a[4], b[4];
    for (int q = 0; q < 4; ++q)
        for (int e = 0; e < 4; ++e) {
            unsigned ra = mix(threadIdx.x * 977u + q * 31u + e * 7u + 1u), rb = mix(threadIdx.x * 1361u + q * 17u + e * 3u + 5u);
            a[q][e] = (ra & 0x83ff83ffu) | 0x38003800u;
            b[q][e] = (rb & 0x83ff83ffu) | 0x2c002c00u;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SMALL) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(u + i) & 3]), __builtin_bit_cast(f16x8, b[(u * 3 + i) & 3]), acc4[i], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ia = REUSE == 2 ? (u & 3) : ((u + i) & 3), ib = REUSE == 1 ? (u & 3) : REUSE == 2 ? ((u * 3 + i) & 3) : ((u * 3 + i) & 3);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ia]), __builtin_bit_cast(f16x8, b[ib]), acc[i], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(a[q]), "+v"(b[q]));
        if ((it & 255) == 255) {
#pragma unroll
            for (int q = 0; q < 4; ++q) for (int e = 0; e < 4; ++e) a[q][e] ^= 0x80008000u;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int REUSE, bool SMALL>
void runb(const char* name, int iters) {
    float* out; hipMalloc(&out, (size_t)512 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) loop16b<REUSE, SMALL><<<512, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) loop16b<REUSE, SMALL><<<512, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double fl = (double)512 * 4 * iters * 8 * (SMALL ? 8 * 16384.0 : 4 * 32768.0);
    printf("%-44s random %8.3f ms  %7.1f TF  (%.3f of 2500)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0);
    hipFree(out);
}

template <int NACC, bool BF16>
void run(const char* name, int blocks, int iters, int mode) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) loop16<NACC, BF16><<<blocks, 256>>>(out, clk, iters, mode);       // warm: >= 0.2 s at the power state
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) loop16<NACC, BF16><<<blocks, 256>>>(out, clk, iters, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double fl = (double)blocks * 4 * iters * 8 * NACC * 32768.0;
    const char* modes[] = {"zeros", "random", "const"};
    printf("%-24s %-6s blocks=%4d  %8.3f ms  %7.1f TF  (%.3f of 2500)  shader clock ~%.2f GHz\n", name, modes[mode], blocks, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0,
           (double)h[0] / ((double)h[1] / 100e6) / 1e9);
    hipFree(out); hipFree(clk);
}

int main() {
    const int it = 40000;
    for (int mode = 0; mode < 3; ++mode) {
        run<4, false>("f16 4acc 2w/SIMD", 512, it, mode);
        run<8, false>("f16 8acc 2w/SIMD", 512, it / 2, mode);
        run<4, false>("f16 4acc 1w/SIMD", 256, it, mode);
        run<4, true>("bf16 4acc 2w/SIMD", 512, it, mode);
    }
    runb<0, false>("f16 32x32x16, A and B change every MFMA", it);
    runb<1, false>("f16 32x32x16, B fixed for runs of 4", it);
    runb<2, false>("f16 32x32x16, A fixed for runs of 4", it);
    runb<0, true>("f16 16x16x32 (8 acc)", it);
    return 0;
}
