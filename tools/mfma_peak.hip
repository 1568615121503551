// Standalone f32-MFMA ceiling probe (gfx950): hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256, 2) void mfma_loop_lb2(float* out, int iters, float a, float b) {
    // same loop under the launch bound my kernels use: does hipcc switch to the VGPR-form MFMA, and what does it cost?
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av[8], bv[8];
    for (int q = 0; q < 8; ++q) { av[q] = a + threadIdx.x + q; bv[q] = b + threadIdx.x * q; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(u + i) & 7], bv[(u * 3 + i) & 7], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(av[q]), "+v"(bv[q]));
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f; }
    float av = a + threadIdx.x, bv = b + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 32) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc4[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; for (int r = 0; r < 4; ++r) s += acc4[i][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int SHAPE>
void run(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<NACC, SHAPE><<<blocks, 256>>>(out, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<NACC, SHAPE><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop_per = SHAPE == 32 ? 2.0 * 32 * 32 * 2 : 2.0 * 16 * 16 * 4;
    double fl = (double)blocks * 4 * iters * 16 * NACC * flop_per;
    printf("%-28s blocks=%5d  %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}

template <int NACC>
void run_lb2(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop_lb2<NACC, 32><<<blocks, 256>>>(out, iters / 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop_lb2<NACC, 32><<<blocks, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
    printf("%-28s blocks=%5d  %.3f ms  %.1f TF\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    run_lb2<4>("LB(256,2) 4acc 2 waves/SIMD", 512, 2000);
    run_lb2<4>("LB(256,2) 4acc 1 wave/SIMD", 256, 2000);
    run_lb2<2>("LB(256,2) 2acc 2 waves/SIMD", 512, 4000);
    run<4, 32>("32x32x2 4acc 1 wave/SIMD", 256, 2000);
    run<4, 32>("32x32x2 4acc 2 waves/SIMD", 512, 2000);
    run<4, 32>("32x32x2 4acc 4 waves/SIMD", 1024, 1000);
    run<1, 32>("32x32x2 1acc 1 wave/SIMD", 256, 4000);
    run<2, 32>("32x32x2 2acc 2 waves/SIMD", 512, 4000);
    run<4, 16>("16x16x4 4acc 1 wave/SIMD", 256, 8000);
    run<4, 16>("16x16x4 4acc 2 waves/SIMD", 512, 8000);
    run<8, 16>("16x16x4 8acc 2 waves/SIMD", 512, 4000);
    return 0;
}
