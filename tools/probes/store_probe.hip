// GPU probe: what does the EPILOGUE of a 256 x 256 GEMM tile cost on gfx950 as a function of the store shape?  One workgroup of 512 threads
// per CU (1024 workgroups, 4 rounds) writes its tile of a [M][N] matrix and nothing else; per-wave cycles from s_memtime around the store
// burst and around the drain (vmcnt 0), plus the wall time.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/store_probe.hip -o tools/bin/store_probe
//   mode 0: fp32, the MFMA accumulator pattern (lane = column, 16 rows per 32 x 32 block: dword stores, 2 rows x 128 B per instruction)
//   mode 1: fp32, row-contiguous dwordx4 (16 lanes x 16 B = one 256-byte row segment of the wave's 64 columns, 4 rows per instruction)
//   mode 2: bf16, the lane-pair pattern of the shipped kernel (dword stores, 4 rows x 64 B per instruction)
//   mode 3: bf16, row-contiguous dwordx4 (8 lanes x 16 B = one 128-byte row segment, 8 rows per instruction)
//   mode 4: bf16, row-contiguous dwordx2 (16 lanes x 8 B, 4 rows per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void store_kernel(void* __restrict__ out, int N, unsigned* __restrict__ stamps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 2, wave_n = wave & 3, half = lane >> 5, l31 = lane & 31;
    const int nb = N / 256;
    const int m_tile0 = (blockIdx.x / nb) * 256, n_tile0 = (blockIdx.x % nb) * 256;
    unsigned long long t0, t1, t2;
    const float v = (float)tid;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE == 0) {
        float* O = (float*)out;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* o = O + (size_t)(m_tile0 + wave_m * 128 + i * 32 + 4 * half) * N + n_tile0 + wave_n * 64 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * N] = v + r;
            }
    } else if (MODE == 1) {
        float* O = (float*)out;      // wave tile 128 rows x 64 columns: 16 lanes per row (16 B each), 4 rows per instruction, 32 instructions
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int row = q * 4 + (lane >> 4);
            f32x4 w = {v, v + 1, v + 2, v + q};
            *reinterpret_cast<f32x4*>(O + (size_t)(m_tile0 + wave_m * 128 + row) * N + n_tile0 + wave_n * 64 + (lane & 15) * 4) = w;
        }
    } else if (MODE == 2) {
        unsigned* O = (unsigned*)out;     // bf16 pairs; N counts bf16 elements
        const int odd = l31 & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m0 = m_tile0 + wave_m * 128 + i * 32 + 4 * half;
                const int n = n_tile0 + wave_n * 64 + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int m = m0 + ((r + odd) & 3) + 8 * ((r + odd) >> 2);
                    O[((size_t)m * N + (n - odd)) >> 1] = __float_as_uint(v) + r;
                }
            }
    } else if (MODE == 3) {
        unsigned short* O = (unsigned short*)out;      // 128 rows x 64 bf16 = 128 B per row: 8 lanes per row, 8 rows per instruction, 16 instructions
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = q * 8 + (lane >> 3);
            f32x4 w = {v, v + 1, v + 2, v + q};
            *reinterpret_cast<f32x4*>(O + (size_t)(m_tile0 + wave_m * 128 + row) * N + n_tile0 + wave_n * 64 + (lane & 7) * 8) = w;
        }
    } else {
        unsigned short* O = (unsigned short*)out;      // 16 lanes x 8 B per row, 4 rows per instruction, 32 instructions
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int row = q * 4 + (lane >> 4);
            f32x2 w = {v, v + q};
            *reinterpret_cast<f32x2*>(O + (size_t)(m_tile0 + wave_m * 128 + row) * N + n_tile0 + wave_n * 64 + (lane & 15) * 4) = w;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
    if (lane == 0) {
        stamps[((size_t)blockIdx.x * 8 + wave) * 2] = (unsigned)(t1 - t0);
        stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = (unsigned)(t2 - t1);
    }
}

template <int MODE>
void run(const char* name, int esz, int M = 65536) {
    const int N = 1024;                                // M = 65536: 1024 tiles = 4 rounds of 256 CUs; M = 2048: 32 tiles (one CU in eight)
    void* out;
    unsigned* st;
    const int nwg = (M / 256) * (N / 256);
    hipMalloc(&out, (size_t)M * N * esz);
    hipMalloc(&st, (size_t)nwg * 16 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nwg), dim3(512), 0, 0, out, N, st);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(nwg), dim3(512), 0, 0, out, N, st);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h((size_t)nwg * 16);
    hipMemcpy(h.data(), st, h.size() * 4, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (size_t i = 0; i < h.size(); i += 2) { a += h[i]; b += h[i + 1]; }
    a /= h.size() / 2; b /= h.size() / 2;
    const double us = ms * 1e3 / it, bytes = (double)M * N * esz;
    printf("M=%5d %-52s issue %7.0f cycles  drain %7.0f cycles per wave   wall %7.1f us (%5.2f TB/s)  %5.1f us per round of 256 tiles, tile = %d KB\n", M, name, a, b, us,
           bytes / us / 1e6, us / 4, 256 * 256 * esz / 1024);
    hipFree(out);
    hipFree(st);
}

int main() {
    run<0>("fp32 accumulator pattern (dword, 2 x 128 B rows)", 4);
    run<1>("fp32 row-contiguous dwordx4 (4 x 256 B rows)", 4);
    run<2>("bf16 lane-pair pattern (dword, 4 x 64 B rows)", 2);
    run<3>("bf16 row-contiguous dwordx4 (8 x 128 B rows)", 2);
    run<4>("bf16 row-contiguous dwordx2 (4 x 128 B rows)", 2);
    // 32 workgroups: the HBM write rate is out of the picture — what ONE CU's store path moves
    run<0>("fp32 accumulator pattern (dword, 2 x 128 B rows)", 4, 2048);
    run<1>("fp32 row-contiguous dwordx4 (4 x 256 B rows)", 4, 2048);
    run<2>("bf16 lane-pair pattern (dword, 4 x 64 B rows)", 2, 2048);
    run<3>("bf16 row-contiguous dwordx4 (8 x 128 B rows)", 2, 2048);
    run<4>("bf16 row-contiguous dwordx2 (4 x 128 B rows)", 2, 2048);
    return 0;
}
