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

int main() {
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
