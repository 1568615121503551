// GPU probe (round 6): where does the fused optimizer + re-pack kernel lose against a streaming AdamW?  Same data volume in every mode (the
// training model's 49 packed matrices, 85 M parameters), hipEvent wall time per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/adamw_pack_probe.hip -o tools/bin/adamw_pack_probe
//   mode 0: flat AdamW, float4 grid-stride (the shipped adamw_flat_kernel's access pattern)            — the floor for w g m v traffic
//   mode 1: 16 x 128 tiles, one per block, AdamW only (no bf16 output)                                  — cost of the tile order alone
//   mode 2: mode 1 + both packings through LDS (the shipped adamw_pack_tiles_kernel)
//   mode 3: mode 2 with only the [K][N] packing; mode 4: only the [N][K] packing
//   mode 5: 8 x 256 tiles (1 KB row segments), both packings
//   mode 6: row bands: a block owns 16 rows x all columns and walks the 128-column tiles with the next tile's loads in flight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
struct Desc { long long offset; __bf16* kn; __bf16* nk; int rows, cols; };

__device__ __forceinline__ void step(float& w, float gi, float& mi, float& vi) {
#pragma clang fp contract(off)
    const float we = __builtin_fmaf(-1e-5f, w, w);
    mi = __builtin_fmaf(0.9f, mi, 0.1f * gi);
    vi = __builtin_fmaf(0.999f, vi, (0.001f * gi) * gi);
    w = we - (1e-4f * mi) / (sqrtf(vi) + 1e-7f);
}
__global__ void flat_kernel(float* p, const float* g, float* m, float* v, long long n4) {
    for (long long i4 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i4 < n4; i4 += (long long)gridDim.x * blockDim.x) {
        const long long i = i4 * 4;
        f32x4 w = *(const f32x4*)(p + i), gi = *(const f32x4*)(g + i), mi = *(const f32x4*)(m + i), vi = *(const f32x4*)(v + i);
        for (int e = 0; e < 4; ++e) { float a = w[e], b = mi[e], c = vi[e]; step(a, gi[e], b, c); w[e] = a; mi[e] = b; vi[e] = c; }
        *(f32x4*)(m + i) = mi; *(f32x4*)(v + i) = vi; *(f32x4*)(p + i) = w;
    }
}
template <int TR, int TC, bool KN, bool NK>
__global__ __launch_bounds__(256) void tile_kernel(float* p, const float* g, float* m, float* v, const Desc* descs) {
    constexpr int LD = TC + 8, TPR = TC / 8;                        // threads per row
    __shared__ __attribute__((aligned(16))) __bf16 tile[TR * LD];
    const Desc d = descs[blockIdx.y];
    const int t = threadIdx.x, tcols = d.cols / TC, ntiles = (d.rows / TR) * tcols;
    const int nb_kn = d.cols / 128, nb_nk = d.rows / 128;
    for (int ti = blockIdx.x; ti < ntiles; ti += gridDim.x) {
        const int tr = ti / tcols, tc = ti - tr * tcols, r0 = tr * TR, c0 = tc * TC;
        const int r = t / TPR, c8 = (t % TPR) * 8;
        const long long i = d.offset + (long long)(r0 + r) * d.cols + c0 + c8;
        bf16x8_t o;
        for (int h = 0; h < 2; ++h) {
            f32x4 w = *(const f32x4*)(p + i + 4 * h), gi = *(const f32x4*)(g + i + 4 * h), mi = *(const f32x4*)(m + i + 4 * h), vi = *(const f32x4*)(v + i + 4 * h);
            for (int e = 0; e < 4; ++e) { float a = w[e], b = mi[e], c = vi[e]; step(a, gi[e], b, c); w[e] = a; mi[e] = b; vi[e] = c; }
            *(f32x4*)(m + i + 4 * h) = mi; *(f32x4*)(v + i + 4 * h) = vi; *(f32x4*)(p + i + 4 * h) = w;
            for (int e = 0; e < 4; ++e) o[4 * h + e] = (__bf16)w[e];
        }
        if (KN || NK) {
            *(bf16x8_t*)(tile + r * LD + c8) = o;
            __syncthreads();
            if (KN) {                                               // TR / 8 halves x TC columns = 256 groups
                const int hh = t / TC, nl0 = t % TC;
                bf16x8_t q;
                for (int e = 0; e < 8; ++e) q[e] = tile[(hh * 8 + e) * LD + nl0];
                const int k0 = r0 + hh * 8, col = c0 + nl0;
                const long long gi = ((((long long)(k0 / 64) * nb_kn + col / 128) * 4 + ((k0 % 64) >> 4)) * 2 + ((k0 & 15) >> 3)) * 128 + (col % 128);
                *(bf16x8_t*)(d.kn + gi * 8) = q;
            }
            if (NK) {
                const int j = t / TR, rr = t % TR;
                const bf16x8_t q = *(const bf16x8_t*)(tile + rr * LD + j * 8);
                const int c = c0 + j * 8, row = r0 + rr;
                const long long gi = ((((long long)(c / 64) * nb_nk + row / 128) * 4 + ((c % 64) >> 4)) * 2 + ((c & 15) >> 3)) * 128 + (row % 128);
                *(bf16x8_t*)(d.nk + gi * 8) = q;
            }
            __syncthreads();
        }
    }
}
// row bands: block (band of 16 rows) walks its column tiles; loads of tile k + 1 are issued before tile k is processed
__global__ __launch_bounds__(256) void band_kernel(float* p, const float* g, float* m, float* v, const Desc* descs) {
    constexpr int TR = 16, TC = 128, LD = TC + 8;
    __shared__ __attribute__((aligned(16))) __bf16 tile[2][TR * LD];
    const Desc d = descs[blockIdx.y];
    const int t = threadIdx.x, tcols = d.cols / TC, nbands = d.rows / TR;
    const int nb_kn = d.cols / 128, nb_nk = d.rows / 128;
    for (int band = blockIdx.x; band < nbands; band += gridDim.x) {
        const int r0 = band * TR, r = t >> 4, c8 = (t & 15) * 8;
        const long long base = d.offset + (long long)(r0 + r) * d.cols + c8;
        f32x4 w[2], gi[2], mi[2], vi[2], nw[2], ng[2], nm[2], nv[2];
        for (int h = 0; h < 2; ++h) { nw[h] = *(const f32x4*)(p + base + 4 * h); ng[h] = *(const f32x4*)(g + base + 4 * h); nm[h] = *(const f32x4*)(m + base + 4 * h); nv[h] = *(const f32x4*)(v + base + 4 * h); }
        for (int tc = 0; tc < tcols; ++tc) {
            const long long i = base + tc * TC;
            for (int h = 0; h < 2; ++h) { w[h] = nw[h]; gi[h] = ng[h]; mi[h] = nm[h]; vi[h] = nv[h]; }
            if (tc + 1 < tcols)
                for (int h = 0; h < 2; ++h) { nw[h] = *(const f32x4*)(p + i + TC + 4 * h); ng[h] = *(const f32x4*)(g + i + TC + 4 * h); nm[h] = *(const f32x4*)(m + i + TC + 4 * h); nv[h] = *(const f32x4*)(v + i + TC + 4 * h); }
            bf16x8_t o;
            for (int h = 0; h < 2; ++h) {
                for (int e = 0; e < 4; ++e) { float a = w[h][e], b = mi[h][e], c = vi[h][e]; step(a, gi[h][e], b, c); w[h][e] = a; mi[h][e] = b; vi[h][e] = c; }
                *(f32x4*)(m + i + 4 * h) = mi[h]; *(f32x4*)(v + i + 4 * h) = vi[h]; *(f32x4*)(p + i + 4 * h) = w[h];
                for (int e = 0; e < 4; ++e) o[4 * h + e] = (__bf16)w[h][e];
            }
            __bf16* tl = tile[tc & 1];
            *(bf16x8_t*)(tl + r * LD + c8) = o;
            __syncthreads();
            const int c0 = tc * TC;
            {
                const int hh = t >> 7, nl0 = t & 127;
                bf16x8_t q;
                for (int e = 0; e < 8; ++e) q[e] = tl[(hh * 8 + e) * LD + nl0];
                const int k0 = r0 + hh * 8;
                const long long gi2 = ((((long long)(k0 / 64) * nb_kn + tc) * 4 + ((k0 % 64) >> 4)) * 2 + hh) * 128 + nl0;
                *(bf16x8_t*)(d.kn + gi2 * 8) = q;
            }
            {
                const int j = t >> 4, rr = t & 15;
                const bf16x8_t q = *(const bf16x8_t*)(tl + rr * LD + j * 8);
                const int c = c0 + j * 8, row = r0 + rr;
                const long long gi2 = ((((long long)(c / 64) * nb_nk + row / 128) * 4 + ((c % 64) >> 4)) * 2 + ((c & 15) >> 3)) * 128 + (row % 128);
                *(bf16x8_t*)(d.nk + gi2 * 8) = q;
            }
        }
    }
}
int main() {
    std::vector<Desc> hd;
    long long n = 0;
    auto add = [&](int r, int c) { hd.push_back({n, nullptr, nullptr, r, c}); n += (long long)r * c; };
    add(1024, 768);
    for (int l = 0; l < 12; ++l) { add(768, 2304); add(768, 768); add(768, 3072); add(3072, 768); }
    float *p, *g, *m, *v; __bf16 *kn, *nk;
    hipMalloc(&p, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&kn, n * 2); hipMalloc(&nk, n * 2);
    hipMemset(p, 0, n * 4); hipMemset(g, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4);
    for (auto& d : hd) { d.kn = kn + d.offset; d.nk = nk + d.offset; }
    Desc* dd; hipMalloc(&dd, hd.size() * sizeof(Desc)); hipMemcpy(dd, hd.data(), hd.size() * sizeof(Desc), hipMemcpyHostToDevice);
    const unsigned nd = (unsigned)hd.size();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode <= 6; ++mode) {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 12; ++it) {
            hipEventRecord(a);
            switch (mode) {
            case 0: hipLaunchKernelGGL(flat_kernel, dim3(8192), dim3(256), 0, 0, p, g, m, v, n / 4); break;
            case 1: hipLaunchKernelGGL((tile_kernel<16, 128, false, false>), dim3(1024, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            case 2: hipLaunchKernelGGL((tile_kernel<16, 128, true, true>), dim3(1024, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            case 3: hipLaunchKernelGGL((tile_kernel<16, 128, true, false>), dim3(1024, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            case 4: hipLaunchKernelGGL((tile_kernel<16, 128, false, true>), dim3(1024, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            case 5: hipLaunchKernelGGL((tile_kernel<8, 256, true, true>), dim3(1024, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            case 6: hipLaunchKernelGGL(band_kernel, dim3(192, nd), dim3(256), 0, 0, p, g, m, v, dd); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("mode %d: mean %.1f us  best %.1f us   (%.2f TB/s of w g m v traffic)\n", mode, sum / 10 * 1e3, best * 1e3, n * 28.0 / (sum / 10 * 1e-3) / 1e12);
    }
    return hipGetLastError() != hipSuccess;
}
