// probe: semantics of ds_read_b64_tr_b16 on gfx950 (which lane's address / element each result half-word comes from)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned short* out, int stride_b) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * stride_b;   // LDS byte address
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    u16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int stride : {8, 32, 64}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d bytes (lane i address = elements %d*i ..):\n", stride, stride / 2);
        for (int l = 0; l < 64; ++l) { printf("  lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
    }
    return 0;
}
