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

__device__ __forceinline__ float vf_swish(float v) {
    // x * sigmoid(x), precise expf + IEEE division (no fast-math: fp32 parity with the oracle)
    return v / (1.0f + expf(-v));
}

__device__ __forceinline__ float vf_gelu_erf(float v) {
    // tf.nn.gelu(approximate=False): 0.5 x (1 + erf(x / sqrt 2))
    return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
}

__device__ __forceinline__ float vf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float vf_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// shared host helper (defined in igemm_f32.hip): pack [taps][K][N] into the fragment-major B layout
int vf_pack_b_impl(const float* src, float* dst, int K, int N, int taps, long long sk, long long sn, long long st,
                   int BN, int batch, long long src_bstride, hipStream_t stream);
