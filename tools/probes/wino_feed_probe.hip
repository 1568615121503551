// GPU probe (VERDICT r2 'next' #1): what could a Winograd F(2x2,3x3) form of the x3h convolution sustain on gfx950 BEFORE its input
// side is paid for?  Not a convolution: the inner loop a conv3_wino_x3h kernel would have to run, on synthetic operands, with the
// largest tile the register file allows.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/wino_feed_probe.hip -o tools/bin/wino_feed_probe
//
// Structure under test: a workgroup owns T = 32 MI Winograd tiles (T = 64 <=> 16 x 16 output pixels) x 128 output
// channels (4 waves x 32), walks the input channels in 16-deep chunks (V of a chunk: 16 elements x T tiles x 2 fp16 planes in LDS)
// and, per chunk and element e, runs M_e = V_e U_e as 3 MI MFMAs (x3h: al'bh + ah bl' -> accx, ah bh -> acc) on fresh accumulators,
// then folds M_e into the 2 x 2 outputs of every tile (Y[a][b] += +-(acc + 2^-11 accx), the A^T . A output transform: 36 signed adds per
// 16 elements) — the 16 elements' accumulators cannot stay live across chunks (16 x T x 128 fp32 = 512 KB at T = 64: the whole register
// file), so the output transform is paid per chunk.  U streams L2 -> VGPR (1 MB per 128 -> 128 layer, 2 fp16 planes, fragment-major).
//   LEVEL 0: MFMAs + weight-fragment stream + LDS fragment reads          (the feed ceiling)
//   LEVEL 1: + the per-chunk output transform on the vector ALU           (what the arithmetic alone costs)
//   LEVEL 2: + a stand-in for the input side (HBM patch loads, ~640 VALU per thread and chunk, LDS parking, 2 barriers per chunk)
// The real input side (GroupNorm + swish prologue, B^T d B on 16 values per tile and channel, two fp16 splits, LDS parking, two more
// barriers per chunk: ~650 VALU instructions per thread and chunk against 96 MFMAs per wave) is only imitated in cost: the result is an
// upper bound.  Printed: executed 16-bit TFLOP/s and the fp32-EQUIVALENT convolution rate it would correspond to
// (executed / 3 products x 2.25 fewer multiplications than the direct form), next to the direct kernel's measured 352 TF.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CIN = 128, CK = 16, NCH = CIN / CK, NE = 16, BN = 128;
constexpr int W_PLANE = 2 * BN * 16;          // [half][n][8 f16] = 4 KB
constexpr int W_ELEM = 2 * W_PLANE;           // 2 planes per (chunk, element): 8 KB
constexpr int V_LDB = 64;                     // bytes per (element, tile): 2 planes x 16 channels x f16, 16-byte granules XOR-swizzled

__device__ __forceinline__ f16x8 wload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// outputs of element e = (i, j): rows a with A^T[a][i] != 0, columns b likewise; A^T = [[1,1,1,0],[0,1,-1,-1]]
__device__ __forceinline__ constexpr int at(int a, int i) { return a == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : (i == 1 ? 1 : -1)); }

template <int MI, int LEVEL>
__global__ __launch_bounds__(256, 1) void wino_probe(const unsigned char* __restrict__ W, const float* __restrict__ X, float* __restrict__ out, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // V: [NE][32 MI tiles][64 B]
    constexpr int T = 32 * MI;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    // synthetic V: fp16 values of unit scale (both planes: the scaled low piece has the high piece's magnitude)
    for (int i = tid; i < NE * T * V_LDB / 2; i += 256) {
        unsigned h = (unsigned)i * 2654435761u;
        reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(((int)(h >> 20) - 2048) * (1.0f / 1024.0f));
    }
    __syncthreads();
    int a_off[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int tile = mi * 32 + l31;
        a_off[mi] = tile * V_LDB;                                            // + ((plane * 2 + half) ^ ((tile >> 2) & 3)) * 16 below
    }
    const int sw = (l31 >> 2) & 3;
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(W), 0, 0x7fffffff, 0x00020000);
    const unsigned b_lane = (unsigned)((half * BN + wave * 32 + l31) * 16);
    constexpr int RING = 4;                 // 16 stages per chunk: the slot of a stage repeats per chunk; fragments are fetched 2 stages ahead
    f16x8 bring[RING][2];
    f32x16 Y[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[mi][o][r] = 0.f;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int n_stage = tiles_per_wg * NCH * NE;
    auto b_load = [&](f16x8 (&dst)[2], int g) {
        g = g < n_stage ? g : n_stage - 1;
        const unsigned soff = (unsigned)((g % (NCH * NE)) * W_ELEM);
        dst[0] = wload(w_rs, b_lane, soff);
        dst[1] = wload(w_rs, b_lane + W_PLANE, soff);
    };
    b_load(bring[0], 0);
    b_load(bring[1], 1);
    int g = 0;
    // LEVEL 2: a stand-in for the input side, spread over the 16 element stages of a chunk like the real kernel would have to: the raw
    // 18 x 18 x 16-channel fp32 patch of the NEXT chunk from HBM (5 float4 per thread), ~40 vector-ALU instructions per stage and thread
    // (GroupNorm + swish prologue 13 per value, B^T d B 2 adds and two fp16 splits 4 ops per V value = ~650 per chunk, packed where the ISA
    // packs), 2 ds_write_b64 per stage (V parked as h | l' planes) and two workgroup barriers per chunk (raw patch visible; V complete)
    f32x2 dummy[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) dummy[q] = (f32x2){(float)(tid + q), (float)(tid - q)};
    unsigned char* vpark = smem + NE * T * V_LDB + tid * 16;                   // scratch behind V (LEVEL 2 allocates 4 KB more)
    const f32x4* xsrc = reinterpret_cast<const f32x4*>(X) + (size_t)blockIdx.x * tiles_per_wg * NCH * 5 * 256 + tid;
    for (int t = 0; t < tiles_per_wg; ++t) {
        for (int chunk = 0; chunk < NCH; ++chunk) {
            f32x4 raw[5];
            if (LEVEL >= 2) {
#pragma unroll
                for (int q = 0; q < 5; ++q) raw[q] = xsrc[(size_t)((t * NCH + chunk) * 5 + q) * 256];
            }
#pragma unroll
            for (int e = 0; e < NE; ++e, ++g) {
                b_load(bring[(e + 2) % RING], g + 2);
                if (LEVEL >= 2) {
                    if (e == 8) __syncthreads();
                    if (e == 15) {
#pragma unroll
                        for (int q = 0; q < 5; ++q) dummy[q & 3] += (f32x2){raw[q][0] + raw[q][2], raw[q][1] + raw[q][3]};
                    }
#pragma unroll
                    for (int v = 0; v < 10; ++v)
#pragma unroll
                        for (int q = 0; q < 4; ++q) dummy[q] = __builtin_elementwise_fma(dummy[q], (f32x2){1.0001f, 0.9999f}, (f32x2){1e-3f, -1e-3f});
                    *reinterpret_cast<f32x2*>(vpark) = dummy[e & 3];
                    *reinterpret_cast<f32x2*>(vpark + 8) = dummy[(e + 1) & 3];
                }
                f16x8 a[MI][2];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        a[mi][pl] = *reinterpret_cast<const f16x8*>(smem + e * T * V_LDB + a_off[mi] + (((pl * 2 + half) ^ sw) * 16));
                if (LEVEL == 0) {                                            // feed ceiling: the same MFMAs accumulating in place, no vector ALU
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        Y[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][1], bring[e % RING][0], Y[mi][0], 0, 0, 0);
                        Y[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], bring[e % RING][1], Y[mi][1], 0, 0, 0);
                        Y[mi][2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], bring[e % RING][0], Y[mi][2], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    f32x16 acc[MI], accx[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        accx[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][1], bring[e % RING][0], zero, 0, 0, 0);
                        accx[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], bring[e % RING][1], accx[mi], 0, 0, 0);
                        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi][0], bring[e % RING][0], zero, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const int ei = e >> 2, ej = e & 3;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        // packed fp32 (v_pk_fma_f32 / v_pk_add_f32 with neg modifiers): 8 + 8 |outputs of e| instructions per MFMA tile
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const f32x2 ax = {accx[mi][2 * r], accx[mi][2 * r + 1]}, ac = {acc[mi][2 * r], acc[mi][2 * r + 1]};
                            const f32x2 m = __builtin_elementwise_fma(ax, (f32x2){4.8828125e-4f, 4.8828125e-4f}, ac);
#pragma unroll
                            for (int oa = 0; oa < 2; ++oa)
#pragma unroll
                                for (int ob = 0; ob < 2; ++ob) {
                                    const int s = at(oa, ei) * at(ob, ej);
                                    f32x2 y = {Y[mi][oa * 2 + ob][2 * r], Y[mi][oa * 2 + ob][2 * r + 1]};
                                    if (s > 0) y = y + m;
                                    else if (s < 0) y = y - m;
                                    Y[mi][oa * 2 + ob][2 * r] = y[0];
                                    Y[mi][oa * 2 + ob][2 * r + 1] = y[1];
                                }
                        }
                    }
                }
            }
            if (LEVEL >= 2) __syncthreads();
        }
    }
    float s = 0.f;
    if (LEVEL >= 2) s = dummy[0][0] + dummy[1][1] + dummy[2][0] + dummy[3][1];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += Y[mi][o][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MI, int LEVEL>
void run(const unsigned char* W, const float* X, float* out, int wgs, int tiles_per_wg) {
    constexpr int T = 32 * MI;
    const size_t smem = (size_t)NE * T * V_LDB + (LEVEL >= 2 ? 4096 : 0);
    hipFuncSetAttribute(reinterpret_cast<const void*>(wino_probe<MI, LEVEL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    wino_probe<MI, LEVEL><<<wgs, 256, smem>>>(W, X, out, 2);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        wino_probe<MI, LEVEL><<<wgs, 256, smem>>>(W, X, out, tiles_per_wg);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double mfma = (double)wgs * 4 * tiles_per_wg * NCH * NE * 3 * MI;           // per wave: 3 MI MFMAs per (chunk, element)
    const double exec_tf = mfma * 2.0 * 32 * 32 * 16 / best / 1e9;
    const double wbytes = (double)wgs * tiles_per_wg * NCH * NE * W_ELEM;
    printf("T=%3d tiles (%s)  level %d  %8.3f ms  executed %7.1f TF  = %6.1f TF fp32-equivalent conv (x2.25/3)  weight stream %5.2f TB/s"
           "  [err %d]\n", T, MI == 2 ? "16x16 px" : MI == 1 ? " 8x16 px" : "16x32 px", LEVEL, best, exec_tf, exec_tf * 2.25 / 3.0,
           wbytes / best / 1e9, (int)hipGetLastError());
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256 * 8, tiles = argc > 2 ? atoi(argv[2]) : 8;
    const size_t wbytes = (size_t)NCH * NE * W_ELEM;
    std::vector<_Float16> hw(wbytes / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (_Float16)((float)((int)((i * 2654435761u) >> 20 & 4095) - 2048) * (8.0f));   // ~2^13-scaled weights
    unsigned char* W;
    float* out;
    hipMalloc(&W, wbytes);
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipMemcpy(W, hw.data(), wbytes, hipMemcpyHostToDevice);
    float* X;                                                        // LEVEL 2's raw patches: 20 KB per (workgroup, tile, chunk), read once
    const size_t xbytes = (size_t)wgs * (tiles * 2) * NCH * 5 * 256 * 16;
    hipMalloc(&X, xbytes);
    hipMemset(X, 0, xbytes);
    printf("# wino_feed_probe: %d workgroups x %d tiles, Cin = Cout-block = 128, weights %zu KB (L2-resident), direct x3h kernel: 352 TF fp32-equivalent (1056 executed)\n",
           wgs, tiles, wbytes >> 10);
    run<1, 0>(W, X, out, wgs, tiles * 2);
    run<1, 1>(W, X, out, wgs, tiles * 2);
    run<1, 2>(W, X, out, wgs, tiles * 2);
    run<2, 0>(W, X, out, wgs, tiles);
    run<2, 1>(W, X, out, wgs, tiles);
    // (T = 64 at LEVEL 2 does not fit: Y 128 + M 64 + fragment rings 48 leave no arch VGPRs for the input side — LEVEL 1 alone allocates
    //  256 + 34; the compiler's AGPR-spill rewrite crashes on it)
    run<4, 0>(W, X, out, wgs, tiles / 2);          // T = 128: needs 512 accumulator registers for Y alone at LEVEL 1 — feed ceiling only
    return 0;
}
