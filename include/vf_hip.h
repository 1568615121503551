/*
 * vf_hip.h — C-ABI of libvf_hip.so: the MI355X (gfx950) kernels of the ViewFormer
 * novel-view hot path (VQ-VAE codebook encode -> image-token transformer -> decode).
 *
 * The reference (jkulhanek/viewformer) has no FFI/plugin layer: its "operators" are
 * PyTorch / TensorFlow framework calls.  Each entry point below replaces the framework
 * op call site(s) cited next to it (paths relative to the reference root).  The Python
 * host side (viewformer_amd/) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless it says host
 *   - caller owns all buffers; no hidden allocation; workspaces are passed in and sized
 *     by the matching *_workspace_bytes / *_packed_floats query (host-only, no GPU needed)
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *   - returns 0 (VF_OK) on success, <0 for argument errors, >0 = hipError_t of the launch;
 *     never throws across the ABI; thread-safe for distinct streams
 *   - activations are channels-last fp32 ("NHWC": [image][y][x][channel]); token rows are
 *     [row][feature] fp32; codes are int64 like the reference's Torch path
 *   - arithmetic: every entry point names its own.  The fp32-EQUIVALENT families are the
 *     native f32 MFMA (*_f32: v_mfma_f32_32x32x2_f32 = a k-ordered fmaf chain), x6 (*_x6:
 *     three exact bf16 pieces, six products) and x3h (*_x3h: two exact fp16 pieces, three
 *     products — the default of the inference encoder); they differ from the fp32
 *     reference by summation order only.  *_bf16 / *_fp8 entry points round
 *     their operands once and are the tolerance arms.
 */
#ifndef VF_HIP_H
#define VF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VF_OK 0
#define VF_ERR_BAD_ARG (-1)
#define VF_ERR_UNSUPPORTED (-2)

/* library / device introspection (host) */
int vf_abi_version(void);                 /* bumps when a signature changes */
size_t vf_sizeof_igemm_args(void);        /* sizeof(vf_igemm_args) / sizeof(vf_pack_desc) as this library was built: a binding compares */
size_t vf_sizeof_pack_desc(void);         /* its mirror of the two structs with them before the first call */
const char* vf_build_arch(void);          /* "gfx950" */
/* developer switches compiled into this library (ablation / cycle-stamp builds of the kernels, csrc/vf_common.h): the count and
 * the i-th macro name.  A product build returns 0; tests/test_abi.py asserts it of the shipped library. */
int vf_build_flags(void);
const char* vf_build_flag_name(int i);
/* Kernel selection for A/B runs and parity tests.  The library reads NO environment variable; the only run-time switches are these, and each
 * chooses between two kernels whose results the tests assert bit-identical (tests/test_hip_bf16.py, tests/test_train.py) — with TWO exceptions
 * that change numerics at tolerance level: VF_SEL_CONV_X3H_K32 (two MFMA shapes: same fp32-equivalence bound, same reference tokens) and
 * VF_SEL_ATTN_DMA (since round 4 the LDS-DMA kernel pre-scales q by scale*log2(e) and re-rounds it to bf16 and keeps a lazily-updated
 * reference maximum: within 1e-2 * max|out| of the register-staged kernel, tests/test_hip_parity_scale.py; both inside the bf16 arm's
 * stated tolerance).  Training note (round 6): the dQ kernel of the bf16 flash backward re-materialises P from the forward's own operand
 * q' = bf16(q scale log2 e) (the same S product bit for bit); the dK / dV kernel streams q un-rounded (dK = dS^T.Q needs it), so its P differs
 * from the forward's by that one rounding of q — |s| 2^-9 in the exponent; both stay inside the bound tests/test_train.py states at the score
 * magnitudes of a trained model (|s| <= 40: test_bf16_flash_attention_backward_at_trained_scale_scores).
 * Process-wide, atomic; vf_select returns the previous value (or VF_ERR_BAD_ARG), value in {0, 1}; all default to 1 except VF_SEL_GEMM_TAIL (0). */
enum { VF_SEL_ATTN_DMA = 0,          /* vf_attn_blockcausal_bf16_v2: 1 = LDS-DMA ring kernel where it applies, 0 = register-staged kernel (tolerance-level pair) */
       VF_SEL_GEMM_G256 = 1,         /* vf_gemm_bf16: 1 = 256-tile LDS-DMA kernel where it applies, 0 = 128-tile kernel */
       VF_SEL_LN_BWD_TWO_ROWS = 2,   /* vf_layernorm_bwd_f32: 1 = two rows of a wave in flight, 0 = one */
       VF_SEL_ATTN_Q32 = 3,          /* the LDS-DMA attention kernel: 1 = 8 waves x 32 queries per workgroup, 0 = 4 waves x 64 queries */
       VF_SEL_CONV_X3H_K32 = 4,      /* vf_conv3_halo_x3h, stride 1: 1 = the v_mfma_f32_16x16x32_f16 kernel where it applies, 0 = the 32x32x16 kernel.  The ONE
                                      * switch whose two sides differ in the last bits (another accumulation order; same fp32-equivalence bound) */
       VF_SEL_GEMM_TAIL = 5,         /* the 256-tile bf16 GEMM: 1 = launches whose last round of the 256 CUs would be a few tiles to 3/4 full end with ONE round of
                                      * 192- / 128-row tail tiles instead (K is never split: bit-identical results), 0 = the plain grid (the DEFAULT: the policy is
                                      * 2-5 % faster on an isolated launch and nothing in the timed steps, round 6) */
       VF_SEL_COUNT = 6 };
int vf_select(int which, int value);
int vf_selected(int which);

/* ---------------------------------------------------------------------------------------
 * Implicit-GEMM family: conv3x3 (stride 1 / stride-2 with (0,1,0,1) pad / nearest-x2
 * upsample fused), conv1x1 and dense layers, all as
 *     out[m][n] = epi( sum_{tap,c} pro(A_tap[m][c]) * W[tap][c][n] + bias[n] ) + res[m][n]
 * Replaces: torch.nn.Conv2d call sites of viewformer/models/vqgan_th.py:23-27,39-49,
 * 60-76,99-118,159,197,249,285,332-333 (F.pad :46-47 and F.interpolate :30 are folded
 * into the gather), Conv1D.call viewformer/models/migt.py:89-96 (x @ W + b),
 * SharedEmbeddings._linear migt.py:51-56 (h @ wte^T), tf.nn.gelu migt.py:13,70 (epilogue),
 * residual adds vqgan_th.py:90,144 and migt.py:233,237, and — through the prologue —
 * the GroupNorm(32, eps 1e-6)+swish of vqgan_th.py:11-17,80-85,122,222-223,315-316.
 * ------------------------------------------------------------------------------------- */
enum { VF_MODE_GEMM = 0, VF_MODE_CONV3_S1 = 1, VF_MODE_CONV3_S2PAD = 2, VF_MODE_CONV3_UP2 = 3 };
enum { VF_EPI_NONE = 0, VF_EPI_GELU_ERF = 1,
       VF_EPI_GELU_BWD = 2,     /* vf_gemm_bf16 with bf16 output only: out = bf16(acc * gelu'(res[m][n])) — `res` carries the saved fp32
                                   pre-activation, not a residual: the GELU backward of the training step inside the dX GEMM that
                                   produces its input (migt.py:70 under autograd) */
       VF_EPI_GELU_DUAL = 3 };  /* vf_gemm_bf16, bf16 activations in, 256-aligned shapes only: out = acc + bias (fp32, or bf16 with the
                                   bf16-output flag) AND out_aux = bf16(gelu(acc + bias)) — see vf_igemm_args.out_aux.
                                   vf_gemm_bf16's dtype flags live in reserved0: bit 0 bf16 activations in, bit 1 bf16 out, bit 2
                                   (VF_EPI_GELU_BWD only) the pre-activation behind `res` is bf16 [M][ldr]; bit 3 (round 6, 256-tile shapes, bf16
                                   out): the SAVED-DERIVATIVE forms — with VF_EPI_GELU_DUAL `out` receives bf16(gelu'(acc + bias)) instead of the
                                   pre-activation (from the erf / exp evaluation of the GELU beside it, whose bits do not change), with
                                   VF_EPI_GELU_BWD (bit 2 set) `res` holds that table and out = bf16(acc * res[m][n]): the backward's epilogue
                                   loads what the forward already evaluated (VF_ERR_BAD_ARG on other combinations) */

typedef struct vf_igemm_args {
    const float* x;          /* GEMM: [M][lda]; conv: NHWC [Nimg][Hin][Win][Cin] */
    const float* w_packed;   /* from vf_igemm_pack_f32 */
    const float* bias;       /* [Cout] or NULL */
    const float* res;        /* residual [M][ldr] added after the epilogue, or NULL */
    float* out;              /* [M][ldc] (conv: NHWC [Nimg][Hout][Wout][Cout] with ldc=Cout) */
    const float* pro_mean;   /* prologue (GroupNorm apply), all three NULL = off:          */
    const float* pro_scale;  /*   a = (x - mean[img][c]) * scale[img][c] + beta[c]           */
    const float* pro_beta;   /*   (mean/scale [Nimg][Cin] from vf_groupnorm_stats_f32)       */
    int32_t pro_swish;       /* 1: a = a * sigmoid(a) after the affine                      */
    int32_t pro_rows_per_img;/* GEMM mode: rows (pixels) per image for the prologue lookup   */
    int32_t mode;            /* VF_MODE_*                                                   */
    int32_t epilogue;        /* VF_EPI_*                                                    */
    int32_t M;               /* output rows = Nimg*Hout*Wout (conv) */
    int32_t Cin, Cout;       /* Cin must be a multiple of 32 */
    int32_t Hin, Win, Hout, Wout;   /* conv modes only */
    int32_t lda, ldc, ldr;   /* row strides in floats (conv: lda is ignored, = Cin) */
    int32_t batch;           /* >=1: independent problems, pointer strides below (floats) */
    int64_t stride_x, stride_w, stride_out, stride_res;
    /* optional fused GroupNorm statistics of the OUTPUT (vf_conv3_halo_x6 / vf_conv3_halo_bf16 only, others refuse a
     * non-NULL pointer): per-(image, slot, group) partial {sum, sum of squares} of the stored values, laid out
     * [Nimg][gn_slots][32][2] exactly like vf_groupnorm_stats_f32's workspace, to be reduced by
     * vf_groupnorm_finalize_f32.  gn_slots must equal vf_conv3_halo_gn_slots(Hout, Wout). */
    float* gn_part;
    int32_t gn_slots;
    int32_t reserved0;       /* vf_gemm_x6 only: split-K count S > 1 -> S raw partial slabs at out + s*stride_out (no bias /
                              * residual / epilogue), to be summed by vf_sum_slabs_f32; 0 or 1 = off */
    void* out_aux;           /* VF_EPI_GELU_DUAL only (vf_gemm_bf16, bf16 activations in, fp32 or bf16 out): bf16 [M][ldc] that receives
                              * gelu(out) beside the fp32 pre-activation `out` — the training forward's c_fc keeps u for the backward
                              * pass and hands f to mlp.c_proj from ONE epilogue.  NULL otherwise (other entry points refuse it). */
    /* fused OUTPUT dropout of the training step (resid_dropout migt.py:216, the MLP's migt.py:72): vf_gemm_bf16 only, bf16 activations in,
     * fp32 out, VF_EPI_NONE, shapes of the 256-tile kernel (VF_ERR_UNSUPPORTED otherwise: run the GEMM, then vf_dropout_add_f32):
     *     out[m][n] = keep(m, n) ? (acc + bias[n]) / (1 - drop_rate) : 0   (+ res[m][n]),
     * keep as vf_dropout_add_f32 with cols = Cout and row0 = drop_row0 (mask group ((m + drop_row0) >> 2) * Cout + n, position m & 3:
     * csrc/vf_common.h; drop_row0 % 4 == 0 — the first row's index in the GLOBAL batch of a data-parallel step).  drop_rate 0 = off;
     * every other entry point refuses a non-zero rate. */
    float drop_rate;
    uint32_t drop_seed, drop_site;
    int32_t drop_row0;
} vf_igemm_args;

/* floats needed for the packed form of a [taps][K][N] weight (K,N padded to the tile) */
size_t vf_igemm_packed_floats(int K, int N, int taps);
/* pack src (element (tap,k,n) at src[tap*st + k*sk + n*sn]) into the MFMA-fragment-major
 * layout; OIHW conv weight: sk=taps, sn=K*taps, st=1; Conv1D [K][N]: sk=N, sn=1, st=0;
 * transposed [N][K] (tied LM head): sk=1, sn=K, st=0.  batch>1 packs `batch` matrices
 * (src stride src_bstride floats, dst stride = vf_igemm_packed_floats). */
int vf_igemm_pack_f32(const float* src, float* dst, int K, int N, int taps,
                      int64_t sk, int64_t sn, int64_t st, int batch, int64_t src_bstride, void* stream);
int vf_igemm_f32(const vf_igemm_args* args /* host */, void* stream);

/* conv_in special case: uint8 NHWC image -> x*(1/255)*2-1 -> conv3x3 (3 -> Cout), fp32 NHWC out.
 * Replaces evaluate_transformer.py:105-108 (convert_image_dtype, *2-1) + vqgan_th.py:159,205.
 * w is the plain OIHW [Cout][3][3][3] weight.  If img_f32 != NULL it is used instead of img_u8
 * (already-normalised NHWC float input, the Torch-convention entry). */
int vf_conv_in_u8_f32(const uint8_t* img_u8, const float* img_f32, const float* w_oihw, const float* bias,
                      float* out, int n_img, int H, int W, int Cout, void* stream);
/* the same entry on the fp16 matrix pipe with the x3h arithmetic (csrc/conv_in_x3h.hip: fp32-equivalent, inputs in [-1, 1]); the
 * weights are packed once (vf_conv_in_x3h_pack, vf_conv_in_x3h_packed_elems(Cout) f16 elements); optional fused GroupNorm partial
 * statistics of the output as in vf_conv3_halo_x6 (gn_part / gn_slots, NULL / 0 = off).  Needs H % 8 == 0, W % 16 == 0,
 * Cout % 128 == 0 (VF_ERR_UNSUPPORTED otherwise: call vf_conv_in_u8_f32). */
size_t vf_conv_in_x3h_packed_elems(int Cout);
int vf_conv_in_x3h_pack(const float* w_oihw, void* dst, int Cout, void* stream);
int vf_conv_in_x3h(const uint8_t* img_u8, const float* img_f32, const void* w_packed, const float* bias, float* out, float* gn_part,
                   int gn_slots, int n_img, int H, int W, int Cout, void* stream);

/* 3x3 stride-1 pad-1 convolution to 1..4 output channels (the decoder's conv_out, vqgan_th.py:285-289,316-318) with
 * the GroupNorm-apply(+swish) of the preceding norm_out (:313-315) fused; plain fp32 fmaf arithmetic.  x NHWC
 * [n_img][H][W][Cin], w OIHW, out NHWC [n_img][H][W][Cout].  Cin % 32 == 0, H % 8 == 0, W % 32 == 0.  x_bf16 != 0: x is a bf16 NHWC
 * activation (the bf16-activation decoder: vf_conv3_halo_bf16 with reserved0 bit 1). */
int vf_conv3_small_cout_f32(const float* x, const float* w_oihw, const float* bias, const float* pro_mean,
                            const float* pro_scale, const float* pro_beta, int pro_swish, float* out, int n_img, int H, int W,
                            int Cin, int Cout, int x_bf16, void* stream);

/* ---------------------------------------------------------------------------------------
 * GroupNorm(32 groups) statistics.  Replaces torch.nn.GroupNorm vqgan_th.py:16-17.
 * Produces mean_c/scale_c [Nimg][C] (scale = rstd*gamma) consumed by the igemm prologue or
 * by vf_groupnorm_apply_f32.  ws: vf_groupnorm_workspace_bytes(...) bytes.
 * ------------------------------------------------------------------------------------- */
size_t vf_groupnorm_workspace_bytes(int n_img, int HW, int C);
int vf_groupnorm_stats_f32(const float* x, const float* gamma, int n_img, int HW, int C, int groups, float eps,
                           float* mean_c, float* scale_c, void* ws, void* stream);
/* second half of vf_groupnorm_stats_f32 alone: reduce [Nimg][nslots][groups][2] partial {sum, sumsq} (fp64, fixed order)
 * written by a producer's fused epilogue (vf_igemm_args.gn_part) into mean_c / scale_c.  The activation is not re-read. */
int vf_groupnorm_finalize_f32(const float* part, const float* gamma, int n_img, int HW, int C, int groups, int nslots,
                              float eps, float* mean_c, float* scale_c, void* stream);
int vf_groupnorm_apply_f32(const float* x, const float* mean_c, const float* scale_c, const float* beta,
                           float* out, int n_img, int HW, int C, int swish, void* stream);

/* ---------------------------------------------------------------------------------------
 * Codebook lookup.  Replaces QuantizeEMA.forward (eval) viewformer/models/utils_th.py:32-44:
 *   dist = sum(z^2) - 2 z@E + sum(E^2);  idx = first argmax(-dist)  (ties -> lowest index)
 * z [M][D] fp32 rows (the NHWC flatten of :34-35), E packed by vf_igemm_pack_f32(K=D, N=Kc,
 * sk=Kc, sn=1) — use vf_vq_pack_codebook_f32 — from the reference's `embeddings` [D][Kc]; e_sq [Kc]
 * from vf_colsumsq_f32.
 * The [M][Kc] distance matrix is never written.  idx int64 [M].
 * ------------------------------------------------------------------------------------- */
size_t vf_vq_packed_floats(int D, int Kc);
int vf_vq_pack_codebook_f32(const float* E /* [D][Kc] */, float* dst, int D, int Kc, void* stream);
int vf_colsumsq_f32(const float* E /* [D][Kc] */, float* e_sq /* [Kc] */, int D, int Kc, void* stream);
int vf_vq_argmin_f32(const float* z, const float* E_packed, const float* e_sq, int64_t M, int D, int Kc,
                     int64_t* idx, void* stream);
/* The same lookup — same contract, same indices bit for bit (csrc/vq_filter.hip) — as a 16-bit candidate filter on the fp16 matrix
 * pipe (16x the f32 MFMA rate) followed by an exact fp32 re-rank of the few codes inside a proven error window: dist of every
 * candidate is re-evaluated in vf_vq_argmin_f32's own arithmetic, rows the filter cannot certify are scanned exactly.  D must be 256,
 * Kc % 32 == 0, Kc <= 1024 (VF_ERR_UNSUPPORTED otherwise: call vf_vq_argmin_f32).  packed: vf_vq_filter_packed_bytes(D, Kc) bytes
 * written by vf_vq_filter_pack from the reference's `embeddings` [D][Kc].  stats4 (NULL or 4 zero-initialised uint32): rows decided
 * by the filter alone / rows re-ranked / exact distances evaluated in the re-rank / rows scanned over the whole codebook. */
size_t vf_vq_filter_packed_bytes(int D, int Kc);
int vf_vq_filter_pack(const float* E /* [D][Kc] */, void* dst, int D, int Kc, void* stream);
int vf_vq_argmin_filtered_f32(const float* z, const void* packed, int64_t M, int D, int Kc, int64_t* idx, uint32_t* stats4,
                              void* stream);
/* embed_code (utils_th.py:70-72): out[m][:] = E[:, idx[m]]  (NHWC rows) */
int vf_codebook_gather_f32(const float* E /* [D][Kc] */, const int64_t* idx, float* out /* [M][D] */,
                           int64_t M, int D, int Kc, void* stream);

/* ---------------------------------------------------------------------------------------
 * Attention.
 * vf_attn_blockcausal_f32 replaces compute_causal_block_attention + compute_attention
 * (viewformer/models/branching_attention.py:41-61,5-18) as used by the single-stream
 * inference graph (:82-92): scores q.k^T * scale (scale = 1: the reference has NO 1/sqrt(d)),
 * masked entries (view(q) < view(k), view = token / L) take the value -1e4 exactly as
 * `w*m - 1e4*(1-m)`, softmax over keys, times v.  q/k/v/out are [B][T][ld] rows with the head h
 * at column offset h*64 (dh must be 64): it reads the fused c_attn output in place (V|Q|K thirds,
 * migt.py:207-213) and writes merge_heads layout (migt.py:195-199).
 * skip_masked=1 skips key tiles that are masked for the whole query tile (their softmax weight
 * underflows to exactly 0.0f whenever the row max exceeds -1e4+104); 0 = dense reference form.
 * L=0 disables the mask (plain softmax attention).
 * twin_view = Vc >= 0: views Vc, Vc+1, ... are alternative endings of the same sequence position (each
 * sees views < Vc and itself, never a sibling) — the reference's branch streams
 * (branching_attention.py:94-125), used to run the evaluator's generation pass (MASK view) and
 * localization pass (LOC view, evaluate_transformer.py:119-123,134-136) as ONE pass over S+1 views
 * with bit-identical rows.  -1 = plain block-causal.
 * twin_view = -Sv <= -2: STREAMS mode, Sv views per stream, view index = stream*Sv + position: stream 0 is the
 * main block-causal sequence, a branch stream s >= 1 at position i sees main views j < i and its own (s,i)
 * tile — compute_causal_block_multiend_attention (branching_attention.py:82-126) for all streams in one launch
 * (multi-context evaluators evaluate_transformer_multictx.py:60-77; forward of the training graph migt.py:392-401).
 * ------------------------------------------------------------------------------------- */
int vf_attn_blockcausal_f32(const float* q, const float* k, const float* v, float* out,
                            int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                            float scale, int skip_masked, int twin_view, void* stream);
/* same contract on the bf16 matrix pipe (Q, K, V and the probabilities rounded to bf16, fp32 sums and softmax): the tolerance-bounded
 * transformer arm (csrc/attention_lp.hip; with bf16 q / k / v / out, 64-token views and skip_masked the LDS-DMA ring kernel of
 * csrc/attention_dma.hip, bit-identical).  in_bf16: 0 = fp32 q/k/v, 1 = bf16 (ld* in elements, % 8); out_bf16: 0 = fp32 out, 1 = bf16 out
 * (ldo in elements) for a bf16-GEMM consumer.  A wave owns 64 queries as two MFMA tiles sharing every K / V fragment, 5 VALU per score in
 * the softmax.  (The "_v2" is historical: round 1's first kernel, vf_attn_blockcausal_bf16, left the library in round 4 —
 * tools/variants/attention_bf16_record.hip.) */
int vf_attn_blockcausal_bf16_v2(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L,
                                int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view, void* stream);
/* the same kernel with OCP e4m3 operands on v_mfma_f32_32x32x16_fp8_fp8 (BASELINE configs[4] "fp8 MFMA attention"): Q, K, V clamped to
 * +-448 and rounded to e4m3, probabilities carried at 2^8 and rounded to e4m3, fp32 sums and softmax; tolerances in
 * tests/test_hip_fp8.py (the reference's un-scaled logits, branching_attention.py:7, are what makes this arm loose) */
int vf_attn_blockcausal_fp8(const void* q, const void* k, const void* v, int in_bf16, void* out, int out_bf16, int B, int H, int T, int L,
                            int ldq, int ldk, int ldv, int ldo, float scale, int skip_masked, int twin_view, void* stream);
/* same contract, fp32-EQUIVALENT on the bf16 pipe (x6: every operand split into three bf16 pieces, six partial products per
 * fp32 product, fp32 softmax) — the default attention of the fp32 transformer arm */
int vf_attn_blockcausal_x6(const float* q, const float* k, const float* v, float* out,
                           int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                           float scale, int skip_masked, int twin_view, void* stream);
/* Single-head spatial self-attention of the VQGAN AttnBlock, fused (csrc/attn_spatial.hip): replaces the core of AttnBlock.forward
 * (vqgan_th.py:124-141) — scores = q^T k * scale, softmax over the keys, h = v . p^T — per image of HW tokens x C channels, from the
 * fused q|k|v projection qkv [n_img * HW][ld] (q at column 0, k at C, v at 2C); out [n_img * HW][ldo].  Exact fp32
 * (v_mfma_f32_32x32x2_f32 + libm expf); the [HW][HW] score matrix never leaves the CU.  (HW, C) in {(256, 256), (64, 512), (64, 256)}
 * (VF_ERR_UNSUPPORTED otherwise: batched vf_igemm_f32 + vf_softmax_rows_f32). */
int vf_attn_spatial_f32(const float* qkv, float* out, int n_img, int HW, int C, int64_t ld, int64_t ldo, float scale, void* stream);
/* The same contract in fp32-EQUIVALENT "x3h" arithmetic on the fp16 matrix pipe (two fp16 pieces per operand, three exact products; see
 * vf_conv3_halo_x3h): what the encoder's AttnBlocks run under conv_arith = 'x3h'.  Error against fp64 not above vf_attn_spatial_f32's. */
int vf_attn_spatial_x3h(const float* qkv, float* out, int n_img, int HW, int C, int64_t ld, int64_t ldo, float scale, void* stream);
/* row softmax with scale (VQGAN AttnBlock, vqgan_th.py:132-134): x[r][0:n] in place */
int vf_softmax_rows_f32(float* x, int64_t rows, int n, float scale, void* stream);

/* ---------------------------------------------------------------------------------------
 * Transformer glue.
 * ------------------------------------------------------------------------------------- */
/* LayerNormalization(eps) over the last dim (migt.py:225,227,292) */
int vf_layernorm_f32(const float* x, const float* gamma, const float* beta, float* out,
                     int64_t rows, int d, float eps, void* stream);
/* the same with a bf16 output row (rounded to nearest even, exactly as a bf16-MFMA consumer would round the fp32 value on load) */
int vf_layernorm_bf16out_f32(const float* x, const float* gamma, const float* beta, void* out_bf16, int64_t rows, int d, float eps,
                             void* stream);
/* h0[b][s][l][:] = wte[ids[b][s][l]] + wpe[l] + add[b][s][:]   (migt.py:358-368,392) */
int vf_embed_sum_f32(const int32_t* ids, const float* wte, const float* wpe, const float* add,
                     float* out, int64_t BS, int L, int d, int vocab, void* stream);
/* pose MLP first layer, K=7: out[r][j] = gelu(sum_i x[r][i]*W[i][j] + b[j])  (migt.py:291,354,70) */
int vf_dense_small_k_gelu_f32(const float* x, const float* W, const float* b, float* out,
                              int64_t rows, int K, int N, int gelu, void* stream);
/* first-max index over each row of n floats (tf.argmax, evaluate_transformer.py:123; ties -> lowest) */
int vf_argmax_rows_f32(const float* x, int64_t rows, int n, int ld, int64_t* idx, void* stream);
/* tied LM head with the arg-max fused into its epilogue (bf16 arm; csrc/lmhead_argmax.hip): idx[m] = first arg-max over n < N of
 * sum_k bf16(h[m][k]) * bf16(W[n][k]) — SharedEmbeddings._linear (migt.py:51-56) + the :417 slice + tf.argmax
 * (evaluate_transformer.py:123) without writing the [M][N] logits.  h: fp32 rows (h_bf16 = 0) or bf16 rows (1), ldh in elements;
 * w_packed = vf_gemm_bf16_pack of wte[:N] (sk = 1, sn = K).  Same arithmetic as vf_gemm_bf16 on that packing: the index equals
 * vf_argmax_rows_f32 of its logits.  max_logit (NULL or [M]): the winning logit.  K in {128, 768}, N % 128 == 0
 * (VF_ERR_UNSUPPORTED otherwise: compute the logits and call vf_argmax_rows_f32). */
int vf_lmhead_argmax_bf16(const void* h, int h_bf16, int64_t ldh, const void* w_packed, int64_t M, int K, int N, int64_t* idx,
                          float* max_logit, void* stream);
/* host-side CRC-32C (Castagnoli) of a HOST buffer, for the TFRecord / TensorBundle files of the reference's datasets and
 * Keras checkpoints (viewformer_amd/codes_dataset.py, checkpoint.py); crc = 0 starts a new checksum */
uint32_t vf_crc32c(const void* data, size_t n, uint32_t crc);
/* clip[-1,1] -> /2+0.5 -> trunc(x*255.5) uint8  (evaluate_transformer.py:128-129, TF semantics) */
int vf_postprocess_u8(const float* x, uint8_t* out, int64_t n, void* stream);
/* resize of the evaluators' pre-process (viewformer/data/_common.py:19-61 resize / resize_th): NHWC uint8 [n][Hin][Win][C] ->
 * [n][Hout][Wout][C]; bilinear = 0: torch 'nearest' (the reference's choice when enlarging), 1: bilinear, align_corners = False
 * (shrinking); through /255, clamp, *255 and a truncating cast exactly as the reference does (bit-identical uint8). */
int vf_resize_u8(const uint8_t* src, uint8_t* dst, int n_img, int Hin, int Win, int Hout, int Wout, int C, int bilinear, void* stream);

/* ---------------------------------------------------------------------------------------
 * Reduced-precision arm (bf16 MFMA, fp32 activations in HBM, fp32 accumulate / epilogue) for the layers whose
 * outputs the north star bounds by a tolerance rather than bit-exactness: the transformer's dense layers and
 * the decoder's convolutions.  Same vf_igemm_args as the exact path; w_packed points to the bf16 packing.
 * vf_gemm_bf16: VF_MODE_GEMM, Cin % 64 == 0, no prologue.
 * vf_conv3_halo_bf16: VF_MODE_CONV3_S1 / _UP2 with the halo-kernel shape rules (Cin % 32, Cout % 128,
 * Wout % 16, Hout % 8), GroupNorm(+swish) prologue, bias, residual.  reserved0 carries dtype flags like vf_gemm_bf16's: bit 1 = out AND
 * res are bf16 NHWC (ldc / ldr in elements, even); bit 0 (only together with bit 1) = x is bf16 NHWC: the decoder's activations stay bf16
 * between its layers (fused GroupNorm partials are taken from the fp32 values before the output rounding).
 * ------------------------------------------------------------------------------------- */
size_t vf_gemm_bf16_packed_elems(int K, int N);            /* number of bf16 elements of the packed weight */
int vf_gemm_bf16_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, int batch,
                      int64_t src_bstride, void* stream);
/* the same packing for many weights in ONE launch: `descs_device` = n descriptors in DEVICE memory (src [K][N] with element strides sk / sn,
 * dst of vf_gemm_bf16_packed_elems(K, N) bf16) — the training step's per-step refresh of every layer's W and W^T packing */
typedef struct vf_pack_desc {
    const float* src;
    void* dst;
    int32_t K, N;
    int64_t sk, sn;
} vf_pack_desc;
int vf_gemm_bf16_pack_multi(const vf_pack_desc* descs_device, int n, void* stream);
int vf_gemm_bf16(const vf_igemm_args* args /* host */, void* stream);
/* vf_gemm_bf16 reads args->reserved0 as dtype flags: bit 0 = x is bf16 [M][lda] (lda in elements, % 8 == 0), bit 1 = out is bf16
 * [M][ldc] (no residual).  Both need Cin % 128 == 0.  For activations that only bf16 GEMMs consume (LayerNorm / GELU / attention
 * outputs): bit-identical results, half the traffic. */
size_t vf_conv3_bf16_packed_elems(int Cin, int Cout);
int vf_conv3_bf16_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream);
int vf_conv3_halo_bf16(const vf_igemm_args* args /* host */, void* stream);

/* ---------------------------------------------------------------------------------------
 * fp32-EQUIVALENT arm on the bf16 matrix pipe ("x6").  Every fp32 operand is split exactly into three bf16
 * pieces (x = h + m + l) and each fp32 product is evaluated as the six partial products whose weight is
 * >= 2^-24 of it, accumulated in fp32 (small terms first).  Error vs fp64 equals the native f32 MFMA's
 * (profiles/r1_split_bf16_probe.txt); this is NOT a reduced-precision path and replaces the same reference
 * call sites as vf_igemm_f32's 3x3 modes (torch.nn.Conv2d in vqgan_th.py:23-32,60-70,197,249).
 * Same vf_igemm_args and shape rules as vf_conv3_halo_bf16; w_packed points to the 3-plane bf16 packing.
 * ------------------------------------------------------------------------------------- */
/* partial-statistics slots per image the halo kernels write for an Hout x Wout map (2 per 8x16 tile; 2 for an 8x8 map) */
int vf_conv3_halo_gn_slots(int Hout, int Wout);
size_t vf_conv3_x6_packed_elems(int Cin, int Cout);       /* number of bf16 elements (3 planes) */
int vf_conv3_x6_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream);
int vf_conv3_halo_x6(const vf_igemm_args* args /* host */, void* stream);
/* the same 3x3 convolution (stride 1 / nearest-x2 upsample; same call sites) with HALF the matrix instructions: operands split
 * into two fp16 pieces with the low piece carried at 2^11 times its value, 3 products, cross terms in their own accumulator, weights
 * pre-scaled by a power of two at pack time (csrc/conv3_halo_x3h.hip).  fp32-equivalent for |x| in [~6e-5, 65504): the arithmetic
 * of the inference encoder (GroupNorm-normalised / O(1) activations); x6 has no range condition and stays the training arithmetic. */
size_t vf_conv3_x3h_packed_elems(int Cin, int Cout);      /* number of f16 elements (2 planes + the 1/S tail) */
int vf_conv3_x3h_pack(const float* w_oihw, void* dst, int Cin, int Cout, void* stream);
int vf_conv3_halo_x3h(const vf_igemm_args* args /* host */, void* stream);
/* dense / 1x1 sibling (csrc/gemm_x3h.hip): same contract as vf_gemm_x6 incl. split-K (reserved0), half the matrix instructions;
 * forward activations only (range condition above) */
size_t vf_gemm_x3h_packed_elems(int K, int N);
int vf_gemm_x3h_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, void* stream);
int vf_gemm_x3h(const vf_igemm_args* args /* host */, void* stream);
/* dense / 1x1 sibling: VF_MODE_GEMM, Cin % 64 == 0, batch == 1, optional GroupNorm(+swish) prologue, bias / exact-erf GELU /
 * residual epilogue.  Replaces the same call sites as vf_igemm_f32's GEMM mode (Conv1D.call migt.py:89-96,
 * SharedEmbeddings._linear :51-56, the 1x1 convolutions of vqgan_th.py:72-76,99-118,332-333).
 * pack: element (k, n) of the weight at src[k*sk + n*sn] (Conv1D [K][N]: sk=N, sn=1; [N][K]: sk=1, sn=K). */
size_t vf_gemm_x6_packed_elems(int K, int N);             /* number of bf16 elements (3 planes) */
int vf_gemm_x6_pack(const float* src, void* dst, int K, int N, int64_t sk, int64_t sn, void* stream);
int vf_gemm_x6(const vf_igemm_args* args /* host */, void* stream);
/* dst[i] (+)= sum_s slabs[s*stride + i], s ascending (fixed order): the reduction of a split-K GEMM */
int vf_sum_slabs_f32(const float* slabs, int nslabs, int64_t stride, int64_t n, float* dst, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training branch of the codebook quantizer: the EMA codebook update of QuantizeEMA.forward (utils_th.py:46-64).
 * accumulate: counts[k] = #rows with idx == k (:47), embed_sum[d][k] = sum of those rows of z (:48) — rows added in index order,
 * no atomics (bit-reproducible).  The caller all-reduces counts / embed_sum over replicas (:50-52), then
 * update: EMA buffers (:55-56), bias correction corr = 1 - decay^counter (:24-30), Laplace smoothing (:59-62),
 * embeddings = ema_dw / cluster_size (:63-64).  E, embed_sum, dw_hidden are [D][Kc] like the reference's buffers.
 * ------------------------------------------------------------------------------------- */
int vf_vq_ema_accumulate_f32(const float* z, const int64_t* idx, int64_t M, int D, int Kc, float* counts, float* embed_sum,
                             void* stream);
int vf_vq_ema_update_f32(const float* counts, const float* embed_sum, float* cluster_size_hidden, float* dw_hidden,
                         float* embeddings, int D, int Kc, float decay, float eps, float corr, void* stream);

/* ---------------------------------------------------------------------------------------
 * Backward-pass helpers of the codebook (VQGAN) training step (vqgan_th.py:349-368,427-429).  The contractions of that backward
 * pass are launches of the forward conv / GEMM entry points on re-packed operands; these are the remaining HBM-bound pieces.
 * ------------------------------------------------------------------------------------- */
/* dst[c][p] = src[img][y*stride+oy][x*stride+ox][c] (0 outside), p = (img,y,x) over Hout x Wout: the tap-shifted channel-major
 * activation that makes conv dW one GEMM per tap */
int vf_gather_transpose_f32(const float* src, float* dst, int n_img, int Hin, int Win, int C, int Hout, int Wout, int stride, int oy,
                            int ox, int64_t ld_dst, void* stream);
/* weight + bias gradient of a 3x3 convolution (torch.nn.Conv2d backward of vqgan_th.py's ResnetBlock / Downsample / Upsample convs) as
 * ONE split-bf16 GEMM gathered straight from the NHWC input x [n][Hin][Win][Cin] (mode = VF_MODE_CONV3_*: the forward conv's mode):
 *   slabs[s][(ky*3+kx)*Cin + ci][co] = partial sums over output pixels of x[tap-shifted][ci] * dY[p][co],  row 9*Cin = sum_p dY[p][co]
 * dy_packed = vf_gemm_x6_pack of dY [P][Cout] (K = P); sum the `splits` slabs of vf_conv3_wgrad_x6_rows(Cin) * Cout floats with
 * vf_sum_slabs_f32.  Needs Cin % 128 == 0, Hout / Wout powers of two, P % 64 == 0 (VF_ERR_UNSUPPORTED otherwise). */
size_t vf_conv3_wgrad_x6_rows(int Cin);
int vf_conv3_wgrad_x6(const float* x, const void* dy_packed, float* slabs, int n_img, int Hin, int Win, int Cin, int Hout, int Wout,
                      int Cout, int mode, int splits, void* stream);
/* nearest-x2 upsample backward (Upsample.forward vqgan_th.py:29-32): dx = 2x2 block sums of du [n][2H][2W][C] */
int vf_upsample2_bwd_f32(const float* du, float* dx, int n_img, int H, int W, int C, void* stream);
/* GroupNorm(+swish) backward (Normalize / nonlinearity vqgan_th.py:11-17): dx (+=), chan_sums [n_img][C][2] = {dgamma, dbeta} parts */
size_t vf_groupnorm_bwd_workspace_bytes(int n_img, int HW, int C, int groups);
int vf_groupnorm_bwd_f32(const float* x, const float* da, const float* mean_c, const float* scale_c, const float* gamma,
                         const float* beta, float* dx, float* chan_sums, int n_img, int HW, int C, int groups, int swish,
                         int accumulate, void* ws, void* stream);
/* row softmax backward (AttnBlock vqgan_th.py:132-134): dp <- scale * p * (dp - sum p dp) in place */
int vf_softmax_rows_bwd_f32(const float* p, float* dp, int64_t rows, int n, float scale, void* stream);
/* L1 reconstruction loss (vqgan_th.py:355,361): partial sums of |y - x| (vf_l1_loss_partials(n) of them) and dy = sign(y-x)*w */
int vf_l1_loss_partials(int64_t n);
int vf_l1_loss_f32(const float* x, const float* y, float* dy, float* part, int64_t n, float grad_weight, void* stream);

/* ---------------------------------------------------------------------------------------
 * Perceptual loss of the codebook training step: lpips.LPIPS(net='vgg') as called at vqgan_th.py:337,402-404 (and, forward only,
 * the LPIPSMetric of the evaluators, evaluate_transformer.py:34).  The `lpips` package (v0.1.x) is a third-party dependency that is
 * not part of the reference tree; its published algorithm: ScalingLayer (x - shift) / scale, VGG-16 features tapped after
 * relu1_2 / 2_2 / 3_3 / 4_3 / 5_3, per-pixel channel normalisation x / (|x| + 1e-10), squared difference, non-negative 1x1 "lin"
 * weights, spatial mean, sum over the five taps.  The VGG convolutions are vf_igemm_f32 / vf_conv3_halo_x6 launches; these are
 * the remaining HBM-bound pieces (NHWC, fp32).
 * ------------------------------------------------------------------------------------- */
/* ScalingLayer: y = (x - shift[c]) / scale[c] on [npix][3]; backward = 1: y = x / scale[c] (shift3 / scale3 are HOST arrays) */
int vf_lpips_scaling_f32(const float* x, float* y, int64_t npix, const float* shift3, const float* scale3, int backward, void* stream);
/* in-place ReLU, and its backward dy <- dy * (y > 0) with y the ReLU OUTPUT; n % 4 == 0 */
int vf_relu_f32(float* x, int64_t n, void* stream);
int vf_relu_bwd_f32(float* dy, const float* y, int64_t n, void* stream);
/* 2x2 stride-2 max-pool, x [n][2Hout][2Wout][C] -> y [n][Hout][Wout][C] (C % 4 == 0); backward routes dy to the first maximum of
 * each window in row-major order (torch.nn.MaxPool2d) and writes every element of dx */
int vf_maxpool2_f32(const float* x, float* y, int n_img, int Hout, int Wout, int C, void* stream);
int vf_maxpool2_bwd_f32(const float* x, const float* dy, float* dx, int n_img, int Hout, int Wout, int C, void* stream);
/* LPIPS head of one tap: part[blk * n_img + img] = partial sums over pixels of sum_c w[c] (f0n - f1n)^2 (vf_lpips_head_blocks(HW)
 * blocks per image); backward: df1 (+)= gscale * d/df1 of the per-pixel value (f0, w are constants) */
int vf_lpips_head_blocks(int HW);
int vf_lpips_head_f32(const float* f0, const float* f1, const float* w, float* part, int n_img, int HW, int C, void* stream);
int vf_lpips_head_bwd_f32(const float* f0, const float* f1, const float* w, float* df1, int64_t npix, int C, float gscale,
                          int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------
 * Training step of the transformer (MIGT.train_step, viewformer/models/migt.py:464-505).
 * The dense contractions of the backward pass are vf_igemm_f32 calls (dX = dY.W^T with the weight
 * packed transposed, dW = X^T.dY via vf_transpose_f32); these are the remaining pieces.
 * ------------------------------------------------------------------------------------- */
/* attention of the training graph with its backward (flash-style: probabilities re-materialised per tile from the saved
 * per-query log-sum-exp, masked tiles skipped, deterministic — no atomics).  Replaces the autograd of compute_attention /
 * compute_causal_block_multiend_attention (branching_attention.py:5-18,82-126) inside MIGT.train_step (migt.py:464-505).
 *   forward with statistics: vf_attn_blockcausal_f32 + lse[B][H][T] = log sum_k exp(score)
 *   prep:  D[B][H][T] = rowsum(dOut * Out)
 *   bwd:   dq, dk, dv [B*T][ld*] (any column offsets / strides: the (V,Q,K) thirds of one buffer are fine)
 *   dropout (attn_dropout, branching_attention.py:15-17): drop_rate in [0, 1); element (b, h, q, k) of softmax(w) is kept iff
 *   vf_dropout_keep(word, k & 3, floor(rate * 2^32)) with word = the mask word of group q * ceil(T/4) + (k >> 2) in plane b*H + h of
 *   (drop_seed, drop_site) (csrc/vf_common.h: four consecutive keys of a query share one hashed word), and scaled by 1/(1-rate); the
 *   backward recomputes the same mask.  rate 0 = off.  drop_plane0: the mask plane of (b = 0, h = 0) — first scene's index in the GLOBAL
 *   batch times H — so that a data-parallel step draws the masks of the concatenated batch whatever the world size. */
int vf_attn_blockcausal_lse_f32(const float* q, const float* k, const float* v, float* out, float* lse,
                                int B, int H, int T, int L, int ldq, int ldk, int ldv, int ldo,
                                float scale, int skip_masked, int twin_view, float drop_rate, uint32_t drop_seed,
                                uint32_t drop_site, uint32_t drop_plane0, void* stream);
int vf_attn_bwd_prep_f32(const float* dout, const float* out, float* D, int B, int H, int T, int lddo, int ldo, void* stream);
int vf_attn_bwd_f32(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* D,
                    float* dq, float* dk, float* dv, int B, int H, int T, int L, int ldq, int ldk, int ldv, int lddo,
                    int lddq, int lddk, int lddv, float scale, int twin_view, float drop_rate, uint32_t drop_seed,
                    uint32_t drop_site, uint32_t drop_plane0, void* stream);
/* weight + bias gradient of a dense layer in the bf16 training arm (csrc/gemm_tn_bf16.hip): for split s of the M rows,
 * w_slabs[s][K][N] = sum_m x[m][k] * dy[m][n] and (b_slabs != NULL) b_slabs[s][N] = sum_m dy[m][n], with x a saved bf16 activation
 * [M][ldx] and dy the fp32 gradient [M][ldy], both read as they lie (no transposed copy, no packed copy, no separate column-sum pass).
 * Fold the slabs with vf_sum_slabs_f32 (fixed order).  K % 256 == N % 256 == M % 64 == 0; autograd of Conv1D.call (migt.py:89-96).
 * slab_stride (floats): 0 = the two arrays above; > 0 = split s writes its weight slab at w_slabs + s*slab_stride and its bias slab at
 * b_slabs + s*slab_stride, so with b_slabs = w_slabs + K*N and a gradient buffer that holds the bias right after the weight ONE
 * vf_sum_slabs_f32 over K*N + N floats folds both. */
int vf_gemm_tn_bf16(const void* x_bf16, int ldx, const void* dy, int dy_is_bf16, int ldy, int M, int K, int N, int splits, float* w_slabs,
                    float* b_slabs, int64_t slab_stride, void* stream);      /* dy_is_bf16: the gradient arrives already rounded to bf16 (ldy in elements) */
/* bf16 arm of the training step's attention (csrc/attention_dma.hip, attention_train_bf16.hip): bf16 q / k / v / out / dout in HBM
 * (ld* in ELEMENTS), fp32 lse, D and gradients; 64-token views, T % 64 == 0, <= 64 views — VF_ERR_UNSUPPORTED otherwise (callers
 * then take the f32 kernels above).  Forward = vf_attn_blockcausal_bf16_v2's LDS-DMA kernel also writing the per-query
 * log-sum-exp; backward = bf16-MFMA flash kernels (P and dS rounded to bf16 as operands, fp32 sums).  Same masks (twin_view), the
 * same no-scale convention and the same attention dropout (drop_rate / drop_seed / drop_site: identical masks, applied to the
 * bf16-rounded probabilities, 1 / (1 - rate) folded into the output normalisation; rate 0 = off) as the f32 forms; autograd of
 * branching_attention.py:5-18,82-126 under mixed_float16 (migt.py:464-505 with --fp16). */
int vf_attn_blockcausal_bf16_lse(const void* q, const void* k, const void* v, void* out, float* lse, int B, int H, int T, int L,
                                 int ldq, int ldk, int ldv, int ldo, float scale, int twin_view, float drop_rate, uint32_t drop_seed,
                                 uint32_t drop_site, uint32_t drop_plane0, void* stream);
int vf_attn_bwd_prep_bf16(const void* dout, const void* out, float* D, int B, int H, int T, int lddo, int ldo, void* stream);
int vf_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const float* lse, const float* D, void* dq, void* dk,
                     void* dv, int out_bf16 /* gradients written as bf16 (ldd* in elements) instead of fp32 */, int B, int H, int T, int L,
                     int ldq, int ldk, int ldv, int lddo, int lddq, int lddk, int lddv, float scale, int twin_view, float drop_rate,
                     uint32_t drop_seed, uint32_t drop_site, uint32_t drop_plane0, void* stream);
/* (round 6) vf_attn_bwd_bf16 with dq == NULL issues only the dK / dV launch, with dk == dv == NULL only the dQ launch: the two are independent, a caller
 * may put them on two streams (viewformer_amd/train_ops.py: attn_bwd_bf16(kv_stream=...)). */
/* elementwise dropout of the training graph (tf.keras.layers.Dropout at migt.py:72,216,403) on x [rows][cols] row-major:
 * out = keep ? x/(1-rate) : 0 [+ res], keep(m, n) = vf_dropout_keep(word of mask group (m' >> 2) * cols + n, m' & 3, floor(rate * 2^32)),
 * m' = m + row0 (row0: the first row's index in the global batch of a data-parallel step, 0 otherwise) (csrc/vf_common.h: four
 * consecutive rows of a column share one hashed word — what a lane of a GEMM epilogue holds, so vf_igemm_args.drop_rate applies the
 * same mask for free); applying it to a gradient gives the backward */
int vf_dropout_add_f32(const float* x, const float* res, float* out, int64_t rows, int cols, int64_t row0, float rate, uint32_t seed, uint32_t site,
                       void* stream);

/* dst[c][r] = src[r][c], `batch` matrices with strides (floats) */
int vf_transpose_f32(const float* src, float* dst, int rows, int cols, int64_t ld_src, int64_t ld_dst, int batch,
                     int64_t bs_src, int64_t bs_dst, void* stream);
/* the same from a bf16 source (a saved bf16 activation of the bf16 training arm), widened exactly */
int vf_transpose_bf16_f32(const void* src_bf16, float* dst, int rows, int cols, int64_t ld_src, int64_t ld_dst, int batch,
                          int64_t bs_src, int64_t bs_dst, void* stream);
/* out[n] (+)= sum_m x[m][n]  (bias gradients; deterministic two-stage: fixed-order partial sums)  ws: vf_colsum_workspace_bytes(N) */
size_t vf_colsum_workspace_bytes(int N);
int vf_colsum_f32(const float* x, float* out, int64_t M, int N, int64_t ld, int accumulate, void* ws, void* stream);
/* LayerNormalization backward (migt.py:225,227,292): dx, and dgamma/dbeta (+)= */
size_t vf_layernorm_bwd_workspace_bytes(int64_t rows, int d);
int vf_layernorm_bwd_f32(const float* dy, const float* x, const float* gamma, float* dx, float* dgamma, float* dbeta,
                         int64_t rows, int d, float eps, int accumulate, const float* res /* NULL or [rows][d]: dx += res */,
                         void* dx_bf16 /* NULL or [rows][d] bf16: a rounded copy of dx for GEMM consumers */,
                         float drop_rate, uint32_t drop_seed, uint32_t drop_site, int64_t drop_row0 /* rate > 0 (needs dx_bf16): the bf16 copy
                         is vf_dropout_add_f32(dx, cols = d, row0 = drop_row0) of that site — the dY of a layer whose output went through
                         dropout; dx itself, the residual path's gradient, is not masked */, void* ws, void* stream);
/* exact-erf GELU (tf.nn.gelu, migt.py:13,70) forward on a saved pre-activation, and its backward */
int vf_gelu_f32(const float* u, float* f, int64_t n, void* stream);
/* the same value rounded to bf16 on the way out (the bf16 training arm saves the MLP hidden as its next GEMM reads it) */
int vf_gelu_bf16out_f32(const float* u, void* f_bf16, int64_t n, void* stream);
int vf_gelu_bwd_f32(const float* u, const float* df, float* du, int64_t n, void* stream);
int vf_gelu_bwd_bf16out_f32(const float* u, const float* df, void* du_bf16, int64_t n, void* stream);
/* materialised attention probabilities for the backward pass: s[b][q][k] -> softmax(s*scale masked with -1e4) in
 * place (branching_attention.py:5-18,101-117; mask_spec as vf_attn_blockcausal_f32's twin_view), and
 * dS = P*(dP - sum dP*P)*scale (zero where masked) in place on dp */
int vf_softmax_mask_f32(float* s, int64_t batch, int T, int L, int mask_spec, float scale, void* stream);
int vf_softmax_mask_bwd_f32(const float* p, float* dp, int64_t batch, int T, int L, int mask_spec, float scale, void* stream);
/* tf.nn.sparse_softmax_cross_entropy_with_logits (migt.py:423) and its label-smoothed form (:99-104, y = onehot(1-eps) + eps/V):
 * loss[r], dlogits[r][:] = (softmax - y)*row_weight[r] */
int vf_softmax_ce_f32(const float* logits, const int32_t* target, const float* row_weight, float* loss, float* dlogits,
                      int64_t rows, int V, float label_smoothing, void* stream);
/* pose MSE of QuaternionPoseRepresentation.call (migt.py:156-177): raw [rows][7], gt [rows/L][7]; xyz_div [rows] = the per-scene
 * random pose multiplier the predicted position is divided by (:160-161; NULL = 1); the gradient of the position / orientation
 * terms is weighted separately (w_pos, w_ori [rows]: DynamicLossWeightingCriterion :107-120 scales them differently) */
int vf_pose_mse_f32(const float* raw, const float* gt, const float* w_pos, const float* w_ori, const float* xyz_div, float* pos_loss,
                    float* ori_loss, float* draw, int64_t rows, int L, float position_multiplier, void* stream);
/* backward of vf_embed_sum_f32 (the tf.gather / wpe slice of migt.py:358-368 under autograd): dwte[id] += sum of dh over the tokens
 * with that id, dwpe[l] += sum_bs dh, dadd[bs][:] = sum_l dh.  Deterministic (fixed-order partial sums, no float atomics);
 * workspace = vf_embed_bwd_workspace_bytes(d, vocab) bytes.  d % 4 == 0, d <= 2048, dh 16-byte aligned (rows are read as float4), else
 * VF_ERR_UNSUPPORTED. */
size_t vf_embed_bwd_workspace_bytes(int d, int vocab);
int vf_embed_bwd_f32(const float* dh, const int32_t* ids, float* dwte, float* dwpe, float* dadd, int64_t BS, int L, int d,
                     int vocab, void* workspace, void* stream);
/* weight/bias gradient of vf_dense_small_k_gelu_f32's linear part: dW[k][n] += sum_r x[r][k]*dy[r][n], db[n] += sum_r dy */
int vf_dense_small_k_bwd_f32(const float* x, const float* dy, float* dW, float* db, int64_t rows, int K, int N, void* stream);
/* The other tiny dimension (round 6): a dense layer with N <= 8 OUTPUTS — the pose head's c_proj, 1536 -> 7 (viewformer/models/migt.py:291-292,354:
 * MLP(n_state, n_embd = 7) inside QuaternionPoseRepresentation) — as one pass over x at HBM rate, and its weight gradient dW = x^T dy
 * (autograd of Conv1D.call, migt.py:89-96, under train_step :464-505).  out[r][n] = sum_k x[r][k] W[k][n] + b[n] (b may be NULL); K % 4 == 0,
 * K <= 4096, ldx % 4 == 0, x 16-byte aligned.  The gradient writes vf_dense_small_n_wgrad_slabs(rows) slab partial sums of K x N floats into `ws`
 * and folds them in slab order (deterministic); K <= 2048; accumulate != 0: dW += . */
int vf_dense_small_n_f32(const float* x, const float* W, const float* b, float* out, int64_t rows, int K, int N, int64_t ldx, void* stream);
int vf_dense_small_n_wgrad_slabs(int64_t rows);
int vf_dense_small_n_wgrad_f32(const float* x, const float* dy, float* dW, float* ws, int64_t rows, int K, int N, int64_t ldx, int accumulate,
                               void* stream);
/* AdamWeightDecay step (viewformer/models/utils.py:507-537 on Keras Adam): p -= lr_decay*p; m,v update;
 * p -= lr_adam * m / (sqrt(v) + eps)   with lr_adam = lr*sqrt(1-b2^t)/(1-b1^t), lr_decay = lr*weight_decay or 0 */
int vf_adamw_f32(float* param, const float* grad, float* m, float* v, int64_t n, float lr_decay, float lr_adam, float beta1,
                 float beta2, float eps, void* stream);
/* the same step over a whole FLAT buffer in one launch: elements inside one of the sorted, disjoint [start, end) element ranges of
 * nodecay_ranges (device int64 pairs; the "bias" tensors, models/utils.py:424) take lr_decay = 0; bit-identical to per-tensor
 * vf_adamw_f32 calls.  n % 4 == 0, 16-byte aligned buffers, tensors starting on 4-element boundaries, at most 256 ranges. */
int vf_adamw_flat_f32(float* param, const float* grad, float* m, float* v, int64_t n, const int64_t* nodecay_ranges, int nranges,
                      float lr_decay, float lr_adam, float beta1, float beta2, float eps, void* stream);
/* Round 6: the same flat step with the bf16 re-packing of the dense layers' weights folded in (the training step ended with vf_adamw_flat_f32 and
 * vf_gemm_bf16_pack_multi, which read every updated fp32 weight again twice).  Each descriptor names one weight matrix [rows][cols] (row-major,
 * contiguous) at element `offset` of the flat buffer and the packed operands to refresh from its NEW values: dst_kn = what
 * vf_gemm_bf16_pack(src, dst, K = rows, N = cols, sk = cols, sn = 1) writes (x @ W), dst_nk = what vf_gemm_bf16_pack(src, dst, K = cols, N = rows,
 * sk = 1, sn = cols) writes (dY @ W^T; a tied head's h @ E^T); either may be NULL.  Parameters, moments and both packings are bit-identical to
 * vf_adamw_flat_f32 followed by vf_gemm_bf16_pack_multi.  Descriptors: sorted by offset, disjoint, rows % 128 == 0, cols % 128 == 0 (no padding in either packing),
 * offset % 4 == 0, 16-byte aligned destinations, at most 256 — vf_adamw_pack_check validates a HOST copy of the table (call it once when the
 * table is built; the step itself takes the device copy and does not synchronise).  Replaces, for the bf16 arm, the reference's
 * optimizer.apply_gradients (models/utils.py:507-537) + the per-step cast of the variables to the compute dtype that Keras mixed precision
 * does inside every layer call (migt.py:89-96 under the mixed_float16 policy). */
typedef struct vf_adamw_pack_desc {
    int64_t offset;
    void* dst_kn;
    void* dst_nk;
    int32_t rows, cols;
    int32_t nodecay;            /* != 0: the matrix lies in one of nodecay_ranges (the caller resolves it once, on the host) */
    int32_t reserved;
} vf_adamw_pack_desc;
size_t vf_sizeof_adamw_pack_desc(void);
int vf_adamw_pack_check(const vf_adamw_pack_desc* descs_host, int ndesc, int64_t n);
int vf_adamw_flat_pack_f32(float* param, const float* grad, float* m, float* v, int64_t n, const int64_t* nodecay_ranges, int nranges,
                           float lr_decay, float lr_adam, float beta1, float beta2, float eps, const vf_adamw_pack_desc* descs_device, int ndesc,
                           void* stream);
int vf_add_inplace_f32(float* a, const float* b, int64_t n, void* stream);
/* out = a*x + b*y (y may be NULL: out = a*x); out may alias x or y */
int vf_axpby_f32(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream);
/* tf.clip_by_norm per tensor (migt.py:486-487): x *= clip / max(||x||, clip); scratch1 = one float */
int vf_clip_by_norm_f32(float* x, int64_t n, float clip, float* scratch1, void* stream);
/* global-norm clip of a flat gradient buffer as pytorch_lightning's Trainer(gradient_clip_val=...) applies it to the codebook model
 * (train_codebook_th.py:69 -> torch.nn.utils.clip_grad_norm_): x *= max_norm / (||x|| + 1e-6) when that factor is < 1 */
int vf_clip_grad_norm_f32(float* x, int64_t n, float max_norm, float* scratch1, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VF_HIP_H */
