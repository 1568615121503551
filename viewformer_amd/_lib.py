"""ctypes binding of libvf_hip.so (C-ABI declared in include/vf_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is
absent, importing/using the ops raises.  ``EXPORTS`` is the single list of symbols
the header declares; tests check the library exports every one of them.
"""
import ctypes
import os
from ctypes import c_int, c_int32, c_int64, c_float, c_void_p, c_size_t, c_char_p, POINTER

HERE = os.path.dirname(os.path.abspath(__file__))
# VF_HIP_LIB: developer override used by tools/variants.sh to time side-by-side builds of the same ABI
LIB_PATH = os.environ.get('VF_HIP_LIB') or os.path.join(HERE, 'libvf_hip.so')


class VfIgemmArgs(ctypes.Structure):
    """mirror of ``vf_igemm_args`` (include/vf_hip.h)"""
    _fields_ = [
        ('x', c_void_p), ('w_packed', c_void_p), ('bias', c_void_p), ('res', c_void_p), ('out', c_void_p),
        ('pro_mean', c_void_p), ('pro_scale', c_void_p), ('pro_beta', c_void_p),
        ('pro_swish', c_int32), ('pro_rows_per_img', c_int32), ('mode', c_int32), ('epilogue', c_int32),
        ('M', c_int32), ('Cin', c_int32), ('Cout', c_int32),
        ('Hin', c_int32), ('Win', c_int32), ('Hout', c_int32), ('Wout', c_int32),
        ('lda', c_int32), ('ldc', c_int32), ('ldr', c_int32), ('batch', c_int32),
        ('stride_x', c_int64), ('stride_w', c_int64), ('stride_out', c_int64), ('stride_res', c_int64),
        ('gn_part', c_void_p), ('gn_slots', c_int32), ('reserved0', c_int32),
        ('out_aux', c_void_p),
        ('drop_rate', c_float), ('drop_seed', ctypes.c_uint32), ('drop_site', ctypes.c_uint32), ('drop_row0', c_int32),
    ]


# kernel-selection switches (include/vf_hip.h: vf_select): each chooses between two kernels with bit-identical results, except the two
# tolerance-level pairs SEL_CONV_X3H_K32 and SEL_ATTN_DMA (the DMA attention kernel re-rounds a pre-scaled q: within 1e-2 of the other)
SEL_ATTN_DMA, SEL_GEMM_G256, SEL_LN_BWD_TWO_ROWS, SEL_ATTN_Q32 = 0, 1, 2, 3
SEL_CONV_X3H_K32 = 4          # (the one pair that differs in the last bits: 16x16x32 vs 32x32x16 MFMAs in the x3h convolution)
SEL_GEMM_TAIL = 5             # 256-tile bf16 GEMM: tail tiles in the last round (bit-identical)
# developer convenience: these environment variables are translated into vf_select calls ONCE, when the library is loaded (the library
# itself reads no environment variable)
_ENV_SELECT = {'VF_ATTN_DMA': SEL_ATTN_DMA, 'VF_GEMM_G256': SEL_GEMM_G256, 'VF_LN_BWD_TWO_ROWS': SEL_LN_BWD_TWO_ROWS, 'VF_ATTN_Q32': SEL_ATTN_Q32,
               'VF_CONV_X3H_K32': SEL_CONV_X3H_K32, 'VF_GEMM_TAIL': SEL_GEMM_TAIL}

ADAMW_PACK_DESC_BYTES = 40    # vf_adamw_pack_desc (train_ops.adamw_pack_table)
PACK_DESC_BYTES = 40          # vf_pack_desc (ops.pack_bf16_multi builds the table as a numpy record array of this item size)
P = c_void_p
# name -> (restype, argtypes)
EXPORTS = {
    'vf_abi_version': (c_int, []),
    'vf_sizeof_igemm_args': (c_size_t, []),
    'vf_sizeof_pack_desc': (c_size_t, []),
    'vf_build_arch': (c_char_p, []),
    'vf_build_flags': (c_int, []),
    'vf_build_flag_name': (c_char_p, [c_int]),
    'vf_select': (c_int, [c_int, c_int]),
    'vf_selected': (c_int, [c_int]),
    'vf_igemm_packed_floats': (c_size_t, [c_int, c_int, c_int]),
    'vf_igemm_pack_f32': (c_int, [P, P, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int64, P]),
    'vf_igemm_f32': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_conv_in_u8_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'vf_conv_in_x3h_packed_elems': (c_size_t, [c_int]),
    'vf_conv_in_x3h_pack': (c_int, [P, P, c_int, P]),
    'vf_conv_in_x3h': (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'vf_crc32c': (ctypes.c_uint32, [P, c_size_t, ctypes.c_uint32]),
    'vf_conv3_small_cout_f32': (c_int, [P, P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'vf_groupnorm_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vf_groupnorm_stats_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, P, P, P]),
    'vf_groupnorm_finalize_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, P]),
    'vf_conv3_halo_gn_slots': (c_int, [c_int, c_int]),
    'vf_groupnorm_apply_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    'vf_vq_packed_floats': (c_size_t, [c_int, c_int]),
    'vf_vq_pack_codebook_f32': (c_int, [P, P, c_int, c_int, P]),
    'vf_colsumsq_f32': (c_int, [P, P, c_int, c_int, P]),
    'vf_vq_argmin_f32': (c_int, [P, P, P, c_int64, c_int, c_int, P, P]),
    'vf_vq_filter_packed_bytes': (c_size_t, [c_int, c_int]),
    'vf_vq_filter_pack': (c_int, [P, P, c_int, c_int, P]),
    'vf_vq_argmin_filtered_f32': (c_int, [P, P, c_int64, c_int, c_int, P, P, P]),
    'vf_gather_transpose_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, P]),
    'vf_upsample2_bwd_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'vf_groupnorm_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'vf_groupnorm_bwd_f32': (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    'vf_softmax_rows_bwd_f32': (c_int, [P, P, c_int64, c_int, c_float, P]),
    'vf_l1_loss_partials': (c_int, [c_int64]),
    'vf_l1_loss_f32': (c_int, [P, P, P, P, c_int64, c_float, P]),
    'vf_conv3_wgrad_x6_rows': (c_size_t, [c_int]),
    'vf_conv3_wgrad_x6': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    'vf_lpips_scaling_f32': (c_int, [P, P, c_int64, P, P, c_int, P]),
    'vf_relu_f32': (c_int, [P, c_int64, P]),
    'vf_relu_bwd_f32': (c_int, [P, P, c_int64, P]),
    'vf_maxpool2_f32': (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    'vf_maxpool2_bwd_f32': (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    'vf_lpips_head_blocks': (c_int, [c_int]),
    'vf_lpips_head_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    'vf_lpips_head_bwd_f32': (c_int, [P, P, P, P, c_int64, c_int, c_float, c_int, P]),
    'vf_vq_ema_accumulate_f32': (c_int, [P, P, c_int64, c_int, c_int, P, P, P]),
    'vf_vq_ema_update_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_float, c_float, c_float, P]),
    'vf_codebook_gather_f32': (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    'vf_attn_blockcausal_f32': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_float, c_int, c_int, P]),
    'vf_attn_blockcausal_bf16_v2': (c_int, [P, P, P, c_int, P, c_int] + [c_int] * 8 + [c_float, c_int, c_int, P]),
    'vf_attn_blockcausal_fp8': (c_int, [P, P, P, c_int, P, c_int] + [c_int] * 8 + [c_float, c_int, c_int, P]),
    'vf_attn_blockcausal_x6': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_float, c_int, c_int, P]),
    'vf_attn_blockcausal_lse_f32': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_float, c_int, c_int, c_float, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]),
    'vf_dropout_add_f32': (c_int, [P, P, P, c_int64, c_int, c_int64, c_float, ctypes.c_uint32, ctypes.c_uint32, P]),
    'vf_attn_bwd_prep_f32': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'vf_attn_bwd_f32': (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_float, c_int, c_float, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]),
    'vf_gemm_tn_bf16': (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int64, P]),
    'vf_attn_blockcausal_bf16_lse': (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                             c_float, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]),
    'vf_attn_bwd_prep_bf16': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    'vf_attn_bwd_bf16': (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 c_float, c_int, c_float, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, P]),
    'vf_attn_spatial_f32': (c_int, [P, P, c_int, c_int, c_int, c_int64, c_int64, c_float, P]),
    'vf_attn_spatial_x3h': (c_int, [P, P, c_int, c_int, c_int, c_int64, c_int64, c_float, P]),
    'vf_softmax_rows_f32': (c_int, [P, c_int64, c_int, c_float, P]),
    'vf_layernorm_f32': (c_int, [P, P, P, P, c_int64, c_int, c_float, P]),
    'vf_layernorm_bf16out_f32': (c_int, [P, P, P, P, c_int64, c_int, c_float, P]),
    'vf_embed_sum_f32': (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    'vf_dense_small_k_gelu_f32': (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int, P]),
    'vf_lmhead_argmax_bf16': (c_int, [P, c_int, c_int64, P, c_int64, c_int, c_int, P, P, P]),
    'vf_argmax_rows_f32': (c_int, [P, c_int64, c_int, c_int, P, P]),
    'vf_postprocess_u8': (c_int, [P, P, c_int64, P]),
    'vf_resize_u8': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    # ---- bf16 arm (transformer dense layers, decoder convolutions)
    'vf_gemm_bf16_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_gemm_bf16_pack': (c_int, [P, P, c_int, c_int, c_int64, c_int64, c_int, c_int64, P]),
    'vf_gemm_bf16_pack_multi': (c_int, [P, c_int, P]),
    'vf_gemm_bf16': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_conv3_bf16_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_conv3_bf16_pack': (c_int, [P, P, c_int, c_int, P]),
    'vf_conv3_halo_bf16': (c_int, [POINTER(VfIgemmArgs), P]),
    # ---- fp32-equivalent split-bf16 arm
    'vf_conv3_x6_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_conv3_x6_pack': (c_int, [P, P, c_int, c_int, P]),
    'vf_conv3_halo_x6': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_conv3_x3h_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_conv3_x3h_pack': (c_int, [P, P, c_int, c_int, P]),
    'vf_conv3_halo_x3h': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_gemm_x3h_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_gemm_x3h_pack': (c_int, [P, P, c_int, c_int, c_int64, c_int64, P]),
    'vf_gemm_x3h': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_gemm_x6_packed_elems': (c_size_t, [c_int, c_int]),
    'vf_gemm_x6_pack': (c_int, [P, P, c_int, c_int, c_int64, c_int64, P]),
    'vf_gemm_x6': (c_int, [POINTER(VfIgemmArgs), P]),
    'vf_sum_slabs_f32': (c_int, [P, c_int, c_int64, c_int64, P, c_int, P]),
    # ---- training step
    'vf_transpose_f32': (c_int, [P, P, c_int, c_int, c_int64, c_int64, c_int, c_int64, c_int64, P]),
    'vf_transpose_bf16_f32': (c_int, [P, P, c_int, c_int, c_int64, c_int64, c_int, c_int64, c_int64, P]),
    'vf_colsum_workspace_bytes': (c_size_t, [c_int]),
    'vf_colsum_f32': (c_int, [P, P, c_int64, c_int, c_int64, c_int, P, P]),
    'vf_layernorm_bwd_workspace_bytes': (c_size_t, [c_int64, c_int]),
    'vf_layernorm_bwd_f32': (c_int, [P, P, P, P, P, P, c_int64, c_int, c_float, c_int, P, P, c_float, ctypes.c_uint32, ctypes.c_uint32, c_int64, P, P]),
    'vf_gelu_f32': (c_int, [P, P, c_int64, P]),
    'vf_gelu_bf16out_f32': (c_int, [P, P, c_int64, P]),
    'vf_gelu_bwd_bf16out_f32': (c_int, [P, P, P, c_int64, P]),
    'vf_gelu_bwd_f32': (c_int, [P, P, P, c_int64, P]),
    'vf_softmax_mask_f32': (c_int, [P, c_int64, c_int, c_int, c_int, c_float, P]),
    'vf_softmax_mask_bwd_f32': (c_int, [P, P, c_int64, c_int, c_int, c_int, c_float, P]),
    'vf_softmax_ce_f32': (c_int, [P, P, P, P, P, c_int64, c_int, c_float, P]),
    'vf_pose_mse_f32': (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_float, P]),
    'vf_embed_bwd_workspace_bytes': (c_size_t, [c_int, c_int]),
    'vf_embed_bwd_f32': (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_int, P, P]),
    'vf_dense_small_k_bwd_f32': (c_int, [P, P, P, P, c_int64, c_int, c_int, P]),
    'vf_dense_small_n_f32': (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int64, P]),
    'vf_dense_small_n_wgrad_slabs': (c_int, [c_int64]),
    'vf_dense_small_n_wgrad_f32': (c_int, [P, P, P, P, c_int64, c_int, c_int, c_int64, c_int, P]),
    'vf_adamw_f32': (c_int, [P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, P]),
    'vf_adamw_flat_f32': (c_int, [P, P, P, P, c_int64, P, c_int, c_float, c_float, c_float, c_float, c_float, P]),
    'vf_sizeof_adamw_pack_desc': (c_size_t, []),
    'vf_adamw_pack_check': (c_int, [P, c_int, c_int64]),
    'vf_adamw_flat_pack_f32': (c_int, [P, P, P, P, c_int64, P, c_int, c_float, c_float, c_float, c_float, c_float, P, c_int, P]),
    'vf_axpby_f32': (c_int, [c_float, P, c_float, P, P, c_int64, P]),
    'vf_add_inplace_f32': (c_int, [P, P, c_int64, P]),
    'vf_clip_by_norm_f32': (c_int, [P, c_int64, c_float, P, P]),
    'vf_clip_grad_norm_f32': (c_int, [P, c_int64, c_float, P, P]),
}

_lib = None


class VfError(RuntimeError):
    pass


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def load_variant(path):
    """a SECOND handle on a side-by-side build of the same ABI (viewformer_amd.build.build_variant): tests and in-process A/B tools only"""
    return _bind(os.path.abspath(path))


class use:
    """``with _lib.use(handle): ...`` — route every op of this module's callers through another build of the library for the block (tests / A/B)"""

    def __init__(self, handle):
        self.handle = handle

    def __enter__(self):
        global _lib
        load()
        self.prev, _lib = _lib, self.handle
        return self.handle

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def load():
    """Load libvf_hip.so (building nothing: see viewformer_amd.build).  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfError(f'{LIB_PATH} not found — run `python -m viewformer_amd.build` (hipcc --offload-arch=gfx950). '
                      'There is no CPU/PyTorch fallback for the hot path.')
    lib = _bind(LIB_PATH)
    # the two structs that cross the boundary by pointer: this module's mirrors must have the library's layout
    if int(lib.vf_sizeof_igemm_args()) != ctypes.sizeof(VfIgemmArgs):
        raise VfError(f'vf_igemm_args is {int(lib.vf_sizeof_igemm_args())} bytes in {LIB_PATH} and {ctypes.sizeof(VfIgemmArgs)} in '
                      'viewformer_amd/_lib.py: rebuild the library or update the mirror')
    if int(lib.vf_sizeof_pack_desc()) != PACK_DESC_BYTES:
        raise VfError(f'vf_pack_desc is {int(lib.vf_sizeof_pack_desc())} bytes in {LIB_PATH}, {PACK_DESC_BYTES} expected')
    if int(lib.vf_sizeof_adamw_pack_desc()) != ADAMW_PACK_DESC_BYTES:
        raise VfError(f'vf_adamw_pack_desc is {int(lib.vf_sizeof_adamw_pack_desc())} bytes in {LIB_PATH}, {ADAMW_PACK_DESC_BYTES} expected')
    for env, which in _ENV_SELECT.items():
        if os.environ.get(env) in ('0', '1'):
            lib.vf_select(which, int(os.environ[env]))
    _lib = lib
    return lib


def select(which, value):
    """vf_select: switch between two kernels (bit-identical pairs, except CONV_X3H_K32 and ATTN_DMA: tolerance-level) for A/B runs and parity tests; returns the previous value"""
    prev = load().vf_select(int(which), 1 if value else 0)
    if prev < 0:
        raise VfError(f'vf_select({which}, {value}): bad argument')
    return prev


def check(status: int, what: str):
    if status == 0:
        return
    if status < 0:
        msg = {-1: 'bad argument', -2: 'unsupported shape'}.get(status, 'error')
        raise VfError(f'{what}: {msg} ({status})')
    raise VfError(f'{what}: HIP error {status}')
