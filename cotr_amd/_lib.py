"""ctypes binding of libcotr_hip.so (include/cotr_hip.h).

The product path has no fallback: if the shared library is missing or does not
load, every model call raises.  ``import torch`` happens before the library is
opened so that the HIP runtime torch bundles (same soname, libamdhip64.so.7) is
the one instance in the process - stream handles from torch are then valid here.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the HIP library is opened)

from .build import LIB

c_float_p = ctypes.c_void_p  # raw device/host addresses from tensor.data_ptr()

_PROTOS = {
    'cotr_abi_version': (ctypes.c_int, []),
    'cotr_create': (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]),
    'cotr_destroy': (None, [ctypes.c_void_p]),
    'cotr_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'cotr_load_weights': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p),
                                         ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.c_int]),
    'cotr_encode': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_backbone': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_void_p]),
    'cotr_backbone_upto': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p]),
    'cotr_decode': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_void_p]),
    'cotr_forward': (ctypes.c_int, [ctypes.c_void_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, c_float_p,
                                    ctypes.c_void_p]),
    'cotr_workspace_bytes': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    'cotr_scratch_bytes': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    'cotr_set_workspace': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]),
    'cotr_debug_tap': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, c_float_p, ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]),
    'cotr_set_debug_taps': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'cotr_set_profiling': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'cotr_get_profile': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    'cotr_op_linear': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p,
                                      ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_conv': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                    ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_stem': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_stem_pool': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_maxpool': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p]),
    'cotr_op_attention': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_attention_fused': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_float,
                                               c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p, c_float_p,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_dec_head': (ctypes.c_int, [c_float_p] * 11 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ln_reduce': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_layernorm': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ffn_block': (ctypes.c_int, [c_float_p] * 9 + [ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ffn_chunks': (ctypes.c_int, [ctypes.c_int]),
    'cotr_op_posenc': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_add_rowmod': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_add_drop_ln_fwd': (ctypes.c_int, [c_float_p] * 7 + [ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]),
    'cotr_train_ln_bwd_parts': (ctypes.c_int, [ctypes.c_int]),
    'cotr_train_ln_bwd': (ctypes.c_int, [c_float_p] * 8 + [ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]),
    'cotr_train_set_dropout_salt': (ctypes.c_int, [ctypes.c_void_p]),
    'cotr_train_clear_dropout_salt': (ctypes.c_int, [ctypes.c_void_p]),
    'cotr_train_dropout_fwd': (ctypes.c_int, [c_float_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_uint32, ctypes.c_void_p]),
    'cotr_train_relu_drop_bwd': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p]),
    'cotr_train_colsum_parts': (ctypes.c_int, [ctypes.c_int]),
    'cotr_train_colsum': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_transpose': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_im2col': (ctypes.c_int, [c_float_p, c_float_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    'cotr_train_col2im': (ctypes.c_int, [c_float_p, c_float_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p]),
    'cotr_train_scale_rows': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_transpose_batched': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_gemm_tn_splits': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'cotr_train_gemm_tn': (ctypes.c_int, [c_float_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_gemm_tn_parts': (ctypes.c_int, [c_float_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    'cotr_train_reduce_jobs': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_adam': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.POINTER(ctypes.c_float), ctypes.c_int,
                                       ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p,
                                       ctypes.c_void_p]),
    'cotr_train_head_fwd': (ctypes.c_int, [c_float_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_head_bwd_parts': (ctypes.c_int, [ctypes.c_int]),
    'cotr_train_head_bwd': (ctypes.c_int, [c_float_p] * 6 + [ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_attention_fwd': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p,
                                                ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_uint32, ctypes.c_void_p]),
    'cotr_train_attention_bwd': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, c_float_p,
                                                c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                                ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_uint32, c_float_p, ctypes.c_void_p]),
    'cotr_train_attention_bwd_scratch': (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    'cotr_crop_resize_pairs': (ctypes.c_int, [c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int,
                                              c_float_p, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_dense_cycle': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, ctypes.c_void_p]),
    'cotr_dense_merge': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        c_float_p, c_float_p, ctypes.c_void_p]),
    'cotr_resize_f32': (ctypes.c_int, [c_float_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_float_p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_void_p]),
    'cotr_set_encode_chunk': (ctypes.c_int, [ctypes.c_int]),
    'cotr_gemm_num_configs': (ctypes.c_int, []),
    'cotr_set_ffn_fusion_max_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_fusion_max_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_head_fusion_max_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_splits': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_fused_splits': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_wide_min_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_pos_table_min_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_conv_patch': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_wide_occupancy': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_xcd_mapping': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_fused_stem': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_dual_conv': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_ks3': (ctypes.c_int, [ctypes.c_int]),
    'cotr_op_conv_dual_cfg': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_set_ffn_tail': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_conv1x1_dense': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_ws_flags': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_coop_tail': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_coop_tail_spin': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_train_attention_form': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_attention_resident': (ctypes.c_int, [ctypes.c_int]),
    'cotr_set_gemm_ln_min_rows': (ctypes.c_int, [ctypes.c_int]),
    'cotr_op_linear_ln': (ctypes.c_int, [c_float_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_set_bottleneck_max_pairs': (ctypes.c_int, [ctypes.c_int]),
    'cotr_op_bottleneck': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int] + [c_float_p] * 12 + [ctypes.c_void_p]),
    'cotr_knob_count': (ctypes.c_int, []),
    'cotr_knob_name': (ctypes.c_char_p, [ctypes.c_int]),
    'cotr_get_knob': (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'cotr_set_knob': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'cotr_reset_knobs': (ctypes.c_int, []),
    'cotr_set_ffn_preln': (ctypes.c_int, [ctypes.c_int]),
    'cotr_bench_linear': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]),
    'cotr_bench_conv': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p] + [ctypes.c_int] * 9 +
                        [ctypes.POINTER(ctypes.c_float)]),
    'cotr_op_linear_cfg': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                          ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_conv_cfg': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p] +
                         [ctypes.c_int] * 8 + [ctypes.c_void_p]),
    'cotr_debug_ffn_times': (ctypes.c_int, [ctypes.c_void_p]),
    'cotr_debug_attention_times': (ctypes.c_int, [ctypes.c_void_p]),
    'cotr_gemm_pick_conv': (ctypes.c_int, [ctypes.c_int] * 7),
    'cotr_debug_conv_times': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p] + [ctypes.c_int] * 8 +
                              [ctypes.c_void_p, ctypes.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)
_lib = None


class CotrHipError(RuntimeError):
    pass


def load_library():
    """Open libcotr_hip.so and declare every prototype; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise CotrHipError(
            f'{LIB} is missing: build it with `python -m cotr_amd.build` (or __graft_entry__.build()). '
            'cotr_amd has no CPU or PyTorch fallback for the forward path.')
    lib = ctypes.CDLL(LIB)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.cotr_abi_version() != 1:
        raise CotrHipError('libcotr_hip.so ABI version mismatch; rebuild with `python -m cotr_amd.build --force`')
    _lib = lib
    return lib


def check(rc, handle=None, what=''):
    if rc != 0:
        lib = load_library()
        msg = lib.cotr_last_error(handle)
        raise CotrHipError(f'{what} failed (code {rc}): {msg.decode() if msg else "?"}')


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def current_stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw-handle call avoids building a Stream object:
    a training step asks ~1500 times)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def knobs():
    """{name: (current, default)} of every process-wide tuning switch of the library (cotr_set_<name>)."""
    lib = load_library()
    out = {}
    for i in range(lib.cotr_knob_count()):
        name = lib.cotr_knob_name(i)
        cur, dflt = ctypes.c_int(), ctypes.c_int()
        check(lib.cotr_get_knob(name, ctypes.byref(cur), ctypes.byref(dflt)), None, 'cotr_get_knob')
        out[name.decode()] = (cur.value, dflt.value)
    return out


def set_knob(name, value):
    check(load_library().cotr_set_knob(name.encode(), int(value)), None, f'cotr_set_knob({name}, {value})')


def reset_knobs():
    """Every tuning switch back to its shipped default (tests call this after each test that touched one)."""
    check(load_library().cotr_reset_knobs(), None, 'cotr_reset_knobs')

