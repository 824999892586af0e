"""ctypes binding of libcotr_hip.so (include/cotr_hip.h).

The product path has no fallback: if the shared library is missing or does not
load, every model call raises.  ``import torch`` happens before the library is
opened so that the HIP runtime torch bundles (same soname, libamdhip64.so.7) is
the one instance in the process - stream handles from torch are then valid here.
"""
import ctypes
import os

import torch  # noqa: F401  (must be imported before the HIP library is opened)

from .build import LIB, LIB_EXP

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
    'cotr_batch_chunks': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
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
    'cotr_op_ln_reduce': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                         ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_layernorm': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ffn_block': (ctypes.c_int, [c_float_p] * 9 + [ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ffn_chunks': (ctypes.c_int, [ctypes.c_int]),
    'cotr_op_att_rows': (ctypes.c_int, [c_float_p, ctypes.c_int, c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_float, c_float_p, c_float_p,
                                        ctypes.c_int] + [c_float_p] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_ffn_rows': (ctypes.c_int, [c_float_p] * 10 + [ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_conv23': (ctypes.c_int, [c_float_p] * 9 + [ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_conv23m': (ctypes.c_int, [c_float_p] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_expand': (ctypes.c_int, [c_float_p, ctypes.c_int] + ([c_float_p] * 3 + [ctypes.c_int, c_float_p, ctypes.c_int]) * 2 + [ctypes.c_void_p]),
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
    'cotr_train_conv_wgrad_parts': (ctypes.c_int, [c_float_p] * 3 + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    'cotr_train_sum_parts': (ctypes.c_int, [c_float_p, ctypes.c_int, ctypes.c_size_t, c_float_p, ctypes.c_void_p]),
    'cotr_train_reduce_jobs': (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_train_perm_jobs': (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
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
    'cotr_gemm_num_configs': (ctypes.c_int, []),
    'cotr_op_conv_dual_cfg': (ctypes.c_int, [c_float_p, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, c_float_p, c_float_p, c_float_p, ctypes.c_int, c_float_p,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_bottleneck': (ctypes.c_int, [c_float_p, c_float_p, ctypes.c_int, ctypes.c_int] + [c_float_p] * 12 + [ctypes.c_void_p]),
    'cotr_knob_count': (ctypes.c_int, []),
    'cotr_knob_name': (ctypes.c_char_p, [ctypes.c_int]),
    'cotr_get_knob': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'cotr_set_knob': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),
    'cotr_check_knob': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    'cotr_reset_knobs': (ctypes.c_int, [ctypes.c_void_p]),
    'cotr_is_experimental': (ctypes.c_int, []),
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

# exported by libcotr_hip_exp.so only (the experimental build: cotr_amd/csrc/experimental/)
_EXP_PROTOS = {
    'cotr_op_dec_head': (ctypes.c_int, [c_float_p] * 11 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_linear_ln': (ctypes.c_int, [c_float_p] * 7 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    'cotr_op_split_h2': (ctypes.c_int, [c_float_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    'cotr_op_unsplit_h2': (ctypes.c_int, [ctypes.c_void_p, c_float_p, ctypes.c_size_t, ctypes.c_void_p]),
    'cotr_op_set_h2_flags': (ctypes.c_int, [ctypes.c_int]),
    'cotr_h2_fallbacks': (ctypes.c_long, []),
    'cotr_op_attention_h2': (ctypes.c_int, [c_float_p, ctypes.c_int, ctypes.c_int, c_float_p, c_float_p, ctypes.c_int, c_float_p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)
EXPERIMENTAL_SYMBOLS = tuple(_EXP_PROTOS)
ABI_VERSION = 2
_lib = None


class CotrHipError(RuntimeError):
    pass


def experimental_selected():
    """COTR_HIP_EXPERIMENTAL=1 in the environment makes this PROCESS load libcotr_hip_exp.so (the product library plus the
    measured dead ends and their knobs) instead of libcotr_hip.so: tests/test_experimental_gpu.py and A/B tools only."""
    return os.environ.get('COTR_HIP_EXPERIMENTAL', '0') not in ('', '0')


def library_path():
    return LIB_EXP if experimental_selected() else LIB


def load_library():
    """Open libcotr_hip.so and declare every prototype; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise CotrHipError(
            f'{path} is missing: build it with `python -m cotr_amd.build{" --experimental" if experimental_selected() else ""}` '
            '(or __graft_entry__.build()). cotr_amd has no CPU or PyTorch fallback for the forward path.')
    lib = ctypes.CDLL(path)
    protos = dict(_PROTOS)
    if experimental_selected():
        protos.update(_EXP_PROTOS)
    for name, (res, args) in protos.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.cotr_abi_version() != ABI_VERSION:
        raise CotrHipError(f'{os.path.basename(path)} has ABI version {lib.cotr_abi_version()}, this binding needs {ABI_VERSION}; '
                           'rebuild with `python -m cotr_amd.build --force`')
    if bool(lib.cotr_is_experimental()) != experimental_selected():
        raise CotrHipError(f'{path}: experimental flag of the library does not match the file name')
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


def _h(handle):
    return handle if handle is None or isinstance(handle, ctypes.c_void_p) else ctypes.c_void_p(handle)


def knobs(handle=None):
    """{name: (current, default)} of every tuning knob of a handle (cotr_set_knob(h, ...)); handle None = the process-wide
    set of the handle-less op-level entry points (cotr_op_*, cotr_bench_*, cotr_train_*)."""
    lib = load_library()
    out = {}
    for i in range(lib.cotr_knob_count()):
        name = lib.cotr_knob_name(i)
        cur, dflt = ctypes.c_int(), ctypes.c_int()
        check(lib.cotr_get_knob(_h(handle), name, ctypes.byref(cur), ctypes.byref(dflt)), handle, 'cotr_get_knob')
        out[name.decode()] = (cur.value, dflt.value)
    return out


def set_knob(name, value, handle=None):
    check(load_library().cotr_set_knob(_h(handle), name.encode(), int(value)), handle, f'cotr_set_knob({name}, {value})')


def validate_knob(name, value):
    """Raise CotrHipError unless the library's registry has the knob and accepts the value (cotr_check_knob: no knob set is touched):
    what a model does with a knob set before its handle exists."""
    if load_library().cotr_check_knob(name.encode(), int(value)) != 0:
        raise CotrHipError(f'cotr_set_knob({name}, {value}): no such knob in this library, or the value is outside its range')


def reset_knobs(handle=None):
    """Every knob of the set back to its shipped default (tests call this for the process-wide set after each GPU test)."""
    check(load_library().cotr_reset_knobs(_h(handle)), handle, 'cotr_reset_knobs')
