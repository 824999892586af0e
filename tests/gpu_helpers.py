"""Shared helpers of the GPU parity tests: layout conversions between the reference's
NCHW / sequence-first tensors and the library's NHWC side-by-side / batch-major ones,
and thin wrappers over the cotr_op_* entry points."""
import contextlib
import ctypes

import torch

from cotr_amd import _lib


def dev():
    return torch.device('cuda:0')


def sptr():
    return _lib.current_stream_ptr()


def P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def nchw_to_sbs(x):  # [B,C,H,2W] -> [B,H,2W,C]
    return x.permute(0, 2, 3, 1).contiguous()


def sbs_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def per_half(fn, x_nchw):
    """Apply an NCHW op to each 256-wide half separately and re-join on W (backbone.py:81-85)."""
    w = x_nchw.shape[-1] // 2
    return torch.cat([fn(x_nchw[..., :w]), fn(x_nchw[..., w:])], dim=-1)


def seq_to_rows(x):  # [L,B,E] -> [B*L,E]
    return x.permute(1, 0, 2).reshape(-1, x.shape[-1]).contiguous()


def op_linear(x, w, x2=None, x2_row_mod=0, scale=None, bias=None, residual=None, relu=False):
    lib = _lib.load_library()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty(M, N, device=x.device)
    rc = lib.cotr_op_linear(P(x), P(x2), x2_row_mod, P(w), P(scale), P(bias), P(residual), int(relu), P(y), M, N, K, sptr())
    assert rc == 0, rc
    return y


def op_conv(x_sbs, w_packed, scale, bias, residual, relu, cout, k, stride):
    lib = _lib.load_library()
    B, H, W2, cin = x_sbs.shape
    W = W2 // 2
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.empty(B, Ho, 2 * Wo, cout, device=x_sbs.device)
    rc = lib.cotr_op_conv(P(x_sbs), P(w_packed), P(scale), P(bias), P(residual), int(relu), P(y), B, H, W, cin, cout, k,
                          stride, sptr())
    assert rc == 0, rc
    return y


def pack_conv_weight(w):  # [Cout,Cin,k,k] -> [Cout,k,k,Cin]
    return w.permute(0, 2, 3, 1).contiguous()


@contextlib.contextmanager
def model_knobs(model, **knobs):
    """Tuning knobs of ONE model's library handle (cotr_set_knob(h, ...)) for the duration of a with-block; always put back to
    the shipped defaults afterwards (models are cached across tests)."""
    try:
        for name, value in knobs.items():
            model.set_knob(name, value)
        yield model
    finally:
        model.reset_knobs()
