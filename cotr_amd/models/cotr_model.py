"""The COTR model object, MI355X edition.

Same seam as the reference (``COTR/models/cotr_model.py:15-51``): ``build(args)`` returns an
``nn.Module`` whose ``state_dict()`` has the reference's keys (a reference checkpoint loads
with ``utils.safe_load_weights``), which exposes ``.transformer .corr_embed .query_proj
.input_proj .backbone`` (``train_cotr.py:49-55``) and whose
``forward(samples, queries) -> {'pred_corrs': [B,Q,2]}`` is what ``SparseEngine.infer_batch``
(``COTR/inference/sparse_engine.py:47-56``) and friends call.

The sub-modules here are PARAMETER CONTAINERS only.  All arithmetic of ``forward`` runs in
``libcotr_hip.so`` (hand-written gfx950 kernels, ``cotr_amd/csrc``) through the C ABI of
``include/cotr_hip.h``; there is no PyTorch or CPU fallback - calling the model with CPU
tensors, or without the built library, raises.
"""
import ctypes

import torch
from torch import nn

from .. import _lib
from .misc import NestedTensor
from .spec import LAYER_CHANNELS, resnet_stages

MAX_SIZE = 256  # COTR/utils/constants.py:2


class FrozenBatchNorm2d(nn.Module):
    """Four fixed buffers per norm (COTR/models/backbone.py:21-44); applied inside the HIP
    convolution epilogue as x*scale+bias with scale = w*rsqrt(var+1e-5) (backbone.py:46-56)."""

    def __init__(self, n):
        super().__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def _load_from_state_dict(self, state_dict, prefix, *rest):
        state_dict.pop(prefix + 'num_batches_tracked', None)  # torchvision checkpoints carry it
        super()._load_from_state_dict(state_dict, prefix, *rest)


def _conv(cin, cout, k, stride):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)


class _Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1, self.bn1 = _conv(inplanes, planes, 1, 1), FrozenBatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride), FrozenBatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1, 1), FrozenBatchNorm2d(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(_conv(inplanes, planes * 4, 1, stride), FrozenBatchNorm2d(planes * 4))


class _ResNetBody(nn.Module):
    """Parameters of torchvision resnet50 children conv1 .. ``layer`` (what the reference's
    IntermediateLayerGetter keeps, COTR/models/backbone.py:71)."""

    def __init__(self, layer):
        super().__init__()
        self.conv1, self.bn1 = _conv(3, 64, 7, 2), FrozenBatchNorm2d(64)
        inplanes = 64
        for name, planes, blocks, stride in resnet_stages(layer):
            seq = nn.Sequential(*[_Bottleneck(inplanes if b == 0 else planes * 4, planes,
                                              stride if b == 0 else 1, b == 0) for b in range(blocks)])
            setattr(self, name, seq)
            inplanes = planes * 4
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')


class _Backbone(nn.Module):
    def __init__(self, layer, train_backbone):
        super().__init__()
        self.body = _ResNetBody(layer)
        self.num_channels = LAYER_CHANNELS[layer]
        for name, p in self.body.named_parameters():  # COTR/models/backbone.py:66-69
            if not train_backbone or ('layer2' not in name and 'layer3' not in name and 'layer4' not in name):
                p.requires_grad_(False)


class _NoParams(nn.Module):
    """Stand-in for the parameter-free encodings (``query_proj``, ``backbone[1]``)."""

    def __init__(self, what):
        super().__init__()
        self.what = what

    def extra_repr(self):
        return self.what


class _EncoderLayer(nn.Module):
    def __init__(self, d, heads, ffn, dropout):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d, heads, dropout=dropout)
        self.linear1, self.linear2 = nn.Linear(d, ffn), nn.Linear(ffn, d)
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)


class _DecoderLayer(nn.Module):
    def __init__(self, d, heads, ffn, dropout):
        super().__init__()
        self.multihead_attn = nn.MultiheadAttention(d, heads, dropout=dropout)
        self.linear1, self.linear2 = nn.Linear(d, ffn), nn.Linear(ffn, d)
        # norm1 is a parameter of the reference that its forward never applies (transformer.py:173)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d), nn.LayerNorm(d), nn.LayerNorm(d)


class _Stack(nn.Module):
    def __init__(self, layers, norm=None):
        super().__init__()
        self.layers = nn.ModuleList(layers)
        if norm is not None:
            self.norm = norm


class _Transformer(nn.Module):
    def __init__(self, d, heads, enc_layers, dec_layers, ffn, dropout):
        super().__init__()
        self.encoder = _Stack([_EncoderLayer(d, heads, ffn, dropout) for _ in range(enc_layers)])
        self.decoder = _Stack([_DecoderLayer(d, heads, ffn, dropout) for _ in range(dec_layers)], nn.LayerNorm(d))
        self.d_model, self.nhead = d, heads
        for p in self.parameters():  # COTR/models/transformer.py:42-45
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class _MLP(nn.Module):
    def __init__(self, dims):
        super().__init__()
        self.num_layers = len(dims) - 1
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))


class COTR(nn.Module):
    def __init__(self, args):
        super().__init__()
        d = args.hidden_dim
        layer = getattr(args, 'layer', 'layer3')
        ffn = args.dim_feedforward
        unsupported = []
        if args.backbone != 'resnet50': unsupported.append(f'backbone={args.backbone}')
        if layer != 'layer3' or ffn != 1024: unsupported.append(f'layer={layer}/dim_feedforward={ffn}')
        if d != 256 or args.nheads != 8: unsupported.append(f'hidden_dim={d}/nheads={args.nheads}')
        if args.dilation: unsupported.append('dilation')
        if args.position_embedding != 'lin_sine': unsupported.append(f'position_embedding={args.position_embedding}')
        if unsupported:
            raise NotImplementedError(
                'libcotr_hip implements COTR\'s published configuration (resnet50/layer3, hidden 256, 8 heads, '
                'lin_sine; COTR/options/options.py:41-51); not: ' + ', '.join(unsupported))
        self.transformer = _Transformer(d, args.nheads, args.enc_layers, args.dec_layers, ffn, args.dropout)
        self.corr_embed = _MLP([d, d, d, 2])
        self.query_proj = _NoParams('lin_sine, depth 64')
        self.input_proj = nn.Conv2d(LAYER_CHANNELS[layer], d, kernel_size=1)
        train_backbone = getattr(args, 'lr_backbone', 0) > 0
        self.backbone = nn.Sequential(_Backbone(layer, train_backbone), _NoParams('lin_sine image grid encoding'))
        self.backbone.num_channels = LAYER_CHANNELS[layer]
        self._handle = None
        self._handle_device = None
        self._weights_dirty = True
        self._encoded_batch = 0
        self._ws = None             # scratch handed to the library (torch caching allocator), see _ensure_workspace
        self._ws_shape = (0, 0)
        self._ws_pins = set()       # ids of captured graphs that have the workspace's addresses baked in (pin_workspace)
        self._knobs = {}            # tuning knobs of THIS model's handle (set_knob); re-applied when the handle is re-created

    # ------------------------------------------------------------------ weight synchronisation
    def _apply(self, fn, *a, **kw):  # .cuda() / .to() / .float() move or replace the storage
        self._weights_dirty = True
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self._weights_dirty = True
        return super().load_state_dict(*a, **kw)

    def refresh_weights(self):
        """Re-pack the weights into the HIP library at the next call (needed only after
        modifying parameters in place; .to()/.cuda()/load_state_dict() do it themselves)."""
        self._weights_dirty = True

    def _ensure_ready(self, device):
        lib = _lib.load_library()
        if device.type != 'cuda':
            raise _lib.CotrHipError(
                'cotr_amd runs the COTR forward path on an MI355X only (HIP kernels, no CPU/PyTorch '
                f'fallback); got tensors on {device}. Move the model and inputs with .cuda().')
        index = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is None or self._handle_device != index:
            self._release()
            handle = ctypes.c_void_p()
            _lib.check(lib.cotr_create(ctypes.byref(handle), index), None, 'cotr_create')
            try:                                          # the remembered knobs go on BEFORE the handle is published: a knob the library
                for name, value in self.__dict__.get('_knobs', {}).items():   # refuses must not leave a half-configured handle behind
                    _lib.set_knob(name, value, handle)
            except Exception:
                lib.cotr_destroy(handle)
                raise
            self._handle, self._handle_device = handle, index
            self._weights_dirty = True
            self._ws, self._ws_shape = None, (0, 0)
        if self._weights_dirty:
            sd = {k: v.detach() for k, v in self.state_dict().items()}
            bad = [k for k, v in sd.items() if v.dtype != torch.float32]
            if bad:
                raise _lib.CotrHipError(f'libcotr_hip is fp32 (1e-3 px parity bar); non-fp32 tensors: {bad[:3]}...')
            keep = [v.contiguous() for v in sd.values()]
            n = len(keep)
            names = (ctypes.c_char_p * n)(*[k.encode() for k in sd])
            ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in keep])
            numels = (ctypes.c_int64 * n)(*[t.numel() for t in keep])
            torch.cuda.synchronize(index)
            _lib.check(lib.cotr_load_weights(self._handle, names, ptrs, numels, n), self._handle, 'cotr_load_weights')
            self._weights_dirty = False
            self._encoded_batch = 0
        return lib

    def _ensure_workspace(self, lib, device, b, q, keep_encode=False):
        """The library's encode cache + scratch come from torch's caching allocator (cotr_set_workspace): a larger batch then
        costs one torch allocation instead of hipFree + hipMalloc (device synchronisations) inside the library.  The
        workspace only grows; a cached encode is carried over into the new one (stream-ordered device copy; the old tensor
        goes back to torch's pool, which is stream-ordered too)."""
        b, q = max(b, self._ws_shape[0]), max(q, self._ws_shape[1])
        stale = self.__dict__.get('_ws_stale', False)      # a knob changed: the library's carving of the workspace is re-done
        if (b, q) == self._ws_shape and not stale:
            return
        need = ctypes.c_size_t()
        _lib.check(lib.cotr_scratch_bytes(self._handle, b, max(q, 1), ctypes.byref(need)), self._handle, 'cotr_scratch_bytes')
        if self._ws is None or self._ws.numel() < need.value + 256:
            if self._ws is not None and self.__dict__.get('_ws_pins'):
                raise _lib.CotrHipError(
                    f'the workspace would have to grow to {need.value} bytes for {b} pairs x {q} queries, but a captured training '
                    'step (GraphedTrainStep) has its addresses baked in: call model.reserve(max_pairs, max_queries) BEFORE '
                    'capturing, or close() the captured step first')
            ws = torch.empty(need.value + 256, dtype=torch.uint8, device=device)
            if self._ws is not None:
                # the old workspace goes back to torch's caching allocator, which may hand it out on ANOTHER stream while
                # kernels enqueued here still use it
                self._ws.record_stream(torch.cuda.current_stream(device))
            off = (-ws.data_ptr()) % 256
            keep = int(keep_encode and b == self._ws_shape[0])     # same pairs, more queries: the cached encode moves along
            _lib.check(lib.cotr_set_workspace(self._handle, ctypes.c_void_p(ws.data_ptr() + off), need.value, keep,
                                              _lib.current_stream_ptr()), self._handle, 'cotr_set_workspace')
            self._ws = ws
            if not keep:
                self._encoded_batch = 0
        elif stale and not self.__dict__.get('_ws_pins'):
            # same buffer, new knobs: the three regions grow in place and never shrink, so regions carved under the old knobs plus one
            # that is larger under the new ones can exceed what cotr_scratch_bytes promises for either - start the carving afresh
            off = (-self._ws.data_ptr()) % 256
            _lib.check(lib.cotr_set_workspace(self._handle, ctypes.c_void_p(self._ws.data_ptr() + off), self._ws.numel() - 256, 0,
                                              _lib.current_stream_ptr()), self._handle, 'cotr_set_workspace')
            self._encoded_batch = 0
        self._ws_stale = False
        self._ws_shape = (b, q)

    def _release(self):
        handle = self.__dict__.get('_handle')
        if handle is not None:
            self.__dict__['_handle'] = None  # (nn.Module.__setattr__ may already be torn down at interpreter exit)
            try:
                _lib.load_library().cotr_destroy(handle)
            except Exception:
                pass

    def __del__(self):
        self._release()

    def __getstate__(self):  # the HIP handle is per process: never pickled / deep-copied
        state = self.__dict__.copy()
        state['_handle'], state['_handle_device'], state['_weights_dirty'], state['_encoded_batch'] = None, None, True, 0
        state['_ws'], state['_ws_shape'], state['_ws_pins'] = None, (0, 0), set()
        state['_knobs'] = dict(state.get('_knobs', {}))
        return state

    # ------------------------------------------------------------------ the path
    def _check_mode(self):
        if self.training:
            raise NotImplementedError('encode()/decode() are the inference split (model.eval()); in training mode call '
                                      'model(img, queries) - see cotr_amd/training.py')

    def train(self, mode=True):
        # parameters updated by an optimiser while training must be re-packed into the HIP library before the next
        # inference call (the frozen backbone the training step uses is not touched by the optimiser)
        if self.training and not mode:
            self._weights_dirty = True
        return super().train(mode)

    @staticmethod
    def _as_batch(samples):
        if isinstance(samples, NestedTensor):
            # the reference turns a NestedTensor mask into a key-padding mask (transformer.py:49-55); every caller of
            # the reference feeds exactly 256x512 pixels, i.e. an all-False mask, and the HIP path has no masked
            # attention - refuse a real mask instead of silently ignoring it
            if samples.mask is not None and bool(samples.mask.any()):
                raise NotImplementedError('libcotr_hip has no key-padding mask: NestedTensor.mask must be all False '
                                          '(the reference always feeds full 256x512 inputs, backbone.py:80)')
            samples = samples.tensors
        elif isinstance(samples, (list, tuple)):
            samples = torch.stack(list(samples))
        # same hard shape contract as COTR/models/backbone.py:80
        assert samples.ndim == 4 and tuple(samples.shape[-2:]) == (MAX_SIZE, MAX_SIZE * 2) and samples.shape[1] == 3
        return samples

    @torch.no_grad()
    def encode(self, samples):
        """Query-independent half (backbone, input_proj, encoder, decoder K/V), cached in the
        HIP handle; follow with any number of ``decode(queries)``."""
        self._check_mode()
        img = self._as_batch(samples)
        lib = self._ensure_ready(img.device)
        img = img.contiguous().float()
        with torch.cuda.device(img.device):
            self._ensure_workspace(lib, img.device, img.shape[0], self._ws_shape[1])
            _lib.check(lib.cotr_encode(self._handle, img.data_ptr(), img.shape[0], _lib.current_stream_ptr()),
                       self._handle, 'cotr_encode')
        self._encoded_batch = img.shape[0]
        return self

    @torch.no_grad()
    def decode(self, queries):
        """pred_corrs [B,Q,2] for ``queries`` [B,Q,2] against the last ``encode``."""
        self._check_mode()
        b, q, two = queries.shape
        assert two == 2
        if self._encoded_batch != b:
            raise _lib.CotrHipError(f'decode of {b} pairs but the cached encode holds {self._encoded_batch}')
        lib = self._ensure_ready(queries.device)
        qs = queries.contiguous().float()
        out = torch.empty((b, q, 2), dtype=torch.float32, device=qs.device)
        with torch.cuda.device(qs.device):
            self._ensure_workspace(lib, qs.device, b, q, keep_encode=True)   # the cached encode moves along if it has to grow
            if self._encoded_batch != b:
                raise _lib.CotrHipError(f'decode of {b} pairs: the cached encode was dropped by a workspace change')
            _lib.check(lib.cotr_decode(self._handle, qs.data_ptr(), b, q, out.data_ptr(), _lib.current_stream_ptr()),
                       self._handle, 'cotr_decode')
        return out

    def pin_workspace(self, owner):
        """A captured HIP graph (training.GraphedTrainStep) holds the workspace's addresses: until unpin_workspace(owner) the
        workspace may not be replaced - a call that needs a larger one raises instead of silently freeing memory the graph writes."""
        self.__dict__.setdefault('_ws_pins', set()).add(id(owner))

    def unpin_workspace(self, owner):
        self.__dict__.setdefault('_ws_pins', set()).discard(id(owner))

    def reserve(self, pairs, queries):
        """Size the scratch workspace for calls of up to ``pairs`` x ``queries`` (optional; it otherwise grows on demand)."""
        dev = next(self.parameters()).device
        lib = self._ensure_ready(dev)
        with torch.cuda.device(dev):
            self._ensure_workspace(lib, dev, int(pairs), int(queries))

    def forward(self, samples, queries):
        if self.training:       # stage-1 training step: HIP backbone + HIP GEMMs under an autograd tape (training.py)
            from .. import training
            img = self._as_batch(samples)
            assert queries.ndim == 3 and queries.shape[2] == 2 and queries.shape[0] == img.shape[0]
            return {'pred_corrs': training.forward_train(self, img, queries)}
        return self._forward_eval(samples, queries)

    @torch.no_grad()
    def _forward_eval(self, samples, queries):
        img = self._as_batch(samples)
        b, q, two = queries.shape
        assert two == 2 and b == img.shape[0]
        if img.device != queries.device:
            raise _lib.CotrHipError(f'samples on {img.device} but queries on {queries.device}')
        lib = self._ensure_ready(img.device)
        img = img.contiguous().float()
        qs = queries.contiguous().float()
        out = torch.empty((b, q, 2), dtype=torch.float32, device=img.device)
        with torch.cuda.device(img.device):
            self._ensure_workspace(lib, img.device, b, q)
            _lib.check(lib.cotr_forward(self._handle, img.data_ptr(), qs.data_ptr(), b, q, out.data_ptr(),
                                        _lib.current_stream_ptr()), self._handle, 'cotr_forward')
        self._encoded_batch = b
        return {'pred_corrs': out}

    # ------------------------------------------------------------------ tuning knobs (per handle: include/cotr_hip.h)
    def set_knob(self, name, value):
        """One tuning knob of this model's library handle (cotr_set_knob(h, ...)): other models - and other threads - keep theirs.
        Remembered, so it survives a move to another GPU; fusion thresholds / encode_chunk change the scratch the library needs,
        so the workspace is re-sized at the next call."""
        if self.__dict__.get('_ws_pins'):
            raise _lib.CotrHipError('tuning knobs change the carving of the workspace, and a captured training step (GraphedTrainStep) has '
                                    'its addresses baked in: set knobs before capturing, or close() the captured step first')
        if self._handle is not None:
            _lib.set_knob(name, value, self._handle)
        else:
            _lib.validate_knob(name, value)               # no handle yet: checked against the library's registry now, applied later
        self._knobs[name] = int(value)
        self._ws_stale = True

    def knobs(self):
        """{name: (current, default)} of this model's handle (the shipped defaults + set_knob calls before the handle exists)."""
        if self._handle is not None:
            return _lib.knobs(self._handle)
        return {k: (self._knobs.get(k, v[1]), v[1]) for k, v in _lib.knobs(None).items()}

    def reset_knobs(self):
        if self.__dict__.get('_ws_pins') and self._knobs:
            raise _lib.CotrHipError('tuning knobs cannot change while a captured training step has the workspace pinned')
        if self._handle is not None:
            _lib.reset_knobs(self._handle)
        self._knobs = {}
        self._ws_stale = True

    # ------------------------------------------------------------------ test / profiling hooks
    def debug_tap(self, name):
        lib = _lib.load_library()
        n = ctypes.c_size_t()
        _lib.check(lib.cotr_debug_tap(self._handle, name.encode(), None, 0, ctypes.byref(n), None), self._handle, 'tap')
        out = torch.empty(n.value, dtype=torch.float32, device=f'cuda:{self._handle_device}')
        _lib.check(lib.cotr_debug_tap(self._handle, name.encode(), out.data_ptr(), n.value, ctypes.byref(n),
                                      _lib.current_stream_ptr()), self._handle, 'tap')
        return out

    def set_debug_taps(self, enable=True):
        self._ensure_ready(next(self.parameters()).device)
        _lib.check(_lib.load_library().cotr_set_debug_taps(self._handle, int(enable)), self._handle, 'debug taps')

    def set_profiling(self, level=1):
        """0 off, 1 HIP-event timing per stage, 2 per kernel launch (see get_profile)."""
        self._ensure_ready(next(self.parameters()).device)
        _lib.check(_lib.load_library().cotr_set_profiling(self._handle, int(level)), self._handle, 'profiling')

    def get_profile(self):
        lib = _lib.load_library()
        names = (ctypes.c_char_p * 512)()
        ms = (ctypes.c_float * 512)()
        n = ctypes.c_int()
        _lib.check(lib.cotr_get_profile(self._handle, names, ms, 512, ctypes.byref(n)), self._handle, 'profile')
        return [(names[i].decode(), ms[i]) for i in range(n.value)]


def build(args):
    return COTR(args)
