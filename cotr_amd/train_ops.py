"""torch.autograd Functions of the training step whose forward AND backward are hand-written HIP kernels
(``cotr_amd/csrc/train.hip``, ``attention_train.hip`` and the inference GEMM kernels), called through the C ABI of
``include/cotr_hip.h``.  PyTorch contributes the autograd tape, tensor allocation, the loss and the optimiser - no arithmetic.

Every Function maps to a piece of the reference's training forward (COTR/models/transformer.py with dropout active,
COTR/models/position_encoding.py:23-26), named in its docstring.  Dropout masks are counter-based (``train.h``): a site
gets a fresh 32-bit seed from ``next_seed()`` in the forward and hands the same seed to its backward kernel, which
recomputes the mask.
"""
import ctypes
import os

import torch

from . import _lib


def _P(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _sp():
    return _lib.current_stream_ptr()


def _chk(rc, what):
    if rc != 0:
        raise _lib.CotrHipError(f'{what} failed (code {rc})')


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes ``device`` current for a kernel call - a no-op when it already is (torch.cuda.device() costs ~15 us of
    host time per use; a training step makes ~1500 kernel calls)."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


# ---- dropout seeds ----------------------------------------------------------------------------------------------------
_seed = {'base': None, 'count': 0}


def reseed(base=None):
    """Start a new deterministic sequence of dropout seeds (default: torch's initial seed)."""
    _seed['base'] = (torch.initial_seed() if base is None else int(base)) & 0xFFFFFFFF
    _seed['count'] = 0


def next_seed():
    if _seed['base'] is None:
        reseed()
    _seed['count'] += 1
    x = (_seed['base'] * 0x9E3779B1 + _seed['count'] * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x2C1B3C6D) & 0xFFFFFFFF
    x ^= x >> 12
    return x


# ---- contractions ------------------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, relu=False, residual=None):
    """y[M,N] = relu?(a[M,K] . w[N,K]^T + bias (+ residual[M,N])) on the library's fp32-MFMA GEMM kernels."""
    lib = _lib.load_library()
    m, k = a.shape
    n = w.shape[0]
    assert a.is_contiguous() and w.is_contiguous() and w.shape[1] == k and k % 32 == 0 and n % 16 == 0, (a.shape, w.shape)
    assert residual is None or (residual.is_contiguous() and residual.numel() == m * n)
    y = _empty((m, n), a)
    if m:
        with _on(a.device):
            _chk(lib.cotr_op_linear(_P(a), None, 0, _P(w), None, _P(bias), _P(residual), int(relu), _P(y), m, n, k, _sp()),
                 f'cotr_op_linear {m}x{n}x{k}')
    return y


def gemm_tn(dy, x, with_colsum=False):
    """dW[N,K] = dy[M,N]^T . x[M,K] (operands row-major as they are: split over M, fixed-order sum of the partials) and, with
    ``with_colsum``, db[N] = column sums of dy from the same pass.  -> (dW, db or None), views of one buffer."""
    lib = _lib.load_library()
    m, n = dy.shape
    k = x.shape[1]
    assert dy.is_contiguous() and x.is_contiguous() and x.shape[0] == m
    extra = n if with_colsum else 0
    buf = _empty((n * k + extra,), dy)
    part = _empty((max(1, lib.cotr_train_gemm_tn_splits(m, n, k)) * (n * k + extra),), dy)
    cs = ctypes.c_void_p(buf.data_ptr() + n * k * 4) if with_colsum else None
    with _on(dy.device):
        _chk(lib.cotr_train_gemm_tn(_P(dy), _P(x), _P(part), _P(buf), cs, m, n, k, _sp()), f'cotr_train_gemm_tn {m}x{n}x{k}')
    return buf[:n * k].view(n, k), (buf[n * k:] if with_colsum else None)


# ---- deferred gradient reduction --------------------------------------------------------------------------------------
_JOB_DT = None
_SRC_DT = None


def _record_dtypes():
    """numpy layouts of cotr_reduce_job / cotr_reduce_src (include/cotr_hip.h)."""
    global _JOB_DT, _SRC_DT
    if _JOB_DT is None:
        import numpy as np
        _SRC_DT = np.dtype([('part', '<u8'), ('pstride', '<u8'), ('nparts', '<u4'), ('pad', '<u4')])
        _JOB_DT = np.dtype([('dst', '<u8'), ('scale', '<u8'), ('numel', '<u4'), ('first_src', '<u4'), ('n_src', '<u4'),
                            ('chunk0', '<u4'), ('row_len', '<u4'), ('cin', '<u4'), ('taps', '<u4'), ('vec', '<u4')])
        assert _SRC_DT.itemsize == 24 and _JOB_DT.itemsize == 48
    return _JOB_DT, _SRC_DT


class GradSink:
    """Persistent flat gradient buffer + ONE reduction launch per backward pass.

    Without it every weight gradient costs: the split-M partials of ``gemm_tn`` (or the per-workgroup partials of the LayerNorm /
    head backward), a ``sum_parts`` launch, for the convolutions a row scale and a layout transpose, a ``cat`` for the packed
    projections and autograd's own ``add_`` into ``.grad`` - ~150 reduction launches + ~230 accumulations per step.  With a sink
    installed (``with sink.collect(): loss.backward()``) the backward Functions below leave their partials where the kernels wrote
    them (a step's worth is 1-2 GB; the HBM holds 288), register (destination, partials) with the sink and return no gradient for
    the parameter; ``flush()`` then runs ``cotr_train_reduce_jobs`` once: every destination += its sources in the order autograd
    produced them, each source's partials in split order - the additions ``sum_parts`` + ``add_`` would have made, so the
    gradients are bit-identical to the Function-by-Function path (tests/test_training_gpu.py).

    The gradients of all parameters are views of one buffer (``flat``): ``zero()`` is one memset and the gradient exchange
    between ranks one collective.  A parameter the sink does not own (or a Function given a non-leaf weight) takes the ordinary
    path."""
    CHUNK = 1024

    def __init__(self, params, capacity=2048):
        params = [p for p in params if p.requires_grad]
        assert params, 'no trainable parameter'
        dev = params[0].device
        self.params, offs, total = params, [], 0
        for p in params:
            assert p.is_leaf and p.dtype == torch.float32 and p.device == dev and p.is_contiguous()
            offs.append(total)
            total += (p.numel() + 63) // 64 * 64                       # every gradient starts on a 256-byte boundary
        self.offsets = offs
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self._lo, self._hi = self.flat.data_ptr(), self.flat.data_ptr() + total * 4
        self.capacity = capacity
        job_dt, src_dt = _record_dtypes()
        # records (up to 4 uses of a parameter per step on average) + one word per 1024-element chunk
        nbytes = capacity * (job_dt.itemsize + 4 * src_dt.itemsize) + 4 * (total // self.CHUNK + 2 * capacity)
        self._host = torch.empty(nbytes, dtype=torch.uint8)
        if dev.type == 'cuda':
            self._host = self._host.pin_memory()
        self._dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._uploaded = None                                          # event after the last table upload
        self._jobs, self._keep = {}, []
        self.frozen = False                                            # set by GraphedTrainStep after capture
        self.last = (0, 0)                                             # (jobs, sources) of the last flush, for tests / reports
        self._exchange, self._exchange_parts = None, 1                 # set_exchange(): gradient averaging started from flush()
        self.last_parts = []                                           # [(lo, hi) element ranges of flat] of the last chunked flush

    # -- ownership ------------------------------------------------------------------------------------------------------
    def grad_of(self, w):
        """The gradient tensor to accumulate into for weight ``w`` (a parameter of this sink, or a same-size contiguous view of
        one, e.g. ``input_proj.weight.view(256, 1024)``), or None."""
        base = w
        if not w.is_leaf:
            base = w._base
            if base is None or not base.is_leaf or base.numel() != w.numel() or not w.is_contiguous():
                return None
        g = base.grad
        if g is None or not (self._lo <= g.data_ptr() < self._hi):
            return None
        return g.view(w.shape)

    def zero(self):
        self.flat.zero_()

    def attach(self):
        """Re-install the gradient views (after an ``optim.zero_grad()`` with ``set_to_none`` dropped them)."""
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self._lo + off * 4:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += (p.numel() + 63) // 64 * 64

    # -- collection -----------------------------------------------------------------------------------------------------
    def collect(self):
        """Context: the backward Functions of this module hand their parameter gradients to this sink while it is active
        (process-wide - the autograd engine runs them on its own thread); flushes on exit."""
        return _Collect(self)

    def add(self, dst, part, part_off, nparts, pstride, numel, scale=None, row_len=0, cin=0, taps=1):
        """dst (a contiguous view into ``flat``, ``numel`` elements) += sum over the ``nparts`` partials
        ``part[part_off + p * pstride : ... + numel]`` (x ``scale[row]`` / re-laid-out, see cotr_reduce_job)."""
        assert not self.frozen, 'this sink belongs to a captured step'
        assert dst.is_contiguous() and dst.numel() == numel and self._lo <= dst.data_ptr() < self._hi
        if nparts <= 0:
            return
        key = dst.data_ptr()
        job = self._jobs.get(key)
        meta = (numel, 0 if scale is None else scale.data_ptr(), row_len, cin, taps)
        if job is None:
            job = self._jobs[key] = {'meta': meta, 'srcs': []}
        else:
            assert job['meta'] == meta, 'two uses of one parameter disagree on the gradient layout'
        job['srcs'].append((part.data_ptr() + part_off * 4, pstride, nparts))
        self._keep.append(part)
        if scale is not None:
            self._keep.append(scale)

    # -- gradient exchange overlapped with the reduction (several ranks) ------------------------------------------------------
    def set_exchange(self, start, parts=3):
        """``start(flat_slice) -> finish()`` begins averaging a contiguous slice of ``flat`` over the ranks
        (``dist.flat_exchange_async``) and returns the callable that waits for it.  With an exchange installed ``flush()`` reduces
        the gradients in ``parts`` address-ordered ranges of the buffer - one reduction launch each - and starts the exchange of
        range k right behind its launch, so that it runs (on the collective's own stream) while range k+1 is being reduced; without
        one the single deferred reduction of a backward pass would leave the whole exchange exposed after it.  Every gradient
        element still receives the same sources in the same order and the same sum over the ranks: values unchanged.
        ``set_exchange(None)`` goes back to the single launch."""
        self._exchange, self._exchange_parts = start, max(1, int(parts)) if start is not None else 1

    def partition(self, parts):
        """The registered jobs split into <= ``parts`` groups by destination address -> [(lo, hi, [(dst, job)])]: element ranges of
        ``flat`` cut at gradient boundaries (every parameter's gradient - hence every job's destination - lies in exactly one),
        of roughly equal size, covering the whole buffer in address order."""
        total = self.flat.numel()
        starts = self.offsets + [total]
        cuts = [0]
        for k in range(1, parts):
            want = total * k // parts
            best = min(starts, key=lambda o: abs(o - want))            # the gradient boundary nearest to an even split
            if best > cuts[-1] and best < total:
                cuts.append(best)
        cuts.append(total)
        groups = [(lo, hi, []) for lo, hi in zip(cuts[:-1], cuts[1:])]
        for dst, job in self._jobs.items():
            el = (dst - self._lo) // 4
            for lo, hi, items in groups:
                if lo <= el < hi:
                    items.append((dst, job))
                    break
        return groups

    def tables(self, items=None):
        """-> (jobs, srcs, nchunks) as numpy record arrays (host side of flush(); also what the CPU test inspects); ``items``: a
        subset of the registered jobs as [(dst, job)] (one part of a chunked flush), default all."""
        import numpy as np
        job_dt, src_dt = _record_dtypes()
        items = list(self._jobs.items()) if items is None else items
        nj = len(items)
        ns = sum(len(j['srcs']) for _, j in items)
        jobs, srcs = np.zeros(nj, dtype=job_dt), np.zeros(ns, dtype=src_dt)
        chunk = si = 0
        # the longest per-thread walks first (a LayerNorm weight: ~500 partials per use): they would otherwise be the launch's tail.
        # (Order of the JOBS only - within a job the sources keep autograd's order, so the arithmetic does not change.)
        order = sorted(items, key=lambda kv: -sum(n for _, _, n in kv[1]['srcs']))
        for ji, (dst, j) in enumerate(order):
            numel, scale, row_len, cin, taps = j['meta']
            vec = dst % 16 == 0 and numel % 4 == 0 and row_len % 4 == 0 and cin % 4 == 0
            for ptr, pstride, nparts in j['srcs']:
                srcs[si] = (ptr, pstride, nparts, 0)
                vec = vec and ptr % 16 == 0 and pstride % 4 == 0
                si += 1
            jobs[ji] = (dst, scale, numel, si - len(j['srcs']), len(j['srcs']), chunk, row_len, cin, taps, int(vec))
            chunk += (numel + self.CHUNK - 1) // self.CHUNK
        return jobs, srcs, chunk

    @staticmethod
    def chunk_map(jobs, nchunks):
        """chunk -> job index (one workgroup of cotr_train_reduce_jobs per chunk)."""
        import numpy as np
        counts = np.diff(np.append(jobs['chunk0'], np.uint32(nchunks))).astype(np.int64)
        return np.repeat(np.arange(len(jobs), dtype=np.uint32), counts)

    def _reduce(self, jobs_ptr, srcs_ptr, cmap_ptr, njobs, nchunks):
        """cotr_train_reduce_jobs on the tables at these device addresses (tests of the host logic replace this)."""
        lib = _lib.load_library()
        with _on(self.flat.device):
            _chk(lib.cotr_train_reduce_jobs(ctypes.c_void_p(jobs_ptr), ctypes.c_void_p(srcs_ptr), ctypes.c_void_p(cmap_ptr), njobs,
                                            nchunks, _sp()), 'cotr_train_reduce_jobs')

    def flush(self):
        """Every registered partial is summed into its gradient: ONE launch - or, with an exchange installed (set_exchange), one
        launch per address range of the buffer, each followed by the start of that range's exchange between the ranks; returns
        when all exchanges have been waited for (stream-ordered).  The tables of all launches go up in one copy.  The partial
        buffers are released afterwards (the caching allocator is stream-ordered: they are not reused before the launches ran)."""
        self.last_parts = []
        if not self._jobs:
            self.last = (0, 0)
            if self._exchange is not None:                             # every rank must still take part in the collectives
                self._exchange(self.flat)()
                self.last_parts = [(0, self.flat.numel())]
            return
        import numpy as np
        capturing = self.flat.is_cuda and torch.cuda.is_current_stream_capturing()
        chunked = self._exchange is not None and not capturing and not self.frozen
        groups = self.partition(self._exchange_parts) if chunked else [(0, self.flat.numel(), list(self._jobs.items()))]
        tabs, off = [], 0
        for lo, hi, items in groups:
            jobs, srcs, nchunks = self.tables(items)
            cmap = self.chunk_map(jobs, nchunks)
            tabs.append((lo, hi, jobs, srcs, cmap, nchunks, off))
            off += (jobs.nbytes + srcs.nbytes + cmap.nbytes + 15) // 16 * 16
        assert off <= self._host.numel(), 'GradSink capacity exceeded'
        if self._uploaded is not None and not capturing:
            self._uploaded.synchronize()                               # the previous upload has left the pinned buffer
        host = self._host.numpy()
        for lo, hi, jobs, srcs, cmap, nchunks, o in tabs:
            jb, sb, cb = jobs.nbytes, srcs.nbytes, cmap.nbytes
            host[o:o + jb] = jobs.view(np.uint8)
            host[o + jb:o + jb + sb] = srcs.view(np.uint8)
            host[o + jb + sb:o + jb + sb + cb] = cmap.view(np.uint8)
        self._dev[:off].copy_(self._host[:off], non_blocking=True)
        if self.flat.is_cuda and not capturing:
            self._uploaded = torch.cuda.Event()
            self._uploaded.record()
        base = self._dev.data_ptr()
        finish = []
        for lo, hi, jobs, srcs, cmap, nchunks, o in tabs:
            if len(jobs):
                self._reduce(base + o, base + o + jobs.nbytes, base + o + jobs.nbytes + srcs.nbytes, len(jobs), nchunks)
            if chunked:
                finish.append(self._exchange(self.flat[lo:hi]))        # runs beside the next range's reduction launch
                self.last_parts.append((lo, hi))
        for f in finish:
            f()
        self.last = (sum(len(t[2]) for t in tabs), sum(len(t[3]) for t in tabs))
        self._jobs, self._keep = {}, []

    def discard(self):
        self._jobs, self._keep = {}, []


_sink = None      # the GradSink collecting right now (process-wide: backward runs on the autograd engine's thread)


class _Collect:
    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        global _sink
        assert _sink is None, 'another GradSink is collecting'
        self.sink.attach()
        _sink = self.sink
        return self.sink

    def __exit__(self, exc_type, exc, tb):
        global _sink
        _sink = None
        if exc_type is None:
            self.sink.flush()
        else:
            self.sink.discard()
        return False


def gemm_tn_parts(dy, x, with_colsum=False):
    """The split-M partials of ``gemm_tn`` alone -> (part, number of partials, floats per partial = n*k (+ n))."""
    lib = _lib.load_library()
    m, n = dy.shape
    k = x.shape[1]
    assert dy.is_contiguous() and x.is_contiguous() and x.shape[0] == m
    pstride = n * k + (n if with_colsum else 0)
    part = _empty((max(1, lib.cotr_train_gemm_tn_splits(m, n, k)) * pstride,), dy)
    with _on(dy.device):
        rc = lib.cotr_train_gemm_tn_parts(_P(dy), _P(x), _P(part), m, n, k, int(with_colsum), _sp())
    if rc < 0:
        _chk(rc, f'cotr_train_gemm_tn_parts {m}x{n}x{k}')
    return part, rc, pstride


def conv_wgrad_parts(dz2, x, b, h, wd, cin, cout, k, stride):
    """The split-M partials of a convolution's weight gradient straight from its input x (no im2col image: the kernel gathers the
    shifted pixels itself; csrc/train.hip gemm_tn_big_kernel<true>) -> (part, number of partials, floats per partial), the same
    values ``gemm_tn_parts(dz2, im2col(x))`` returns; None where that form does not apply (Cin % 128, small shapes)."""
    lib = _lib.load_library()
    m, kk = dz2.shape[0], k * k * cin
    assert dz2.is_contiguous() and x.is_contiguous() and dz2.shape[1] == cout
    part = _empty((max(1, lib.cotr_train_gemm_tn_splits(m, cout, kk)) * cout * kk,), dz2)
    with _on(dz2.device):
        rc = lib.cotr_train_conv_wgrad_parts(_P(dz2), _P(x), _P(part), b, h, wd, cin, cout, k, stride, _sp())
    if rc == -1:
        return None
    if rc < 0:
        _chk(rc, f'cotr_train_conv_wgrad_parts {m}x{cout}x{kk}')
    return part, rc, cout * kk


def sum_parts(part, nparts, n):
    """part [nparts][n] summed in split order -> [n] (what gemm_tn does behind its partials)."""
    lib = _lib.load_library()
    out = _empty((n,), part)
    with _on(part.device):
        _chk(lib.cotr_train_sum_parts(_P(part), nparts, n, _P(out), _sp()), 'cotr_train_sum_parts')
    return out


def colsum(x, out):
    lib = _lib.load_library()
    m, n = x.shape
    part = _empty((lib.cotr_train_colsum_parts(m) * n,), x)
    with _on(x.device):
        _chk(lib.cotr_train_colsum(_P(x), _P(part), _P(out), m, n, _sp()), 'cotr_train_colsum')
    return out


# ---- weight-shaped operands DERIVED from the parameters ---------------------------------------------------------------------
# A training step needs, besides the parameters themselves, ~130 re-laid-out copies of them: the W^T slices of the dX GEMMs (the
# GEMM kernels take their weight operand as [N][K]), the convolution weights packed [Cout][k][k][Cin] (forward, implicit GEMM) and
# their FrozenBN-scaled transposes [k*k*Cin][Cout] (dgrad).  They change exactly when the optimiser steps.  This registry keeps each
# in a PERSISTENT buffer, knows how it is derived (a batched strided transpose with an optional row factor: cotr_perm_job) and
# re-derives ALL of them in one launch after the optimiser step (refresh_derived) - instead of ~130 transposes / row scalings of a
# few microseconds each spread over the next backward pass (0.6 ms of a 22 ms stage-2 step).  A miss (first use, a new slice, a
# parameter that moved) is served by the same kernel with a one-job table.
_PERM_DT = None


def _perm_dtype():
    global _PERM_DT
    if _PERM_DT is None:
        import numpy as np
        _PERM_DT = np.dtype([('src', '<u8'), ('dst', '<u8'), ('scale', '<u8'), ('Z', '<u4'), ('R', '<u4'), ('C', '<u4'), ('sz', '<u4'),
                             ('sr', '<u4'), ('sc', '<u4'), ('dz', '<u4'), ('dc', '<u4'), ('tile0', '<u4'), ('tiles_r', '<u4'),
                             ('tiles_c', '<u4'), ('pad', '<u4')])
        assert _PERM_DT.itemsize == 72
    return _PERM_DT


class _Derived:
    def __init__(self):
        self.entries = {}      # key -> entry dict
        self.tables = {}       # device -> (signature, device table tensor, byte offset of the tile map, njobs, ntiles) of its last full refresh

    def get(self, base, view, kind, scale, spec, shape):
        """The derived operand of ``view`` (a view of parameter ``base``), current with respect to the parameter."""
        import weakref
        key = (id(base), view.storage_offset(), tuple(view.shape), kind, 0 if scale is None else id(scale))
        e = self.entries.get(key)
        if e is not None and (e['ref']() is not base or e['src'] != view.data_ptr() or (scale is not None and e['scale_ref']() is not scale)):
            e = None                                                       # a dead parameter's address, or the parameter moved
        if e is None:
            for k in [k for k, v in self.entries.items() if v['ref']() is None]:
                del self.entries[k]
            e = {'ref': weakref.ref(base), 'src': view.data_ptr(), 'dst': _empty(shape, view), 'spec': spec, 'version': None,
                 'scale_ref': None if scale is None else weakref.ref(scale), 'scale': 0 if scale is None else scale.data_ptr(),
                 'fresh': False, 'device': view.device}
            self.entries[key] = e
            self.tables.pop(view.device, None)
        if not e['fresh'] or e['version'] != base._version:
            self._run([e], cache=False)
            e['fresh'], e['version'] = True, base._version
        return e['dst']

    def _build(self, entries):
        import numpy as np
        jobs = np.zeros(len(entries), dtype=_perm_dtype())
        tile = 0
        owners = []
        for i, e in enumerate(entries):
            Z, R, C, sz, sr, sc, dz, dc = e['spec'][:8]
            flags = e['spec'][8] if len(e['spec']) > 8 else 0                # bit 0: source batch Z-1-z -> destination batch z
            tr, tc = (R + 31) // 32, (C + 31) // 32
            jobs[i] = (e['src'], e['dst'].data_ptr(), e['scale'], Z, R, C, sz, sr, sc, dz, dc, tile, tr, tc, flags)
            owners.append(np.full(Z * tr * tc, i, dtype=np.uint32))
            tile += Z * tr * tc
        cmap = np.concatenate(owners) if owners else np.zeros(0, np.uint32)
        raw = np.concatenate([jobs.view(np.uint8), cmap.view(np.uint8)])
        return raw, jobs.nbytes, len(entries), tile

    def _run(self, entries, cache):
        if not entries:
            return
        dev = entries[0]['device']
        sig = tuple((e['src'], e['dst'].data_ptr(), e['scale']) for e in entries)
        cached = self.tables.get(dev) if cache else None
        if cached is not None and cached[0] == sig:
            _, table, off, njobs, ntiles = cached
        else:
            # a miss allocates, pins and uploads a job table: not something a stream capture may contain (a captured step must find
            # every operand registered and its device's table cached - GraphedTrainStep warms up eagerly for exactly that)
            if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
                raise RuntimeError('a derived weight operand was registered or its job table rebuilt during stream capture')
            raw, off, njobs, ntiles = self._build(entries)
            host = torch.from_numpy(raw)
            if dev.type == 'cuda':
                host = host.pin_memory()
            table = torch.empty(raw.nbytes, dtype=torch.uint8, device=dev)
            table.copy_(host, non_blocking=True)
            if dev.type == 'cuda':
                host_keep = (host, torch.cuda.Event())                    # the pinned staging buffer outlives the copy
                host_keep[1].record()
                self._staging = getattr(self, '_staging', [])[-8:] + [host_keep]
            if cache:
                self.tables[dev] = (sig, table, off, njobs, ntiles)
        lib = _lib.load_library()
        with _on(dev):
            base = table.data_ptr()
            _chk(lib.cotr_train_perm_jobs(ctypes.c_void_p(base), ctypes.c_void_p(base + off), njobs, ntiles, _sp()), 'cotr_train_perm_jobs')
        self._last_table = table                                           # (stream-ordered allocator: alive until the launch ran)

    def refresh(self):
        """Every registered operand re-derived from its parameter in ONE launch (call after the optimiser step)."""
        live = [e for e in self.entries.values() if e['ref']() is not None]
        live = [e for e in live if e['scale_ref'] is None or e['scale_ref']() is not None]
        if len(live) != len(self.entries):
            self.entries = {k: v for k, v in self.entries.items() if any(v is e for e in live)}
            self.tables = {}
        by_dev = {}
        for e in live:
            by_dev.setdefault(e['device'], []).append(e)
        for dev_entries in by_dev.values():
            self._run(dev_entries, cache=True)                             # one cached table per device
        for e in live:
            e['fresh'], e['version'] = True, e['ref']()._version
        return len(live)

    def mark_stale(self, keep=()):
        keep = set(keep)
        for k, e in self.entries.items():
            if k not in keep:
                e['fresh'] = False

    def clear(self):
        self.entries, self.tables = {}, {}

    def hold(self):
        """Strong references to everything a captured step has baked the ADDRESS of into its graph: the destination buffers of the
        registered operands and the cached job tables.  Whoever replays such a graph keeps this list alive (GraphedTrainStep):
        a later miss, refresh or clear_weight_cache() then only drops the registry's own references, never the memory."""
        keep = [e['dst'] for e in self.entries.values()]
        keep += [t[1] for t in self.tables.values()]
        if getattr(self, '_last_table', None) is not None:
            keep.append(self._last_table)
        return keep


_derived = _Derived()


def clear_weight_cache():
    """Forget every derived operand (their buffers are re-made at the next use)."""
    _derived.clear()


def refresh_derived():
    """Re-derive every registered weight-shaped operand (W^T slices, packed / BN-scaled convolution weights) in one launch: call
    after the optimiser step.  -> number of operands."""
    return _derived.refresh()


def derived_keys():
    return tuple(_derived.entries)


def hold_derived():
    """References that keep every registered operand buffer and job table alive (for a captured step: its graph holds their addresses)."""
    return _derived.hold()


def mark_derived_stale(keep=()):
    """The parameters changed behind the registry's back (a replayed graph that does not refresh these operands itself)."""
    _derived.mark_stale(keep)


def _base_of(w):
    return w._base if w._base is not None else w


def weight_t(w):
    """w [N,K] (a parameter, or a contiguous row slice of one) -> contiguous w^T [K,N], kept current by the registry above."""
    n, k = w.shape
    assert w.is_contiguous()
    return _derived.get(_base_of(w), w.detach(), 'T', None, (1, n, k, 0, k, 1, 0, n), (k, n))


def conv_packed(w):
    """torch's [Cout, Cin, k, k] -> the kernels' [Cout][k*k*Cin] (tap-major, channel fastest)."""
    cout, cin, k, _ = w.shape
    if k == 1:
        return w.detach().reshape(cout, cin)
    kk = k * k
    return _derived.get(_base_of(w), w.detach(), 'P', None, (cout, cin, kk, cin * kk, kk, 1, kk * cin, cin), (cout, kk * cin))


def conv_scaled_t(w, scale):
    """(packed weight * FrozenBN scale per output channel)^T -> [k*k*Cin][Cout]: the weight operand of the dgrad GEMM."""
    cout, cin, k, _ = w.shape
    kk = k * k
    return _derived.get(_base_of(w), w.detach(), 'S', scale, (kk, cout, cin, 1, cin * kk, kk, cin * cout, cout), (kk * cin, cout))


def conv_dgrad_w(w, scale):
    """The data gradient of a 3 x 3 stride-1 convolution IS a 3 x 3 stride-1 convolution of dz with the kernel flipped and its channel
    axes exchanged: dx = conv(dz, W'), W'[cin][2-ky][2-kx][cout] = W[cout][cin][ky][kx] * scale[cout] - in the kernels' packed
    layout [Cin rows][k*k*Cout], derived once per optimiser step like the other weight-shaped operands."""
    cout, cin, k, _ = w.shape
    kk = k * k
    return _derived.get(_base_of(w), w.detach(), 'D', scale, (kk, cout, cin, 1, cin * kk, kk, cout, kk * cout, 1), (cin, kk * cout))


class Proj(torch.autograd.Function):
    """Row slices of ONE weight matrix applied to (possibly different) inputs: y_i = relu?(x_i . W[lo_i:hi_i]^T + b[lo_i:hi_i]),
    optionally followed by dropout.  nn.Linear is the one-slice case; the packed in_proj of nn.MultiheadAttention
    (transformer.py:149-153 with q = k = src + pos, v = src; :192-195 with q = tgt + query_pos, k = memory + pos, v = memory)
    is the two / three-slice case.  Backward: dx_i = dy_i . W[lo:hi] (GEMM on the cached W^T slice), dW[lo:hi] = dy_i^T . x_i
    (gemm_tn, written straight into its rows of the full-size gradient), db[lo:hi] = column sums."""

    @staticmethod
    def forward(ctx, w, b, ranges, relu, p, *xs):
        assert len(xs) == len(ranges) and (not relu or len(xs) == 1)
        xs = [x.contiguous() for x in xs]
        wd, bd = w.detach(), None if b is None else b.detach()
        ys = []
        for x, (lo, hi) in zip(xs, ranges):
            ys.append(gemm(x, wd[lo:hi], None if bd is None else bd[lo:hi], relu))
        seed = 0
        if relu and p > 0:
            seed = next_seed()
            lib = _lib.load_library()
            with _on(ys[0].device):
                _chk(lib.cotr_train_dropout_fwd(_P(ys[0]), ys[0].numel(), float(p), seed, _sp()), 'cotr_train_dropout_fwd')
        ctx.ranges, ctx.relu, ctx.p, ctx.has_bias = ranges, relu, float(p), b is not None
        ctx.bias = b                                       # (the parameter itself: GradSink needs to know whose gradient db is)
        ctx.save_for_backward(w, *xs, *(ys if relu else []))
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        w = saved[0]
        n = len(ctx.ranges)
        xs = saved[1:1 + n]
        lib = _lib.load_library()
        need_w = ctx.needs_input_grad[0]
        need_b = ctx.has_bias and ctx.needs_input_grad[1]
        dws, dbs, dxs = [], [], []
        sink, gw, gb = _sink, None, None
        if sink is not None and need_w:                   # gradients straight into the sink's buffers (GradSink)
            gw = sink.grad_of(w)
            gb = sink.grad_of(ctx.bias) if need_b else None
            if gw is None or (need_b and gb is None):
                gw = gb = None
        for i, ((lo, hi), x, dy) in enumerate(zip(ctx.ranges, xs, dys)):
            if dy is None:      # an output nobody used
                dy = torch.zeros((x.shape[0], hi - lo), dtype=torch.float32, device=x.device)
            dy = dy.contiguous()
            if ctx.relu:
                y = saved[1 + n]
                dh = torch.empty_like(dy)
                with _on(dy.device):
                    _chk(lib.cotr_train_relu_drop_bwd(_P(dy), _P(y), _P(dh), dy.numel(), ctx.p, _sp()), 'cotr_train_relu_drop_bwd')
                dy = dh
            dxs.append(gemm(dy, weight_t(w[lo:hi])) if ctx.needs_input_grad[5 + i] else None)
            if gw is not None:
                part, nparts, pstride = gemm_tn_parts(dy, x, with_colsum=need_b)
                nk = (hi - lo) * w.shape[1]
                sink.add(gw[lo:hi], part, 0, nparts, pstride, nk)
                if need_b:
                    sink.add(gb[lo:hi], part, nk, nparts, pstride, hi - lo)
            elif need_w:
                dwi, dbi = gemm_tn(dy, x, with_colsum=need_b)      # dW and db of this slice from one pass over dy
                dws.append(dwi)
                dbs.append(dbi)
            elif need_b:
                dbs.append(colsum(dy, _empty((hi - lo,), dy)))
        if gw is not None:
            return (None, None, None, None, None, *dxs)
        tiles = [lo for lo, _ in ctx.ranges] == [0] + [hi for _, hi in ctx.ranges[:-1]] and ctx.ranges[-1][1] == w.shape[0]
        if tiles:
            dw = (dws[0] if n == 1 else torch.cat(dws, dim=0)) if need_w else None
            db = (dbs[0] if n == 1 else torch.cat(dbs, dim=0)) if need_b else None
        else:       # some rows of the weight are used elsewhere (the decoder's k / v rows: ProjKV): zero gradient from here
            dw = db = None
            if need_w:
                dw = torch.zeros_like(w)
                for (lo, hi), dwi in zip(ctx.ranges, dws):
                    dw[lo:hi] = dwi
            if need_b:
                db = torch.zeros_like(ctx.bias)
                for (lo, hi), dbi in zip(ctx.ranges, dbs):
                    db[lo:hi] = dbi
        return (dw, db, None, None, None, *dxs)


def linear(x, w, b=None, relu=False, p=0.0):
    return Proj.apply(w, b, ((0, w.shape[0]),), relu, p, x)[0]


class AddRows(torch.autograd.Function):
    """x + x2[row % mod] over rows of 256 (mod == 0: same row): ``src + pos`` / ``memory + pos`` with the constant image
    position table, ``tgt + query_pos`` with the (no-grad) query encoding.  Gradient: identity to x."""

    @staticmethod
    def forward(ctx, x, x2, mod):
        x = x.contiguous()
        y = torch.empty_like(x)
        with _on(x.device):
            _chk(_lib.load_library().cotr_train_add_rowmod(_P(x), _P(x2), int(mod), _P(y), x.shape[0], _sp()), 'cotr_train_add_rowmod')
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


class AddDropLN(torch.autograd.Function):
    """y = LayerNorm(x + dropout(a)): the tail of every sub-layer (transformer.py:154-155,157-158,196-198,200-201); with
    x = None, p = 0 a plain LayerNorm (decoder.norm, :110-111)."""

    @staticmethod
    def forward(ctx, x, a, w, b, p):
        lib = _lib.load_library()
        a = a.contiguous()
        rows = a.shape[0]
        seed = next_seed() if p > 0 else 0
        s, y, stats = torch.empty_like(a), torch.empty_like(a), _empty((rows, 2), a)
        with _on(a.device):
            _chk(lib.cotr_train_add_drop_ln_fwd(_P(None if x is None else x.contiguous()), _P(a), _P(w.detach()), _P(b.detach()),
                                                _P(s), _P(y), _P(stats), rows, float(p), seed, _sp()), 'cotr_train_add_drop_ln_fwd')
        ctx.p, ctx.seed, ctx.has_x = float(p), seed, x is not None
        ctx.bias = b
        ctx.save_for_backward(s, stats, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load_library()
        s, stats, w = ctx.saved_tensors
        dy = dy.contiguous()
        rows = dy.shape[0]
        ds = torch.empty_like(dy)
        da = torch.empty_like(dy) if ctx.p > 0 else None          # p == 0: da == ds
        nparts = lib.cotr_train_ln_bwd_parts(rows)
        part = _empty((nparts * 512,), dy)
        sink = _sink
        gw = sink.grad_of(w) if sink is not None else None
        gb = sink.grad_of(ctx.bias) if gw is not None else None
        deferred = gb is not None and rows > 0
        dwb = None if deferred else _empty((512,), dy)
        with _on(dy.device):
            _chk(lib.cotr_train_ln_bwd(_P(dy), _P(s), _P(stats), _P(w), _P(ds), _P(da), _P(part), _P(dwb), rows, ctx.p, ctx.seed,
                                       _sp()), 'cotr_train_ln_bwd')
        if deferred:
            sink.add(gw, part, 0, nparts, 512, 256)
            sink.add(gb, part, 256, nparts, 512, 256)
            return (ds if ctx.has_x else None), (ds if da is None else da), None, None, None
        return (ds if ctx.has_x else None), (ds if da is None else da), dwb[:256], dwb[256:], None


def _rows_view(t):
    """``t`` [rows, 256] usable by the attention kernels in place: unit column stride, 16-byte aligned rows; else a contiguous copy."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


class ProjKV(torch.autograd.Function):
    """The k and v projections of ALL decoder layers (rows d:2d and 2d:3d of each layer's packed in_proj, transformer.py:192-195
    with k = memory + pos, v = memory) for all rows of ``memory`` - both passes of a training step - as TWO GEMMs
    [rows, 256] x [L*256, 256]^T instead of 2 L per pass: the projections do not depend on the queries (the inference path hoists
    them into the encode step the same way, csrc/api.hip).  Backward: two K = L*256 deep GEMMs give d(memory + pos) and d(memory)
    (instead of 4 L small ones and as many gradient additions), two transpose-free gemm_tn passes give the weight / bias gradients
    of all layers.  -> (K_all, V_all), each [rows, L*256]; layer l's block is columns l*256:(l+1)*256 (ColBlocks)."""

    @staticmethod
    def forward(ctx, mem_pos, memory, *wb):
        ws, bs = wb[0::2], wb[1::2]
        d = ws[0].shape[1]
        mem_pos, memory = mem_pos.contiguous(), memory.contiguous()
        with torch.no_grad():
            wk = torch.cat([w.detach()[d:2 * d] for w in ws], dim=0)
            wv = torch.cat([w.detach()[2 * d:3 * d] for w in ws], dim=0)
            bk = torch.cat([b.detach()[d:2 * d] for b in bs], dim=0)
            bv = torch.cat([b.detach()[2 * d:3 * d] for b in bs], dim=0)
        k_all, v_all = gemm(mem_pos, wk, bk), gemm(memory, wv, bv)
        ctx.params, ctx.d = (ws, bs), d
        ctx.save_for_backward(mem_pos, memory, wk, wv)
        return k_all, v_all

    @staticmethod
    def backward(ctx, dk_all, dv_all):
        lib = _lib.load_library()
        mem_pos, memory, wk, wv = ctx.saved_tensors
        ws, bs = ctx.params
        d, nl = ctx.d, len(ws)
        need_x = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_w = any(ctx.needs_input_grad[2:])
        sink = _sink
        gws = gbs = None
        if sink is not None and need_w:
            gws, gbs = [sink.grad_of(w) for w in ws], [sink.grad_of(b) for b in bs]
            if any(g is None for g in gws + gbs):
                gws = gbs = None
        dxs, grads = [], [None] * (2 * nl)
        for which, (dy, x, w_all, need) in enumerate(((dk_all, mem_pos, wk, need_x[0]), (dv_all, memory, wv, need_x[1]))):
            dy = dy.contiguous()
            dx = None
            if need:
                wt = _empty((d, nl * d), w_all)
                with _on(dy.device):
                    _chk(lib.cotr_train_transpose(_P(w_all), _P(wt), nl * d, d, _sp()), 'cotr_train_transpose')
                dx = gemm(dy, wt)
            dxs.append(dx)
            if not need_w:
                continue
            lo = (1 + which) * d                                     # rows of the packed in_proj this pass owns
            if gws is not None:
                part, nparts, pstride = gemm_tn_parts(dy, x, with_colsum=True)
                for l in range(nl):
                    sink.add(gws[l][lo:lo + d], part, l * d * d, nparts, pstride, d * d)
                    sink.add(gbs[l][lo:lo + d], part, nl * d * d + l * d, nparts, pstride, d)
            else:
                dw_all, db_all = gemm_tn(dy, x, with_colsum=True)
                for l in range(nl):
                    if grads[2 * l] is None:
                        grads[2 * l], grads[2 * l + 1] = torch.zeros_like(ws[l]), torch.zeros_like(bs[l])
                    grads[2 * l][lo:lo + d] = dw_all[l * d:(l + 1) * d]
                    grads[2 * l + 1][lo:lo + d] = db_all[l * d:(l + 1) * d]
        return (dxs[0], dxs[1], *grads)


class ColBlocks(torch.autograd.Function):
    """x [rows, cols] -> its row_blocks x col_blocks blocks as views (block (h, l) = rows h*R:(h+1)*R, columns l*C:(l+1)*C), in
    (h, l) order.  Each view carries ``_grad_dst``, its block of ``gbuf``: a consumer that can write its gradient anywhere
    (Attention.backward: dk / dv through a leading dimension) writes it THERE and returns that view, so the gradient of x is
    assembled without a copy or an addition; a gradient that arrives elsewhere is copied in, a block nobody used is zeroed."""

    @staticmethod
    def forward(ctx, x, gbuf, row_blocks, col_blocks):
        rows, cols = x.shape
        assert gbuf.shape == x.shape and gbuf.is_contiguous() and x.is_contiguous() and rows % row_blocks == 0 and cols % col_blocks == 0
        r, c = rows // row_blocks, cols // col_blocks
        ctx.gbuf, ctx.blocks = gbuf, [(h * r, l * c, r, c) for h in range(row_blocks) for l in range(col_blocks)]
        return tuple(x[r0:r0 + r, c0:c0 + c] for r0, c0, _, _ in ctx.blocks)

    @staticmethod
    def backward(ctx, *grads):
        gbuf = ctx.gbuf
        for (r0, c0, r, c), g in zip(ctx.blocks, grads):
            dst = gbuf[r0:r0 + r, c0:c0 + c]
            if g is None:
                dst.zero_()
            elif g.data_ptr() != dst.data_ptr() or g.stride() != dst.stride():
                dst.copy_(g)
        return gbuf, None, None, None


def col_blocks(x, row_blocks, col_blocks_):
    """ColBlocks with the gradient destinations attached -> list over row blocks of lists over column blocks."""
    gbuf = torch.empty_like(x)
    outs = ColBlocks.apply(x, gbuf, row_blocks, col_blocks_)
    r, c = x.shape[0] // row_blocks, x.shape[1] // col_blocks_
    res = []
    for h in range(row_blocks):
        row = []
        for l in range(col_blocks_):
            o = outs[h * col_blocks_ + l]
            o._grad_dst = gbuf[h * r:(h + 1) * r, l * c:(l + 1) * c]
            row.append(o)
        res.append(row)
    return res


def _claim_grad_dst(t):
    """The in-place gradient destination ColBlocks attached to this block view, for its FIRST consumer; None afterwards."""
    dst = getattr(t, '_grad_dst', None) if t is not None else None
    if dst is None or getattr(t, '_grad_dst_claimed', False):
        return None
    t._grad_dst_claimed = True
    return dst


class Attention(torch.autograd.Function):
    """o = dropout(softmax(q k^T * scale)) v per head - the core of nn.MultiheadAttention (transformer.py:149-153, 192-195).
    ``qk`` given: q and k are the two column halves of one [rows, 512] tensor (encoder self-attention: one projection launch
    produced both) and their gradient comes back as one tensor; otherwise q [nb*nq, 256] and k [nb*512, 256] are separate
    (decoder cross-attention).  v [nb*512, 256].  The kernels take a pointer and a leading dimension per operand, so
    nothing is copied or re-laid-out on the way in or out."""

    @staticmethod
    def forward(ctx, qk, q, k, v, nb, nq, scale, p):
        lib = _lib.load_library()
        packed = qk is not None
        if packed:
            qk = qk.contiguous()
            q_t, k_t, ldq, ldk = qk, qk, 512, 512
            k_ptr = ctypes.c_void_p(qk.data_ptr() + 256 * 4)
        else:
            # k / v may be column blocks of one wide projection output (ColBlocks: the decoder's K / V of all layers from one GEMM):
            # the kernels take a leading dimension, so such a view is used in place
            q_t, k_t, ldq = q.contiguous(), _rows_view(k), 256
            ldk = k_t.stride(0)
            k_ptr = _P(k_t)
        # A block's gradient destination can be written in place by ONE consumer only: the first Attention to use a k / v block
        # claims it, a second consumer of the same block (decode_train called twice with one kv list) gets no destination, returns
        # a fresh dk / dv, and autograd adds the two before ColBlocks.backward copies the sum in - instead of the second write
        # silently replacing the first.
        ctx.k_dst, ctx.v_dst = _claim_grad_dst(k), _claim_grad_dst(v)
        v = _rows_view(v)
        ldv = v.stride(0)
        rows = nb * nq
        o, lse = _empty((rows, 256), v), _empty((rows, 8), v)
        seed = next_seed() if p > 0 else 0
        with _on(v.device):
            _chk(lib.cotr_train_attention_fwd(_P(q_t), ldq, k_ptr, ldk, _P(v), ldv, _P(o), 256, _P(lse), nb, nq, float(scale),
                                              float(p), seed, _sp()), 'cotr_train_attention_fwd')
        ctx.meta = (packed, nb, nq, float(scale), float(p), seed)
        ctx.save_for_backward(q_t, k_t, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, d_o):
        lib = _lib.load_library()
        packed, nb, nq, scale, p, seed = ctx.meta
        q_t, k_t, v, o, lse = ctx.saved_tensors
        d_o = d_o.contiguous()
        rows = nb * nq
        delta = _empty((rows, 8), o)
        # (gradient destinations handed over by ColBlocks: dk / dv are written straight into their column block of the wide
        # gradient buffer, nothing is copied or added afterwards)
        dv = ctx.v_dst if ctx.v_dst is not None else _empty((nb * 512, 256), o)
        lddv, ldv = dv.stride(0), v.stride(0)
        if packed:
            dqk = _empty((rows, 512), o)
            k_ptr = ctypes.c_void_p(k_t.data_ptr() + 256 * 4)
            dq_ptr, dk_ptr, ldq, ldk = _P(dqk), ctypes.c_void_p(dqk.data_ptr() + 256 * 4), 512, 512
            lddq = lddk = 512
        else:
            dq = _empty((rows, 256), o)
            dk = ctx.k_dst if ctx.k_dst is not None else _empty((nb * 512, 256), o)
            k_ptr = _P(k_t)
            dq_ptr, dk_ptr, ldq, ldk, lddq, lddk = _P(dq), _P(dk), 256, k_t.stride(0), 256, dk.stride(0)
        # room for the dQ partials of the key-split one-pass backward (decoder: few pairs); not needed from 24 pairs up
        scratch = None if (packed or nb * 8 >= 192) else _empty((lib.cotr_train_attention_bwd_scratch(nb, nq),), o)
        with _on(o.device):
            _chk(lib.cotr_train_attention_bwd(_P(q_t), ldq, k_ptr, ldk, _P(v), ldv, _P(o), _P(d_o), 256, _P(lse), _P(delta),
                                              dq_ptr, lddq, dk_ptr, lddk, _P(dv), lddv, nb, nq, scale, p, seed, _P(scratch), _sp()),
                 'cotr_train_attention_bwd')
        if packed:
            return dqk, None, None, dv, None, None, None, None
        return None, dq, dk, dv, None, None, None, None


IMPLICIT_WGRAD = os.environ.get('COTR_IMPLICIT_WGRAD', '1') not in ('', '0')   # tools / tests: 0 = the explicit im2col image everywhere (A/B, bit-identity check)


IMPLICIT_DGRAD = os.environ.get('COTR_IMPLICIT_DGRAD', '1') not in ('', '0')   # 0 = dz . W^T + col2im everywhere (A/B)


class ConvBN(torch.autograd.Function):
    """One convolution of a trainable bottleneck (layer2 / layer3: COTR/models/backbone.py:66-69 trains only these) with its
    FrozenBatchNorm2d affine (backbone.py:46-56), optional residual and ReLU, on the NHWC side-by-side layout of the inference
    kernels: y = relu?(conv(x, W) * scale + bias (+ res)).  Forward = the inference implicit-GEMM kernel (cotr_op_conv).
    Backward: dz = dy * (y > 0); wgrad = dz^T . im2col(x) * scale (transpose-free split-M GEMM); dgrad = col2im(dz . (W * scale));
    1 x 1 stride-1 convolutions skip the im2col / col2im (the activation matrix is its own im2col)."""

    @staticmethod
    def forward(ctx, x, w, scale, bias, res, relu, stride):
        lib = _lib.load_library()
        x = x.contiguous()
        b, h, w2, cin = x.shape
        wd = w2 // 2
        cout, _, k, _ = w.shape
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
        wp = conv_packed(w)
        y = _empty((b, ho, 2 * wo, cout), x)
        with _on(x.device):
            _chk(lib.cotr_op_conv(_P(x), _P(wp), _P(scale), _P(bias), _P(None if res is None else res.contiguous()), int(relu), _P(y),
                                  b, h, wd, cin, cout, k, stride, _sp()), 'cotr_op_conv')
        ctx.meta = (b, h, wd, cin, cout, k, stride, ho, wo, bool(relu), res is not None)
        ctx.weight = w
        ctx.save_for_backward(x, scale, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, y = ctx.saved_tensors
        dx, dw, dz = _conv_backward(ctx.meta, x, scale, y, ctx.weight, dy, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return dx, dw, None, None, (dz if ctx.meta[10] else None), None, None


def _conv_backward(meta, x, scale, y, weight, dy, need_dx, need_dw, dx_residual=None):
    """Backward of y = relu?(conv(x, W) * scale + bias (+ res)) -> (dx, dW or None, dz): dz = dy * (y > 0) (the gradient of a residual
    input, if there was one); the weight gradient goes to the GradSink where one is installed (dW is None then); ``dx_residual`` (1 x 1
    stride-1 convolutions) is added to dx in the epilogue of the data-gradient GEMM - a gradient that arrives at x by another path
    (the bottleneck's identity branch) without an add launch of its own, the same single fp32 addition."""
    lib = _lib.load_library()
    b, h, wd, cin, cout, k, stride, ho, wo, relu, has_res = meta
    m, kk = b * ho * 2 * wo, k * k * cin
    dy = dy.contiguous()
    dz = dy
    with _on(dy.device):
        if relu:
            dz = torch.empty_like(dy)
            _chk(lib.cotr_train_relu_drop_bwd(_P(dy), _P(y), _P(dz), dy.numel(), 0.0, _sp()), 'cotr_train_relu_drop_bwd')
        dz2 = dz.view(m, cout)
        direct = k == 1 and stride == 1
        assert dx_residual is None or direct
        # weight-gradient partials: 1 x 1 stride-1 - the activation matrix is its own im2col image; else straight from x where the
        # implicit form applies (round 6: no im2col launch, no [m, k*k*cin] image - the same bits), else from an explicit image
        wparts = None
        if need_dw:
            wparts = gemm_tn_parts(dz2, x.view(m, cin)) if direct else (IMPLICIT_WGRAD and conv_wgrad_parts(dz2, x, b, h, wd, cin, cout, k, stride)) or None
            if wparts is None:
                col = _empty((m, kk), x)
                _chk(lib.cotr_train_im2col(_P(x), _P(col), b, h, wd, cin, k, stride, _sp()), 'cotr_train_im2col')
                wparts = gemm_tn_parts(dz2, col)
        dw = None
        gw = _sink.grad_of(weight) if (_sink is not None and need_dw) else None
        if gw is not None:                                                # scale + layout inside the deferred reduction
            part, nparts, pstride = wparts
            _sink.add(gw, part, 0, nparts, pstride, cout * kk, scale=scale, row_len=kk, cin=cin, taps=k * k)
        elif need_dw:
            dwp = sum_parts(wparts[0], wparts[1], cout * kk).view(cout, kk)   # d(W * scale), packed layout
            _chk(lib.cotr_train_scale_rows(_P(dwp), _P(scale), _P(dwp), cout, kk, _sp()), 'cotr_train_scale_rows')
            if k == 1:
                dw = dwp.view(cout, cin, 1, 1)
            else:
                dw = _empty((cout, cin, k, k), x)
                _chk(lib.cotr_train_transpose_batched(_P(dwp), _P(dw), cout, k * k, cin, _sp()), 'cotr_train_transpose_batched')
        dx = None
        if need_dx and IMPLICIT_DGRAD and k == 3 and stride == 1:
            # round 6: the data gradient as ONE implicit-GEMM launch of the forward's convolution kernel on the flipped kernel - no
            # [m, 9 * cin] image of partial products (75 MB per layer3 convolution at 16 pairs) and no col2im pass over it
            dx = torch.empty_like(x)
            _chk(lib.cotr_op_conv(_P(dz), _P(conv_dgrad_w(weight, scale)), None, None, None, 0, _P(dx), b, ho, wo, cout, cin, 3, 1, _sp()),
                 'cotr_op_conv (dgrad)')
        elif need_dx:
            wst = conv_scaled_t(weight, scale)                            # (W * scale)^T, derived once per optimiser step
            dcol = gemm(dz2, wst, residual=None if dx_residual is None else dx_residual.contiguous())   # [m, k*k*cin]
            if direct:
                dx = dcol.view(b, h, 2 * wd, cin)
            else:
                dx = torch.empty_like(x)
                _chk(lib.cotr_train_col2im(_P(dcol), _P(dx), b, h, wd, cin, k, stride, _sp()), 'cotr_train_col2im')
    return dx, dw, dz


def _conv_forward(x, w, scale, bias, res, relu, stride):
    """-> (y, meta) of one convolution + FrozenBN affine (+ residual) (+ ReLU) on the inference kernel (cotr_op_conv)."""
    lib = _lib.load_library()
    b, h, w2, cin = x.shape
    wd = w2 // 2
    cout, _, k, _ = w.shape
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
    y = _empty((b, ho, 2 * wo, cout), x)
    with _on(x.device):
        _chk(lib.cotr_op_conv(_P(x), _P(conv_packed(w)), _P(scale), _P(bias), _P(None if res is None else res.contiguous()), int(relu), _P(y),
                              b, h, wd, cin, cout, k, stride, _sp()), 'cotr_op_conv')
    return y, (b, h, wd, cin, cout, k, stride, ho, wo, bool(relu), res is not None)


BOTTLENECK_FN = os.environ.get('COTR_BOTTLENECK_FN', '1') not in ('', '0')   # 0 = four ConvBN nodes per block (A/B, bit-identity check)


class Bottleneck(torch.autograd.Function):
    """One trainable ResNet bottleneck (torchvision v1.5: stride on the 3x3; FrozenBN; COTR/models/backbone.py:46-56, 66-69) as ONE
    autograd node: conv1 1x1 + ReLU, conv2 3x3 (stride) + ReLU, [downsample 1x1 (stride)], conv3 1x1 + identity + ReLU - the same
    twelve-odd launches forward and backward as four ConvBN nodes, minus the add launch autograd makes where the two paths meet at
    the block's input: the identity branch's gradient enters conv1's data-gradient GEMM as its residual (one fp32 addition either
    way: the same bits).  Arguments: x, then (weight, scale, bias) of conv1, conv2, conv3 and of the downsample (None x 3 without
    one), then the stride."""

    @staticmethod
    def forward(ctx, x, w1, s1, b1, w2, s2, b2, w3, s3, b3, wd, sd, bd, stride):
        x = x.contiguous()
        t1, m1 = _conv_forward(x, w1, s1, b1, None, True, 1)
        t2, m2 = _conv_forward(t1, w2, s2, b2, None, True, stride)
        idt, md = (x, None) if wd is None else _conv_forward(x, wd, sd, bd, None, False, stride)
        y, m3 = _conv_forward(t2, w3, s3, b3, idt, True, 1)
        ctx.metas = (m1, m2, m3, md)
        ctx.weights = (w1, w2, w3, wd)
        ctx.save_for_backward(x, t1, t2, y, s1, s2, s3, *((sd,) if wd is not None else ()))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, t1, t2, y, s1, s2, s3, *rest = ctx.saved_tensors
        m1, m2, m3, md = ctx.metas
        w1, w2, w3, wd = ctx.weights
        need = ctx.needs_input_grad
        # conv3 (+ identity + ReLU): dz3 is also the gradient of the identity input
        dt2, dw3, dz3 = _conv_backward(m3, t2, s3, y, w3, dy, True, need[7])
        # identity branch: the block input itself, or the downsample convolution of it
        dwd = None
        if wd is None:
            didt = dz3
        else:
            didt, dwd, _ = _conv_backward(md, x, rest[0], None, wd, dz3, need[0], need[10])
        dt1, dw2, _ = _conv_backward(m2, t1, s2, t2, w2, dt2, True, need[4])
        dx, dw1, _ = _conv_backward(m1, x, s1, t1, w1, dt1, need[0], need[1], dx_residual=didt if need[0] else None)
        return dx, dw1, None, None, dw2, None, None, dw3, None, None, dwd, None, None, None


class Head(torch.autograd.Function):
    """pred = h . W2^T + b2, the last corr_embed layer (256 -> 2, position_encoding.py:23-26) -> [nb, nq, 2]."""

    @staticmethod
    def forward(ctx, h, w2, b2, nb, nq):
        lib = _lib.load_library()
        h = h.contiguous()
        y = _empty((nb, nq, 2), h)
        with _on(h.device):
            _chk(lib.cotr_train_head_fwd(_P(h), _P(w2.detach().contiguous()), _P(b2.detach()), _P(y), nb, nq, _sp()), 'cotr_train_head_fwd')
        ctx.bias = b2
        ctx.save_for_backward(h, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load_library()
        h, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        rows = h.shape[0]
        dh = torch.empty_like(h)
        nparts = lib.cotr_train_head_bwd_parts(rows)
        part = _empty((nparts * 514,), h)
        sink = _sink
        gw = sink.grad_of(w2) if sink is not None else None
        gb = sink.grad_of(ctx.bias) if gw is not None else None
        deferred = gb is not None and rows > 0
        dwb = None if deferred else _empty((514,), h)
        with _on(h.device):
            _chk(lib.cotr_train_head_bwd(_P(dy), _P(h), _P(w2.contiguous()), _P(dh), _P(part), _P(dwb), rows, _sp()), 'cotr_train_head_bwd')
        if deferred:
            sink.add(gw, part, 0, nparts, 514, 512)
            sink.add(gb, part, 512, nparts, 514, 2)
            return dh, None, None, None, None
        return dh, dwb[:512].view(2, 256), dwb[512:], None, None
