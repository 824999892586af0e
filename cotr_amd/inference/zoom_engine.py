"""Recursive zoom-in refinement, one launch per zoom level.

Mirrors, for ALL queries at once (vectorised numpy state instead of one Python object per query):

* ``RefinementTask``  (COTR/inference/refinement_task.py:15-188): per-query zoom state machine - crop both
  images around (loc_from, cur_loc_to) at scale s*zoom, network on the 256x512 side-by-side crop with ONE query,
  map the answer back to pixels (``scale_to_loc`` :145-151), advance / detect loops / finish (``step`` :153-182),
  ``conclude`` (:184-188).
* ``get_patch_centered_at`` (COTR/inference/inference_helper.py:78-102): the crop-box arithmetic.
* the hot loop of ``SparseEngine.cotr_corr_multiscale`` (COTR/inference/sparse_engine.py:208-218) for tasks with
  known scale (``gen_tasks_w_known_scale`` :100-106) and ``conclude_tasks`` (:58-84).

The reference walks tasks 32 at a time, and for every task and level does two PIL resizes on the host, a 1.5 MB
H2D copy and a full backbone+encoder pass inside ``model(img[<=32], q[<=32,1])``.  Here a level is: box arithmetic
on the host (a few numpy ops on [N] arrays), ONE ``cotr_crop_resize_pairs`` launch that builds all N network inputs
on the device (bit-exact with Pillow), and the model called on chunks of ``max_pairs`` crops.  Queries never
interact, so per-query results are the reference's; only the batching differs.

The arithmetic follows the reference's dtypes: the network answer is float32, ``(x - 0.5) * 2`` is done in float32,
the multiplication by the (integer) patch size and the offset addition in float64.
"""
import ctypes
from collections import namedtuple

import numpy as np
import torch

BASE_ZOOM = 1.0                     # COTR/inference/inference_helper.py:17
THRESHOLD_PIXELS_RELATIVE = 0.02    # :16
THRESHOLD_SPARSE = 0.02             # :15
THRESHOLD_AREA = 0.02               # :18
MAX_SIZE = 256                      # COTR/utils/constants.py:2


def _affine_from_3pts(src, dst):
    """2x3 affine map through three point pairs - what ``cv2.getAffineTransform`` returns (float64)."""
    a = np.concatenate([np.asarray(src, dtype=np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, np.asarray(dst, dtype=np.float64)).T


def _patch_affines(p_i, p_j, shape_a, shape_b):
    """(T_i, T_j) of inference_helper.py:151-156: network frame [-1,1]^2 -> normalised image frame of the patch in
    image b (for the left half's answers) and in image a (for the right half's)."""
    base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]])

    def corners(p, shape):
        x, y, s = p
        return (np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s]]) / np.array([shape[1], shape[0]])) * 2 + np.array([-1, -1])
    real_j, real_i = corners(p_j, shape_b), corners(p_i, shape_a)
    return (_affine_from_3pts(base[:3].astype(np.float32), real_j[:3].astype(np.float32)),
            _affine_from_3pts(base[:3].astype(np.float32), real_i[:3].astype(np.float32)))


class _DeviceDensePost:
    """Dense-pass post-processing on the GPU: cotr_dense_cycle (1 launch) + cotr_dense_merge (1 launch per image);
    only the merged maps come back to the host."""

    def __init__(self, device):
        from .. import _lib
        self._lib = _lib
        self.lib = _lib.load_library()
        self.device = device

    def device_maps(self, pred, pairs, shape_a, shape_b):
        """-> (corr_a [Ha,Wa,2], con_a [Ha,Wa], corr_b, con_b) float32 tensors ON THE DEVICE (nothing is copied back)."""
        lib, dev = self.lib, self.device
        n = len(pairs)
        pred = pred.detach().to(dev, torch.float32).contiguous()
        assert pred.shape == (n, MAX_SIZE, MAX_SIZE * 2, 2) or pred.shape == (n, MAX_SIZE * MAX_SIZE * 2, 2)
        aff = np.stack([np.stack(_patch_affines(p_i, p_j, shape_a, shape_b)) for p_i, p_j in pairs])     # [n,2,2,3] f64
        aff_d = torch.from_numpy(np.ascontiguousarray(aff)).to(dev)
        maps = torch.empty((n, MAX_SIZE, MAX_SIZE * 2, 3), dtype=torch.float32, device=dev)
        out = []
        with torch.cuda.device(dev):
            stream = self._lib.current_stream_ptr()
            rc = lib.cotr_dense_cycle(pred.data_ptr(), n, aff_d.data_ptr(), maps.data_ptr(), stream)
            if rc != 0:
                raise self._lib.CotrHipError(f'cotr_dense_cycle failed (code {rc})')
            boxes = torch.tensor([list(p[0]) + list(p[1]) for p in pairs], dtype=torch.int32).to(dev)   # one upload for both sides
            for side, shape in ((0, shape_a), (1, shape_b)):
                bx = boxes[:, 3 * side:3 * side + 3].contiguous()
                flow = torch.empty((shape[0], shape[1], 2), dtype=torch.float32, device=dev)
                conf = torch.empty((shape[0], shape[1]), dtype=torch.float32, device=dev)
                rc = lib.cotr_dense_merge(maps.data_ptr(), bx.data_ptr(), n, side, shape[0], shape[1], flow.data_ptr(),
                                          conf.data_ptr(), stream)
                if rc != 0:
                    raise self._lib.CotrHipError(f'cotr_dense_merge failed (code {rc})')
                out += [flow, conf]
        return tuple(out)

    def to_host(self, tensors):
        """Device float32 maps -> float64 numpy arrays (the reference's maps are float64 arrays holding float32 values): widened on the
        device, copied through ONE pinned staging buffer with ONE synchronisation (pageable D2H of the four 5-10 MB maps + four host
        astype passes were a third of the dense pass at engine level)."""
        wide = [t.double().contiguous() for t in tensors]
        total = sum(w.numel() for w in wide)
        stage = torch.empty(total, dtype=torch.float64, pin_memory=True)
        off = 0
        views = []
        for w in wide:
            v = stage[off:off + w.numel()].view(w.shape)
            v.copy_(w, non_blocking=True)
            views.append(v)
            off += w.numel()
        torch.cuda.current_stream(self.device).synchronize()
        return tuple(v.numpy().copy() for v in views)

    def __call__(self, pred, pairs, shape_a, shape_b):
        return self.to_host(self.device_maps(pred, pairs, shape_a, shape_b))

    def resize(self, arr, shape):
        """``utils.float_image_resize`` (utils.py:69-83) of a [H,W] or [H,W,C] map to ``shape`` = (H', W'): float32 result,
        like the Pillow mode-'F' round trip of the reference."""
        a = np.ascontiguousarray(arr, dtype=np.float32)
        src = torch.from_numpy(a.reshape(a.shape[0], a.shape[1], -1)).to(self.device)
        dst = torch.empty((shape[0], shape[1], src.shape[2]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.cotr_resize_f32(src.data_ptr(), src.shape[0], src.shape[1], src.shape[2], dst.data_ptr(), shape[0],
                                          shape[1], self._lib.current_stream_ptr())
        if rc != 0:
            raise self._lib.CotrHipError(f'cotr_resize_f32 failed (code {rc})')
        out = dst.cpu().numpy()
        return out[..., 0] if a.ndim == 2 else out


RefineResult = namedtuple('RefineResult', ['loc_from', 'loc_to', 'good', 'loc_history', 'model_calls', 'crops', 'steps',
                                           'last_iters'], defaults=(None,))


def _reference_schedule_steps(steps, good, batch_size, max_corrs, total):
    """How many steps has the reference's loop given each task when it exits?  (sparse_engine.py:208-218: every
    iteration steps the first ``batch_size`` unfinished tasks of the list once; it stops when none is left or when
    ``max_corrs`` finished tasks are 'good'.)  ``steps``/``good`` cover the first len(steps) of ``total`` tasks.
    -> executed step count per task of the known prefix, or None if the answer depends on tasks beyond the prefix."""
    steps = np.asarray(steps, dtype=np.int64)
    remaining = steps.copy()
    n_good = 0
    while True:
        window = np.flatnonzero(remaining > 0)[:batch_size]
        if n_good >= max_corrs:
            return steps - remaining            # stops here whatever follows in the list
        if window.size < batch_size and len(remaining) < total:
            return None                         # the next batch would reach into tasks not refined yet
        if window.size == 0:
            return steps - remaining
        remaining[window] -= 1
        n_good += int(np.asarray(good)[window[remaining[window] == 0]].sum())


def _reference_schedule(steps, good, batch_size, max_corrs, total):
    """-> mask of the tasks the reference's loop has FINISHED when it exits (None: needs more of the list)."""
    executed = _reference_schedule_steps(steps, good, batch_size, max_corrs, total)
    return None if executed is None else executed == np.asarray(steps, dtype=np.int64)


def patch_boxes(img_shape, pos, scale):
    """``get_patch_centered_at(None, pos, scale, return_content=False, img_shape=...)`` for an [N,2] array of
    (x, y) positions -> (x, y, size) int arrays.  inference_helper.py:78-102."""
    h, w = img_shape[0], img_shape[1]
    short = min(h, w)
    scale = float(np.clip(scale, 0.0, 1.0))
    size = short * scale
    size = int((size // 2) * 2)
    pos = np.asarray(pos, dtype=np.float64)
    lu_x = np.trunc(pos[:, 0] - size // 2).astype(np.int64)      # int(): truncation toward zero
    lu_y = np.trunc(pos[:, 1] - size // 2).astype(np.int64)
    lu_x = np.where(lu_x < 0, 0, lu_x)
    lu_y = np.where(lu_y < 0, 0, lu_y)
    lu_x = np.where(lu_x + size > w, w - size, lu_x)
    lu_y = np.where(lu_y + size > h, h - size, lu_y)
    return lu_x, lu_y, size


class _DeviceCropper:
    """All network inputs of a level in one launch of the HIP crop+resize kernel (cotr_crop_resize_pairs)."""

    def __init__(self, img_a, img_b, device):
        from .. import _lib
        self._lib = _lib
        self.lib = _lib.load_library()
        self.device = device
        for name, im in (('img_a', img_a), ('img_b', img_b)):    # what PIL.Image.fromarray(patch) accepts in the reference
            if not (isinstance(im, np.ndarray) and im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3):
                raise ValueError(f'{name} must be an HxWx3 uint8 array (got {getattr(im, "dtype", type(im))}, '
                                 f'shape {getattr(im, "shape", None)})')
        if device.type != 'cuda':
            raise _lib.CotrHipError('the device-side crop kernel needs the model (and its inputs) on the MI355X; for host-side '
                                    'experiments pass make_cropper=...')
        self.a = torch.from_numpy(np.ascontiguousarray(img_a)).to(device)
        self.b = torch.from_numpy(np.ascontiguousarray(img_b)).to(device)
        self.shape_a, self.shape_b = img_a.shape, img_b.shape

    def __call__(self, boxes, out):
        """boxes int32 [n,6] (xa, ya, sa, xb, yb, sb) host array -> fills out[:n] ([n,3,256,512] device tensor)."""
        n = boxes.shape[0]
        bx = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.int32)).to(self.device)
        max_size = int(max(boxes[:, 2].max(), boxes[:, 5].max()))
        with torch.cuda.device(self.device):
            rc = self.lib.cotr_crop_resize_pairs(
                ctypes.c_void_p(self.a.data_ptr()), self.shape_a[0], self.shape_a[1],
                ctypes.c_void_p(self.b.data_ptr()), self.shape_b[0], self.shape_b[1],
                ctypes.c_void_p(bx.data_ptr()), n, ctypes.c_void_p(out.data_ptr()), max_size,
                self._lib.current_stream_ptr())
        if rc != 0:
            raise self._lib.CotrHipError(f'cotr_crop_resize_pairs failed (code {rc})')
        return out[:n]


class ZoomTask:
    """State of ONE query's recursive zoom - the fields and methods of the reference's ``RefinementTask``
    (COTR/inference/refinement_task.py:15-188) that the engines and their callers read: ``loc_from, best_loc_to,
    cur_loc_to, loc_history, status, result, submitted, identifier, total_iter, cur_zoom, conclude(force)``.  Crops are
    boxes (x, y, size) - the pixels are cut by the device kernel for a whole batch at once - so there is no image
    content in here.  ``ZoomEngine`` keeps this state in arrays and only materialises ZoomTask objects for
    ``return_tasks_only``; ``FasterSparseEngine`` (whose grouping couples tasks) steps them one by one like the reference."""

    def __init__(self, shape_from, shape_to, loc_from, loc_to, area_from, area_to, converge_iters, zoom_ins, identifier=None):
        self.identifier = identifier
        self.shape_from, self.shape_to = tuple(shape_from), tuple(shape_to)
        self.loc_from = np.asarray(loc_from, dtype=np.float64)
        self.best_loc_to = self.cur_loc_to = np.asarray(loc_to, dtype=np.float64)
        if area_from < area_to:                                     # refinement_task.py:25-30
            self.s_from, self.s_to = BASE_ZOOM, BASE_ZOOM * np.sqrt(area_to / area_from)
        else:
            self.s_to, self.s_from = BASE_ZOOM, BASE_ZOOM * np.sqrt(area_from / area_to)
        self.status, self.result, self.submitted = 'unfinished', 'unknown', False
        self.converge_iters, self.zoom_ins = converge_iters, list(zoom_ins)
        self.cur_zoom_idx = self.cur_iter = self.total_iter = 0
        self.loc_to_at_zoom = []
        self.loc_history = [self.cur_loc_to]
        self.job = None             # (xa, ya, sa, xb, yb, sb) of the crop pair the pending answer refers to

    @property
    def cur_zoom(self):
        return self.zoom_ins[self.cur_zoom_idx]

    def peek(self):
        """Boxes this task would crop now (``peek`` :58-67): (xa, ya, sa, xb, yb, sb)."""
        assert self.status == 'unfinished'
        xa, ya, sa = patch_boxes(self.shape_from, self.loc_from[None], self.s_from * self.cur_zoom)
        xb, yb, sb = patch_boxes(self.shape_to, self.cur_loc_to[None], self.s_to * self.cur_zoom)
        return (int(xa[0]), int(ya[0]), sa, int(xb[0]), int(yb[0]), sb)

    def submit(self, boxes=None):
        """``get_task`` (own crops, :105-132) / ``get_task_pilot`` (somebody else's crops, :69-85) -> (boxes, query):
        the query in the frame of the left crop, float64 math then float32 (:110)."""
        assert self.status == 'unfinished' and not self.submitted
        self.job = self.peek() if boxes is None else tuple(boxes)
        xa, ya, sa = self.job[:3]
        query = ((self.loc_from - np.array([xa, ya])) / np.array([sa * 2, sa])).astype(np.float32)
        self.submitted = True
        return self.job, query

    def step(self, raw):
        """``step`` (:153-182) with ``scale_to_loc`` (:145-151): float32 answer, (x - 0.5) * 2 in float32, patch scaling
        and offset in float64."""
        assert self.submitted
        self.submitted = False
        raw = np.array(raw, dtype=np.float32)
        raw[0] = (raw[0] - np.float32(0.5)) * np.float32(2)
        xb, yb, sb = self.job[3:]
        loc_to = raw.astype(np.float64) * np.array([sb, sb]) + np.array([xb, yb])
        self.total_iter += 1
        self.loc_to_at_zoom.append(loc_to)
        self.cur_loc_to = loc_to
        finished = True                                             # every level but the last takes one step
        if self.cur_zoom_idx == len(self.zoom_ins) - 1:
            prev = np.array(self.loc_to_at_zoom[:-1]).reshape(-1, 2)
            finished = bool(len(prev) and (prev == loc_to).all(axis=1).any()) or self.cur_iter >= self.converge_iters - 1
            self.cur_iter += 1
        if finished:
            arr = np.array(self.loc_to_at_zoom)
            if len(arr) >= 2 and (arr[:-1] == arr[-1]).all(axis=1).any():       # find_prediction_loop
                start = np.where((arr[:-1] == arr[-1]).all(axis=1))[0][0]
                loc_to = arr[start:-1].mean(axis=0)
            self.loc_history.append(loc_to)
            self.best_loc_to = self.cur_loc_to = loc_to
            if self.cur_zoom_idx >= len(self.zoom_ins) - 1:                     # next_zoom :134-143
                self.status = 'finished'
                self.result = 'bad' if self.conclude() is None else 'good'
            self.cur_zoom_idx += 1
            self.cur_iter = 0
            self.loc_to_at_zoom = []

    def conclude(self, force=False):
        """:184-188 -> [x_a, y_a, x_b, y_b] or None when the levels disagree by more than 2 % of the image."""
        hist = np.array(self.loc_history)
        if not force and max(hist.std(axis=0)) >= THRESHOLD_PIXELS_RELATIVE * max(*self.shape_to):
            return None
        return np.concatenate([self.loc_from, self.best_loc_to])


def conclude_tasks(tasks, return_idx=False, force=False, img_a_shape=None, img_b_shape=None):
    """``SparseEngine.conclude_tasks`` (sparse_engine.py:58-84) for a list of ZoomTask."""
    corrs, idx = [], []
    for t in tasks:
        if t.status == 'finished':
            out = t.conclude(force)
            if out is not None:
                corrs.append(np.array(out))
                idx.append(t.identifier)
    corrs, idx = np.array(corrs), np.array(idx)
    if corrs.shape[0] > 0 and img_a_shape is not None and img_b_shape is not None and not force:
        lim = np.concatenate([np.array(img_a_shape[:2])[::-1], np.array(img_b_shape[:2])[::-1]])
        keep = (corrs < lim).all(axis=1) & (corrs > 0).all(axis=1)
        corrs, idx = corrs[keep], idx[keep]
    return (corrs, idx) if return_idx else corrs


class ZoomEngine:
    """``ZoomEngine(model).refine(...)`` / ``.cotr_corr_multiscale(...)``.

    model       the object returned by ``cotr_amd.models.build_model`` on the GPU (any callable with the reference's
                ``model(img, queries) -> {'pred_corrs'}`` contract works: the tests drive the state machine with a
                deterministic stand-in).
    max_pairs   crops per model call (the library itself walks them 32 at a time through the backbone).
    make_cropper  factory (img_a, img_b, device) -> callable(boxes, out); default: the HIP kernel.
    make_dense_post  factory (device) -> callable(pred, pairs, shape_a, shape_b) -> corr_a, con_a, corr_b, con_b;
                default: the HIP kernels (cotr_dense_cycle / cotr_dense_merge).
    """

    def __init__(self, model, max_pairs=256, make_cropper=None, batch_size=32, mode='tile', make_dense_post=None):
        if mode not in ('stretching', 'tile'):                      # sparse_engine.py:19
            raise ValueError(f'unsupported mode: {mode}')
        self.mode = mode
        self.model = model
        self.batch_size = int(batch_size)   # the reference walks tasks in groups of this size: decides where it stops
        self.max_pairs = int(max_pairs)
        self.make_cropper = make_cropper or _DeviceCropper
        self.make_dense_post = make_dense_post or _DeviceDensePost
        self.total_tasks = 0       # same bookkeeping as SparseEngine.total_tasks: crops pushed through the model

    # ------------------------------------------------------------------------------------------------
    def _infer(self, cropper, boxes, queries, device, buf):
        """-> float32 [n,2] network answers for n (box, query) tasks, chunked by max_pairs."""
        n = boxes.shape[0]
        outs = []
        for lo in range(0, n, self.max_pairs):
            hi = min(n, lo + self.max_pairs)
            img = cropper(boxes[lo:hi], buf)
            q = torch.from_numpy(queries[lo:hi]).to(device)[:, None, :]
            pred = self.model(img, q)['pred_corrs']
            outs.append(pred.detach().cpu().numpy()[:, 0, :])
            self.total_tasks += hi - lo
        out = np.concatenate(outs, axis=0)
        if np.isnan(out).any():
            raise ValueError('NaN in prediction')          # sparse_engine.py:54-55
        return out

    def refine(self, img_a, img_b, loc_from, loc_to, area_from=1.0, area_to=1.0, zoom_ins=(1.0,), converge_iters=1,
               force=False):
        """Run every (loc_from -> loc_to) task through all zoom levels.  Returns RefineResult with
        loc_to = best_loc_to of each task and good = what ``conclude(force)`` would keep."""
        img_a = np.ascontiguousarray(img_a)
        img_b = np.ascontiguousarray(img_b)
        loc_from = np.array(loc_from, dtype=np.float64).reshape(-1, 2)
        cur = np.array(loc_to, dtype=np.float64).reshape(-1, 2)
        n = loc_from.shape[0]
        zoom_ins = [float(z) for z in zoom_ins]
        # RefinementTask.__init__ :25-30
        if area_from < area_to:
            s_from, s_to = BASE_ZOOM, BASE_ZOOM * np.sqrt(area_to / area_from)
        else:
            s_to, s_from = BASE_ZOOM, BASE_ZOOM * np.sqrt(area_from / area_to)
        device = next(self.model.parameters()).device
        cropper = self.make_cropper(img_a, img_b, device)
        buf = torch.empty((min(n, self.max_pairs), 3, 256, 512), dtype=torch.float32, device=device) if n else None
        history = [cur.copy()]
        last_iters = [[] for _ in range(n)]
        steps = np.zeros(n, dtype=np.int64)
        calls0, crops0 = 0, self.total_tasks
        for zi, zoom in enumerate(zoom_ins):
            last = zi == len(zoom_ins) - 1
            active = np.arange(n)
            at_zoom = [[] for _ in range(n)]          # loc_to_at_zoom per task (only the last level keeps > 1)
            it = 0
            while active.size:
                ax, ay, asz = patch_boxes(img_a.shape, loc_from[active], s_from * zoom)
                bx, by, bsz = patch_boxes(img_b.shape, cur[active], s_to * zoom)
                boxes = np.stack([ax, ay, np.full_like(ax, asz), bx, by, np.full_like(bx, bsz)], axis=1)
                # query in the crop's frame, refinement_task.py:110 (float64 math, then .float())
                q = ((loc_from[active] - np.stack([ax, ay], 1)) / np.array([asz * 2, asz])).astype(np.float32)
                raw = self._infer(cropper, boxes, q, device, buf)
                calls0 += 1
                steps[active] += 1
                # scale_to_loc :145-151
                raw = raw.copy()
                raw[:, 0] = (raw[:, 0] - np.float32(0.5)) * np.float32(2)
                loc = raw.astype(np.float64) * np.array([bsz, bsz]) + np.stack([bx, by], 1)
                cur[active] = loc
                done = np.ones(active.size, dtype=bool)
                if last:
                    for j, t in enumerate(active):
                        prev = at_zoom[t]
                        repeat = len(prev) >= 1 and any((p == loc[j]).all() for p in prev)
                        at_zoom[t].append(loc[j].copy())
                        last_iters[t].append(loc[j].copy())
                        done[j] = repeat or it >= converge_iters - 1
                else:
                    for j, t in enumerate(active):
                        at_zoom[t].append(loc[j].copy())
                for j, t in enumerate(active):
                    if done[j]:
                        arr = np.array(at_zoom[t])
                        final = loc[j]
                        if len(arr) >= 2 and (arr[:-1] == arr[-1]).all(axis=1).any():
                            start = np.where((arr[:-1] == arr[-1]).all(axis=1))[0][0]     # find_prediction_loop
                            final = arr[start:-1].mean(axis=0)
                        cur[t] = final
                active = active[~done]
                it += 1
            history.append(cur.copy())
        hist = np.stack(history, axis=0)                          # [levels+1, N, 2] == loc_history per task
        if force:
            good = np.ones(n, dtype=bool)
        else:                                                      # conclude :184-188
            good = hist.std(axis=0).max(axis=1) < THRESHOLD_PIXELS_RELATIVE * max(*img_b.shape)
        return RefineResult(loc_from, cur.copy(), good, hist, calls0, self.total_tasks - crops0, steps, last_iters)

    # ------------------------------------------------------------------------------------------------
    _grids = {}

    @classmethod
    def _query_grid(cls, device):
        """The dense pass's constant queries (inference_helper.py:116-120): (j / 512, i / 256) for the 256 x 512 pixels of the network
        frame, [1, 131072, 2] float32, built ON THE DEVICE once per device (every value is k / 2^n: exact in float32, the same bits as
        the reference's float64 grid cast with .float()); the per-call numpy meshgrid + 1 MB upload is gone."""
        key = str(device)
        g = cls._grids.get(key)
        if g is None:
            jj = torch.arange(MAX_SIZE * 2, dtype=torch.float32, device=device) / (MAX_SIZE * 2)
            ii = torch.arange(MAX_SIZE, dtype=torch.float32, device=device) / MAX_SIZE
            g = torch.stack([jj[None, :].expand(MAX_SIZE, -1), ii[:, None].expand(-1, MAX_SIZE * 2)], dim=-1).reshape(1, -1, 2).contiguous()
            cls._grids[key] = g
        return g

    @staticmethod
    def _square_patches(img):
        """``to_square_patches`` (inference_helper.py:41-58) as boxes: [(x, y, size)], one or two per image."""
        h, w = img.shape[:2]
        size = min(h, w)
        if max(h, w) == size:
            return [(0, 0, size)]
        if max(h, w) <= size * 2:
            return [(0, 0, size), (w - size, h - size, size)]
        raise NotImplementedError('aspect ratio above 2 (the reference raises here as well)')

    def corr_base(self, img_a, img_b, queries_a):
        """``cotr_corr_base`` (inference_helper.py:185-232): for every pair of square patches of the two images one
        forward for the queries and one for the answers (cycle check), the patch pair with the smallest cycle error
        wins per query.  -> [N,4] (x_a, y_a, x_b, y_b).  The two forwards share ONE encode (``model.encode`` +
        two ``model.decode``) when the model offers the split, and all patch pairs are cropped in one launch."""
        img_a = np.ascontiguousarray(img_a)
        img_b = np.ascontiguousarray(img_b)
        queries_a = np.asarray(queries_a, dtype=np.float64)
        pa, pb = self._square_patches(img_a), self._square_patches(img_b)
        pairs = [(i, j) for i in pa for j in pb]
        boxes = np.array([[i[0], i[1], i[2], j[0], j[1], j[2]] for i, j in pairs], dtype=np.int32)
        device = next(self.model.parameters()).device
        cropper = self.make_cropper(img_a, img_b, device)
        buf = torch.empty((len(pairs), 3, 256, 512), dtype=torch.float32, device=device)
        img = cropper(boxes, buf)
        qn = np.empty((len(pairs),) + queries_a.shape, dtype=np.float32)
        masks = []
        for k, (i, _) in enumerate(pairs):
            q = queries_a.copy()
            masks.append((q[:, 0] >= i[0]) & (q[:, 1] >= i[1]) & (q[:, 0] <= i[0] + i[2]) & (q[:, 1] <= i[1] + i[2]))
            q[:, 0] = (q[:, 0] - i[0]) / (2 * i[2])
            q[:, 1] = (q[:, 1] - i[1]) / i[2]
            qn[k] = q.astype(np.float32)
        qt = torch.from_numpy(qn).to(device)
        if hasattr(self.model, 'encode') and hasattr(self.model, 'decode'):
            self.model.encode(img)                      # backbone + encoder + K/V once, reused by both passes
            out = self.model.decode(qt)
            cyc = self.model.decode(out)
        else:
            out = self.model(img, qt)['pred_corrs']
            cyc = self.model(img, out)['pred_corrs']
        self.total_tasks += len(pairs)
        out, cyc = out.detach().cpu().numpy(), cyc.detach().cpu().numpy()
        preds = []
        for k, (_, j) in enumerate(pairs):
            conf = np.linalg.norm(qn[k] - cyc[k], axis=1, keepdims=True)
            pred = np.concatenate([out[k], conf], axis=1)          # float32, like the reference's one_pass
            pred[~masks[k], 2] = np.inf
            pred[:, 0] -= 0.5
            pred[:, 0] *= 2 * j[2]
            pred[:, 0] += j[0]
            pred[:, 1] *= j[2]
            pred[:, 1] += j[1]
            preds.append(pred)
        preds = np.stack(preds).transpose(1, 0, 2)                  # [N, pairs, 3]
        best = np.array([item[np.argmin(item[..., 2], axis=0)] for item in preds])[..., :2]
        return np.concatenate([queries_a, best], axis=1)

    # ------------------------------------------------------------------------------------------------
    def flow(self, img_a, img_b, resample=True):
        """``cotr_flow`` (inference_helper.py:168-182): the dense initial pass.  Every pair of square patches of the two
        images gets ONE forward with the 256x512 grid of queries (131072, both halves; :116-127); the self-composition
        of the answer gives a per-pixel cycle error (:137-145); the maps are moved to image coordinates, resized to the
        patch and merged by lowest cycle error.  All patch pairs are cropped in one launch, go through the model as
        one batch, and the post-processing stays on the device (2 + 1 launches; only the merged maps come back).
        -> corr_a, con_a, resample_a, corr_b, con_b, resample_b like the reference (resample_* = the other image
        warped by the flow, a visualisation aid: None with resample=False)."""
        img_a = np.ascontiguousarray(img_a)
        img_b = np.ascontiguousarray(img_b)
        pa, pb = self._square_patches(img_a), self._square_patches(img_b)
        pairs = [(i, j) for i in pa for j in pb]
        boxes = np.array([[i[0], i[1], i[2], j[0], j[1], j[2]] for i, j in pairs], dtype=np.int32)
        device = next(self.model.parameters()).device
        cropper = self.make_cropper(img_a, img_b, device)
        buf = torch.empty((len(pairs), 3, 256, 512), dtype=torch.float32, device=device)
        img = cropper(boxes, buf)
        q = self._query_grid(device).expand(len(pairs), -1, -1).contiguous()
        # (cotr_amd.dist.sharded_zoom_engine routes this one call through a PairShardedModel: pairs / queries over the ranks)
        pred = getattr(self, '_dense_model', self.model)(img, q)['pred_corrs']
        self.total_tasks += len(pairs)
        post = self.make_dense_post(device)
        res_a = res_b = None
        if hasattr(post, 'device_maps') and hasattr(cropper, 'a'):
            # device post-processing: the merged maps stay on the device until here; the visualisation warps (:178-181) read them and
            # the cropper's device copies of the images in place; everything the caller gets comes back in one pinned copy
            maps = post.device_maps(pred, pairs, img_a.shape, img_b.shape)
            outs = list(maps)
            if resample:
                def warp_d(img_dev, corr_dev):
                    t = img_dev.permute(2, 0, 1)[None].float()
                    return torch.nn.functional.grid_sample(t, corr_dev[None], align_corners=False)[0].permute(1, 2, 0)
                outs += [warp_d(cropper.b, maps[0]), warp_d(cropper.a, maps[2])]
            host = post.to_host(outs)
            corr_a, con_a, corr_b, con_b = host[:4]
            if resample:
                res_a, res_b = (np.ascontiguousarray(r, dtype=np.float32) for r in host[4:])
            return corr_a, con_a, res_a, corr_b, con_b, res_b
        corr_a, con_a, corr_b, con_b = post(pred, pairs, img_a.shape, img_b.shape)
        if resample:                                                                    # :178-181
            def warp(img_src, corr):
                t = torch.from_numpy(np.transpose(img_src, (2, 0, 1)))[None].float().to(device)
                g = torch.from_numpy(corr)[None].float().to(device)
                r = torch.nn.functional.grid_sample(t, g, align_corners=False)[0]
                return np.transpose(r.cpu().numpy(), (1, 2, 0))
            res_a, res_b = warp(img_b, corr_a), warp(img_a, corr_b)
        return corr_a, con_a, res_a, corr_b, con_b, res_b

    def gen_tasks(self, img_a, img_b, max_corrs, queries_a, force):
        """``SparseEngine.gen_tasks`` without ``areas`` (sparse_engine.py:108-195): dense pass, confident
        pixels, relative scale from the confident areas.  -> loc_from [N,2], loc_to [N,2], identifier [N] (-1 = None),
        area_a, area_b.  Uses numpy's global RNG exactly where the reference does (np.random.choice, :151,154)."""
        if self.mode == 'stretching' and (img_a.shape[0] != img_a.shape[1] or img_b.shape[0] != img_b.shape[1]):
            # sparse_engine.py:114-129: dense pass on the images stretched to squares (Pillow 8-bit bilinear, once per
            # pair, on the host like the reference), maps brought back to the image shape by the float resize
            import PIL.Image

            def stretch(img):
                size = max(*img.shape[:2])
                return np.array(PIL.Image.fromarray(img).resize((size, size), resample=PIL.Image.BILINEAR))
            corr_a, con_a, _, corr_b, con_b, _ = self.flow(stretch(img_a), stretch(img_b), resample=False)
            device = next(self.model.parameters()).device
            post = self.make_dense_post(device)
            corr_a, con_a = post.resize(corr_a, img_a.shape[:2]), post.resize(con_a, img_a.shape[:2])
            corr_b, con_b = post.resize(corr_b, img_b.shape[:2]), post.resize(con_b, img_b.shape[:2])
        else:
            corr_a, con_a, _, corr_b, con_b, _ = self.flow(img_a, img_b, resample=False)
        mask_a, mask_b = con_a < THRESHOLD_SPARSE, con_b < THRESHOLD_SPARSE
        area_a = (con_a < THRESHOLD_AREA).sum() / mask_a.size
        area_b = (con_b < THRESHOLD_AREA).sum() / mask_b.size
        size_a, size_b = np.array(img_a.shape[:2][::-1]), np.array(img_b.shape[:2][::-1])
        loc_from, loc_to, ident = [], [], []
        if queries_a is None:
            index_a = np.array(np.where(mask_a)).T
            index_a = index_a[np.random.choice(len(index_a), min(max_corrs, len(index_a)))]
            index_b = np.array(np.where(mask_b)).T
            index_b = index_b[np.random.choice(len(index_b), min(max_corrs, len(index_b)))]
            for pos in index_a:
                loc_from.append(pos[::-1].astype(np.float64))
                loc_to.append((corr_a[tuple(np.floor(pos).astype('int'))].copy() * 0.5 + 0.5) * size_b)
                ident.append(-1)
            for pos in index_b:   # "trick": the first guess is fixed instead of the query (sparse_engine.py:160-166)
                loc_from.append((corr_b[tuple(np.floor(pos).astype('int'))].copy() * 0.5 + 0.5) * size_a)
                loc_to.append(pos[::-1].astype(np.float64))
                ident.append(-1)
        elif force:
            for i, lf in enumerate(queries_a):
                pos = lf[::-1]
                pos = np.array([np.clip(pos[0], 0, corr_a.shape[0] - 1), np.clip(pos[1], 0, corr_a.shape[1] - 1)], dtype=int)
                loc_from.append(np.asarray(lf, dtype=np.float64))
                loc_to.append((corr_a[tuple(pos)].copy() * 0.5 + 0.5) * size_b)
                ident.append(i)
        else:
            def inside(pos):
                return not ((pos > np.array(img_a.shape[:2]) - 1).any() or (pos < 0).any())
            for i, lf in enumerate(queries_a):
                pos = lf[::-1]
                if inside(pos) and mask_a[tuple(np.floor(pos).astype('int'))]:
                    loc_from.append(np.asarray(lf, dtype=np.float64))
                    loc_to.append((corr_a[tuple(np.floor(pos).astype('int'))].copy() * 0.5 + 0.5) * size_b)
                    ident.append(i)
            if len(loc_from) < max_corrs:
                extra, counter = max_corrs - len(loc_from), 0
                for i, lf in enumerate(queries_a):
                    if counter >= extra:
                        break
                    pos = lf[::-1]
                    if inside(pos) and not mask_a[tuple(np.floor(pos).astype('int'))]:
                        loc_from.append(np.asarray(lf, dtype=np.float64))
                        loc_to.append((corr_a[tuple(np.floor(pos).astype('int'))].copy() * 0.5 + 0.5) * size_b)
                        ident.append(i)
                        counter += 1
        n = len(loc_from)
        return (np.array(loc_from, dtype=np.float64).reshape(n, 2), np.array(loc_to, dtype=np.float64).reshape(n, 2),
                np.array(ident, dtype=np.int64), area_a, area_b)

    def cotr_corr_multiscale(self, img_a, img_b, zoom_ins=(1.0,), converge_iters=1, max_corrs=1000, queries_a=None,
                             return_idx=False, force=False, return_tasks_only=False, areas=None, init_b=None):
        """``SparseEngine.cotr_corr_multiscale`` (sparse_engine.py:197-233), same arguments and result ([M,4] rows
        (x_a, y_a, x_b, y_b), at most max_corrs, in task order; with return_idx also the task identifiers).
        ``init_b`` (extra): initial estimates for the known-scale path instead of running ``corr_base``.
        ``return_tasks_only``: a list of ``ZoomTask`` (the reference returns its RefinementTask objects)."""
        img_a, img_b = np.ascontiguousarray(img_a), np.ascontiguousarray(img_b)
        if areas is not None:                                      # gen_tasks / gen_tasks_w_known_scale :100-114
            assert queries_a is not None
            assert force == True                                   # noqa: E712  (sparse_engine.py:110)
            assert max_corrs >= len(queries_a)
            if init_b is None:
                base = self.corr_base(img_a, img_b, queries_a)
                loc_from, loc_to = base[:, :2], base[:, 2:]
            else:
                loc_from, loc_to = np.asarray(queries_a, dtype=np.float64), np.asarray(init_b, dtype=np.float64)
            ident = np.full(len(loc_from), -1, dtype=np.int64)     # the reference leaves identifier = None here
            area_a, area_b = areas
        else:
            loc_from, loc_to, ident, area_a, area_b = self.gen_tasks(img_a, img_b, max_corrs, queries_a, force)
        n = len(loc_from)
        # The reference refines tasks in groups of batch_size in list order and stops as soon as max_corrs of them are
        # "good" (:208-218).  Same outcome here, but a chunk of max_pairs tasks at a time and one launch per level.
        chunk = max(self.batch_size, (self.max_pairs // self.batch_size) * self.batch_size)
        final = np.zeros((n, 2))
        good = np.zeros(n, dtype=bool)
        steps = np.zeros(n, dtype=np.int64)
        results = []
        done = 0
        while True:
            executed = _reference_schedule_steps(steps[:done], good[:done], self.batch_size, max_corrs, n)
            if executed is not None:
                break
            hi = min(n, done + chunk)
            res = self.refine(img_a, img_b, loc_from[done:hi], loc_to[done:hi], area_a, area_b, zoom_ins, converge_iters, False)
            final[done:hi], good[done:hi], steps[done:hi] = res.loc_to, res.good, res.steps
            results.append((done, res))
            done = hi
        finished = executed == steps[:done]
        if return_tasks_only:                                          # :218-219: the task objects, finished or not
            return self._task_objects(img_a.shape, img_b.shape, loc_from, loc_to, ident, area_a, area_b, zoom_ins,
                                      converge_iters, results, executed, areas is None)
        keep = np.zeros(n, dtype=bool)
        keep[:done] = finished if force else (finished & good[:done])  # status == 'finished' and conclude(force) :67-72
        corrs = np.concatenate([loc_from, final], axis=1)
        if not force:                                                  # conclude_tasks border mask :75-80
            lim = np.concatenate([np.array(img_a.shape[:2])[::-1], np.array(img_b.shape[:2])[::-1]])
            keep &= (corrs < lim).all(axis=1) & (corrs > 0).all(axis=1)
        corrs, idx = corrs[keep][:max_corrs], ident[keep][:max_corrs]
        if (idx < 0).any():     # tasks made without an identifier carry None in the reference (:63-69 -> array of None)
            idx = np.array([None if i < 0 else int(i) for i in idx], dtype=object)
        return (corrs, idx) if return_idx else corrs

    @staticmethod
    def _task_objects(shape_a, shape_b, loc_from, loc_to, ident, area_a, area_b, zoom_ins, converge_iters, results, executed,
                      with_ident):
        """``return_tasks_only``: the per-task state the reference's loop leaves behind, rebuilt from the array state
        (level results of ``refine`` + the number of steps the reference's schedule gave each task before it stopped)."""
        levels = len(zoom_ins)
        tasks = [ZoomTask(shape_a, shape_b, loc_from[i], loc_to[i], area_a, area_b, converge_iters, zoom_ins,
                          int(ident[i]) if with_ident and ident[i] >= 0 else None) for i in range(len(loc_from))]
        for lo, res in results:
            for j in range(len(res.steps)):
                t, k, total = tasks[lo + j], int(executed[lo + j]), int(res.steps[j])
                done_levels = levels if k == total else min(k, levels - 1)
                t.loc_history = [t.loc_history[0]] + [res.loc_history[lv + 1, j].copy() for lv in range(done_levels)]
                t.best_loc_to = t.cur_loc_to = t.loc_history[-1]
                t.total_iter, t.cur_zoom_idx = k, done_levels
                if k == total:
                    t.status = 'finished'
                    t.result = 'bad' if t.conclude() is None else 'good'
                elif k > levels - 1 and res.last_iters is not None:     # stopped between iterations of the last level
                    t.loc_to_at_zoom = [p.copy() for p in res.last_iters[j][:k - (levels - 1)]]
                    t.cur_loc_to, t.cur_iter = t.loc_to_at_zoom[-1], k - (levels - 1)
        return tasks

    def cotr_corr_multiscale_with_cycle_consistency(self, img_a, img_b, zoom_ins=(1.0,), converge_iters=1, max_corrs=1000,
                                                    queries_a=None, return_idx=False, return_cycle_error=False):
        """``SparseEngine.cotr_corr_multiscale_with_cycle_consistency`` (sparse_engine.py:235-264): a -> b with
        max_corrs/0.3 candidates, b -> a seeded with the answers, keep the max_corrs with the smallest cycle error."""
        temp_max_corrs = int(max_corrs / 0.3)
        if queries_a is not None:
            temp_max_corrs = min(temp_max_corrs, queries_a.shape[0])
            queries_a = queries_a.copy()
        corr_f, idx_f = self.cotr_corr_multiscale(img_a.copy(), img_b.copy(), zoom_ins, converge_iters, temp_max_corrs,
                                                  queries_a, return_idx=True)
        assert corr_f.shape[0] > 0
        corr_b, idx_b = self.cotr_corr_multiscale(img_b.copy(), img_a.copy(), zoom_ins, converge_iters, corr_f.shape[0],
                                                  corr_f[:, 2:].copy(), return_idx=True)
        assert corr_b.shape[0] > 0
        cycle_errors = np.linalg.norm(corr_f[idx_b][:, :2] - corr_b[:, 2:], axis=1)
        order = np.argsort(cycle_errors)
        out = [corr_f[idx_b][order][:max_corrs]]
        if return_idx:
            out.append(idx_f[idx_b][order][:max_corrs])
        if return_cycle_error:
            out.append(cycle_errors[order][:max_corrs])
        return out[0] if len(out) == 1 else out


class SparseEngine(ZoomEngine):
    """Same constructor as ``COTR.inference.sparse_engine.SparseEngine(model, batch_size, mode='stretching')`` (:18-21), so
    the reference's demos switch engines by changing one import; ``batch_size`` only decides where the early exit falls
    (the reference's group size), the launches themselves are one per zoom level."""

    def __init__(self, model, batch_size, mode='stretching'):
        super().__init__(model, batch_size=batch_size, mode=mode)


class FasterSparseEngine(SparseEngine):
    """``COTR.inference.sparse_engine.FasterSparseEngine`` (:267-427): "search and merge nearby tasks to accelerate
    inference speed.  It will make spatial accuracy slightly worse."  A *pilot* task cuts its crop pair; every other
    task of the same zoom level whose query AND current estimate fall into the central half of the pilot's two crops
    (``form_squad`` :295-337, at most ``max_load`` of them) is answered from the pilot's crops in the same forward, so a
    model call is img[<=batch_size, 3, 256, 512] x q[<=batch_size, <=max_load+1, 2] (zero-padded queries, :363-366).

    Grouping couples the tasks (who rides with whom depends on ``np.random.permutation`` and on every earlier answer), so
    unlike ``ZoomEngine`` this engine keeps the reference's control flow call for call - same pilots in the same order,
    same squads, same RNG draws (:339-427) - and therefore the same correspondences.  What is MI355X-specific: all crop
    pairs of a model call are cut, resized (Pillow-exact), laid side by side and normalised by ONE launch of the device
    crop kernel instead of 2 PIL resizes + an H2D copy per pilot, and the forward is the HIP library's batched
    encode + decode (queries of a squad share their pilot's encode)."""

    def __init__(self, model, batch_size, mode='stretching', max_load=256):
        super().__init__(model, batch_size, mode)
        self.max_load = int(max_load)

    # -- task list --------------------------------------------------------------------------------------------
    def _make_tasks(self, img_a, img_b, zoom_ins, converge_iters, max_corrs, queries_a, force, areas):
        """``gen_tasks`` (:108-195) -> [ZoomTask]."""
        if areas is not None:
            assert queries_a is not None
            assert force == True                                    # noqa: E712  (:110)
            assert max_corrs >= queries_a.shape[0]
            base = self.corr_base(img_a, img_b, queries_a)
            loc_from, loc_to, ident = base[:, :2], base[:, 2:], None
            area_a, area_b = areas
        else:
            loc_from, loc_to, ident, area_a, area_b = self.gen_tasks(img_a, img_b, max_corrs, queries_a, force)
        return [ZoomTask(img_a.shape, img_b.shape, loc_from[i], loc_to[i], area_a, area_b, converge_iters, zoom_ins,
                         None if ident is None or ident[i] < 0 else int(ident[i])) for i in range(len(loc_from))]

    def _form_grouped_batch(self, zoom, tasks):
        """``form_grouped_batch`` (:339-369) + ``form_squad`` (:295-337) -> ([members], boxes [P,6], queries [P,Qmax,2])."""
        cand = [i for i, t in enumerate(tasks) if t.status == 'unfinished' and not t.submitted and t.cur_zoom == zoom]
        tasks_map = np.array([np.concatenate([tasks[i].loc_from, tasks[i].cur_loc_to]) for i in cand]).reshape(-1, 4)
        task_ids = np.array(cand, dtype=np.int64)
        shuffle = np.random.permutation(tasks_map.shape[0])          # the reference's draw from numpy's global RNG
        tasks_map, task_ids = np.take(tasks_map, shuffle, axis=0), np.take(task_ids, shuffle, axis=0)
        free = np.ones(len(task_ids), dtype=bool)
        squads, boxes, queries = [], [], []
        for i, ti in enumerate(task_ids):
            pilot = tasks[ti]
            if not (pilot.status == 'unfinished' and not pilot.submitted and pilot.cur_zoom == zoom):
                continue                                             # already riding with an earlier pilot
            xa, ya, sa, xb, yb, sb = pilot.peek()
            safe = 0.5                                               # SAFE_AREA: the central half of both crops
            ca = (xa + sa / 2, ya + sa / 2)
            cb = (xb + sb / 2, yb + sb / 2)
            job, q = pilot.submit()
            members, qs = [pilot], [q]
            free[i] = False
            inside = ((tasks_map[:, 0] > ca[0] - sa / 2 * safe) & (tasks_map[:, 0] < ca[0] + sa / 2 * safe) &
                      (tasks_map[:, 1] > ca[1] - sa / 2 * safe) & (tasks_map[:, 1] < ca[1] + sa / 2 * safe) &
                      (tasks_map[:, 2] > cb[0] - sb / 2 * safe) & (tasks_map[:, 2] < cb[0] + sb / 2 * safe) &
                      (tasks_map[:, 3] > cb[1] - sb / 2 * safe) & (tasks_map[:, 3] < cb[1] + sb / 2 * safe))
            loads = np.where(inside * free)[0][: self.max_load]
            for tj in task_ids[loads]:
                _, q = tasks[tj].submit(job)                         # get_task_pilot: the pilot's crops, own query
                members.append(tasks[tj])
                qs.append(q)
            free[loads] = False
            squads.append(members)
            boxes.append(job)
            queries.append(np.stack(qs))
            if len(squads) >= self.batch_size:
                break
        if not squads:
            return [], None, None
        qmax = max(len(q) for q in queries)
        qpad = np.zeros((len(squads), qmax, 2), dtype=np.float32)    # zero-padded like torch.zeros (:363-366)
        for k, q in enumerate(queries):
            qpad[k, :len(q)] = q
        return squads, np.array(boxes, dtype=np.int32), qpad

    def _form_batch(self, tasks, zoom):
        """``form_batch`` (:23-45): the first ``batch_size`` open tasks of this zoom level, each with its own crops."""
        ref, boxes, queries = [], [], []
        for t in tasks:
            if t.status == 'unfinished' and not t.submitted and (zoom is None or t.cur_zoom == zoom):
                job, q = t.submit()
                ref.append(t)
                boxes.append(job)
                queries.append(q[None])
                if len(ref) >= self.batch_size:
                    break
        if not ref:
            return [], None, None
        return ref, np.array(boxes, dtype=np.int32), np.stack(queries)

    def _forward(self, cropper, boxes, queries, device, count):
        """One model call on the crop pairs ``boxes`` [P,6] with ``queries`` [P,Q,2] -> float32 [P,Q,2]."""
        buf = torch.empty((len(boxes), 3, 256, 512), dtype=torch.float32, device=device)
        img = cropper(boxes, buf)
        pred = self.model(img, torch.from_numpy(queries).to(device))['pred_corrs']
        if count:
            self.total_tasks += len(boxes)                           # only infer_batch counts (:48), infer_batch_grouped not
        return pred.detach().cpu().numpy()

    def cotr_corr_multiscale(self, img_a, img_b, zoom_ins=(1.0,), converge_iters=1, max_corrs=1000, queries_a=None,
                             return_idx=False, force=False, return_tasks_only=False, areas=None):
        """``FasterSparseEngine.cotr_corr_multiscale`` (:371-427)."""
        img_a, img_b = np.ascontiguousarray(img_a), np.ascontiguousarray(img_b)
        if queries_a is not None:
            queries_a = np.array(queries_a, dtype=np.float64)
        zoom_ins = list(zoom_ins)
        tasks = self._make_tasks(img_a, img_b, zoom_ins, converge_iters, max_corrs, queries_a, force, areas)
        device = next(self.model.parameters()).device
        cropper = self.make_cropper(img_a, img_b, device)
        num_good = lambda: sum(t.result == 'good' for t in tasks)    # noqa: E731
        zm = None
        for zm in zoom_ins:
            while True:
                num_g = num_good()
                squads, boxes, queries = self._form_grouped_batch(zm, tasks)
                if not squads or num_g >= max_corrs:                 # (a batch formed before the exit test stays submitted,
                    break                                            #  exactly as in the reference)
                out = self._forward(cropper, boxes, queries, device, count=False)
                num_steps = 0
                for i, members in enumerate(squads):
                    for j, t in enumerate(members):
                        t.step(out[i, j])
                        num_steps += 1
                if num_steps <= self.batch_size:                     # too few tasks group together: next level
                    break
        while True:                                                  # "rollback to default inference" (:400-411), last level only
            num_g = num_good()
            ref, boxes, queries = self._form_batch(tasks, zm)
            if not ref or num_g >= max_corrs:
                break
            out = self._forward(cropper, boxes, queries, device, count=True)[:, 0, :]
            if np.isnan(out).any():
                raise ValueError('NaN in prediction')                # infer_batch (:54-55)
            for t, o in zip(ref, out):
                t.step(o)
        if return_tasks_only:
            return tasks
        corrs, idx = conclude_tasks(tasks, True, force, img_a.shape[:2], img_b.shape[:2])
        corrs, idx = corrs[:max_corrs], idx[:max_corrs]
        return (corrs, idx) if return_idx else corrs
