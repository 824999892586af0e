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

RefineResult = namedtuple('RefineResult', ['loc_from', 'loc_to', 'good', 'loc_history', 'model_calls', 'crops'])


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


class ZoomEngine:
    """``ZoomEngine(model).refine(...)`` / ``.cotr_corr_multiscale(...)``.

    model       the object returned by ``cotr_amd.models.build_model`` on the GPU (any callable with the reference's
                ``model(img, queries) -> {'pred_corrs'}`` contract works: the tests drive the state machine with a
                deterministic stand-in).
    max_pairs   crops per model call (the library itself walks them 32 at a time through the backbone).
    make_cropper  factory (img_a, img_b, device) -> callable(boxes, out); default: the HIP kernel.
    """

    def __init__(self, model, max_pairs=256, make_cropper=None):
        self.model = model
        self.max_pairs = int(max_pairs)
        self.make_cropper = make_cropper or _DeviceCropper
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
        return RefineResult(loc_from, cur.copy(), good, hist, calls0, self.total_tasks - crops0)

    # ------------------------------------------------------------------------------------------------
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
    def cotr_corr_multiscale(self, img_a, img_b, zoom_ins=(1.0,), converge_iters=1, max_corrs=1000, queries_a=None,
                             return_idx=False, force=False, areas=None, init_b=None):
        """``SparseEngine.cotr_corr_multiscale`` for tasks with known scale: ``queries_a`` [N,2] pixel positions in
        img_a, ``areas`` = (area_a, area_b) as the reference requires for this path (:108-114), initial estimates
        ``init_b`` [N,2] in img_b (default: ``corr_base``, as the reference does).  Returns [M,4]
        (x_a, y_a, x_b, y_b), at most max_corrs rows, in task order."""
        if queries_a is None or areas is None:
            raise NotImplementedError('ZoomEngine batches the refinement of tasks with known scale (queries_a + areas, '
                                      'sparse_engine.py:100-114); the dense initial pass (cotr_flow) is not part of it yet')
        if init_b is None:                                         # gen_tasks_w_known_scale :100-106
            base = self.corr_base(img_a, img_b, queries_a)
            queries_a, init_b = base[:, :2], base[:, 2:]
        res = self.refine(img_a, img_b, queries_a, init_b, areas[0], areas[1], zoom_ins, converge_iters, force)
        corrs = np.concatenate([res.loc_from, res.loc_to], axis=1)
        idx = np.arange(corrs.shape[0])
        keep = res.good.copy()
        if not force:                                              # conclude_tasks border mask :75-80
            lim = np.concatenate([np.array(img_a.shape[:2])[::-1], np.array(img_b.shape[:2])[::-1]])
            keep &= (corrs < lim).all(axis=1) & (corrs > 0).all(axis=1)
        corrs, idx = corrs[keep][:max_corrs], idx[keep][:max_corrs]
        return (corrs, idx) if return_idx else corrs
