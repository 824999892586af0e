"""CPU restatement of the reference's dense-pass post-processing, literally as the reference does it on the host
(torch CPU ``grid_sample``, numpy, Pillow mode-'F' resize):

* ``cycle_maps``        <- ``cotr_patch_flow_exhaustive.one_pass`` tail, COTR/inference/inference_helper.py:137-145
* ``to_image_frame``    <- :150-160 (3-point affine of the patch corners, ``utils.float_image_resize``)
* ``merge_flow_patches``<- :61-75
* ``dense_post``        the three chained for a list of patch pairs, what ``cotr_flow`` (:168-176) returns as
                         (corr_a, con_a, corr_b, con_b)

Pinned: tests/golden/engine_dense_*.npz hold the output of the reference's own ``cotr_flow`` (run unchanged in the
authoring container by tests/golden/make_engine_golden.py); tests/test_zoom_engine_cpu.py checks this restatement
against them bit for bit (sha256 of the whole maps).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the product (cotr_amd/inference/zoom_engine.py) does this on the
device with cotr_dense_cycle / cotr_dense_merge and never imports this module.
"""
import numpy as np
import torch

MAX_SIZE = 256


def affine_from_3pts(src, dst):
    """What ``cv2.getAffineTransform`` solves: the 2x3 map through three point pairs, float64."""
    a = np.concatenate([np.asarray(src, dtype=np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, np.asarray(dst, dtype=np.float64)).T


def float_image_resize(img, shape):
    """COTR/utils/utils.py:69-83: per-channel Pillow mode-'F' bilinear resize."""
    import PIL.Image
    layers = [np.array(PIL.Image.fromarray(l).resize(shape[::-1], resample=PIL.Image.BILINEAR)) for l in img.transpose(2, 0, 1)]
    return np.stack(layers, axis=-1)


def cycle_maps(out_list):
    """out_list float32 [256,512,2] (answer for the grid (j/512, i/256)) -> (left [256,256,3], right [256,256,3]):
    x re-centred per half, third channel = cycle error.  inference_helper.py:137-145."""
    jj, ii = np.meshgrid(np.arange(MAX_SIZE * 2), np.arange(MAX_SIZE))
    q_grid = np.stack([jj / (MAX_SIZE * 2), ii / MAX_SIZE], axis=-1)
    in_grid = torch.from_numpy(q_grid).float()[None] * 2 - 1
    out_grid = torch.from_numpy(np.ascontiguousarray(out_list)).float()[None] * 2 - 1
    cycle_grid = torch.nn.functional.grid_sample(out_grid.permute(0, 3, 1, 2), out_grid, align_corners=False).permute(0, 2, 3, 1)
    confidence = torch.norm(cycle_grid[0, ...] - in_grid[0, ...], dim=-1)
    corr = out_grid[0].clone()
    corr[:, :MAX_SIZE, 0] = corr[:, :MAX_SIZE, 0] * 2 - 1
    corr[:, MAX_SIZE:, 0] = corr[:, MAX_SIZE:, 0] * 2 + 1
    corr = torch.cat([corr, confidence[..., None]], dim=-1).numpy()
    return corr[:, :MAX_SIZE, :], corr[:, MAX_SIZE:, :]


def patch_affines(p_i, p_j, shape_a, shape_b):
    """(T_i, T_j) of inference_helper.py:151-156 for patches (x, y, size) of images a and b."""
    base = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]])

    def corners(p, shape):
        x, y, s = p
        return (np.array([[x, y], [x + s, y], [x + s, y + s], [x, y + s]]) / np.array([shape[1], shape[0]])) * 2 + np.array([-1, -1])
    real_j, real_i = corners(p_j, shape_b), corners(p_i, shape_a)
    t_i = affine_from_3pts(base[:3].astype(np.float32), real_j[:3].astype(np.float32))
    t_j = affine_from_3pts(base[:3].astype(np.float32), real_i[:3].astype(np.float32))
    return t_i, t_j


def merge_flow_patches(corrs):
    """inference_helper.py:61-75.  corrs: list of (patch [h,w,3], x, y, w, h, ow, oh)."""
    oh, ow = corrs[0][6], corrs[0][5]
    confidence = np.ones([oh, ow]) * 100
    flow = np.zeros([oh, ow, 2])
    cmap = np.ones([oh, ow]) * -1
    for i, (patch, x, y, w, h, _ow, _oh) in enumerate(corrs):
        temp = np.ones([oh, ow]) * 100
        temp[y:y + h, x:x + w] = patch[..., 2]
        tempf = np.zeros([oh, ow, 2])
        tempf[y:y + h, x:x + w] = patch[..., :2]
        min_ind = np.stack([temp, confidence], axis=-1).argmin(axis=-1) == 0
        confidence[min_ind] = temp[min_ind]
        flow[min_ind] = tempf[min_ind]
        cmap[min_ind] = i
    return flow, confidence, cmap


def dense_post(pred, pairs, shape_a, shape_b):
    """pred [P,256,512,2] (tensor or array), pairs = [((xa,ya,sa), (xb,yb,sb))] -> corr_a, con_a, corr_b, con_b
    (float64, like ``cotr_flow``)."""
    pred = pred.detach().cpu().numpy() if isinstance(pred, torch.Tensor) else np.asarray(pred)
    (ha, wa), (hb, wb) = shape_a[:2], shape_b[:2]
    corrs_a, corrs_b = [], []
    for k, (p_i, p_j) in enumerate(pairs):
        c_i, c_j = cycle_maps(pred[k].reshape(MAX_SIZE, MAX_SIZE * 2, -1))
        t_i, t_j = patch_affines(p_i, p_j, shape_a, shape_b)
        c_i[..., :2] = c_i[..., :2] @ t_i[:2, :2] + t_i[:, 2]
        c_j[..., :2] = c_j[..., :2] @ t_j[:2, :2] + t_j[:, 2]
        corrs_a.append((float_image_resize(c_i, (p_i[2], p_i[2])), p_i[0], p_i[1], p_i[2], p_i[2], wa, ha))
        corrs_b.append((float_image_resize(c_j, (p_j[2], p_j[2])), p_j[0], p_j[1], p_j[2], p_j[2], wb, hb))
    corr_a, con_a, _ = merge_flow_patches(corrs_a)
    corr_b, con_b, _ = merge_flow_patches(corrs_b)
    return corr_a, con_a, corr_b, con_b


class _HostDensePost:
    def __call__(self, pred, pairs, shape_a, shape_b):
        return dense_post(pred, pairs, shape_a, shape_b)

    def resize(self, arr, shape):
        """``utils.float_image_resize`` incl. its 2-D special case (utils.py:69-83); float64 input goes through Pillow's
        'F;64F' conversion to float32, as in the reference."""
        arr = np.asarray(arr)
        if arr.ndim == 2:
            return float_image_resize(arr[..., None], shape)[..., 0]
        return float_image_resize(arr, shape)


def host_dense_post_factory(device):
    """Drop-in for ZoomEngine(make_dense_post=...) in CPU tests of the engine logic."""
    return _HostDensePost()
