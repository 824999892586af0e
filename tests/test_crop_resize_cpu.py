"""The Pillow-bilinear restatement (oracle/pil_resize.py) against the installed Pillow, bit for bit. CPU."""
import numpy as np
import PIL.Image
import pytest

from oracle.pil_resize import resize_bilinear_u8, crop_resize_normalize


@pytest.mark.parametrize('size', [2, 6, 62, 64, 100, 254, 256, 258, 300, 512, 514, 700])
def test_resize_matches_pillow_bit_for_bit(size):
    rng = np.random.default_rng(size)
    img = rng.integers(0, 256, (size, size, 3), dtype=np.uint8)
    ref = np.array(PIL.Image.fromarray(img).resize((256, 256), resample=PIL.Image.BILINEAR))
    assert np.array_equal(resize_bilinear_u8(img), ref)


def test_task_input_matches_reference_recipe():
    """refinement_task.py:105-120: crop, resize x2, side by side, to_tensor, normalize."""
    import torch
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (350, 330, 3), dtype=np.uint8)
    out = crop_resize_normalize(a, b, (17, 40, 150), (100, 7, 222))
    halves = [np.array(PIL.Image.fromarray(im[y:y + s, x:x + s]).resize((256, 256), resample=PIL.Image.BILINEAR))
              for im, (x, y, s) in ((a, (17, 40, 150)), (b, (100, 7, 222)))]
    canvas = np.concatenate(halves, axis=1)
    t = torch.from_numpy(canvas.transpose(2, 0, 1).copy()).float().div(255)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    ref = ((t - mean) / std).numpy()
    assert out.shape == (3, 256, 512) and np.array_equal(out, ref)
