"""``NestedTensor`` container the reference's callers may pass to ``model(samples, queries)``
(COTR/models/misc.py:35-55, :58-80).  On the COTR path the mask is all-False for every caller
(inputs are always exactly 256x512), so it is carried but never consumed."""
from typing import List, Optional

import torch
from torch import Tensor


class NestedTensor(object):
    def __init__(self, tensors: Tensor, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask

    def to(self, device):
        return NestedTensor(self.tensors.to(device), None if self.mask is None else self.mask.to(device))

    def decompose(self):
        return self.tensors, self.mask

    def __repr__(self):
        return str(self.tensors)


def nested_tensor_from_tensor_list(tensor_list: List[Tensor]) -> NestedTensor:
    """Batch of equally sized [3,H,W] images -> NestedTensor with an all-False mask."""
    if isinstance(tensor_list, Tensor):
        if tensor_list.ndim != 4:
            raise ValueError('not supported')
        batch = tensor_list
    else:
        if tensor_list[0].ndim != 3:
            raise ValueError('not supported')
        shapes = {tuple(t.shape) for t in tensor_list}
        if len(shapes) != 1:
            raise ValueError('COTR feeds equally sized 256x512 pairs; ragged batches are not supported')
        batch = torch.stack(list(tensor_list))
    b, _, hh, ww = batch.shape
    return NestedTensor(batch, torch.zeros((b, hh, ww), dtype=torch.bool, device=batch.device))
