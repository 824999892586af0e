"""``from cotr_amd.models import build_model`` is the drop-in for ``from COTR.models import build_model``
(COTR/models/__init__.py:9-10)."""
from .cotr_model import COTR, build
from .misc import NestedTensor, nested_tensor_from_tensor_list


def build_model(args):
    return build(args)


__all__ = ['build_model', 'COTR', 'NestedTensor', 'nested_tensor_from_tensor_list']
