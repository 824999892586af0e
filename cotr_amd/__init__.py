"""cotr_amd - COTR's batched correspondence-query forward path as hand-written gfx950 (MI355X) HIP
kernels behind a C ABI (``include/cotr_hip.h``), presented through the reference's own Python seam
(``cotr_amd.models.build_model`` == ``COTR.models.build_model``)."""
import argparse

__version__ = '0.1.0'


def default_args(**over):
    """COTR's model defaults (COTR/options/options.py:41-51; dim_feedforward derived from --layer as in
    demo_single_pair.py:58-62)."""
    a = argparse.Namespace(backbone='resnet50', hidden_dim=256, dilation=False, dropout=0.1, nheads=8,
                           layer='layer3', enc_layers=6, dec_layers=6, position_embedding='lin_sine',
                           dim_feedforward=1024)
    for k, v in over.items():
        setattr(a, k, v)
    return a
