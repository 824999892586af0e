"""Host-side engines for the MI355X model object.

The reference's own engines (``COTR/inference/sparse_engine.py``) keep working unchanged against
``cotr_amd.models.build_model`` (same call contract).  ``ZoomEngine`` is the MI355X-native way to run the
same recursive zoom-in: it stays host-side Python as the reference's is, but every zoom level is ONE
device-side crop+resize launch and one batched encode/decode instead of one PIL resize, one H2D copy and
one backbone pass per query per level."""
from .zoom_engine import FasterSparseEngine, RefineResult, SparseEngine, ZoomEngine, patch_boxes

__all__ = ['ZoomEngine', 'SparseEngine', 'FasterSparseEngine', 'patch_boxes', 'RefineResult']
