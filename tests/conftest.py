import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(autouse=True)
def _library_knobs_back_to_default(request):
    """Tuning knobs live in the library handle of each model (cotr_set_knob(h, ...)); the handle-less op-level entry points
    (cotr_op_*, cotr_train_*) read ONE process-wide set (cotr_set_knob(NULL, ...)).  A GPU test that flips a process-wide knob must
    not leak it into the tests that follow (round 2: a hand-written "restore" constant left the rest of the suite on a non-shipped
    kernel): after every GPU test that set is reset through the library's own registry and checked to report exactly the
    defaults; models cached across tests (tests/test_parity_gpu.py) are checked the same way."""
    yield
    if request.node.get_closest_marker('gpu') is None:
        return
    from cotr_amd import _lib
    if _lib._lib is None:          # the test never opened the library
        return
    _lib.reset_knobs()
    off = {k: v for k, v in _lib.knobs().items() if v[0] != v[1]}
    assert not off, f'process-wide knobs not at their defaults after reset: {off}'
    mod = sys.modules.get('tests.test_parity_gpu')
    for m in (getattr(mod, '_models', {}) if mod else {}).values():
        off = {k: v for k, v in m.knobs().items() if v[0] != v[1]}
        assert not off, f'a cached model was left with non-default knobs: {off}'
