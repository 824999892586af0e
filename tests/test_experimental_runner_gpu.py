"""One `pytest -m gpu` run covers BOTH libraries: this test starts a second pytest process with COTR_HIP_EXPERIMENTAL=1 (which makes
cotr_amd load libcotr_hip_exp.so - the product sources + cotr_amd/csrc/experimental/ compiled with -DCOTR_EXPERIMENTAL) on
tests/test_experimental_gpu.py, the tests of the measured dead ends that no longer live in libcotr_hip.so, plus the golden-vector
parity test, so the experimental build is also pinned end to end."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_experimental_library_suite():
    from cotr_amd import _lib
    from cotr_amd.build import LIB_EXP
    if _lib.experimental_selected():
        pytest.skip('already inside the experimental run')
    assert os.path.exists(LIB_EXP), 'libcotr_hip_exp.so missing: python -m cotr_amd.build --experimental (build() makes it)'
    env = dict(os.environ, COTR_HIP_EXPERIMENTAL='1')
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', 'tests/test_experimental_gpu.py',
           'tests/test_parity_gpu.py::test_golden_vectors_from_the_reference', 'tests/test_parity_gpu.py::test_knobs_are_per_handle']
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = '\n'.join(r.stdout.splitlines()[-25:])
    assert r.returncode == 0, tail
    assert ' passed' in tail and 'skipped' not in tail.splitlines()[-1], tail
