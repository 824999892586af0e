"""Compile the oracle's plain-C restatements with gcc (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

    python -m oracle.build_c

``oracle/dense_cycle_ref.c`` -> ``oracle/_build/libdense_cycle_ref.so`` (git-ignored, travels with the gpurun snapshot).
"""
import ctypes
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libdense_cycle_ref.so')
SRC = os.path.join(HERE, 'dense_cycle_ref.c')


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    gcc = shutil.which('gcc')
    if gcc is None:
        raise RuntimeError('gcc not found (needed for the oracle C restatement)')
    os.makedirs(OUT, exist_ok=True)
    # no FMA contraction, no fast-math: every FMA of the restatement is an explicit fmaf() (software fmaf is exact where
    # the host has no FMA unit; -mfma is deliberately NOT passed so the binary runs on any x86-64 / aarch64 host)
    subprocess.run([gcc, '-O1', '-ffp-contract=off', '-shared', '-fPIC', '-o', LIB, SRC, '-lm'], check=True)
    return LIB


def load():
    lib = ctypes.CDLL(build())
    lib.dense_cycle_ref.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.dense_cycle_ref.restype = None
    return lib


def dense_cycle_ref(pred):
    """pred float32 [P,256,512,2] -> (cycle_grid [P,256,512,2], cycle error [P,256,512]) by the C restatement."""
    import numpy as np
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    P = pred.shape[0]
    cyc = np.empty((P, 256, 512, 2), np.float32)
    err = np.empty((P, 256, 512), np.float32)
    load().dense_cycle_ref(pred.ctypes.data, cyc.ctypes.data, err.ctypes.data, P)
    return cyc, err


if __name__ == '__main__':
    print(build(force=True))
