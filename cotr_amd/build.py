"""Build libcotr_hip.so (gfx950) in-tree with hipcc.

    python -m cotr_amd.build [--force]

The shared object lands next to the sources (``cotr_amd/csrc/libcotr_hip.so``): it is
git-ignored but travels with the gpurun snapshot, so the GPU box never compiles.
hipcc cross-compiles for gfx950 without a GPU.
"""
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libcotr_hip.so')
SOURCES = ['gemm.hip', 'gemm_big.hip', 'gemm_wp.hip', 'bottleneck.hip', 'gemm_ln.hip', 'attention.hip', 'pointwise.hip', 'stem_pool.hip', 'crop_resize.hip', 'dense_post.hip', 'ffn.hip', 'head.hip', 'train.hip', 'attention_train.hip', 'api.hip']
# Pillow-exact resamples (8-bit and float): double-precision coefficient code must not be contracted into FMAs
EXTRA_FLAGS = {'crop_resize.hip': ['-ffp-contract=off'], 'dense_post.hip': ['-ffp-contract=off']}
HEADERS = ['common.h', 'train.h', 'gemm_tuned.inc', os.path.join('..', '..', 'include', 'cotr_hip.h')]
# code-object v5: loadable by the ROCm 7.0 runtime torch bundles as well as by ROCm 7.2's
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-mcode-object-version=5',
         '-Wall', '-Wno-unused-function']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (needed to build libcotr_hip.so)')
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link the C-ABI shared library. Returns its path."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', '.o'))
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{out}')
        if verbose and out.strip():
            print(out)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
