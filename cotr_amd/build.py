"""Build libcotr_hip.so (gfx950) in-tree with hipcc.

    python -m cotr_amd.build [--force] [--experimental | --experimental-only] [--refresh-patches NAME ...]

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
# The research library (split-f16 products, the measured dead ends and their knobs): built with -DCOTR_EXPERIMENTAL from PATCHED copies
# of the translation units it changes (FORKED below) + the product's other translation units (which reach its declarations through the
# redirect in common.h) + csrc/experimental/*.hip.  The patched copies are not in the repository: csrc/experimental/patches/<name>.patch
# (a unified diff against the product file) is applied to the CURRENT product source at build time into csrc/experimental/gen/
# (git-ignored), so a fix to a product kernel reaches the research library with the next build, and a product edit a patch no longer
# fits fails that build loudly instead of leaving a stale fork behind (tests/test_abi_cpu.py checks that every patch applies).
# The product sources contain none of the research code.  Never loaded by the product path; tests/test_experimental_gpu.py and the
# A/B tools select it with COTR_HIP_EXPERIMENTAL=1.
LIB_EXP = os.path.join(CSRC, 'libcotr_hip_exp.so')
FORKED = ['gemm.hip', 'gemm_big.hip', 'attention.hip', 'pointwise.hip', 'ffn.hip', 'api.hip']
PATCH_DIR = os.path.join(CSRC, 'experimental', 'patches')
GEN_DIR = os.path.join(CSRC, 'experimental', 'gen')
# generated name -> (product source, patch)
GENERATED = {**{f: (f, f + '.patch') for f in FORKED}, 'common_exp.h': ('common.h', 'common_exp.h.patch')}
SOURCES = ['gemm.hip', 'gemm_big.hip', 'gemm_wp.hip', 'bottleneck.hip', 'attention.hip', 'pointwise.hip', 'stem_pool.hip', 'crop_resize.hip',
           'dense_post.hip', 'ffn.hip', 'ffn_rows.hip', 'att_rows.hip', 'conv23.hip', 'conv23m.hip', 'expand.hip', 'train.hip', 'attention_train.hip', 'api.hip']
EXP_SOURCES = [os.path.join('experimental', 'head.hip'), os.path.join('experimental', 'gemm_ln.hip'), os.path.join('experimental', 'gemm_pp.hip'),
               os.path.join('experimental', 'gemm_h2.hip'), os.path.join('experimental', 'attention_h2.hip'), os.path.join('experimental', 'gemm_h2r.hip'),
               os.path.join('experimental', 'linear_rows.hip')]
# Pillow-exact resamples and the torch-CPU-exact cycle map (8-bit, float and double code whose products must not be contracted into
# FMAs behind the source's back; the FMAs that belong there are explicit)
EXTRA_FLAGS = {'crop_resize.hip': ['-ffp-contract=off'], 'dense_post.hip': ['-ffp-contract=off'],
               # att_rows.hip: matrix-instruction results in VGPRs, not AccVGPRs - its softmax VALU sits between the matrix instructions of ONE
               # wavefront per SIMD, and any AccVGPR access (v_accvgpr_read / write) waits for the matrix instruction in flight: with the scores in
               # AccVGPRs the softmax of a key block ran entirely BEHIND its 64 matrix instructions (K/V phase 205 k cycles against 131 k of matrix
               # work, whatever the interleaving; profiles/r5_att_rows_probe.txt)
               'att_rows.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form']}
HEADERS = ['common.h', 'train.h', 'gemm_tuned.inc', 'enc_split.inc', os.path.join('..', '..', 'include', 'cotr_hip.h')]
EXP_HEADERS = [os.path.join('experimental', f) for f in ['coop_tail.h', 'experimental.h', 'api_exp.inc', 'gemm_h2.h']] + \
              [os.path.join('experimental', 'patches', pt) for _, pt in GENERATED.values()]
# code-object v5: loadable by the ROCm 7.0 runtime torch bundles as well as by ROCm 7.2's
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-mcode-object-version=5', '-fvisibility=hidden',
         '-Wall', '-Wno-unused-function']


def declared_symbols(experimental=False):
    """The C ABI: every function include/cotr_hip.h declares (the #ifdef COTR_EXPERIMENTAL block only for the experimental library)."""
    import re
    src = open(os.path.join(CSRC, '..', '..', 'include', 'cotr_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    if not experimental:
        src = re.sub(r'#ifdef COTR_EXPERIMENTAL.*?#endif', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(cotr_[a-z0-9_]+)\s*\(', src)))


def _version_script(objdir, experimental):
    """A linker version script that exports exactly the header's names: kernel host stubs (_Z...), the thread-local knob / device
    slots and every other internal symbol stay local - the shared object is a sealed C ABI."""
    path = os.path.join(objdir, 'exports_exp.map' if experimental else 'exports.map')
    with open(path, 'w') as f:
        f.write('{\n  global:\n' + ''.join(f'    {n};\n' for n in declared_symbols(experimental)) + '  local:\n    *;\n};\n')
    return path


def generate_forks(dest=None, verbose=False):
    """Apply csrc/experimental/patches/*.patch to the current product sources -> dest (default csrc/experimental/gen/).  Raises with
    patch(1)'s report when a hunk no longer fits.  Returns the directory."""
    dest = dest or GEN_DIR
    os.makedirs(dest, exist_ok=True)
    exe = shutil.which('patch')
    if not exe:
        raise RuntimeError('patch(1) not found (needed to derive the research library\'s sources from the product sources)')
    for name, (src, pt) in GENERATED.items():
        out = os.path.join(dest, name)
        r = subprocess.run([exe, '--no-backup-if-mismatch', '-s', '-F', '1', '-o', out, os.path.join(CSRC, src), os.path.join(PATCH_DIR, pt)],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0 or os.path.exists(out + '.rej'):
            raise RuntimeError(f'experimental/patches/{pt} no longer applies to csrc/{src} (edit the patch, or regenerate it with '
                               f'`python -m cotr_amd.build --refresh-patches` after fixing {out} by hand):\n{r.stdout}')
        if verbose:
            print(f'generated {out}')
    return dest


def refresh_patches(names):
    """After editing generated sources under csrc/experimental/gen/ by hand: rewrite the patches of the NAMED ones (e.g. api.hip
    common_exp.h) as the diff against the product file.  Named explicitly on purpose: a generated file that was not regenerated since
    its product file changed would otherwise turn the product's change into a reverse hunk of the patch."""
    for name, (src, pt) in GENERATED.items():
        gen = os.path.join(GEN_DIR, name)
        if name not in names:
            continue
        if not os.path.exists(gen):
            raise RuntimeError(f'{gen} does not exist')
        r = subprocess.run(['diff', '-u', '--label', src, '--label', name, os.path.join(CSRC, src), gen], stdout=subprocess.PIPE, text=True)
        open(os.path.join(PATCH_DIR, pt), 'w').write(r.stdout)
        print(f'refreshed experimental/patches/{pt}')


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (needed to build libcotr_hip.so)')
    return exe


_probed = {}


def _supported(hipcc, extra):
    """Do these extra flags compile an empty translation unit with this hipcc?  (-mllvm options are specific to a compiler version:
    att_rows.hip is built WITH -amdgpu-mfma-vgpr-form where the compiler has it - 2107 -> 2016 us at 32 pairs x 1000 queries - and
    without it elsewhere: same results, the softmax of a key block then runs behind its matrix instructions.)"""
    key = (hipcc,) + tuple(extra)
    if key not in _probed:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, 'probe.hip')
            open(src, 'w').write('__global__ void probe() {}\n')
            r = subprocess.run([hipcc, '--offload-arch=gfx950', '-c', src, '-o', os.path.join(d, 'probe.o')] + list(extra),
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        _probed[key] = r.returncode == 0
        if r.returncode != 0:
            print(f'[build] {" ".join(extra)} is not accepted by this hipcc: building without it', file=sys.stderr)
    return _probed[key]


def needs_build(experimental=False):
    lib = LIB_EXP if experimental else LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS + (EXP_SOURCES + EXP_HEADERS if experimental else [])]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, experimental=False):
    """Compile every HIP source for gfx950 and link the C-ABI shared library (the experimental one with experimental=True).
    Returns its path."""
    lib = LIB_EXP if experimental else LIB
    if not force and not needs_build(experimental):
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, 'experimental', 'obj') if experimental else CSRC
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    if experimental:
        generate_forks(verbose=verbose)
    sources = [os.path.join('experimental', 'gen', f) if f in FORKED else f for f in SOURCES] + EXP_SOURCES if experimental else SOURCES
    # (-I csrc/experimental: the generated sources keep the relative includes of a file that lives in csrc/experimental/)
    exp_flags = ['-DCOTR_EXPERIMENTAL', '-I' + os.path.join(CSRC, 'experimental')] if experimental else []
    for src in sources:
        obj = os.path.join(objdir, os.path.basename(src).replace('.hip', '.o'))
        extra = EXTRA_FLAGS.get(os.path.basename(src), [])
        if '-mllvm' in extra and not _supported(hipcc, extra):
            extra = []
        cmd = [hipcc] + FLAGS + exp_flags + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
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
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-Wl,--version-script=' + _version_script(objdir, experimental), '-o', lib] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}')
    return lib


if __name__ == '__main__':
    if '--refresh-patches' in sys.argv:
        refresh_patches(sys.argv[sys.argv.index('--refresh-patches') + 1:])
        sys.exit(0)
    force = '--force' in sys.argv
    if '--experimental-only' not in sys.argv:
        print(build_library(force=force, verbose=True))
    if '--experimental' in sys.argv or '--experimental-only' in sys.argv:
        print(build_library(force=force, verbose=True, experimental=True))
