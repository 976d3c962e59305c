"""Builds the HIP shared library in-tree (proteus_amd/_lib/libdswx_hip.so).

hipcc cross-compiles gfx950 without a GPU; the built .so is git-ignored but
travels with the gpurun snapshot.  `python -m proteus_amd.build` or
__graft_entry__.build() call this.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
SOURCES = [os.path.join(CSRC, n) for n in ('dswx_hip.hip', 'dswx_classify_lut.hip', 'dswx_layers.hip', 'dswx_host_path.hip',
                                                  'dswx_variants.hip', 'dswx_probes.hip')]
HEADERS = [os.path.join(CSRC, n) for n in ('dswx_device.h', 'dswx_host.h', 'dswx_tables.h')]
INCLUDE = os.path.join(ROOT, 'include')
LIB_DIR = os.path.join(PKG, '_lib')
LIB_PATH = os.path.join(LIB_DIR, 'libdswx_hip.so')

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm for the gfx950 build)')


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = SOURCES + HEADERS + [os.path.join(INCLUDE, 'dswx_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the library if missing or older than its sources; return its path."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [find_hipcc()] + HIPCC_FLAGS + ['-I', INCLUDE, '-I', CSRC] + SOURCES + ['-o', LIB_PATH + '.tmp']
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    os.replace(LIB_PATH + '.tmp', LIB_PATH)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
