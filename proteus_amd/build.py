"""Builds the HIP shared libraries in-tree (proteus_amd/_lib/).

  libdswx_hip.so   the product: production kernels + the C-ABI of include/dswx_hip.h
  libdswx_lab.so   experiments only (csrc/lab/: losing kernel structures, roofline probes, A/B switches);
                   links against libdswx_hip.so; loaded by tools/ and the variant tests, never by the product

hipcc cross-compiles gfx950 without a GPU; the built .so files are git-ignored but travel with the
gpurun snapshot.  `python -m proteus_amd.build` or __graft_entry__.build() call this.  Every entry
(tests, bench.py, smoke) calls build(): it is a no-op when the libraries are newer than their sources,
so an edited kernel can never be measured or tested through a stale binary.
"""
import contextlib
import fcntl
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LAB = os.path.join(CSRC, 'lab')
SOURCES = [os.path.join(CSRC, n) for n in ('dswx_hip.hip', 'dswx_classify_lut.hip', 'dswx_cover.hip',
                                           'dswx_layers.hip', 'dswx_host_path.hip', 'dswx_batch.hip')]
HEADERS = [os.path.join(CSRC, n) for n in ('dswx_device.h', 'dswx_host.h', 'dswx_tables.h')]
LAB_SOURCES = [os.path.join(LAB, n) for n in ('dswx_variants.hip', 'dswx_probes.hip')]
LAB_HEADERS = [os.path.join(LAB, 'dswx_lab.h')]
INCLUDE = os.path.join(ROOT, 'include')
PUBLIC_HEADER = os.path.join(INCLUDE, 'dswx_hip.h')
LIB_DIR = os.path.join(PKG, '_lib')
LIB_PATH = os.path.join(LIB_DIR, 'libdswx_hip.so')
LAB_PATH = os.path.join(LIB_DIR, 'libdswx_lab.so')

HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-ffp-contract=off', '-Wall', '-Wno-unused-function']
# the translation units that make up the headline kernel (bench.py ties profiles/pmc_traffic.json to them)
HOT_KERNEL_SOURCES = [os.path.join(CSRC, n) for n in ('dswx_classify_lut.hip', 'dswx_tables.h', 'dswx_device.h')]


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    return None


def _stale(lib, deps):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in deps)


def is_stale():
    return _stale(LIB_PATH, SOURCES + HEADERS + [PUBLIC_HEADER])


def lab_is_stale():
    return _stale(LAB_PATH, LAB_SOURCES + LAB_HEADERS + HEADERS + [PUBLIC_HEADER, LIB_PATH])


def hot_kernel_hash():
    """sha256 over the sources of the headline kernel (the files named in VERDICT r01 item 6)."""
    h = hashlib.sha256()
    for path in HOT_KERNEL_SOURCES:
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _run(cmd, verbose):
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)


@contextlib.contextmanager
def _build_lock():
    """One builder at a time across processes (ranks, xdist workers, batch workers all call build()): the
    others wait here and then find the library fresh."""
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, '.build.lock'), 'w') as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def _need_hipcc(lib):
    hipcc = find_hipcc()
    if hipcc is None:
        # never a stale binary: a library older than its sources is not what the sources say
        raise RuntimeError(f'hipcc not found and {os.path.relpath(lib, ROOT)} is '
                           + ('older than its sources' if os.path.exists(lib) else 'missing')
                           + ' (need ROCm for the gfx950 build)')
    return hipcc


def build(force=False, verbose=False):
    """Compile the product library if missing or older than its sources; return its path.
    Stale or missing without hipcc raises: no entry point ever runs an old binary."""
    if not force and not is_stale():
        return LIB_PATH
    with _build_lock():
        if not force and not is_stale():            # another process built it while this one waited
            return LIB_PATH
        hipcc = _need_hipcc(LIB_PATH)
        tmp = f'{LIB_PATH}.{os.getpid()}.tmp'
        _run([hipcc] + HIPCC_FLAGS + ['-I', INCLUDE, '-I', CSRC] + SOURCES + ['-o', tmp], verbose)
        os.replace(tmp, LIB_PATH)
    return LIB_PATH


def build_lab(force=False, verbose=False):
    """Compile libdswx_lab.so (experiments; links against the product library)."""
    build(force=False, verbose=verbose)
    if not force and not lab_is_stale():
        return LAB_PATH
    with _build_lock():
        if not force and not lab_is_stale():
            return LAB_PATH
        hipcc = _need_hipcc(LAB_PATH)
        tmp = f'{LAB_PATH}.{os.getpid()}.tmp'
        _run([hipcc] + HIPCC_FLAGS + ['-I', INCLUDE, '-I', CSRC, '-I', LAB] + LAB_SOURCES +
             ['-L', LIB_DIR, '-ldswx_hip', '-Wl,-rpath,$ORIGIN', '-o', tmp], verbose)
        os.replace(tmp, LAB_PATH)
    return LAB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
    print(build_lab(force='--force' in sys.argv, verbose=True))
