"""ASan + UBSan build of the virtual-memory layer's host code (proteus_amd/csrc/dswx_vmm.h) against an
in-test fake of the HIP virtual-memory calls that fails the k-th call (tests/native/vmm_fault_injection.cpp, which
documents the fake's rules and the scenario).  CPU only: gcc, no GPU, no libamdhip64."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'native', 'vmm_fault_injection.cpp')
OUT_DIR = os.path.join(ROOT, 'tests', 'native', '_build')
HIP_INCLUDE = '/opt/rocm/include'


def _build(tmp_path, include_dir, name='vmm_fault_injection'):
    gxx = shutil.which('g++')
    if gxx is None or not os.path.exists(os.path.join(HIP_INCLUDE, 'hip', 'hip_runtime_api.h')):
        pytest.skip('needs g++ and the HIP runtime API header')
    os.makedirs(OUT_DIR, exist_ok=True)
    exe = os.path.join(OUT_DIR, name)
    cmd = [gxx, '-std=c++17', '-g', '-O1', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-Wall',
           '-D__HIP_PLATFORM_AMD__', '-I', HIP_INCLUDE, '-I', include_dir, SRC, '-o', exe]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def _run(exe):
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:halt_on_error=1', UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1')
    env.pop('LD_PRELOAD', None)
    return subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)


def test_every_hip_call_of_a_placement_life_cycle_may_fail(tmp_path):
    """create -> wide create -> rehome (two intervals, a hole) -> drop both old ranges -> second range from the pool -> trim
    with live ranges -> drop everything -> trim: once without faults (64 HIP calls), then with the k-th call failing, for
    every k, once and from then on.  After every step of every run the library's account (live / retired / pooled /
    leaked) equals what the fake driver holds, the fake saw no rule broken (incl. "an address a kernel used is never mapped
    onto other memory" and "no reservation is freed with a mapping in it"), and ASan / UBSan / LeakSanitizer are silent."""
    exe = _build(tmp_path, os.path.join(ROOT, 'proteus_amd', 'csrc'))
    res = _run(exe)
    assert res.returncode == 0, (res.stdout[-1000:], res.stderr[-4000:])
    assert 'ERROR: AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr and 'LeakSanitizer' not in res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out['failures'] == 0 and out['hip_calls_fault_free'] >= 60
    assert out['runs_with_a_fault'] == 2 * out['hip_calls_fault_free']


def test_the_harness_sees_an_unchecked_unmap(tmp_path):
    """The harness must be able to fail: the same scenario against a copy of the layer in which destroy() ignores a failed
    hipMemUnmap again (as rounds 3 - 4 did: the still-mapped chunk went into the pool and the reservation was freed by the
    next trim with the mapping in it)."""
    src = open(os.path.join(ROOT, 'proteus_amd', 'csrc', 'dswx_vmm.h')).read()
    needle = 'if (hipMemUnmap(va + i * chunk, chunk) != hipSuccess) {'
    assert src.count(needle) == 1
    mutated = src.replace(needle, '(void)hipMemUnmap(va + i * chunk, chunk);\n                if (false) {')
    inc = tmp_path / 'mutant'
    inc.mkdir()
    (inc / 'dswx_vmm.h').write_text(mutated)
    exe = _build(tmp_path, str(inc), name='vmm_fault_injection_mutant')
    res = _run(exe)
    assert res.returncode != 0
    assert 'the reservation still has mappings' in res.stderr or 'driver memory' in res.stderr
