"""The native DEFLATE codec's thread pool (proteus_amd/csrc/dswx_codec.cpp, round 6) under sanitizers on the CPU: the source
is compiled INTO a stress harness (tests/native/codec_stress.cpp: several caller threads, random batches, both engines,
failing calls mixed in, the CPU budget changed while calls run) once with ThreadSanitizer and once with ASan + UBSan +
LeakSanitizer.  GPU sanitizers are not available on this pool; this code never touches the GPU."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, 'tests', 'native', 'codec_stress.cpp'), os.path.join(ROOT, 'proteus_amd', 'csrc', 'dswx_codec.cpp')]
OUT_DIR = os.path.join(ROOT, 'tests', 'native', '_build')


def _build(name, flags):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('needs g++')
    os.makedirs(OUT_DIR, exist_ok=True)
    exe = os.path.join(OUT_DIR, name)
    cmd = [gxx, '-std=c++17', '-g', '-O1', '-pthread', '-Wall'] + flags + ['-I', os.path.join(ROOT, 'include')] + SRC + \
        ['-o', exe, '-lz', '-ldl']
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if res.returncode != 0 and ('cannot find' in res.stderr or 'No such file' in res.stderr):
        pytest.skip(f'sanitizer runtime missing: {res.stderr[-300:]}')
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


def _run(exe, env_extra, args=('6', '25')):
    env = dict(os.environ, **env_extra)
    env.pop('LD_PRELOAD', None)
    res = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=900, env=env)
    return res


def test_codec_pool_under_thread_sanitizer():
    exe = _build('codec_stress_tsan', ['-fsanitize=thread'])
    res = _run(exe, {'TSAN_OPTIONS': 'halt_on_error=1:second_deadlock_stack=1'})
    if res.returncode != 0 and 'unexpected memory mapping' in res.stderr:
        pytest.skip('ThreadSanitizer cannot map its shadow in this container')
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-4000:])
    assert 'WARNING: ThreadSanitizer' not in res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out['failures'] == 0 and out['calls'] >= 2 * 2 * 6 * 25 and out['expected_errors'] > 10


def test_codec_pool_under_address_and_undefined_behaviour_sanitizers():
    exe = _build('codec_stress_asan', ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined'])
    res = _run(exe, {'ASAN_OPTIONS': 'detect_leaks=1:halt_on_error=1', 'UBSAN_OPTIONS': 'print_stacktrace=1:halt_on_error=1'})
    assert res.returncode == 0, (res.stdout[-500:], res.stderr[-4000:])
    assert 'ERROR: AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out['failures'] == 0 and out['blocks'] > 5000
