"""The N > 1 path on the GPU: `bench.py --gpus 2` as a child process with both ranks on
device 0 (DSWX_BENCH_SHARE_DEVICE=1: gloo control plane, RCCL refuses two ranks on one GPU).  Not a measurement --
it checks what a 1-GPU run never exercises: rank-offset tile indices, the chunked strong-scaling walk, per-rank
results, and parity ON EVERY RANK (each rank checks its own tiles against the C oracle and the numpy generator;
the records are gathered into the line)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(*argv, share_device=True, force_dist=False, extra_env=None, expect_rc=0):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'DSWX_BENCH_SHARE_DEVICE', 'DSWX_FORCE_DIST'):
        env.pop(k, None)
    if share_device:
        env['DSWX_BENCH_SHARE_DEVICE'] = '1'
    if force_dist:
        env['DSWX_FORCE_DIST'] = '1'
    env.pop('DSWX_BENCH_INJECT', None)
    env.update(extra_env or {})
    # a child process (never exec from a process that touched the GPU); the parent bench process itself starts the
    # ranks with torch.distributed.run before it imports torch
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), capture_output=True, text=True,
                         timeout=1500, cwd=ROOT, env=env)
    assert (res.returncode == 0) == (expect_rc == 0), res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-3000:])
    return json.loads(lines[0])


def test_two_ranks_strong_scaling_every_rank_checks_every_chunk():
    out = _bench('--gpus', '2', '--total-tiles', '16', '--tiles', '4', '--steps', '2', '--warmup', '1',
                 '--no-cpu-baseline', '--distinct-chunks', '--tile-size', '1024')
    # n_gpus counts DISTINCT devices (PCI address + UUID), not ranks: both ranks share device 0 here, and the line says so
    assert out['n_ranks'] == 2 and out['n_gpus'] == 1 and 'DSWX_BENCH_SHARE_DEVICE' in out['n_gpus_note']
    assert out['scaling'] == 'strong'
    assert out['config']['tiles_per_step_all_ranks'] == 16 and out['config']['launches_per_step'] == 2
    assert out['config']['control_plane'] == 'gloo'
    par = out['parity_check']
    assert par['result'] == 'bit-exact', par
    r0, r1 = par['ranks']
    assert (r0['rank'], r0['first_tile'], r1['rank'], r1['first_tile']) == (0, 0, 1, 8)
    # resident chunk: first / middle / last; then the second chunk generated with ITS indices: first / last
    assert r0['tiles'] == [0, 2, 3, 4, 7] and r1['tiles'] == [8, 10, 11, 12, 15]
    assert r0['distinct_chunks'] == 2 and r1['distinct_chunks'] == 2
    assert out['value'] > 0 and out['roofline']['pixels_per_launch'] == 4 * 1024 * 1024
    # every rank's own numbers are in the line, the slowest one is named (value is bounded by it)
    ranks = out['ranks']
    assert [r['rank'] for r in ranks] == [0, 1] and ranks[0]['device'] == ranks[1]['device']
    for r in ranks:
        assert r['tiles_per_step'] == 8 and r['launches_per_step'] == 2 and r['launch_ms_avg'] > 0 and 0 < r['frac'] < 1
        assert r['placement']['how'] == 'first' and r['wall_ms_per_step'] > 0
    assert out['slowest_rank']['rank'] in (0, 1)
    assert out['slowest_rank']['wall_ms_per_step'] == max(r['wall_ms_per_step'] for r in ranks)
    # the end-to-end leg: both ranks at once, rates summed
    hp = out['host_path']
    assert hp['ranks'] == 2 and hp['zero_copy_Gpx_s'] > 0 and hp['pageable_Gpx_s'] > 0
    assert 'MISMATCH' not in hp['parity'] and 'bit-exact vs the C oracle' in hp['parity']
    assert 'zero copy' in hp['zero_copy_kernel'] and 'dswx_classify_lut' in hp['zero_copy_kernel']


def test_chain_mode_shadow_and_land_layers_into_the_batch():
    """`bench.py --chain` (BASELINE configs[4]'s per-pixel chain, one GPU's share): terrain shadow layer and LAND
    aggregation written straight into the SHAD / LAND planes of the resident batch (dswx_shadow_layer_batch,
    dswx_landcover_mask_batch with the batch's tile stride), then the classifier with SHAD + LAND + OCEAN.  The line's
    parity record compares SHAD and LAND with the numpy oracle's layers and everything downstream with the C oracle."""
    out = _bench('--chain', '--tiles', '3', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--tile-size', '1200',
                 share_device=False)
    assert out['n_gpus'] == 1 and out['n_ranks'] == 1 and 'configs[4]' in out['config']['workload'] and out['config']['planes_in'] == 10
    assert out['parity_check']['result'] == 'bit-exact', out['parity_check']
    assert out['parity_check']['ranks'][0]['tiles'] == [0, 2]
    r = out['roofline']
    assert abs(r['algorithmic_bytes_per_pixel'] - (24 + 11 + 4 * 1300 ** 2 / 1200 ** 2 + 1)) < 1e-3
    assert all(r['chain'][k] > 0 for k in ('terrain_shadow_ms', 'land_aggregation_ms', 'classify_ms'))


def test_two_ranks_weak_scaling_rank_offsets():
    out = _bench('--gpus', '2', '--tiles', '3', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--masks',
                 '--no-host-path', '--tile-size', '1024')
    assert out['n_ranks'] == 2 and out['scaling'] == 'weak' and out['config']['tiles_per_step_all_ranks'] == 6
    assert 'host_path' not in out and 'strong' not in out
    par = out['parity_check']
    assert par['result'] == 'bit-exact', par
    assert [r['first_tile'] for r in par['ranks']] == [0, 3]
    assert [r['tiles'] for r in par['ranks']] == [[0, 1, 2], [3, 4, 5]]


def test_rccl_control_plane_world_of_one():
    """the RCCL code path of an N > 1 run had never executed (the two-rank tests above force gloo:
    RCCL refuses two ranks on one GPU).  DSWX_FORCE_DIST=1 gives bench.py a process group of ONE rank over the 'nccl'
    backend -- init_process_group with device_id, barrier, all_reduce MAX / SUM on device tensors, all_gather_object,
    destroy -- around a real (small) measurement whose kernels run on the library's own stream beside torch's.  Every
    record that an N > 1 line gathers through the control plane must come back through RCCL intact."""
    out = _bench('--tiles', '2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-single-tile',
                 '--realloc-repeats', '0', share_device=False, force_dist=True)
    assert out['config']['control_plane'] == 'nccl' and out['rccl_ranks'] == 1
    pre = out['preflight']                      # taken whenever a process group exists
    assert pre['ok'] and pre['rccl_ranks'] == 1 and pre['ranks'][0]['hbm_free_GiB'] > 100 and pre['ranks'][0]['slide_slack_GiB'] == 48.0
    assert out['n_gpus'] == 1 and out['n_ranks'] == 1 and out['config']['tiles_per_step_all_ranks'] == 2
    assert out['parity_check']['result'] == 'bit-exact' and out['parity_check']['ranks'][0]['tiles'] == [0, 1]
    assert [r['rank'] for r in out['ranks']] == [0] and out['ranks'][0]['placement']['how'] == 'slide'
    assert out['value'] > 0 and out['ms_per_step'] > 0
    hp = out['host_path']                       # barriers, MAX and SUM of the leg went through RCCL too
    assert hp['ranks'] == 1 and hp['zero_copy_Gpx_s'] > 0 and 'MISMATCH' not in hp['parity']


def test_rccl_control_plane_methods_beside_the_library_stream():
    """The ControlPlane object itself over RCCL (world of one), every method, interleaved with launches of the library on
    its own non-blocking stream: values survive the device round trip, close() leaves no process group behind."""
    code = (
        "import os, sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np, torch\n"
        "from proteus_amd import _capi, shard\n"
        "from proteus_amd.synth import SEED\n"
        "torch.cuda.set_device(0)\n"
        "cp = shard.ControlPlane(backend='nccl', device=torch.device('cuda', 0))\n"
        "assert cp.dist is not None and cp.backend == 'nccl'\n"
        "ctx = _capi.Context(0)\n"
        "b = _capi.DeviceBatch(ctx, 2, 200, 300)\n"
        "b.synth(SEED)\n"
        "p = _capi.default_params()\n"
        "cp.barrier()\n"
        "b.classify(p)\n"
        "worst = cp.max_over_ranks(1.25)\n"
        "b.classify(p)\n"
        "total = cp.sum_over_ranks(2 ** 40 + 3)\n"
        "ctx.synchronize()\n"
        "objs = cp.gather_objects({'rank': cp.rank, 'nested': [1, 2.5, 'x']})\n"
        "cnt = cp.gather_counters(b.read_counters())\n"
        "cp.close()\n"
        "import torch.distributed as dist\n"
        "print(json.dumps({'worst': worst, 'total': total, 'objs': objs, 'cnt': np.asarray(cnt).tolist(), 'left': dist.is_initialized()}))\n"
        "b.free(); ctx.close()\n")
    env = dict(os.environ, DSWX_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"worst"')][-1])
    assert out['worst'] == 1.25 and out['total'] == 2 ** 40 + 3 and out['left'] is False
    assert out['objs'] == [{'rank': 0, 'nested': [1, 2.5, 'x']}]
    assert len(out['cnt']) == 2 and all(len(c) == 3 and c[0] > 0 for c in out['cnt'])


def test_address_space_budget_spent_falls_back_to_packed_planes():
    """ADVICE r03 (medium) / VERDICT r03 next-2: every sliding placement retires ~100 GiB of address space (addresses a
    kernel has used must never be re-mapped on this stack: tools/vmm_reuse_repro.hip), so a long-lived service must not
    die when that runs out.  With the library's address-space budget capped, dswx_batch_create(SLIDING) allocates the
    planes PACKED and says why, dswx_batch_place_slide leaves them where they are and succeeds, and the results are
    bit-exact.  A fresh process, so that the account starts at zero."""
    code = (
        "import sys, json\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np\n"
        "from oracle import c_oracle\n"
        "from proteus_amd import _capi\n"
        "from proteus_amd.synth import SEED, synth_tile\n"
        "ctx = _capi.Context(0)\n"
        "p = _capi.default_params()\n"
        "rec = {}\n"
        "def run(tag):\n"
        "    b = _capi.DeviceBatch(ctx, 3, 300, 400, sliding_outputs=True)\n"
        "    b.synth(SEED, tile0=40)\n"
        "    r = b.place_slide(p, slack_bytes=8 << 20, step_bytes=1 << 20, launches=1)\n"
        "    b.classify(p); ctx.synchronize()\n"
        "    ok = True\n"
        "    cnt = b.read_counters()\n"
        "    for t in range(3):\n"
        "        s = synth_tile(40 + t, 300, 400)\n"
        "        e = c_oracle.classify(p, s['bands'], s['fmask'])\n"
        "        ok = ok and all(np.array_equal(b.read_tile(k, t), e[k]) for k in ('diag', 'wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'))\n"
        "        ok = ok and cnt[t].tolist() == e['counters'].tolist()\n"
        "    rec[tag] = dict(place=r, info=b.info(), exact=ok, account=_capi.va_budget())\n"
        "    b.free()\n"
        "run('roomy')\n"
        "acct = _capi.va_budget()\n"
        "_capi.va_budget(acct['retired_bytes'] + (1 << 20))      # room for 1 MiB more: no range of this batch fits\n"
        "run('capped')\n"
        "_capi.va_budget(64 << 40)\n"
        "run('restored')\n"
        "print(json.dumps(rec))\n")
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    rec = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"roomy"')][-1])
    SLIDING = 1 << 11
    roomy, capped, restored = rec['roomy'], rec['capped'], rec['restored']
    assert roomy['exact'] and capped['exact'] and restored['exact']
    assert roomy['info']['flags'] & SLIDING and roomy['place']['positions'] > 0 and roomy['info']['note'] == ''
    assert roomy['info']['va_reserved_bytes'] > 0
    # after the batch is gone its ranges are retired (address space, not memory), none is live
    assert roomy['account']['live_bytes'] > 0 and capped['account']['live_bytes'] == 0
    assert not capped['info']['flags'] & SLIDING and capped['info']['n_allocations'] == 1
    assert 'address-space budget' in capped['info']['note'] and 'packed' in capped['info']['note']
    assert capped['place']['positions'] == 0 and capped['place']['first_come_launch_ms'] > 0
    assert capped['info']['va_reserved_bytes'] == 0
    assert capped['account']['retired_bytes'] >= roomy['info']['va_reserved_bytes']      # the dropped ranges: retired, not freed
    assert restored['info']['flags'] & SLIDING and restored['place']['positions'] > 0


def test_four_ranks_on_one_device_plain_command():
    """VERDICT r03 next-1b on the GPU, at toy sizes (hidden --plain-tiles / --strong-total / --strong-chunk): `bench.py
    --gpus N` with no workload named measures the weak record AND BASELINE configs[3]'s strong walk in the one line.
    More than two ranks (sharing this box's one device: gloo control plane), the plain two-record command; the two-rank
    form of it runs in test_a_failing_rank_is_a_record_in_the_line_real_kernels.
    Every rank must hold and check ITS tiles -- weak: tiles 2r, 2r + 1; strong: 20 tiles split five per rank, walked as
    chunks of 2 + 2 + 1 with every chunk generated under its own indices -- and report its own record.  (The same with
    EIGHT ranks and the host-path leg was run by hand -- profiles/r04_bench_8ranks_one_device_toy.json -- and is not part
    of the suite: nine HIP processes on one device took between 27 s and 6 min on the test boxes.)"""
    out = _bench('--gpus', '4', '--plain-tiles', '2', '--strong-total', '20', '--strong-chunk', '2', '--steps', '3',
                 '--warmup', '1', '--no-cpu-baseline', '--no-host-path', '--tile-size', '1024')
    assert out['n_ranks'] == 4 and out['n_gpus'] == 1 and out['config']['tiles_per_step_all_ranks'] == 8
    assert out['parity_check']['result'] == 'bit-exact'
    assert [r['tiles'] for r in out['parity_check']['ranks']] == [[2 * r, 2 * r + 1] for r in range(4)]
    assert [r['rank'] for r in out['ranks']] == list(range(4))
    st = out['strong']
    assert st['config']['tiles_per_step_all_ranks'] == 20 and st['config']['launches_per_step'] == 3
    assert st['parity_check']['result'] == 'bit-exact'
    assert [r['tiles'] for r in st['parity_check']['ranks']] == [list(range(5 * r, 5 * r + 5)) for r in range(4)]
    assert all(r['distinct_chunks'] == 3 for r in st['parity_check']['ranks'])
    assert [r['tiles_per_step'] for r in st['ranks']] == [5] * 4


def test_a_failing_rank_is_a_record_in_the_line_real_kernels():
    """VERDICT r04 next-1b on the GPU: two ranks on one device, the plain two-record command at toy sizes, and rank 1
    raises inside the timed region of the FIRST case (DSWX_BENCH_INJECT).  Rank 0's kernels and parity check are real.
    The line is printed, names the failure, has no whole-job value for that case, keeps rank 0's own roofline, and the
    second case -- measured after the failure by BOTH ranks, on real kernels -- is intact and bit-exact; the exit code
    is non-zero, after the line."""
    out = _bench('--gpus', '2', '--plain-tiles', '3', '--strong-total', '16', '--strong-chunk', '4', '--steps', '2',
                 '--warmup', '1', '--no-cpu-baseline', extra_env={'DSWX_BENCH_INJECT': '1:0:timed region'}, expect_rc=1)
    assert out['value'] is None and out['failed_ranks'] == [1] and 'rank 1 failed in timed region' in out['error']
    assert out['roofline']['frac'] > 0 and out['ranks'][0]['frac'] > 0 and 'injected' in out['ranks'][1]['error']
    assert out['parity_check']['ranks'][0]['result'] == 'bit-exact'
    assert out['parity_check']['ranks'][1]['result'].startswith('not checked')
    assert 'host_path' not in out                   # the collective legs after a failure are skipped by every rank
    st = out['strong']
    assert st['value'] > 0 and 'error' not in st and st['parity_check']['result'] == 'bit-exact'
    assert [r['tiles'] for r in st['parity_check']['ranks']] == [[0, 2, 3, 4, 7], [8, 10, 11, 12, 15]]
    assert out['preflight']['distinct_devices'] == 1 and not out['preflight']['ok']      # both ranks on device 0: said


def test_a_rank_that_dies_hard_in_the_second_case_leaves_the_first_in_the_line():
    """What no try / except catches: rank 1's process ENDS (os._exit, as a GPU fault would end it) while it places the
    batch of the second case.  torchrun terminates rank 0, which sits in a collective that will never complete; its
    wake-up-pipe thread (or the exception gloo raises about the lost peer) prints the line as far as it got: the weak
    record measured on real kernels by both ranks, bit-exact, and `strong.value: null` with the reason."""
    out = _bench('--gpus', '2', '--plain-tiles', '2', '--strong-total', '8', '--strong-chunk', '2', '--steps', '2',
                 '--warmup', '1', '--no-cpu-baseline', '--tile-size', '1024',
                 extra_env={'DSWX_BENCH_INJECT': '1:1:place:hard'}, expect_rc=1)
    assert out['value'] > 0 and out['n_ranks'] == 2 and out['parity_check']['result'] == 'bit-exact'
    assert [r['rank'] for r in out['ranks']] == [0, 1] and all(r['frac'] > 0 for r in out['ranks'])
    assert out['strong']['value'] is None and 'terminated while case 1 (strong) was running' in out['strong']['error']


def test_a_rank_whose_allocation_fails_is_a_record_and_the_memory_comes_back():
    """The failure VERDICT r04 names: the resident chunk does not fit (here: 40 000 tiles = 10 TB asked of the library).
    Both ranks fail in 'place' with the library's own error, the line says so, nothing is left allocated, exit non-zero."""
    out = _bench('--gpus', '2', '--tiles', '40000', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-host-path',
                 expect_rc=1)
    assert out['value'] is None and out['failed_ranks'] == [0, 1] and out['roofline'] is None
    assert all(r['phase'] == 'place' and 'DswxError' in r['error'] for r in out['ranks'])
    assert any('likely to fail' in r.get('warning', '') for r in out['preflight']['ranks'])
