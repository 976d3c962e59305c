"""The N > 1 path on CPU: static tile sharding + the gloo control plane
(world_size 2), i.e. everything bench.py does across ranks except the kernel."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from proteus_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('n_tiles', [0, 1, 7, 8, 256, 4096, 4097])
@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_tile_range_partitions_exactly(n_tiles, world):
    seen = []
    sizes = []
    for r in range(world):
        lo, hi = shard.tile_range(n_tiles, r, world)
        assert 0 <= lo <= hi <= n_tiles
        seen += list(range(lo, hi))
        sizes.append(hi - lo)
    assert seen == list(range(n_tiles))
    assert max(sizes) - min(sizes) <= 1


def test_tile_range_rejects_bad_rank():
    with pytest.raises(ValueError):
        shard.tile_range(8, 2, 2)
    with pytest.raises(ValueError):
        shard.tile_range(-1, 0, 1)


def test_weak_range():
    assert shard.weak_tile_range(256, 3) == (768, 1024)


WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %r)
    import numpy as np
    from proteus_amd import shard
    from proteus_amd.synth import synth_tile
    from oracle import dswx_oracle as o
    cp = shard.ControlPlane(backend='gloo')
    lo, hi = shard.tile_range(5, cp.rank, cp.world)
    # each rank classifies ITS tiles (the oracle stands in for the kernel on CPU)
    rows = []
    for t in range(lo, hi):
        s = synth_tile(t, 24, 40)
        c = o.classify_tile(s['bands'], s['fmask'])['counters']
        rows.append([c['n_valid'], c['n_cloud_and_valid'], c['n_not_ocean']])
    local = np.asarray(rows, dtype=np.int64).reshape(-1, 3)
    cp.barrier()
    elapsed = 1.0 + cp.rank          # pretend rank 1 is slower
    worst = cp.max_over_ranks(elapsed)
    allc = cp.gather_counters(local)
    cp.barrier()
    if cp.rank == 0:
        print(json.dumps({'worst': worst, 'counters': allc.tolist(), 'world': cp.world}))
    cp.close()
''') % ROOT


def test_two_rank_gloo_control_plane(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['world'] == 2 and out['worst'] == 2.0
    from oracle import dswx_oracle as o
    from proteus_amd.synth import synth_tile
    exp = []
    for t in range(5):
        s = synth_tile(t, 24, 40)
        c = o.classify_tile(s['bands'], s['fmask'])['counters']
        exp.append([c['n_valid'], c['n_cloud_and_valid'], c['n_not_ocean']])
    assert out['counters'] == exp


@pytest.mark.parametrize('n,chunk', [(0, 4), (1, 4), (4, 4), (9, 4), (512, 512), (513, 512)])
def test_chunk_sizes(n, chunk):
    c = shard.chunk_sizes(n, chunk)
    assert sum(c) == n and all(0 < x <= chunk for x in c) and c[:-1] == [chunk] * (len(c) - 1)


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), capture_output=True,
                          text=True, timeout=300, cwd=ROOT, env=e)


def test_bench_self_launches_its_ranks_strong_scaling():
    """`python bench.py --gpus 2 --total-tiles 13` with no torchrun environment: the parent starts the two
    ranks itself (torch.distributed.run as a child) and relays rank 0's line.  --plan-only stops before any
    GPU work, so this runs here: BASELINE configs[3]'s split, chunked walk and control plane."""
    import json
    res = _bench('--gpus', '2', '--total-tiles', '13', '--tiles', '4', '--plan-only')
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong' and out['tiles_per_step_all_ranks'] == 13
    r0, r1 = out['ranks']
    assert (r0['first_tile'], r0['tiles'], r0['launches']) == (0, 6, [4, 2])
    assert (r1['first_tile'], r1['tiles'], r1['launches']) == (6, 7, [4, 3])


def test_bench_self_launch_weak_scaling_and_single_rank_plan():
    import json
    res = _bench('--gpus', '2', '--tiles', '3', '--plan-only')
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['scaling'] == 'weak' and out['tiles_per_step_all_ranks'] == 6
    assert [r['first_tile'] for r in out['ranks']] == [0, 3]
    # N = 1 walks configs[3] alone, in resident chunks
    res = _bench('--total-tiles', '4096', '--plan-only')         # default chunk in strong mode: 512 resident tiles
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['ranks'][0]['launches'] == [512] * 8 and out['tiles_per_step_all_ranks'] == 4096


def test_plain_gpus_command_plans_the_weak_and_the_strong_record():
    """`bench.py --gpus N` with nothing else -- what the driver runs for the scaling curve -- yields
    BOTH records in its one line: the weak one at the top level (256 tiles per GPU = configs[2] x N) and BASELINE
    configs[3] as the `strong` sub-record (4096 tiles split over the ranks, 512-tile resident chunks).  Naming a workload
    (--tiles / --total-tiles / --masks / --chain) or --no-strong keeps the one record that was asked for."""
    import json
    res = _bench('--gpus', '2', '--plan-only')
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['scaling'] == 'weak' and out['tiles_per_step_all_ranks'] == 512
    assert [(r['first_tile'], r['tiles'], r['launches']) for r in out['ranks']] == [(0, 256, [256]), (256, 256, [256])]
    st = out['strong']
    assert st['scaling'] == 'strong' and st['tiles_per_step_all_ranks'] == 4096 and st['n_gpus'] == 2
    assert [(r['first_tile'], r['tiles'], r['launches']) for r in st['ranks']] == \
        [(0, 2048, [512] * 4), (2048, 2048, [512] * 4)]
    for extra in (['--no-strong'], ['--tiles', '8'], ['--masks']):
        res = _bench('--gpus', '2', '--plan-only', *extra)
        out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
        assert 'strong' not in out and out['scaling'] == 'weak', extra
    res = _bench('--plan-only')                                     # N = 1: configs[2] alone
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert 'strong' not in out and out['tiles_per_step_all_ranks'] == 256
    res = _bench('--plan-only', '--also-strong')                    # ... unless asked for: the one rank walks all 4096 tiles
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])
    assert out['tiles_per_step_all_ranks'] == 256 and out['strong']['ranks'][0]['launches'] == [512] * 8


def test_forced_world_of_one_goes_through_the_process_group(tmp_path):
    """DSWX_FORCE_DIST=1: a process group of ONE rank outside torchrun (private tcp rendezvous), so that every call an
    N > 1 run makes -- init, barrier, all_reduce MAX / SUM, all_gather_object, destroy -- runs on a single box; here
    over gloo, on the GPU box over RCCL (tests/test_gpu_multirank.py)."""
    script = tmp_path / 'one.py'
    script.write_text(textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        from proteus_amd import shard
        cp = shard.ControlPlane(backend='gloo')
        assert cp.dist is not None and cp.backend == 'gloo' and cp.world == 1
        cp.barrier()
        assert cp.max_over_ranks(2.5) == 2.5 and cp.sum_over_ranks(7) == 7
        assert cp.gather_objects({'rank': 0}) == [{'rank': 0}]
        assert cp.gather_counters(np.arange(6).reshape(2, 3)).tolist() == [[0, 1, 2], [3, 4, 5]]
        cp.close()
        assert cp.dist is None
        print('OK')
    ''') % ROOT)
    e = dict(os.environ, DSWX_FORCE_DIST='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert res.returncode == 0 and 'OK' in res.stdout, res.stderr[-2000:]


def test_bench_rank_without_a_gpu_fails_loudly():
    """A real (not --plan-only) run on a box without a GPU must fail, not fall back to anything: exit code non-zero,
    and the line it still prints (a failure is a record) carries no number, only the reason."""
    import json
    from proteus_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip('a GPU is present')
    res = _bench('--steps', '1', '--warmup', '0', '--tiles', '1')
    assert res.returncode != 0
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert out['value'] is None and out['ms_per_step'] is None and out['roofline'] is None
    assert 'no GPU visible to this rank: the DSWx HIP path has no CPU fallback' in out['error']
    assert 'cpu_baseline' not in out and 'single_tile' not in out and 'host_path' not in out


def test_control_plane_does_not_fall_back_to_gloo_silently(tmp_path):
    """RCCL cannot come up here (no GPU): with allow_fallback=False (or require=True) the control plane raises on every
    rank; by default the backend string says what happened (more cases: tests/test_bench_survival.py)."""
    script = tmp_path / 'cp.py'
    script.write_text(textwrap.dedent('''
        import sys
        sys.path.insert(0, %r)
        from proteus_amd import shard
        allow = sys.argv[1] == '1'
        try:
            cp = shard.ControlPlane(backend='nccl', device=None, allow_fallback=allow)
        except Exception as e:
            print('RAISED', type(e).__name__)
            raise SystemExit(3)
        if cp.rank == 0:
            print('BACKEND', cp.backend)
        cp.close()
    ''') % ROOT)
    for allow, rc in (('0', 3), ('1', 0)):
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
               '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), allow]
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
        if rc:
            assert res.returncode != 0 and 'RAISED' in res.stdout
        else:
            assert res.returncode == 0, res.stderr[-2000:]
            assert 'BACKEND gloo (fallback: nccl bring-up failed on rank(s) [0, 1]' in res.stdout


def test_batch_plan():
    from proteus_amd import batch
    rcs = [f'rc{i}.yaml' for i in range(10)]
    p = batch.plan(rcs, 4)
    assert [g for g, _ in p] == [0, 1, 2, 3]
    assert sum((c for _, c in p), []) == rcs
    assert {len(c) for _, c in p} <= {2, 3}
    few = batch.plan(rcs[:2], 8)
    assert sum((c for _, c in few), []) == rcs[:2] and max(len(c) for _, c in few) == 1


def test_batch_plan_several_workers_per_gpu():
    from proteus_amd import batch
    rcs = [f'rc_{i}.yaml' for i in range(11)]
    p = batch.plan(rcs, 2, workers_per_gpu=3)
    assert [g for g, _ in p] == [0, 0, 0, 1, 1, 1]
    assert [x for _, chunk in p for x in chunk] == rcs               # contiguous, complete, in order
    assert max(len(c) for _, c in p) - min(len(c) for _, c in p) <= 1
