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
