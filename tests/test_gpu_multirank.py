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


def _bench(*argv):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['DSWX_BENCH_SHARE_DEVICE'] = '1'
    # a child process (never exec from a process that touched the GPU); the parent bench process itself starts the
    # ranks with torch.distributed.run before it imports torch
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), capture_output=True, text=True,
                         timeout=1500, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith('{"metric"')][-1])


def test_two_ranks_strong_scaling_every_rank_checks_every_chunk():
    out = _bench('--gpus', '2', '--total-tiles', '16', '--tiles', '4', '--steps', '2', '--warmup', '1',
                 '--no-cpu-baseline', '--distinct-chunks')
    assert out['n_gpus'] == 2 and out['scaling'] == 'strong'
    assert out['config']['tiles_per_step_all_ranks'] == 16 and out['config']['launches_per_step'] == 2
    assert out['config']['control_plane'] == 'gloo'
    par = out['parity_check']
    assert par['result'] == 'bit-exact', par
    r0, r1 = par['ranks']
    assert (r0['rank'], r0['first_tile'], r1['rank'], r1['first_tile']) == (0, 0, 1, 8)
    # resident chunk: first / middle / last; then the second chunk generated with ITS indices: first / last
    assert r0['tiles'] == [0, 2, 3, 4, 7] and r1['tiles'] == [8, 10, 11, 12, 15]
    assert r0['distinct_chunks'] == 2 and r1['distinct_chunks'] == 2
    assert out['value'] > 0 and out['roofline']['pixels_per_launch'] == 4 * 3660 * 3660


def test_chain_mode_shadow_and_land_layers_into_the_batch():
    """`bench.py --chain` (BASELINE configs[4]'s per-pixel chain, one GPU's share): terrain shadow layer and LAND
    aggregation written straight into the SHAD / LAND planes of the resident batch (dswx_shadow_layer_batch,
    dswx_landcover_mask_batch with the batch's tile stride), then the classifier with SHAD + LAND + OCEAN.  The line's
    parity record compares SHAD and LAND with the numpy oracle's layers and everything downstream with the C oracle."""
    out = _bench('--chain', '--tiles', '3', '--steps', '2', '--warmup', '1', '--no-cpu-baseline')
    assert out['n_gpus'] == 1 and 'configs[4]' in out['config']['workload'] and out['config']['planes_in'] == 10
    assert out['parity_check']['result'] == 'bit-exact', out['parity_check']
    assert out['parity_check']['ranks'][0]['tiles'] == [0, 2]
    r = out['roofline']
    assert abs(r['algorithmic_bytes_per_pixel'] - (24 + 11 + 4 * 3760 ** 2 / 3660 ** 2 + 1)) < 1e-3
    assert all(r['chain'][k] > 0 for k in ('terrain_shadow_ms', 'land_aggregation_ms', 'classify_ms'))


def test_two_ranks_weak_scaling_rank_offsets():
    out = _bench('--gpus', '2', '--tiles', '3', '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--masks')
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['config']['tiles_per_step_all_ranks'] == 6
    par = out['parity_check']
    assert par['result'] == 'bit-exact', par
    assert [r['first_tile'] for r in par['ranks']] == [0, 3]
    assert [r['tiles'] for r in par['ranks']] == [[0, 1, 2], [3, 4, 5]]
