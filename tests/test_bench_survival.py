"""the driver's one-shot `bench.py --gpus N` line must be impossible to lose.
  (a) the RCCL control plane falls back to gloo LOUDLY (config.control_plane says why, rccl_ranks = 0); --require-rccl is strict
  (b) a rank's exception becomes that rank's record: the line is still printed, `value: null` for the case it failed in,
      the other case intact, exit code non-zero AFTER the line
  (c) the preflight record (devices, free HBM, control-plane round trip) is in every N > 1 line
All on CPU: two ranks under torch.distributed.run over gloo.  The ControlPlane tests run the real class; the bench tests
run bench.py's own control flow with the device replaced by tests/helpers/fake_bench_rank.py (no GPU work is faked INTO
the product: the stand-ins live under tests/).  The GPU counterpart (two ranks on one device, real kernels) is in
tests/test_gpu_multirank.py."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_RANK = os.path.join(ROOT, 'tests', 'helpers', 'fake_bench_rank.py')


def _torchrun(script, *argv, nproc=2, env=None, timeout=300):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'DSWX_BENCH_INJECT'):
        e.pop(k, None)
    e.update(env or {})
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)] + list(argv)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)


def _line(res, key='{"metric"'):
    lines = [l for l in res.stdout.splitlines() if l.startswith(key)]
    assert len(lines) == 1, (res.stdout[-2000:], res.stderr[-3000:])
    return json.loads(lines[0])


# ------------------------------------------------------------------ (a) the control plane itself

CP_WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from proteus_amd import shard
    mode = sys.argv[1]
    rank = shard.env_rank()[0]

    def pretend_rccl(self):
        """An 'RCCL' group made of gloo (every rank creates it: new_group is collective), then the verdict of the mode."""
        group = dist.new_group(backend='gloo')
        if mode == 'rank1_fails' and rank == 1:
            return 'RuntimeError: injected: ncclCommInitRank failed on this rank'
        self.fast = group
        return None

    if not mode.startswith('real'):
        shard.ControlPlane._bring_up_rccl = pretend_rccl
    if mode.startswith('late_failure'):
        real_all_reduce = dist.all_reduce
        state = {'calls': 0}
        def flaky(t, op=None, group=None, **kw):
            if group is not None and rank == 1:
                state['calls'] += 1
                if state['calls'] == 2:
                    if mode == 'late_failure_after':        # the collective ran, the error surfaced afterwards
                        real_all_reduce(t, op=op, group=group, **kw)
                    # else: rank 1 never enters the collective and rank 0 waits in it until the time limit
                    raise RuntimeError('injected: ncclAllReduce failed')
            return real_all_reduce(t, op=op, group=group, **kw)
        dist.all_reduce = flaky
    if mode.startswith('barrier_failure'):
        real_barrier = dist.barrier
        bstate = {'calls': 0}
        def flaky_barrier(group=None, **kw):
            if group is not None and rank == 1:
                bstate['calls'] += 1
                if bstate['calls'] == 2:                    # the barrier that CLOSES the timed region
                    if mode == 'barrier_failure_after':
                        real_barrier(group=group, **kw)
                    raise RuntimeError('injected: ncclBarrier failed')
            return real_barrier(group=group, **kw)
        dist.barrier = flaky_barrier
    try:
        cp = shard.ControlPlane(backend='nccl', device=None, require=(mode == 'real_strict'))
    except Exception as e:
        sys.stdout.write(f'RAISED {type(e).__name__} {str(e)[:200]}\\n'); sys.stdout.flush()
        raise SystemExit(3)
    seen = [cp.backend]
    degraded = [cp.barrier()]
    a = cp.max_over_ranks(1.0 + rank)
    seen.append(cp.backend)
    b = cp.sum_over_ranks(2 ** 40 + rank)
    seen.append(cp.backend)
    c = cp.max_over_ranks(10.0 - rank)
    objs = cp.gather_objects({'rank': rank})
    degraded.append(cp.barrier())
    d = cp.max_over_ranks(3.0 + rank)           # the ranks are still in step after the barrier: values right on every rank
    degraded_all = cp.gather_objects(degraded)
    if rank == 0:
        print(json.dumps({'seen': seen, 'a': a, 'b': b, 'c': c, 'd': d, 'objs': objs, 'final': cp.backend, 'rccl_ranks': cp.rccl_ranks,
                          'hung': cp.hung, 'degraded': degraded_all}), flush=True)
    hung = cp.hung
    cp.close()
    if hung:                # a helper thread is still inside the abandoned collective (bench.py leaves the same way)
        import os
        sys.stdout.flush()
        os._exit(0)
''') % ROOT


@pytest.fixture()
def cp_worker(tmp_path):
    path = tmp_path / 'cp_worker.py'
    path.write_text(CP_WORKER)
    return path


def test_rccl_that_cannot_come_up_falls_back_to_gloo_loudly(cp_worker):
    """No GPU here: the real RCCL bring-up fails on both ranks.  Default: gloo, the backend string says why, rccl_ranks 0,
    stderr carries the notice; strict (bench.py --require-rccl): every rank raises."""
    res = _torchrun(cp_worker, 'real')
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['final'].startswith('gloo (fallback: nccl bring-up failed on rank(s) [0, 1]:') and out['rccl_ranks'] == 0
    assert (out['a'], out['b'], out['c']) == (2.0, 2 ** 41 + 1, 10.0) and out['objs'] == [{'rank': 0}, {'rank': 1}]
    assert '[dswx control plane] gloo (fallback' in res.stderr
    res = _torchrun(cp_worker, 'real_strict')
    assert res.returncode != 0 and 'RAISED RuntimeError' in res.stdout and '{"seen"' not in res.stdout


def test_one_rank_failing_the_bring_up_moves_every_rank_to_gloo(cp_worker):
    """The asymmetric case a plain try/except around init_process_group cannot handle: rank 0's RCCL is fine, rank 1's is
    not.  The verdicts are exchanged over gloo and BOTH ranks end on gloo (a split would hang the first reduction)."""
    res = _torchrun(cp_worker, 'rank1_fails')
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['final'].startswith('gloo (fallback: nccl bring-up failed on rank(s) [1]: RuntimeError: injected')
    assert out['rccl_ranks'] == 0 and (out['a'], out['b'], out['c']) == (2.0, 2 ** 41 + 1, 10.0)


def test_all_ranks_good_use_the_fast_group_and_a_late_failure_still_agrees(cp_worker):
    res = _torchrun(cp_worker, 'all_good')
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['final'] == 'nccl' and out['rccl_ranks'] == 2 and out['seen'] == ['nccl'] * 3
    assert (out['a'], out['b'], out['c']) == (2.0, 2 ** 41 + 1, 10.0)
    # rank 1's SECOND reduction over the fast group raises: the agreement after it moves both ranks to gloo, the
    # reduction is repeated there, its value is right, and so is the next one
    res = _torchrun(cp_worker, 'late_failure_after')
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['seen'][:2] == ['nccl', 'nccl'] and out['seen'][2].startswith('gloo (fallback: nccl collective failed on rank(s) [1]')
    assert (out['a'], out['b'], out['c']) == (2.0, 2 ** 41 + 1, 10.0) and out['rccl_ranks'] == 0 and not out['hung']
    # ... and when rank 1 raises BEFORE entering the collective, rank 0 is alone in it: its call is abandoned after the
    # time limit (here 4 s), it reports that, and both ranks go on over gloo with the right values
    res = _torchrun(cp_worker, 'late_failure_before', env={'DSWX_RCCL_PROBE_TIMEOUT_S': '4'})
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['seen'][2].startswith('gloo (fallback: nccl collective failed on rank(s) [0, 1]: RCCL all_reduce SUM did not return within 4 s')
    assert (out['a'], out['b'], out['c']) == (2.0, 2 ** 41 + 1, 10.0) and out['rccl_ranks'] == 0 and out['hung']


def test_a_barrier_that_fails_on_one_rank_leaves_the_ranks_in_step(cp_worker):
    """the RCCL barrier raises on rank 1 and completes on rank 0 (or rank 0 waits in it alone until
    the time limit).  Every barrier over RCCL is followed by the one-integer agreement over gloo on EVERY rank, so both
    ranks report the barrier as degraded, both move to gloo, and the reductions after it carry the right values -- the
    ranks are not one gloo collective apart."""
    res = _torchrun(cp_worker, 'all_good')
    out = _line(res, '{"seen"')
    assert out['degraded'] == [[False, False], [False, False]] and out['d'] == 4.0
    res = _torchrun(cp_worker, 'barrier_failure_after')
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['degraded'] == [[False, True], [False, True]], out
    assert out['seen'] == ['nccl'] * 3 and out['final'].startswith('gloo (fallback: nccl collective failed on rank(s) [1]: RuntimeError: injected: ncclBarrier')
    assert (out['a'], out['b'], out['c'], out['d']) == (2.0, 2 ** 41 + 1, 10.0, 4.0) and not out['hung']
    res = _torchrun(cp_worker, 'barrier_failure_before', env={'DSWX_RCCL_PROBE_TIMEOUT_S': '4'})
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res, '{"seen"')
    assert out['degraded'] == [[False, True], [False, True]], out
    assert 'RCCL barrier did not return within 4 s' in out['final'] and out['hung']
    assert (out['a'], out['b'], out['c'], out['d']) == (2.0, 2 ** 41 + 1, 10.0, 4.0)


def test_world_above_one_without_master_port_fails_at_once(tmp_path):
    """the private rendezvous is for the forced world of ONE only; a launcher that sets RANK / WORLD_SIZE
    but not MASTER_PORT must get env://'s immediate error, not N ranks each waiting on a port of its own."""
    script = tmp_path / 'w.py'
    script.write_text(f"import sys\nsys.path.insert(0, {ROOT!r})\nfrom proteus_amd import shard\nshard.ControlPlane(backend='gloo')\n")
    e = {k: v for k, v in os.environ.items() if k not in ('MASTER_PORT', 'MASTER_ADDR')}
    e.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='2')
    res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=e)
    assert res.returncode != 0 and 'MASTER_PORT' in res.stderr


# ------------------------------------------------------------------ (b) + (c): bench.py's own control flow, two ranks

TOY = ['--gpus', '2', '--plain-tiles', '3', '--strong-total', '16', '--strong-chunk', '4', '--steps', '3', '--warmup', '1',
       '--no-host-path']


def test_healthy_two_rank_line_has_both_records_the_fallback_notice_and_the_preflight():
    res = _torchrun(FAKE_RANK, *TOY)
    assert res.returncode == 0, res.stderr[-3000:]
    out = _line(res)
    assert out['value'] > 0 and out['strong']['value'] > 0 and 'error' not in out
    assert out['rccl_ranks'] == 0 and out['config']['control_plane'].startswith('gloo (fallback: nccl bring-up failed')
    assert out['n_gpus'] == 2 and [r['rank'] for r in out['ranks']] == [0, 1]
    pre = out['preflight']
    assert pre['ok'] and pre['distinct_devices'] == 2 and [r['rank'] for r in pre['ranks']] == [0, 1]
    for r in pre['ranks']:
        assert r['hbm_free_GiB'] == 280.0 and r['control_plane_round_trip_ms'] > 0 and r['slide_slack_GiB'] == 48.0
    assert '[bench partial]' in res.stderr


@pytest.mark.parametrize('phase', ['place', 'warm-up', 'timed region'])
def test_a_rank_failing_in_the_first_case_leaves_the_second_intact(phase):
    res = _torchrun(FAKE_RANK, *TOY, env={'DSWX_BENCH_INJECT': f'1:0:{phase}'})
    assert res.returncode != 0                      # non-zero, but AFTER the line:
    out = _line(res)
    assert out['value'] is None and out['ms_per_step'] is None and out['failed_ranks'] == [1]
    assert f'rank 1 failed in {phase}: RuntimeError: injected failure' in out['error']
    r0, r1 = out['ranks']
    assert 'error' not in r0 and r0['frac'] > 0 and r1['rank'] == 1 and r1['phase'] == phase and 'injected' in r1['error']
    assert out['roofline'] is not None and out['slowest_rank']['rank'] == 0         # rank 0's own figures survive
    assert out['parity_check']['ranks'][1]['result'].startswith('not checked (the rank failed in')
    st = out['strong']                              # the case after the failure: measured by BOTH ranks
    assert st['value'] > 0 and 'error' not in st and [r['rank'] for r in st['ranks']] == [0, 1]
    assert st['parity_check']['result'] == 'bit-exact'


def test_rank_zero_failing_in_the_second_case_still_prints_the_line():
    res = _torchrun(FAKE_RANK, *TOY, env={'DSWX_BENCH_INJECT': '0:1:place'})
    assert res.returncode != 0
    out = _line(res)
    assert out['value'] > 0 and 'error' not in out and out['parity_check']['result'] == 'bit-exact'
    st = out['strong']
    assert st['value'] is None and st['failed_ranks'] == [0] and st['roofline'] is None
    assert 'error' in st['ranks'][0] and st['ranks'][1]['frac'] > 0 and st['slowest_rank']['rank'] == 1


def test_a_rank_without_a_working_device_is_a_record_not_a_hang():
    res = _torchrun(FAKE_RANK, *TOY, env={'FAKE_BOOT_ERROR_RANK': '1'})
    assert res.returncode != 0
    out = _line(res)
    assert out['value'] is None and out['strong']['value'] is None
    assert 'rank 1 failed in library context' in out['error'] and not out['preflight']['ok']
    assert 'error' in out['preflight']['ranks'][1] and 'error' not in out['preflight']['ranks'][0]


@pytest.mark.parametrize('case_no', [2, 1])
def test_a_rank_that_dies_hard_still_leaves_rank_zero_its_last_words(case_no):
    """What RankGuard cannot catch: rank 1 ends with os._exit in the middle of a case (as a GPU fault or a signal would end
    it).  torchrun then terminates rank 0, which is waiting in a collective that will never complete; its wake-up-pipe
    thread prints the line as far as it got.  Dying in the SECOND case leaves the first (the weak record, the point of the
    scaling curve) intact in the line; dying in the first leaves a line with `value: null` and the reason."""
    res = _torchrun(FAKE_RANK, *TOY, env={'FAKE_HARD_EXIT': f'1:{case_no}'}, timeout=180)
    assert res.returncode != 0
    out = _line(res)
    if case_no == 2:
        assert out['value'] > 0 and 'error' not in out and out['parity_check']['result'] == 'bit-exact'
        assert out['strong']['value'] is None and 'terminated while case 1 (strong) was running' in out['strong']['error']
    else:
        assert out['value'] is None and 'terminated while case 0 was running' in out['error']


def test_preflight_alone_and_a_rank_short_of_hbm_shrinks_its_slack():
    res = _torchrun(FAKE_RANK, '--gpus', '2', '--preflight')
    assert res.returncode == 0, res.stderr[-3000:]
    pre = _line(res, '{"preflight"')['preflight']
    # the plain command: the largest resident chunk is the strong case's 512 tiles = 134.1 GiB + 48 GiB of slack
    assert pre['ok'] and [r['resident_chunk_GiB'] for r in pre['ranks']] == [134.14, 134.14]
    res = _torchrun(FAKE_RANK, '--gpus', '2', '--preflight', env={'FAKE_FREE_HBM_GIB': '160'})
    pre = _line(res, '{"preflight"')['preflight']
    assert pre['ok'] and all(r['slide_slack_GiB'] == 19.0 and '48 -> 19' in r['adjusted'] for r in pre['ranks'])
    res = _torchrun(FAKE_RANK, '--gpus', '2', '--preflight', env={'FAKE_FREE_HBM_GIB': '100'})
    assert res.returncode != 0
    pre = _line(res, '{"preflight"')['preflight']
    assert not pre['ok'] and all('likely to fail' in r['warning'] for r in pre['ranks'])


def test_require_rccl_is_strict():
    """--require-rccl: no measurement without RCCL.  Every rank raises; rank 0's last words are a line with no number and
    the reason."""
    res = _torchrun(FAKE_RANK, *TOY, '--require-rccl')
    assert res.returncode != 0 and 'strict control plane' in res.stderr
    out = _line(res)
    assert out['value'] is None and 'strict control plane' in out['error'] and 'strong' not in out
