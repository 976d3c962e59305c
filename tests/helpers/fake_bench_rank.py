"""A bench.py rank WITHOUT a GPU (CPU tests of the N > 1 control flow, tests/test_bench_survival.py): bench.bring_up
and bench.place_batch are replaced by stand-ins that hold no device memory and launch nothing, everything else --
the real proteus_amd.shard.ControlPlane (gloo + the RCCL probe, which cannot pass here), the case list, RankGuard,
the preflight, the gathers, the assembly of the line, the exit code -- is bench.py's own code.  The numbers such a run
prints mean nothing; the STRUCTURE of the line and who survives what is the subject.  Test scaffolding: nothing in the
product or in bench.py imports this file."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench                                                # noqa: E402


class FakeCtx:
    def __init__(self):
        self.events = 0

    def event(self):
        self.events += 1
        return self.events

    def record(self, event, stream=None):
        pass

    def elapsed_ms(self, a, b):
        return 1.0

    def destroy_event(self, event):
        self.events -= 1

    def synchronize(self, stream=None):
        pass

    def last_kernel_info(self):
        return 'fake kernel (tests/helpers/fake_bench_rank.py)'

    def close(self):
        assert self.events == 0, 'events leaked'


class FakeBatch:
    masks = False
    tile_stride = 13395712
    live = 0

    def __init__(self):
        FakeBatch.live += 1
        self.launches = []

    def classify(self, params, n_tiles=None, **kw):
        self.launches.append(n_tiles)

    def free(self):
        FakeBatch.live -= 1


PLACED = [0]


def fake_place_batch(ctx, params, n_tiles, tile0, masks, how, trials, refine=1, slack_gib=48.0):
    # FAKE_HARD_EXIT='<rank>:<n>': that rank dies HARD (no exception, no clean-up) in its n-th placement, i.e. case n - 1
    spec = os.environ.get('FAKE_HARD_EXIT')
    PLACED[0] += 1
    if spec:
        r, n = spec.split(':')
        if int(r) == int(os.environ.get('RANK', '0')) and int(n) == PLACED[0]:
            os._exit(7)
    return FakeBatch(), {'how': how, 'probes': 0, 'seconds': 0.0, 'slack_gib_used': slack_gib}


def fake_rank_parity(ctx, batch, params, rank, tile0, n_tiles, chunks, distinct):
    return {'rank': rank, 'first_tile': tile0, 'tiles': [tile0], 'result': 'bit-exact'}


def fake_bring_up(args, rank, local_rank, world):
    from proteus_amd import shard
    free_gib = float(os.environ.get('FAKE_FREE_HBM_GIB', '280'))
    boot = None
    if os.environ.get('FAKE_BOOT_ERROR_RANK') == str(rank):
        boot = {'phase': 'library context', 'error': 'DswxError: injected: no device'}
    cp = shard.ControlPlane(backend='nccl', device=None, require=args.require_rccl)
    return argparse.Namespace(ctx=FakeCtx(), cp=cp, rank=rank, world=world, params=None, share_device=False,
                              boot_error=boot, slack_gib=args.slide_slack_gib,
                              device_id=f'fakehost/0000:{rank:02x}:00/uuid-{rank}',
                              device_synchronize=lambda: None,
                              mem_info=lambda: (int(free_gib * 2 ** 30), 288 << 30))


bench.bring_up = fake_bring_up
bench.place_batch = fake_place_batch
bench.rank_parity = fake_rank_parity

if __name__ == '__main__':
    code = bench.main()
    assert FakeBatch.live == 0, 'a batch was left allocated'
    sys.exit(code)
