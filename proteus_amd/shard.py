"""Tile sharding across the GPUs of one node.

Every pixel -- hence every tile -- is independent in the 'mask' / 'ignore' modes, so
the batch is split statically and contiguously over ranks and no data-path
collective exists.  torch.distributed (gloo always, RCCL beside it on the GPU box)
is used only as the control plane: a barrier around the timed region and a MAX
over ranks of the elapsed time.
"""
import os


def tile_range(n_tiles, rank, world):
    """Contiguous static split: rank r owns tiles [lo, hi).  Sizes differ by at
    most one; the union over ranks is exactly range(n_tiles)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f'bad rank {rank} of {world}')
    if n_tiles < 0:
        raise ValueError('n_tiles must be >= 0')
    lo = (n_tiles * rank) // world
    hi = (n_tiles * (rank + 1)) // world
    return lo, hi


def weak_tile_range(tiles_per_rank, rank):
    """Weak scaling (bench.py): every rank owns `tiles_per_rank` tiles."""
    return rank * tiles_per_rank, (rank + 1) * tiles_per_rank


def chunk_sizes(n_tiles, chunk):
    """A rank's share walked in resident chunks: [chunk, chunk, ..., remainder]; sums to n_tiles."""
    if chunk <= 0:
        raise ValueError('chunk must be positive')
    full, rest = divmod(max(n_tiles, 0), chunk)
    return [chunk] * full + ([rest] if rest else [])


class _stdout_to_stderr:
    """gloo and RCCL print banners ('[Gloo] Rank 0 is connected to ...', the RCCL version block) through C stdio on fd 1;
    bench.py's contract is ONE JSON line on stdout.  While a process group comes up, fd 1 points at fd 2."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        import sys
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)          # C stdio is block-buffered when stdout is a pipe: empty it into fd 2
        except Exception:                           # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def env_rank():
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


class ControlPlane:
    """Barrier + max-over-ranks; a no-op object for world == 1 -- unless DSWX_FORCE_DIST=1 asks for a process group of
    ONE rank: the same bring-up, barrier, all_reduce on device tensors, all_gather_object and destroy an N > 1 run goes
    through, on a box with a single GPU (tests/test_gpu_multirank.py).

    Bring-up.  The default process group is ALWAYS gloo over the launcher's TCP rendezvous: it
    carries the pickled per-rank records and is the channel on which the ranks AGREE about everything else.  With
    backend='nccl' an RCCL group is created beside it and probed (barrier + all_reduce of a device tensor, inside a
    helper thread with a time limit); the ranks then take the MIN of their verdicts over gloo.  All ranks good: the
    barriers and the MAX / SUM reductions run over RCCL on device tensors.  Any rank bad: EVERY rank uses gloo, loudly --
    `backend` becomes 'gloo (fallback: nccl ... failed on rank(s) [..]: <first error>)' and `rccl_ranks` is 0 -- unless
    `require=True` (bench.py --require-rccl), which raises on every rank.  RCCL is not on the data path (tiles are
    independent), so the fallback loses nothing of the measurement.  A reduction that fails over RCCL later in the run
    is caught the same way: every reduction AND every barrier over RCCL is followed by a one-integer agreement over gloo,
    and one bad rank moves all ranks to gloo, where a reduction is repeated (the agreement is itself a barrier)."""

    PROBE_TIMEOUT_S = 120.0

    def __init__(self, backend=None, device=None, allow_fallback=True, require=False):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        self.backend = None
        self.rccl_ranks = 0
        self.fast = None                # the RCCL group, when every rank's probe passed
        self.hung = False               # an RCCL call never returned: leave with os._exit, never through destroy
        if not (self.world > 1 or os.environ.get('DSWX_FORCE_DIST') == '1'):
            return
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ['MASTER_ADDR'] in ('127.0.0.1', 'localhost', '::1'):
            # one node, rendezvous on loopback: gloo otherwise picks its interface by resolving the host name, which a
            # container's may not do ("Unable to resolve hostname to a (local) address")
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        kw = {}
        if self.world == 1 and 'MASTER_PORT' not in os.environ:
            # a forced world of one outside torchrun: a private rendezvous.  A world > 1 keeps env:// so that a launcher
            # which forgot MASTER_PORT fails at once instead of every rank waiting on its own port
            import socket
            with socket.socket() as sock:
                sock.bind(('127.0.0.1', 0))
                port = sock.getsockname()[1]
            kw.update(init_method=f'tcp://127.0.0.1:{port}', rank=self.rank, world_size=self.world)
        with _stdout_to_stderr():
            if not dist.is_initialized():
                dist.init_process_group('gloo', **kw)
            self.dist = dist
            self.backend = 'gloo'
            err = self._bring_up_rccl() if backend == 'nccl' else None
        if backend == 'nccl':
            verdicts = self.gather_objects(err)
            bad = [(r, e) for r, e in enumerate(verdicts) if e]
            if not bad:
                self.backend, self.rccl_ranks = 'nccl', self.world
            else:
                self.fast = None
                why = f'nccl bring-up failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}'
                if require or not allow_fallback:
                    raise RuntimeError(why + ' (and the strict control plane was asked for)')
                self._fall_back(why)

    # ---- RCCL beside gloo
    def _timed_call(self, fn, what):
        """(result, None) or (None, why): `fn` in a helper thread with a time limit, so that an RCCL call which never
        returns (a peer that raised before entering the collective, a wedged link) becomes an error on this rank after
        DSWX_RCCL_PROBE_TIMEOUT_S instead of the end of the run.  After a timeout the thread is abandoned inside RCCL
        (`hung`: the process must leave through os._exit) and the RCCL group is not used again."""
        import threading
        box = {}

        def body():
            try:
                if self.device is not None:         # the current device is per THREAD: this one starts on device 0
                    import torch
                    torch.cuda.set_device(self.device)
                box['result'] = fn()
            except BaseException as e:                      # noqa: BLE001
                box['error'] = f'{type(e).__name__}: {e}'[:300]

        limit = float(os.environ.get('DSWX_RCCL_PROBE_TIMEOUT_S', self.PROBE_TIMEOUT_S))
        th = threading.Thread(target=body, daemon=True, name='dswx-rccl-call')
        th.start()
        th.join(limit)
        if th.is_alive():
            self.hung = True
            self.fast = None
            return None, f'RCCL {what} did not return within {limit:.0f} s'
        if 'error' in box:
            return None, box['error']
        return box.get('result'), None

    def _bring_up_rccl(self):
        """None, or why RCCL is not usable on THIS rank: group creation, a barrier and an all_reduce whose value is
        checked."""
        def probe():
            import torch
            dist = self.dist
            kw = {'device_id': self.device} if self.device is not None else {}
            group = dist.new_group(backend='nccl', **kw)
            dev = self.device if self.device is not None else torch.device('cuda', self.local_rank)
            dist.barrier(group=group, device_ids=[dev.index if dev.index is not None else 0])
            t = torch.full((1,), float(self.rank), dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            if float(t.item()) != float(self.world - 1):
                raise RuntimeError(f'all_reduce MAX of the ranks gave {float(t.item())}, not {self.world - 1}')
            return group

        group, err = self._timed_call(probe, 'bring-up')
        if err is None:
            self.fast = group
        return err

    def _fall_back(self, why):
        import sys
        self.fast = None
        self.rccl_ranks = 0
        self.backend = f'gloo (fallback: {why})'[:400]
        if self.rank == 0:
            print(f'[dswx control plane] {self.backend}', file=sys.stderr, flush=True)

    def _agree(self, my_error):
        """After a reduction over RCCL: did it work on EVERY rank?  One int64 MIN over gloo; if not, all ranks move to
        gloo (the caller repeats its reduction there)."""
        import torch
        err = my_error
        ok = torch.tensor([0 if err else 1], dtype=torch.int64)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            return True
        errs = self.gather_objects(err)
        bad = [(r, e) for r, e in enumerate(errs) if e]
        self._fall_back(f'nccl collective failed on rank(s) {[r for r, _ in bad]}: {bad[0][1]}')
        return False

    def _reduce(self, value, dtype, op_name):
        import torch
        op = getattr(self.dist.ReduceOp, op_name)
        if self.fast is not None:           # (the same on every rank: only an agreement clears it)
            group = self.fast

            def over_rccl():
                t = torch.tensor([value], dtype=dtype, device=self.device)
                self.dist.all_reduce(t, op=op, group=group)
                return t.item()
            out, err = self._timed_call(over_rccl, f'all_reduce {op_name}')
            if self._agree(err):
                return out
        t = torch.tensor([value], dtype=dtype)
        self.dist.all_reduce(t, op=op)
        return t.item()

    # ---- what bench.py calls
    def barrier(self):
        """All ranks meet.  Returns True when the control plane DEGRADED during this barrier (an RCCL barrier failed or
        did not return on some rank: every rank is on gloo afterwards) -- a timed region that such a barrier closes has
        waited for the time limit of the failing call and is not a measurement (bench.py: value null, with the reason).

        ADVICE r05: while RCCL is in use, the RCCL barrier is followed on EVERY rank by the one-integer agreement over
        gloo that the reductions use (itself a barrier), so the ranks leave with the same view of the control plane and
        the same number of gloo collectives behind them -- a rank whose RCCL barrier raised no longer runs one gloo
        barrier more than a rank whose barrier completed.  Cost: one gloo all_reduce of 8 bytes per barrier."""
        if self.dist is None:
            return False
        if self.fast is None:
            self.dist.barrier()
            return False
        group = self.fast
        idx = self.device.index if self.device is not None and self.device.index is not None else 0
        _, err = self._timed_call(lambda: self.dist.barrier(group=group, device_ids=[idx]), 'barrier')
        return not self._agree(err)

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        return float(self._reduce(float(value), torch.float64, 'MAX'))

    def sum_over_ranks(self, value):
        """Integer sum over ranks (e.g. tiles processed per step by the whole job)."""
        if self.dist is None:
            return int(value)
        import torch
        return int(self._reduce(int(value), torch.int64, 'SUM'))

    def gather_objects(self, obj):
        """Every rank's small Python object, in rank order, on every rank (pickled, over gloo)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def gather_counters(self, local):
        """All ranks' per-tile counters ([t_r, 3] int64 each) on every rank, in tile order.
        (Host-side metadata only; 24 bytes per tile.)"""
        if self.dist is None:
            return local
        import numpy as np
        return np.concatenate(self.gather_objects(local), axis=0)

    def close(self):
        if self.dist is not None:
            self.barrier()              # (over gloo on a rank whose RCCL group is gone: the others wait here too)
            if self.hung:               # a thread is still inside RCCL: destroy_process_group would wait for it
                self.dist = None
                return
            self.dist.destroy_process_group()
            self.dist = None
            self.fast = None
