"""Tile sharding across the GPUs of one node.

Every pixel -- hence every tile -- is independent in the 'mask' / 'ignore' modes, so
the batch is split statically and contiguously over ranks and no data-path
collective exists.  torch.distributed (RCCL on the GPU box, gloo in the CPU tests)
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


def env_rank():
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


class ControlPlane:
    """Barrier + max-over-ranks; a no-op object for world == 1 -- unless DSWX_FORCE_DIST=1 asks for a process group of
    ONE rank: the same init (RCCL with device_id), barrier, all_reduce on device tensors, all_gather_object and destroy
    an N > 1 run goes through, on a box with a single GPU (tests/test_gpu_multirank.py)."""

    def __init__(self, backend=None, device=None, allow_fallback=False):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        self.backend = None
        if self.world > 1 or os.environ.get('DSWX_FORCE_DIST') == '1':
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            kw = {}
            if 'MASTER_PORT' not in os.environ:            # a forced world of one outside torchrun: a private rendezvous
                import socket
                with socket.socket() as sock:
                    sock.bind(('127.0.0.1', 0))
                    port = sock.getsockname()[1]
                kw.update(init_method=f'tcp://127.0.0.1:{port}', rank=self.rank, world_size=self.world)
            dev_kw = {'device_id': device} if backend == 'nccl' and device is not None else {}
            if not dist.is_initialized():
                try:
                    dist.init_process_group(backend, **kw, **dev_kw)
                    self.backend = backend
                except Exception as e:                      # noqa: BLE001
                    # RCCL could not come up.  A measurement must not quietly change its control plane:
                    # fail unless the caller opted in (bench.py --allow-gloo).  With the opt-in, gloo
                    # carries the same barrier and MAX of one double (there is no data-path collective
                    # to lose) and the backend string records what happened.
                    if backend == 'gloo' or not allow_fallback:
                        raise
                    if dist.is_initialized():
                        dist.destroy_process_group()
                    dist.init_process_group('gloo', **kw)
                    self.backend = f'gloo (fallback: {backend} init failed: {str(e)[:120]})'
                    self.device = None
            else:
                self.backend = dist.get_backend()
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64,
                         device=self.device if self.device is not None else 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        """Integer sum over ranks (e.g. tiles processed per step by the whole job)."""
        if self.dist is None:
            return int(value)
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64,
                         device=self.device if self.device is not None else 'cpu')
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def gather_objects(self, obj):
        """Every rank's small Python object, in rank order, on every rank."""
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
        import torch
        out = [None] * self.world
        self.dist.all_gather_object(out, local)
        import numpy as np
        return np.concatenate(out, axis=0)

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
