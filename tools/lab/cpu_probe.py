import os, time, json, numpy as np, sys
sys.path.insert(0, '.')
from proteus_amd import codec
out = {'nproc': os.cpu_count(), 'affinity': len(os.sched_getaffinity(0))}
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: out[f] = open(f).read().strip()
    except Exception as e: out[f] = str(e)[:60]
rng = np.random.default_rng(0)
a = rng.integers(0, 5, size=(512, 512, 512)).astype(np.uint8)
codec.deflate_uniform(a[:8], 512*512, 6, 8)
for th in (1, 8, 16, 32, 64, 128, 256):
    n = min(512, max(32, th * 4))
    t = time.perf_counter(); c0 = time.process_time()
    codec.deflate_uniform(a[:n], 512*512, 6, th)
    dt = time.perf_counter() - t
    out[f'threads_{th}'] = {'blocks': n, 'wall_s': round(dt, 3), 'cpu_s': round(time.process_time() - c0, 3), 'blocks_per_s': round(n / dt, 1)}
print(json.dumps(out, indent=1))
