import sys, json
sys.path.insert(0, '.')
from proteus_amd import _capi
from proteus_amd.synth import SEED
n, h, w = 32, 3660, 3660
P = h * w
ctx = _capi.Context(0)
p = _capi.default_params()
def timed(fn, reps=10):
    for _ in range(30): fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps): fn()
    ctx.record(e1); ctx.synchronize()
    return ctx.elapsed_ms(e0, e1) / reps
out = {}
for name, in_off, out_off, stride in (('all aligned (16 B, not 256)', 16, 16, P), ('inputs odd, outputs 16-B aligned', 2, 16, P), ('inputs 16-B aligned, outputs odd', 16, 3, P),
                                      ('all odd', 2, 3, P), ('all odd, odd stride', 2, 3, P + 3)):
    arena = ctx.malloc(n * stride * 21 + (1 << 20))
    pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
    off = 0
    def place(nbytes, a):
        global off
        off = (off + 255) & ~255
        off += a
        r = arena.ptr + off
        off += nbytes
        return r
    for k in range(6):
        pin.band[k] = place(n * stride * 2, in_off)
    pin.fmask = place(n * stride, in_off + (1 if in_off == 2 else 0))
    pout.diag = place(n * stride * 2, out_off + (out_off & 1))
    for nm in ('wtr1', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud'):
        setattr(pout, nm, place(n * stride, out_off))
    cnt = place(n * 24, 0)
    geom = _capi.BatchGeom(n, h, w, stride)
    ctx.synth_batch(SEED, 0, geom, pin)
    ms = timed(lambda: ctx.classify_batch(p, geom, pin, pout, cnt))
    out[name] = {'ms': round(ms, 4), 'frac': round(n * P * 21 / ms / 1e6 / 8000, 4), 'kernel': ctx.last_kernel_info().split(' grid')[0]}
    arena.free()
print(json.dumps(out, indent=1))
