"""Stage clock of a product run: where the wall time of `dswx_hls.py <runconfig>` goes.

The reference spends a product run in GDAL (read + inflate, `save_as_cog`: overviews, DEFLATE, file output;
core.py:7-91) and in numpy; here the per-pixel chain is milliseconds on the GPU and the host codec is what a
user waits for.  The stages run on several threads at once, so every span is kept
with its start and end: the report gives, per stage, the number of spans, the THREAD seconds (sum of the
spans: the work) and the WALL seconds (length of the union of the spans: what the stage occupied of the
clock), plus the wall time between the first start and the last end.

Off unless start() was called (tools/e2e_time.py, the batch worker with --stages): span() is then a no-op
that costs one attribute test.
"""
import threading
import time
from contextlib import contextmanager

_lock = threading.Lock()
_spans = None          # list of (stage, t0, t1) when recording


def start():
    global _spans
    with _lock:
        _spans = []


def stop():
    """Stops recording and returns the report (see report())."""
    global _spans
    with _lock:
        spans, _spans = _spans, None
    return report(spans or [])


def recording():
    return _spans is not None


@contextmanager
def span(stage):
    if _spans is None:
        yield
        return
    t0 = time.perf_counter()
    try:
        yield
    finally:
        t1 = time.perf_counter()
        with _lock:
            if _spans is not None:
                _spans.append((stage, t0, t1))


def add(stage, t0, t1):
    """A span measured by the caller (perf_counter values)."""
    if _spans is not None:
        with _lock:
            if _spans is not None:
                _spans.append((stage, t0, t1))


def _union(intervals):
    total, end = 0.0, None
    for a, b in sorted(intervals):
        if end is None or a > end:
            total += b - a
            end = b
        elif b > end:
            total += b - end
            end = b
    return total


def report(spans):
    """{'wall_s': first start -> last end, 'stages': {stage: {'spans', 'thread_s', 'wall_s'}}} -- stages in the
    order of their first start."""
    if not spans:
        return {'wall_s': 0.0, 'stages': {}}
    by = {}
    for stage, a, b in sorted(spans, key=lambda s: s[1]):
        by.setdefault(stage, []).append((a, b))
    out = {'wall_s': round(max(b for _, _, b in spans) - min(a for _, a, _ in spans), 4), 'stages': {}}
    for stage, iv in by.items():
        out['stages'][stage] = {'spans': len(iv), 'thread_s': round(sum(b - a for a, b in iv), 4),
                                'wall_s': round(_union(iv), 4)}
    return out
