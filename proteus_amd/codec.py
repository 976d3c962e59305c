"""ctypes binding of libdswx_codec.so (include/dswx_codec.h): DEFLATE of GeoTIFF blocks on native threads.

The host side of a product run is its codec (profiles/r06_product_run.json): the reference does it inside GDAL
(C++), this drop-in in a small native library -- the blocks of a file are compressed / decompressed side by side
without the interpreter lock, by libdeflate when the system has it (as GDAL >= 3.2 does), else zlib.  The
library is host-only; proteus_amd.geotiff falls back to Python's zlib module, loudly once, only if it cannot
be built or loaded (no C++ compiler): the FILES are the same either way, it is not a compute path.
"""
import ctypes
import os
import threading

import numpy as np

from . import build as _build

_lib = None
_lock = threading.Lock()


class CodecError(Exception):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = os.environ.get('DSWX_CODEC_LIB') or _build.build_codec()
        lib = ctypes.CDLL(path)
        vpp, szp = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_size_t)
        lib.dswx_codec_abi_version.restype = ctypes.c_int
        lib.dswx_codec_engine.restype = ctypes.c_char_p
        lib.dswx_codec_last_error.restype = ctypes.c_char_p
        lib.dswx_codec_force_zlib.argtypes = [ctypes.c_int]
        lib.dswx_codec_cpu_budget.restype = ctypes.c_int
        lib.dswx_codec_deflate_bound.restype = ctypes.c_size_t
        lib.dswx_codec_deflate_bound.argtypes = [ctypes.c_size_t]
        lib.dswx_codec_deflate_blocks.restype = ctypes.c_int
        lib.dswx_codec_deflate_blocks.argtypes = [vpp, szp, vpp, szp, szp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        lib.dswx_codec_inflate_blocks.restype = ctypes.c_int
        lib.dswx_codec_inflate_blocks.argtypes = [vpp, szp, vpp, szp, szp, ctypes.c_int32, ctypes.c_int32]
        lib.dswx_codec_set_cpu_budget.argtypes = [ctypes.c_int]
        lib.dswx_codec_unlzw_blocks.restype = ctypes.c_int
        lib.dswx_codec_unlzw_blocks.argtypes = [vpp, szp, vpp, szp, szp, ctypes.c_int32, ctypes.c_int32]
        if lib.dswx_codec_abi_version() != 1:
            raise CodecError(f'{path}: ABI version {lib.dswx_codec_abi_version()}, expected 1')
        try:
            share = int(os.environ.get('DSWX_CPU_SHARE', '0') or 0)  # set by proteus_amd.batch for its worker processes
        except ValueError:
            share = 0
        if share > 0:
            lib.dswx_codec_set_cpu_budget(share)
        _lib = lib
    return _lib


def engine():
    return load().dswx_codec_engine().decode()


def force_zlib(on):
    load().dswx_codec_force_zlib(int(bool(on)))


def cpu_budget():
    """Processors this process may really use: hardware threads cut down to the container's CPU quota (cgroup)."""
    return int(load().dswx_codec_cpu_budget())


def set_cpu_budget(processors):
    """This process's share of the machine (0 = what was detected)."""
    load().dswx_codec_set_cpu_budget(int(processors))


def default_threads():
    """Workers of one call.  DSWX_IO_THREADS overrides (1 = serial); default: the processors this process may use
    (cpu_budget) up to 64 -- several files are read / written side by side, each with its own call, and the pool as a
    whole never exceeds the budget."""
    n = int(os.environ.get('DSWX_IO_THREADS', '0'))
    return n if n > 0 else min(64, cpu_budget())


def _check(rc):
    if rc:
        raise CodecError(load().dswx_codec_last_error().decode('utf-8', 'replace'))


def _vp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_void_p))


def _sz(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_size_t))


def deflate_uniform(src, block_bytes, level=6, threads=None):
    """`src`: C-contiguous uint8-viewable ndarray holding n blocks of `block_bytes` back to back.  Returns
    (out, offsets, sizes): block i compressed = out[offsets[i]: offsets[i] + sizes[i]] (out: uint8 ndarray)."""
    lib = load()
    raw = np.ascontiguousarray(src).reshape(-1).view(np.uint8)
    n = raw.size // block_bytes if block_bytes else 0
    if n * block_bytes != raw.size:
        raise ValueError('src is not a whole number of blocks')
    cap = int(lib.dswx_codec_deflate_bound(block_bytes))
    out = np.empty(n * cap, dtype=np.uint8)
    offsets = np.arange(n, dtype=np.uintp) * np.uintp(cap)
    src_ptrs = np.uintp(raw.ctypes.data) + np.arange(n, dtype=np.uintp) * np.uintp(block_bytes)
    dst_ptrs = np.uintp(out.ctypes.data) + offsets
    sizes_in = np.full(n, block_bytes, dtype=np.uintp)
    caps = np.full(n, cap, dtype=np.uintp)
    sizes = np.zeros(n, dtype=np.uintp)
    _check(lib.dswx_codec_deflate_blocks(_vp(src_ptrs), _sz(sizes_in), _vp(dst_ptrs), _sz(caps), _sz(sizes), n, int(level),
                                         int(threads or default_threads())))
    return out, offsets.astype(np.int64), sizes.astype(np.int64)


def inflate_into(buf, offsets, counts, dst, block_bytes, threads=None, scheme='deflate'):
    """n zlib streams (scheme 'lzw': n TIFF LZW streams) buf[offsets[i]: offsets[i] + counts[i]] (buf: bytes-like) into dst (uint8-viewable, C-contiguous,
    n * block_bytes bytes): block i lands at dst[i * block_bytes].  Returns the produced sizes (int64 [n]); a block that
    inflates to MORE than block_bytes raises."""
    lib = load()
    n = len(offsets)
    src = np.frombuffer(buf, dtype=np.uint8)
    raw = dst.reshape(-1).view(np.uint8)
    if raw.size < n * block_bytes:
        raise ValueError('dst too small')
    offsets = np.asarray(offsets, dtype=np.uintp)
    counts = np.ascontiguousarray(counts, dtype=np.uintp)
    if n and int((offsets + counts).max()) > src.size:
        raise CodecError('block table points outside the file')
    src_ptrs = np.uintp(src.ctypes.data) + offsets
    dst_ptrs = np.uintp(raw.ctypes.data) + np.arange(n, dtype=np.uintp) * np.uintp(block_bytes)
    caps = np.full(n, block_bytes, dtype=np.uintp)
    sizes = np.zeros(n, dtype=np.uintp)
    entry = {'deflate': lib.dswx_codec_inflate_blocks, 'lzw': lib.dswx_codec_unlzw_blocks}[scheme]
    _check(entry(_vp(src_ptrs), _sz(counts), _vp(dst_ptrs), _sz(caps), _sz(sizes), n, int(threads or default_threads())))
    return sizes.astype(np.int64)
