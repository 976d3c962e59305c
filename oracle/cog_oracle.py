"""TEST INFRASTRUCTURE -- CPU restatement of the raster-format steps either side of the per-pixel path (SURVEY.md
section 8 f4), block by block and row by row, sharing no code with the product's writer (proteus_amd/geotiff.py) or its
device kernels (proteus_amd/csrc/dswx_writer.hip).  Only tests/ may import it.

What it restates, and from where:

* `save_as_cog` (/root/reference/src/proteus/core.py:7-91) asks GDAL for overviews 4 / 16 / 64 / 128, NEAREST for the integer
  layers (:37-46), then `gdal.Translate` with TILED=YES, 512 x 512 blocks, DEFLATE, PREDICTOR=2 (integers) / 3 (floating
  point) (:60-75).  GDAL and libtiff are third-party code that is not in the reference tree (setup.py / docker/requirements
  name `gdal`, unpinned there; the PGE image ships GDAL 3.x):
    - NEAREST overview: GDAL's GDALResampleChunk_Near picks, for destination pixel i of an overview of N_ovr = ceil(N / f)
      pixels, the source pixel int(0.5 + i * N / N_ovr), clamped to N - 1 (published algorithm, gcore/overview.cpp);
    - PREDICTOR=2: libtiff horDiff8 / horDiff16 -- every sample of a block row minus its left neighbour, in the sample's
      own width (wrap-around), the first sample of the row as it is; edge blocks are padded (GDAL pads with zeros);
    - PREDICTOR=3: libtiff fpDiff (Adobe TIFF Technical Note 3) -- the bytes of the row's samples regrouped into byte
      planes, most significant byte first, then the whole row of bytes differenced byte-wise.
  The inverses (horAcc / fpAcc) are what `ReadAsArray` applies when a band file is read (dswx_hls.py:2136-2302).
* `_save_output_rgb_file` (dswx_hls.py:3013-3036): scale * (float32(band) - offset), NaN at the invalid pixels.
* CUBICSPLINE overviews of the non-integer layers (core.py:41-46): GDAL's GDALResampleChunk_Convolution with the cubic
  B-spline (GWKBSpline; published algorithm, gcore/overview.cpp + alg/gdalwarpkernel.cpp): a destination pixel is centred on
  source coordinate (i + 0.5) * ratio, the kernel is stretched by the decimation ratio (support 2 * ratio source pixels
  either side), evaluated at the source pixel centres, normalised over the pixels that exist and are not NaN; horizontal
  pass, then vertical pass, in double; levels in cascade (GDALRegenerateCascadingOverviews).
* the Byte bands of `save_dswx_product` (dswx_hls.py:2663-2666: every band GDT_Byte): GDALCopyWords clamps integers to
  0 .. 255 and rounds floating point half up after clamping, NaN -> 0.
* the browse PNG's resize (`geotiff2png`, dswx_hls.py:5335-5349): GDAL RasterIO's nearest pick, src = floor((dst + 0.5) *
  N_src / N_dst).

PINNING.  No GDAL exists in this image, so the NEAREST rule, the CUBICSPLINE convolution, the Byte conversion and the
RasterIO pick are unpinned by execution ("parity unpinned" for them: DESIGN.md section 8 says so).  The two predictors ARE pinned against an independent implementation: libtiff,
through Pillow, decodes files whose blocks were made by these functions' product-side counterparts
(tests/test_host_logic.py) and encodes files these inverses decode (tests/test_cog_oracle.py).
"""
import math

import numpy as np


def nearest_overview(arr, factor):
    """One NEAREST overview level of a 2-D raster, element by element."""
    h, w = arr.shape
    oh, ow = -(-h // factor), -(-w // factor)
    out = np.empty((oh, ow), arr.dtype)
    ys = [min(int(0.5 + i * (h / oh)), h - 1) for i in range(oh)]
    xs = [min(int(0.5 + j * (w / ow)), w - 1) for j in range(ow)]
    for i, y in enumerate(ys):
        row = arr[y]
        for j, x in enumerate(xs):
            out[i, j] = row[x]
    return out


def _hordiff_row(row):
    """libtiff horDiff: row of unsigned samples -> differenced row (same dtype, wrap-around)."""
    bits = 8 * row.dtype.itemsize
    out = row.copy()
    for x in range(len(row) - 1, 0, -1):
        out[x] = (int(row[x]) - int(row[x - 1])) % (1 << bits)
    return out


def _fpdiff_row(row):
    """libtiff fpDiff on one row of float32 samples -> uint8 [4 * n]."""
    n = len(row)
    be = row.astype('>f4').tobytes()                    # byte 0 of every sample = most significant
    planes = bytearray(4 * n)
    for j in range(n):
        for b in range(4):
            planes[b * n + j] = be[4 * j + b]
    out = bytearray(planes)
    for q in range(4 * n - 1, 0, -1):
        out[q] = (planes[q] - planes[q - 1]) & 0xff
    return np.frombuffer(bytes(out), dtype=np.uint8)


def blocks(arr, tile, predictor):
    """2-D raster -> the bytes of its tile x tile blocks in row-major block order (edge blocks zero-padded, predictor
    applied per block row, little endian), block by block, row by row."""
    h, w = arr.shape
    down, across = -(-h // tile), -(-w // tile)
    out = []
    unsigned = {1: np.uint8, 2: np.uint16, 4: np.uint32}[arr.dtype.itemsize]
    for by in range(down):
        for bx in range(across):
            blk = np.zeros((tile, tile), arr.dtype)
            hh, ww = min(tile, h - by * tile), min(tile, w - bx * tile)
            blk[:hh, :ww] = arr[by * tile: by * tile + hh, bx * tile: bx * tile + ww]
            for y in range(tile):
                if predictor == 2:
                    out.append(_hordiff_row(blk[y].view(unsigned)).astype('<' + np.dtype(unsigned).str[1:]).tobytes())
                elif predictor == 3:
                    out.append(_fpdiff_row(blk[y]).tobytes())
                else:
                    out.append(blk[y].astype(arr.dtype.newbyteorder('<')).tobytes())
    return np.frombuffer(b''.join(out), dtype=np.uint8)


def cog_levels(arr, factors, tile=512, predictor=2):
    """[(height, width, block bytes)] for the full-resolution raster and every overview level, as save_as_cog leaves them."""
    rasters = [arr] + [nearest_overview(arr, f) for f in factors if f > 1 and arr.shape != (1, 1)]
    return [(r.shape[0], r.shape[1], blocks(r, tile, predictor)) for r in rasters]


def unblocks(data, dtype, height, width, bw, bh, predictor):
    """The inverse: the bytes of the blocks of one plane (bh x bw samples each; strips: bw = width) -> raster.  horAcc / fpAcc
    row by row."""
    dtype = np.dtype(dtype)
    across, down = -(-width // bw), -(-height // bh)
    es = dtype.itemsize
    out = np.zeros((height, width), dtype)
    raw = np.asarray(data, dtype=np.uint8).tobytes()
    unsigned = {1: np.uint8, 2: np.uint16, 4: np.uint32}[es]
    for by in range(down):
        for bx in range(across):
            base = (by * across + bx) * bh * bw * es
            for y_in in range(bh):
                y = by * bh + y_in
                if y >= height:
                    break
                seg = raw[base + y_in * bw * es: base + (y_in + 1) * bw * es]
                if predictor == 3:
                    acc = bytearray(seg)
                    for q in range(1, len(acc)):
                        acc[q] = (acc[q] + acc[q - 1]) & 0xff
                    be = bytearray(4 * bw)
                    for j in range(bw):
                        for b in range(4):
                            be[4 * j + b] = acc[b * bw + j]
                    row = np.frombuffer(bytes(be), dtype='>f4').astype(np.float32)
                else:
                    row = np.frombuffer(seg, dtype=np.dtype(unsigned).newbyteorder('<')).astype(unsigned)
                    if predictor == 2:
                        vals = [int(v) for v in row]
                        for x in range(1, len(vals)):
                            vals[x] = (vals[x] + vals[x - 1]) % (1 << (8 * es))
                        row = np.array(vals, dtype=unsigned)
                    row = row.view(dtype) if dtype.kind != 'f' else row.view(np.float32)
                ww = min(bw, width - bx * bw)
                out[y, bx * bw: bx * bw + ww] = row[:ww]
    return out


def rgb_planes(bands, diag, scales, offsets, clip=True):
    """_save_output_rgb_file (:3013-3036) on the clipped bands: float32 [3, H, W], NaN where DIAG carries the fill code."""
    out = []
    for b, sc, of in zip(bands, scales, offsets):
        if clip:
            b = np.clip(b, 1, None)
        v = sc * (np.asarray(b, dtype=np.float32) - of)
        v = np.asarray(v, dtype=np.float32)
        if diag is not None:
            v[diag == 65535] = np.nan
        out.append(v)
    return np.stack(out)


def _bspline(x):
    """The cubic B-spline, support |x| < 2."""
    x = abs(x)
    if x <= 1.0:
        return 2.0 / 3.0 + x * x * (0.5 * x - 1.0)
    if x < 2.0:
        return (2.0 - x) ** 3 / 6.0
    return 0.0


def _convolve_line(line, n_out):
    """One line (Python floats) -> n_out values (Python floats): the stretched B-spline over the existing, non-NaN taps."""
    n_in = len(line)
    ratio = n_in / n_out
    scale = min(1.0, 1.0 / ratio)
    radius = 2.0 / scale
    taps = int(math.ceil(2 * radius)) + 1
    out = []
    for i in range(n_out):
        centre = (i + 0.5) * ratio
        first = int(math.floor(centre - radius + 0.5))
        num = den = 0.0
        for j in range(first, first + taps):
            if j < 0 or j >= n_in:
                continue
            v = line[j]
            if v != v:
                continue
            w = _bspline((j + 0.5 - centre) * scale)
            if w > 0.0:
                num += v * w
            den += w
        out.append(num / den if den > 0.0 else float('nan'))
    return out


def cubicspline_overview(arr, factor):
    """One CUBICSPLINE overview level of a 2-D float32 raster, line by line: rows first (kept in double), then columns."""
    h, w = arr.shape
    oh, ow = -(-h // factor), -(-w // factor)
    rows = [_convolve_line([float(v) for v in arr[y]], ow) for y in range(h)]
    out = np.empty((oh, ow), np.float32)
    with np.errstate(over='ignore'):
        for x in range(ow):
            col = _convolve_line([rows[y][x] for y in range(h)], oh)
            for y in range(oh):
                out[y, x] = np.float32(col[y])
    return out


def cubicspline_pyramid(arr, factors):
    """[arr, level f1, level f2, ...]: each level from the previous one when its factor divides (and the sizes agree), else
    from the full-resolution raster."""
    h, w = arr.shape
    levels, prev_f = [arr], 1
    for f in factors:
        if f <= 1 or (h, w) == (1, 1):
            continue
        want = (-(-h // f), -(-w // f))
        lv = None
        if prev_f > 1 and f % prev_f == 0:
            lv = cubicspline_overview(levels[-1], f // prev_f)
            if lv.shape != want:
                lv = None
        if lv is None:
            lv = cubicspline_overview(arr, f)
        levels.append(lv)
        prev_f = f
    return levels


def gdal_byte(arr):
    """A raster as GDAL stores it in a Byte band, element by element."""
    flat = arr.ravel()
    out = np.empty(flat.size, np.uint8)
    for i, v in enumerate(flat):
        if arr.dtype.kind == 'f':
            x = float(v)
            if x != x:
                x = 0.0
            x = min(max(x, 0.0), 255.0)
            out[i] = int(min(math.floor(x + 0.5), 255.0))
        else:
            out[i] = min(max(int(v), 0), 255)
    return out.reshape(arr.shape)


def resample_nearest(arr, out_height, out_width):
    """RasterIO's nearest-neighbour resize, element by element."""
    h, w = arr.shape
    out = np.empty((out_height, out_width), arr.dtype)
    for i in range(out_height):
        y = min(int((i + 0.5) * h / out_height), h - 1)
        for j in range(out_width):
            out[i, j] = arr[y, min(int((j + 0.5) * w / out_width), w - 1)]
    return out
