"""Minimal GeoTIFF reader / writer (no GDAL).

PROTEUS does all raster I/O through GDAL (`_load_hls_band_from_file`
src/proteus/dswx_hls.py:2136, `_save_array` :2893, `save_dswx_product` :2601,
`save_as_cog` core.py:7).  GDAL is not installed in this image, so the host side of
the drop-in carries this small classic-TIFF implementation of exactly what that
path touches:

 read   little/big-endian classic TIFF, strips or tiles, compression none / DEFLATE / LZW
        (horizontal predictor 2, floating-point predictor 3), 1..N samples (chunky or
        planar), u8/i8/u16/i16/u32/i32/f32/f64, GDAL_METADATA (42112), GDAL_NODATA (42113),
        colour map and the GeoTIFF tags (33550, 33922, 34264, 34735-34737).
 write  little-endian classic TIFF, 512x512 tiles, DEFLATE + PREDICTOR=2 for the integer
        layers and PREDICTOR=3 for the Float32 ones (the reference's COG creation options,
        core.py:60-75), internal overviews 4 / 16 / 64 / 128 (NEAREST for integer layers,
        CUBICSPLINE for floating point, core.py:37-46), planar multi-band, nodata,
        per-band descriptions, colour table, metadata, and the GeoTIFF tags copied from the
        input HLS file.  Files appear under their final name only when complete.

The georeferencing is carried opaquely: the projection of the output product is the
set of GeoKey tags of the input (the reference copies `GetProjection()` the same
way, :2256-2258 -> :2668).  BigTIFF is not supported.
"""
import os
import struct
import time
import zlib
from xml.sax.saxutils import escape, unescape

import numpy as np

from . import stages

TAG_NEW_SUBFILE_TYPE = 254
TAG_WIDTH, TAG_LENGTH, TAG_BITS, TAG_COMPRESSION, TAG_PHOTOMETRIC = 256, 257, 258, 259, 262
TAG_STRIP_OFFSETS, TAG_SAMPLES, TAG_ROWS_PER_STRIP, TAG_STRIP_COUNTS = 273, 277, 278, 279
TAG_PLANAR, TAG_PREDICTOR, TAG_COLORMAP = 284, 317, 320
TAG_TILE_W, TAG_TILE_L, TAG_TILE_OFFSETS, TAG_TILE_COUNTS = 322, 323, 324, 325
TAG_EXTRA_SAMPLES, TAG_SAMPLE_FORMAT = 338, 339
TAG_PIXEL_SCALE, TAG_TIEPOINT, TAG_TRANSFORM = 33550, 33922, 34264
TAG_GEOKEYS, TAG_GEO_DOUBLES, TAG_GEO_ASCII = 34735, 34736, 34737
TAG_GDAL_METADATA, TAG_GDAL_NODATA = 42112, 42113
GEO_TAGS = (TAG_PIXEL_SCALE, TAG_TIEPOINT, TAG_TRANSFORM, TAG_GEOKEYS, TAG_GEO_DOUBLES,
            TAG_GEO_ASCII)

_TYPE_FMT = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 6: 'b', 7: 'B', 8: 'h', 9: 'i',
             10: 'ii', 11: 'f', 12: 'd', 16: 'Q'}
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8,
              16: 8}


class GeoTiffError(Exception):
    pass


class GeoTiffInfo:
    """What `gdal.Open` would have told the reference about a file."""

    def __init__(self):
        self.width = self.height = self.bands = 0
        self.dtype = None
        self.nodata = None            # float or None  (GetNoDataValue)
        self.metadata = {}            # dataset-level GDAL metadata (GetMetadata)
        self.descriptions = []        # per band
        self.geo_tags = {}            # raw GeoTIFF tags: tag -> (type, values)
        self.colormap = None          # [256][3] uint8 or None

    @property
    def geotransform(self):
        """GDAL-style 6-tuple (GetGeoTransform); identity-like default if absent."""
        if TAG_TRANSFORM in self.geo_tags:
            m = self.geo_tags[TAG_TRANSFORM][1]
            return (m[3], m[0], m[1], m[7], m[4], m[5])
        if TAG_PIXEL_SCALE in self.geo_tags and TAG_TIEPOINT in self.geo_tags:
            sx, sy = self.geo_tags[TAG_PIXEL_SCALE][1][:2]
            i, j, _, x, y, _ = self.geo_tags[TAG_TIEPOINT][1][:6]
            return (x - i * sx, sx, 0.0, y + j * sy, 0.0, -sy)
        return (0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


def _dtype_of(bits, fmt):
    table = {(8, 1): np.uint8, (8, 2): np.int8, (16, 1): np.uint16, (16, 2): np.int16,
             (32, 1): np.uint32, (32, 2): np.int32, (32, 3): np.float32,
             (64, 3): np.float64}
    try:
        return np.dtype(table[(bits, fmt)])
    except KeyError:
        raise GeoTiffError(f'unsupported sample: {bits} bits, format {fmt}')


def _parse_metadata_xml(text):
    """GDALMetadata XML -> (dataset dict, {band: description})."""
    meta, desc = {}, {}
    pos = 0
    while True:
        a = text.find('<Item', pos)
        if a < 0:
            break
        b = text.find('>', a)
        c = text.find('</Item>', b)
        if b < 0 or c < 0:
            break
        attrs = {}
        for part in text[a + 5:b].replace("'", '"').split('" '):
            if '=' in part:
                k, v = part.split('=', 1)
                attrs[k.strip()] = v.strip().strip('"')
        value = unescape(text[b + 1:c], {'&quot;': '"', '&apos;': "'"})
        name = unescape(attrs.get('name', ''), {'&quot;': '"'})
        if 'sample' in attrs:
            if attrs.get('role') == 'description' or name == 'DESCRIPTION':
                desc[int(attrs['sample'])] = value
        else:
            meta[name] = value
        pos = c + 7
    return meta, desc


class TiffDirectory:
    """One image file directory of a TIFF file, parsed and ready to decode: the file's bytes, what `gdal.Open` would
    report (`info`), and the block geometry.  Decoding is two steps so that the second can run on the GPU
    (proteus_amd.pipeline): inflate() puts every block -- inflated, still predictor-encoded, still in block order -- into
    one staging buffer [n_blocks][block_bytes] (native threads, proteus_amd.codec), untile() undoes the predictor and
    moves the blocks into the row-major raster."""

    def __init__(self, path, overview=None):
        self.path = path
        with stages.span('read: file'), open(path, 'rb') as fh:
            buf = fh.read()
        self.buf = buf
        try:
            self._parse(path, buf, overview)
        except (KeyError, IndexError, ValueError, struct.error, TypeError, UnicodeDecodeError, ZeroDivisionError,
                OverflowError) as e:
            # a damaged directory is an unreadable FILE to the caller (the reference: gdal.Open returns None -> 'ERROR could
            # not open', dswx_hls.py:4988-4990), not a lookup error from inside the parser
            raise GeoTiffError(f'{path}: damaged TIFF directory ({type(e).__name__}: {e})')

    def _parse(self, path, buf, overview):
        if len(buf) < 8:
            raise GeoTiffError(f'{path}: not a TIFF file')
        if buf[:2] == b'II':
            e = '<'
        elif buf[:2] == b'MM':
            e = '>'
        else:
            raise GeoTiffError(f'{path}: not a TIFF file')
        self.e = e
        magic, ifd = struct.unpack(e + 'HI', buf[2:8])
        if magic == 43:
            raise GeoTiffError(f'{path}: BigTIFF is not supported')
        if magic != 42:
            raise GeoTiffError(f'{path}: not a TIFF file')
        for _ in range(0 if overview is None else overview + 1):
            if ifd + 2 > len(buf):
                raise GeoTiffError(f'{path}: directory outside the file')
            (n,) = struct.unpack(e + 'H', buf[ifd:ifd + 2])
            if ifd + 6 + 12 * n > len(buf):
                raise GeoTiffError(f'{path}: truncated directory')
            (ifd,) = struct.unpack(e + 'I', buf[ifd + 2 + 12 * n: ifd + 6 + 12 * n])
            if not ifd:
                raise GeoTiffError(f'{path}: no overview {overview}')
        if ifd + 2 > len(buf):
            raise GeoTiffError(f'{path}: directory outside the file')
        (n,) = struct.unpack(e + 'H', buf[ifd:ifd + 2])
        if ifd + 2 + 12 * n + 4 > len(buf):
            raise GeoTiffError(f'{path}: truncated directory')
        tags = {}
        for k in range(n):
            ent = buf[ifd + 2 + 12 * k: ifd + 14 + 12 * k]
            tag, typ, count = struct.unpack(e + 'HHI', ent[:8])
            if typ not in _TYPE_SIZE:
                continue
            size = _TYPE_SIZE[typ] * count
            if size <= 4:
                raw = ent[8:8 + size]
            else:
                (off,) = struct.unpack(e + 'I', ent[8:12])
                if off + size > len(buf):           # a damaged count / offset: not a reason to build a gigabyte format string
                    raise GeoTiffError(f'{path}: directory entry {tag} points outside the file')
                raw = buf[off:off + size]
            if typ == 2:
                vals = raw.rstrip(b'\x00').decode('latin-1')
            elif typ in (5, 10):
                flat = struct.unpack(e + _TYPE_FMT[typ][0] * (2 * count), raw)
                vals = [flat[2 * i] / flat[2 * i + 1] if flat[2 * i + 1] else 0.0
                        for i in range(count)]
            else:
                vals = list(struct.unpack(e + _TYPE_FMT[typ] * count, raw))
            tags[tag] = (typ, vals)
        self.tags = tags

        def one(tag, default=None):
            if tag not in tags:
                return default
            v = tags[tag][1]
            return v if isinstance(v, str) else v[0]

        info = self.info = GeoTiffInfo()
        info.width, info.height = one(TAG_WIDTH), one(TAG_LENGTH)
        self.spp = spp = one(TAG_SAMPLES, 1)
        info.bands = spp
        bits = one(TAG_BITS, 1)
        fmt = one(TAG_SAMPLE_FORMAT, 1)
        info.dtype = _dtype_of(bits, fmt)
        self.comp = comp = one(TAG_COMPRESSION, 1)
        if comp not in (1, 5, 8, 32946):
            raise GeoTiffError(f'{path}: compression {comp} is not supported')
        predictor = one(TAG_PREDICTOR, 1)
        self.planar = planar = one(TAG_PLANAR, 1)
        if predictor not in (1, 2, 3) or (predictor == 3 and fmt != 3):
            raise GeoTiffError(f'{path}: predictor {predictor} is not supported')
        if comp == 1:
            predictor = 1       # libtiff: the predictor belongs to the LZW / DEFLATE codecs; an uncompressed file's tag is ignored
        self.predictor = predictor
        self.dt = info.dtype.newbyteorder(e)
        H, W = info.height, info.width
        self.tiled = tiled = TAG_TILE_OFFSETS in tags
        if tiled:
            self.bw, self.bh = one(TAG_TILE_W), one(TAG_TILE_L)
            self.offs, self.cnts = tags[TAG_TILE_OFFSETS][1], tags[TAG_TILE_COUNTS][1]
        else:
            self.bw, self.bh = W, min(one(TAG_ROWS_PER_STRIP, H), H)
            self.offs, self.cnts = tags[TAG_STRIP_OFFSETS][1], tags[TAG_STRIP_COUNTS][1]
        if not self.bw or not self.bh:
            raise GeoTiffError(f'{path}: empty block size')
        self.across, self.down = (W + self.bw - 1) // self.bw, (H + self.bh - 1) // self.bh
        self.planes = spp if planar == 2 else 1
        self.chunk_spp = 1 if planar == 2 else spp
        self.n_blocks = self.planes * self.across * self.down
        self.block_bytes = self.bh * self.bw * self.chunk_spp * self.dt.itemsize
        if len(self.offs) < self.n_blocks or len(self.cnts) < self.n_blocks:
            raise GeoTiffError(f'{path}: truncated block table')
        # a damaged header must not make the reader allocate the moon: DEFLATE expands at most ~1032 : 1, so a raster
        # that claims more decoded bytes than that (or, uncompressed, more than the file holds) is not what the file contains
        claimed = self.n_blocks * self.block_bytes
        if info.width < 1 or info.height < 1 or spp < 1 or \
                claimed > ((3000 if comp == 5 else 1100) * len(buf) + (1 << 20) if comp != 1 else len(buf) + self.block_bytes * self.planes * self.across):
            raise GeoTiffError(f'{path}: the directory claims {claimed} bytes of raster, the file has {len(buf)}')

        nod = one(TAG_GDAL_NODATA)
        if nod is not None:
            try:
                info.nodata = float(nod.strip())
            except ValueError:
                info.nodata = None
        if TAG_GDAL_METADATA in tags:
            info.metadata, desc = _parse_metadata_xml(tags[TAG_GDAL_METADATA][1])
            info.descriptions = [desc.get(i, '') for i in range(spp)]
        else:
            info.descriptions = [''] * spp
        for t in GEO_TAGS:
            if t in tags:
                info.geo_tags[t] = tags[t]
        if TAG_COLORMAP in tags:
            cm = np.asarray(tags[TAG_COLORMAP][1], dtype=np.uint32).reshape(3, -1)
            info.colormap = (cm >> 8).astype(np.uint8).T

    # what a block of the last row of strips really holds (a tile is always whole)
    def _rows_of(self, by):
        return self.bh if self.tiled else min(self.bh, self.info.height - by * self.bh)

    def inflate(self, staging=None, alloc=None):
        """Every block, inflated (or copied, for an uncompressed file), at staging[i * block_bytes] in block order
        (plane-major, then row-major).  `staging`: uint8-viewable, C-contiguous, >= n_blocks * block_bytes bytes, or
        None (allocated, through alloc(shape, dtype) if given).  Bytes of a block beyond what the file holds for it
        (the short last strip) are zero."""
        need = self.n_blocks * self.block_bytes
        if staging is None:
            staging = np.empty(need, dtype=np.uint8) if alloc is None else alloc((need,), np.uint8)
        raw = staging.reshape(-1).view(np.uint8)
        if raw.size < need:
            raise GeoTiffError('staging buffer too small')
        offs = np.asarray(self.offs[:self.n_blocks], dtype=np.int64)
        cnts = np.asarray(self.cnts[:self.n_blocks], dtype=np.int64)
        if self.n_blocks and (int((offs + cnts).max()) > len(self.buf) or int(offs.min()) < 0):
            raise GeoTiffError(f'{self.path}: block table points outside the file')
        if self.comp == 1:
            with stages.span('read: copy blocks'):
                src = np.frombuffer(self.buf, dtype=np.uint8)
                for i in range(self.n_blocks):
                    n = min(int(cnts[i]), self.block_bytes)
                    raw[i * self.block_bytes: i * self.block_bytes + n] = src[offs[i]: offs[i] + n]
                    raw[i * self.block_bytes + n: (i + 1) * self.block_bytes] = 0
            return staging
        with stages.span('read: inflate'):
            short = not self.tiled and self.info.height % self.bh
            if short:           # the last strip of every plane inflates to fewer bytes: the rest of its slot is zero
                per_plane = self.across * self.down
                for p in range(self.planes):
                    i = p * per_plane + per_plane - 1
                    raw[i * self.block_bytes: (i + 1) * self.block_bytes] = 0
            native = _codec()
            if self.comp == 5 and native is None:
                raise GeoTiffError(f'{self.path}: LZW needs the native codec (libdswx_codec.so)')
            if native is not None:
                try:
                    sizes = native.inflate_into(self.buf, offs, cnts, raw[:need], self.block_bytes,
                                                scheme='lzw' if self.comp == 5 else 'deflate')
                except native.CodecError as e:
                    raise GeoTiffError(f'{self.path}: {e}')
                # a stream that ends early leaves the rest of its block as zeros (the staging buffer is recycled memory)
                for i in np.flatnonzero(sizes < self.block_bytes):
                    raw[i * self.block_bytes + int(sizes[i]): (i + 1) * self.block_bytes] = 0
            else:
                def one_block(i):
                    data = zlib.decompress(self.buf[offs[i]: offs[i] + cnts[i]])[:self.block_bytes]
                    raw[i * self.block_bytes: i * self.block_bytes + len(data)] = np.frombuffer(data, dtype=np.uint8)
                pool = _io_pool() if self.n_blocks > 1 else None
                if pool:
                    list(pool.map(one_block, range(self.n_blocks)))
                else:
                    for i in range(self.n_blocks):
                        one_block(i)
        return staging

    def untile(self, staging, out=None, alloc=None):
        """staging (inflate()) -> [spp, H, W] in the file's sample type, native byte order: inverse predictor over every
        block row, blocks moved to their windows.  Whole-array numpy operations (the GPU pipeline does the same on the
        device: dswx_untile_device)."""
        info = self.info
        H, W, spp = info.height, info.width, self.spp
        if out is None:
            out = np.zeros((spp, H, W), dtype=info.dtype) if alloc is None else alloc((spp, H, W), info.dtype)
        with stages.span('read: predictor + untile'):
            raw = staging.reshape(-1).view(np.uint8)[:self.n_blocks * self.block_bytes]
            rows_all = self.n_blocks * self.bh
            n = self.bw * self.chunk_spp
            if self.predictor == 3:
                blk = _fp_predictor_decode(raw, rows_all, n, self.chunk_spp, info.dtype)
            else:
                blk = raw.view(self.dt).reshape(rows_all, self.bw, self.chunk_spp)
                if self.dt.byteorder not in ('=', '|') and self.dt != info.dtype:
                    blk = blk.astype(info.dtype)
                if self.predictor == 2:
                    blk = np.cumsum(blk, axis=1, dtype=info.dtype)
            blk = blk.reshape(self.planes, self.down, self.across, self.bh, self.bw, self.chunk_spp)
            # [planes, down, bh, across, bw, spp'] -> rows x columns of the padded raster
            full = blk.transpose(0, 1, 3, 2, 4, 5).reshape(self.planes, self.down * self.bh, self.across * self.bw,
                                                            self.chunk_spp)[:, :H, :W, :]
            if self.planar == 2 or spp == 1:
                out[...] = full[..., 0]
            else:
                out[...] = np.moveaxis(full[0], 2, 0)
        return out


def open_geotiff(path, overview=None):
    return TiffDirectory(path, overview)


def read_geotiff(path, window=None, overview=None, alloc=None):
    """Returns (array, GeoTiffInfo).  array is [H,W] for one band, [B,H,W] otherwise.
    window = (xoff, yoff, xsize, ysize) crops after decoding (the reference's
    flag_debug read, :2187-2190).  overview = k reads the k-th internal overview
    (IFD k + 1) instead of the full-resolution image.  alloc(shape, dtype) -> ndarray lets the
    caller own the destination memory (e.g. page-locked host memory for the GPU path)."""
    d = TiffDirectory(path, overview)
    try:
        out = d.untile(d.inflate(), alloc=alloc)
    except (ValueError, IndexError, MemoryError) as e:              # geometry that does not hold together
        raise GeoTiffError(f'{path}: damaged TIFF ({type(e).__name__}: {e})')
    info = d.info
    arr = out[0] if d.spp == 1 else out
    if window is not None:
        xo, yo, xs, ys = window
        arr = arr[..., yo:yo + ys, xo:xo + xs]
        info.height, info.width = arr.shape[-2:]
    return np.ascontiguousarray(arr), info


def _metadata_xml(metadata, descriptions):
    items = []
    for k, v in (metadata or {}).items():
        items.append(f'  <Item name="{escape(str(k), {chr(34): "&quot;"})}">'
                     f'{escape(str(v))}</Item>')
    for i, d in enumerate(descriptions or []):
        if d:
            items.append(f'  <Item name="DESCRIPTION" sample="{i}" role="description">'
                         f'{escape(str(d))}</Item>')
    return '<GDALMetadata>\n' + '\n'.join(items) + '\n</GDALMetadata>\n'


def geo_tags_from_geotransform(geotransform, epsg=None):
    """GeoTIFF tags for a north-up grid (used by the synthetic HLS writer)."""
    x0, sx, _, y0, _, sy = geotransform
    tags = {TAG_PIXEL_SCALE: (12, [float(sx), float(-sy), 0.0]),
            TAG_TIEPOINT: (12, [0.0, 0.0, 0.0, float(x0), float(y0), 0.0])}
    if epsg is not None:
        tags[TAG_GEOKEYS] = (3, [1, 1, 0, 3,
                                 1024, 0, 1, 1,        # GTModelTypeGeoKey: projected
                                 1025, 0, 1, 1,        # GTRasterTypeGeoKey: pixel is area
                                 3072, 0, 1, int(epsg)])
    return tags


COG_OVERVIEW_FACTORS = (4, 16, 64, 128)      # reference core.py:37


def _fp_predictor_encode(blk):
    """TIFF floating-point predictor (PREDICTOR=3, Adobe TIFF Technical Note 3; libtiff fpDiff): per row,
    the bytes of the samples are regrouped into byte planes, most significant byte first whatever the
    file's byte order, and the whole row of bytes is differenced horizontally.  blk: [rows, n] float32 /
    float64 (one sample per pixel: planar configuration) -> bytes."""
    rows, n = blk.shape
    bps = blk.dtype.itemsize
    be = np.ascontiguousarray(blk.astype(blk.dtype.newbyteorder('>')))          # byte 0 = most significant
    planes = be.view(np.uint8).reshape(rows, n, bps).transpose(0, 2, 1).reshape(rows, bps * n)
    out = planes.copy()
    out[:, 1:] = planes[:, 1:] - planes[:, :-1]                                  # uint8 arithmetic wraps
    return out.tobytes()


def _fp_predictor_decode(raw, rows, n, stride, dtype):
    """Inverse of _fp_predictor_encode (libtiff fpAcc).  `stride` = samples per pixel of the block (the
    byte differencing runs with that stride; 1 for planar files)."""
    bps = np.dtype(dtype).itemsize
    acc = raw.reshape(rows, bps * n).copy()
    for s0 in range(stride):                              # cumulative sum per interleaved sample
        acc[:, s0::stride] = np.cumsum(acc[:, s0::stride], axis=1, dtype=np.uint8)
    be = np.ascontiguousarray(acc.reshape(rows, bps, n).transpose(0, 2, 1))      # [rows, n, bps] big-endian
    return be.view(np.dtype(dtype).newbyteorder('>')).reshape(rows, n).astype(np.dtype(dtype))


def _bspline(x):
    """Cubic B-spline kernel of GDAL's CUBICSPLINE resampling (GWKBSpline, support |x| < 2)."""
    x = np.abs(x)
    return np.where(x <= 1.0, 2.0 / 3.0 + x * x * (0.5 * x - 1.0),
                    np.where(x < 2.0, (2.0 - x) ** 3 / 6.0, 0.0))


def convolve_weights(n_in, n_out):
    """(first [n_out] int64, weights [n_out, taps] float64) of one pass of the CUBICSPLINE overview convolution: destination
    pixel i is centred on source coordinate (i + 0.5) * ratio; the cubic B-spline is stretched by the decimation ratio
    (radius 2 * ratio source pixels) and evaluated at the source pixel centres first[i] ... first[i] + taps - 1; taps that
    fall outside the raster carry weight 0.  The host pass below and the device pass (dswx_convolve_axis_device) both
    take their weights from here."""
    ratio = n_in / n_out
    scale = min(1.0, 1.0 / ratio)                       # < 1 when decimating
    radius = 2.0 / scale
    centre = (np.arange(n_out) + 0.5) * ratio
    first = np.floor(centre - radius + 0.5).astype(np.int64)
    taps = int(np.ceil(2 * radius)) + 1
    idx = first[:, None] + np.arange(taps)[None, :]                       # [n_out, taps]
    w = _bspline((idx + 0.5 - centre[:, None]) * scale)
    w = np.where((idx >= 0) & (idx < n_in), w, 0.0)
    return first, np.ascontiguousarray(w, dtype=np.float64)


def _convolve_axis(a, n_out, axis):
    """One separable pass of GDAL's overview convolution (GDALResampleChunk_Convolution, overview.cpp) along
    `axis`: destination pixel i is centred on source coordinate (i + 0.5) * ratio; the kernel is stretched
    by the decimation ratio (radius 2 * ratio source pixels), evaluated at source pixel centres, and the
    weights are normalised over the pixels that exist and are not NaN (a destination whose whole support is
    NaN / outside is NaN).  float64 accumulation."""
    n_in = a.shape[axis]
    first, w = convolve_weights(n_in, n_out)
    a = np.moveaxis(a, axis, -1)
    num = np.zeros(a.shape[:-1] + (n_out,), dtype=np.float64)
    den = np.zeros_like(num)
    # tap by tap, in tap order: the order (and so the last bit) of the device pass, dswx_convolve_axis_v1
    for k in range(w.shape[1]):
        g = a[..., np.clip(first + k, 0, n_in - 1)].astype(np.float64)    # [..., n_out]
        ww = np.where(np.isnan(g), 0.0, w[:, k])
        with np.errstate(invalid='ignore'):
            num += np.where(ww > 0.0, g, 0.0) * ww                        # inf stays inf; zero-weight taps contribute nothing
        den += ww
    with np.errstate(invalid='ignore', divide='ignore'):
        out = np.where(den > 0.0, num / den, np.nan)
    return np.moveaxis(out, -1, axis)


def overview_cubicspline(arr, factor):
    """One CUBICSPLINE overview level (`gdal.BuildOverviews('CUBICSPLINE', ...)`, what `save_as_cog` asks for
    on non-integer layers, reference core.py:41-46): overview size ceil(N / factor), separable cubic B-spline
    convolution with the kernel stretched by the decimation ratio, horizontal pass then vertical pass.
    GDAL builds the levels of a non-NEAREST pyramid in cascade (each from the previous one); the caller
    does the same.  GDAL is not in the reference tree: the weights follow GDAL's published algorithm, the
    last-ulp agreement of the float32 results is UNPINNED."""
    h, w = arr.shape[-2:]
    oh, ow = (h + factor - 1) // factor, (w + factor - 1) // factor
    tmp = _convolve_axis(np.asarray(arr), ow, arr.ndim - 1)
    return np.ascontiguousarray(_convolve_axis(tmp, oh, arr.ndim - 2).astype(arr.dtype))


def overview_nearest(arr, factor):
    """One NEAREST overview level the way `gdal.BuildOverviews('NEAREST', ...)` picks source
    pixels from the full-resolution band (GDALResampleChunk_Near): overview size
    ceil(N / factor), src = int(0.5 + dst * N / N_ovr).  Used by `save_as_cog`
    (reference core.py:37-46: integer layers -> NEAREST, factors 4, 16, 64, 128).
    GDAL is not in the reference tree: parity of this selection rule is unpinned."""
    h, w = arr.shape[-2:]
    oh, ow = (h + factor - 1) // factor, (w + factor - 1) // factor
    ys = np.minimum((0.5 + np.arange(oh) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum((0.5 + np.arange(ow) * (w / ow)).astype(np.int64), w - 1)
    return np.ascontiguousarray(arr[..., ys[:, None], xs[None, :]])


_pool = None


def _io_pool():
    """Thread pool for block (de)compression: zlib and the numpy block copies release the GIL, so
    DEFLATE -- which dominates the wall time of a product run -- scales over host cores.  The
    encoded bytes do not depend on the thread count.  DSWX_IO_THREADS overrides (1 = serial)."""
    global _pool
    n = int(os.environ.get('DSWX_IO_THREADS', '0')) or min(32, os.cpu_count() or 1)
    if n <= 1:
        return None
    if _pool is None or _pool._max_workers != n:
        from concurrent.futures import ThreadPoolExecutor
        _pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix='dswx-io')
    return _pool


_codec_state = {'mod': None, 'tried': False}


def _codec():
    """proteus_amd.codec (native DEFLATE on a thread pool) or None -- said once on stderr -- when its library can be
    neither loaded nor built (no C++ compiler): Python's zlib on Python threads then writes the same files, slower."""
    if not _codec_state['tried']:
        _codec_state['tried'] = True
        try:
            from . import codec
            codec.load()
            _codec_state['mod'] = codec
        except Exception as e:                  # noqa: BLE001
            import sys
            print(f'[dswx geotiff] native codec unavailable ({type(e).__name__}: {str(e)[:200]}); using the zlib module',
                  file=sys.stderr, flush=True)
    return _codec_state['mod']


class BlockedLevel:
    """One resolution level of a raster as the TIFF writer stores it: `data` = uint8 [n_blocks * block_bytes], the
    tile x tile blocks (band-major, then row-major; edge blocks zero-padded) with the predictor already applied, little
    endian.  blocked_level() makes one on the host; the GPU pipeline makes them on the device (dswx_cog_blocks_device)."""

    def __init__(self, height, width, bands, dtype, tile, predictor, data):
        self.height, self.width, self.bands = int(height), int(width), int(bands)
        self.dtype, self.tile, self.predictor, self.data = np.dtype(dtype), int(tile), int(predictor), data
        self.across, self.down = (self.width + tile - 1) // tile, (self.height + tile - 1) // tile
        self.block_bytes = tile * tile * self.dtype.itemsize
        self.n_blocks = self.bands * self.across * self.down


def blocked_level(arr, tile, predictor):
    """[B,H,W] -> BlockedLevel (whole-array numpy operations)."""
    B, H, W = arr.shape
    dt = arr.dtype.newbyteorder('<')
    across, down = (W + tile - 1) // tile, (H + tile - 1) // tile
    with stages.span('write: tile + predictor'):
        pad = np.zeros((B, down * tile, across * tile), dtype=dt)
        pad[:, :H, :W] = arr
        blk = np.ascontiguousarray(pad.reshape(B, down, tile, across, tile).transpose(0, 1, 3, 2, 4))
        if predictor == 2:
            d = blk.copy()
            d[..., 1:] -= blk[..., :-1]              # integer arithmetic wraps, as libtiff's horizontal differencing does
            blk = d
        if predictor == 3:
            data = np.frombuffer(_fp_predictor_encode(blk.reshape(-1, tile)), dtype=np.uint8)
        else:
            data = blk.reshape(-1).view(np.uint8)
    return BlockedLevel(H, W, B, arr.dtype, tile, predictor, data)


def _encode_level(level, compress):
    """BlockedLevel -> list of the encoded blocks (buffer objects), in block order."""
    bb, n = level.block_bytes, level.n_blocks
    raw = level.data.reshape(-1).view(np.uint8)
    if not compress:
        return [raw[i * bb:(i + 1) * bb] for i in range(n)]
    with stages.span('write: deflate'):
        native = _codec()
        if native is not None:
            out, offs, sizes = native.deflate_uniform(raw[:n * bb], bb, 6)
            return [out[o:o + k] for o, k in zip(offs.tolist(), sizes.tolist())]
        pool = _io_pool() if n > 1 else None
        enc = lambda i: zlib.compress(raw[i * bb:(i + 1) * bb], 6)      # noqa: E731
        return list(pool.map(enc, range(n))) if pool else [enc(i) for i in range(n)]


def write_geotiff(path, array, *, geo_tags=None, metadata=None, nodata=None,
                  descriptions=None, colormap=None, tile=512, compress=True, overviews=None, levels=None):
    """array: [H,W] or [B,H,W] (planar multi-band).  colormap: {value: (r,g,b[,a])} or
    [256][3] (single-band u8 only).  NaN nodata is written as 'nan' like GDAL does.

    `overviews`: decimation factors (e.g. COG_OVERVIEW_FACTORS) -> internal reduced-resolution
    IFDs in cloud-optimized order, as `save_as_cog` (reference core.py:7-91) leaves the file:
    all IFDs first (main, then overviews by descending size), then the block data of the
    smallest overview ... the largest overview, the full-resolution image last; 512 x 512 tiles,
    DEFLATE; integer types: NEAREST overviews + PREDICTOR=2, floating point: CUBICSPLINE
    overviews (cascaded, as GDAL builds them) + PREDICTOR=3 (core.py:37-46, :66-69).
    `levels`: instead of `array` (pass None), the full-resolution image and its overviews ALREADY tiled and
    predictor-encoded (list of BlockedLevel, level 0 first: what dswx_cog_blocks_device leaves in page-locked
    memory) -- the host then only deflates.
    The file is written under a temporary name and renamed when complete, so a reader (or the
    batch driver's --skip-existing) never sees a truncated product."""
    if levels is not None:
        if array is not None or not levels:
            raise GeoTiffError('pass either `array` or a non-empty `levels`')
        lv0 = levels[0]
        B, dtype = lv0.bands, lv0.dtype
        tile = lv0.tile
        predictor = lv0.predictor if compress else 1
        if any(lv.tile != tile or lv.bands != B or lv.dtype != dtype or lv.predictor != lv0.predictor for lv in levels):
            raise GeoTiffError('levels disagree about tile / bands / dtype / predictor')
        if not compress and lv0.predictor != 1:
            raise GeoTiffError('predictor-encoded levels need compress=True')
        kind = {'u': 1, 'i': 2, 'f': 3}.get(dtype.kind)
    else:
        arr = np.asarray(array)
        if arr.ndim == 2:
            arr = arr[None]
        if arr.ndim != 3:
            raise GeoTiffError('array must be [H,W] or [B,H,W]')
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint8)
        B, dtype = arr.shape[0], arr.dtype
        kind = {'u': 1, 'i': 2, 'f': 3}.get(dtype.kind)
    if kind is None or dtype.itemsize not in (1, 2, 4, 8):
        raise GeoTiffError(f'unsupported dtype {dtype}')
    if levels is None:
        predictor = (3 if (kind == 3 and dtype.itemsize in (4, 8)) else 2) if compress else 1
    palette = colormap is not None and B == 1 and dtype == np.uint8
    cm_values = None
    if palette:
        cm = np.zeros((256, 3), dtype=np.uint32)
        if isinstance(colormap, dict):
            for v, rgb in colormap.items():
                cm[int(v)] = rgb[:3]
        else:
            cm[:] = np.asarray(colormap)[:, :3]
        cm_values = (cm.T.reshape(-1) * 257).tolist()
    nodata_text = None
    if nodata is not None:
        nodata_text = 'nan' if (isinstance(nodata, float) and np.isnan(nodata)) else \
            (str(int(nodata)) if float(nodata).is_integer() else repr(float(nodata)))

    if levels is None:
        # level 0 = full resolution, then the overviews by descending size
        rasters = [arr]
        prev_f = 1
        t_ovr = time.perf_counter()
        for f in (overviews or ()):
            if f > 1 and (arr.shape[1] > 1 or arr.shape[2] > 1):
                if kind == 3:
                    # cascade: level f from the previous level when it divides evenly (GDAL's
                    # GDALRegenerateCascadingOverviews), else from the full-resolution image
                    lv = overview_cubicspline(rasters[-1], int(f) // prev_f) if (prev_f > 1 and int(f) % prev_f == 0) \
                        else overview_cubicspline(arr, int(f))
                    want = ((arr.shape[1] + int(f) - 1) // int(f), (arr.shape[2] + int(f) - 1) // int(f))
                    if lv.shape[1:] != want:          # ceil of a ceil can differ by one: take it from the full image
                        lv = overview_cubicspline(arr, int(f))
                    rasters.append(lv)
                    prev_f = int(f)
                else:
                    rasters.append(overview_nearest(arr, int(f)))
        if overviews:
            stages.add('write: overviews (CUBICSPLINE)' if kind == 3 else 'write: overviews (NEAREST)', t_ovr, time.perf_counter())
        levels = [blocked_level(r, tile, predictor) for r in rasters]
    level_blocks = [_encode_level(lv, compress) for lv in levels]

    class _Shape:           # what entries_of reads of a level: [bands, height, width]
        def __init__(self, lv):
            self.shape = (lv.bands, lv.height, lv.width)
    levels = [_Shape(lv) for lv in levels]
    itemsize = dtype.itemsize

    def entries_of(k):
        lv = levels[k]
        ent = []
        if k > 0:
            ent.append((TAG_NEW_SUBFILE_TYPE, 4, [1]))       # reduced-resolution image
        ent += [(TAG_WIDTH, 4, [lv.shape[2]]), (TAG_LENGTH, 4, [lv.shape[1]]),
                (TAG_BITS, 3, [itemsize * 8] * B),
                (TAG_COMPRESSION, 3, [8 if compress else 1]),
                (TAG_PHOTOMETRIC, 3, [3 if palette else 1]),
                (TAG_SAMPLES, 3, [B]), (TAG_PLANAR, 3, [2 if B > 1 else 1])]
        if predictor != 1:
            ent.append((TAG_PREDICTOR, 3, [predictor]))
        if palette:
            ent.append((TAG_COLORMAP, 3, cm_values))
        ent += [(TAG_TILE_W, 3, [tile]), (TAG_TILE_L, 3, [tile]),
                (TAG_TILE_OFFSETS, 4, None),                  # patched below
                (TAG_TILE_COUNTS, 4, [len(x) for x in level_blocks[k]])]
        if B > 1:
            ent.append((TAG_EXTRA_SAMPLES, 3, [0] * (B - 1)))
        ent.append((TAG_SAMPLE_FORMAT, 3, [kind] * B))
        if k == 0:
            for t, (typ, vals) in (geo_tags or {}).items():
                ent.append((t, typ, vals))
            ent.append((TAG_GDAL_METADATA, 2, _metadata_xml(metadata, descriptions)))
        if nodata_text is not None:
            ent.append((TAG_GDAL_NODATA, 2, nodata_text))
        ent.sort(key=lambda x: x[0])
        return ent

    def pack(typ, values):
        if typ == 2:
            return values.encode('latin-1', 'replace') + b'\x00'
        return struct.pack('<' + _TYPE_FMT[typ] * len(values), *values)

    # layout: header | IFD 0 + its out-of-line data | IFD 1 + data | ... | blocks of the
    # smallest level ... blocks of level 0
    all_entries = [entries_of(k) for k in range(len(levels))]
    ifd_offs, payload_offs = [], []
    cursor = 8
    for k, ent in enumerate(all_entries):
        ifd_offs.append(cursor)
        cursor += 2 + 12 * len(ent) + 4
        pay = {}
        for tag, typ, values in ent:
            size = 4 * len(level_blocks[k]) if tag == TAG_TILE_OFFSETS else len(pack(typ, values))
            if size > 4:
                pay[tag] = cursor
                cursor += size + (size & 1)
        payload_offs.append(pay)
    block_offs = [None] * len(levels)
    for k in range(len(levels) - 1, -1, -1):
        offs = []
        for x in level_blocks[k]:
            offs.append(cursor)
            cursor += len(x) + (len(x) & 1)
        block_offs[k] = offs
    if cursor >= 2 ** 32:
        raise GeoTiffError('file would exceed 4 GiB (BigTIFF not supported)')
    tmp_path = f'{path}.{os.getpid()}.{id(level_blocks):x}.tmp'
    with stages.span('write: file'), open(tmp_path, 'wb') as fh:
        fh.write(struct.pack('<2sHI', b'II', 42, ifd_offs[0]))
        for k, ent in enumerate(all_entries):
            fh.write(struct.pack('<H', len(ent)))
            tail = []
            for tag, typ, values in ent:
                if tag == TAG_TILE_OFFSETS:
                    values = block_offs[k]
                raw = pack(typ, values)
                count = len(raw) if typ == 2 else len(values)
                if len(raw) <= 4:
                    fh.write(struct.pack('<HHI', tag, typ, count) + raw.ljust(4, b'\x00'))
                else:
                    fh.write(struct.pack('<HHII', tag, typ, count, payload_offs[k][tag]))
                    tail.append(raw + (b'\x00' if len(raw) & 1 else b''))
            fh.write(struct.pack('<I', ifd_offs[k + 1] if k + 1 < len(levels) else 0))
            for raw in tail:
                fh.write(raw)
        for k in range(len(levels) - 1, -1, -1):
            for x in level_blocks[k]:
                fh.write(x)
                if len(x) & 1:
                    fh.write(b'\x00')
    os.replace(tmp_path, path)


def cog_layout(path):
    """Structural facts the reference's COG validator checks
    (extern/validate_cloud_optimized_geotiff.py:176-297), read straight from the TIFF
    directory chain: [{'ifd_offset', 'width', 'height', 'tile', 'first_block', 'reduced'}]
    for the main image and every overview, in file order."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    e = '<' if buf[:2] == b'II' else '>'
    (ifd,) = struct.unpack(e + 'I', buf[4:8])
    out = []
    while ifd:
        (n,) = struct.unpack(e + 'H', buf[ifd:ifd + 2])
        tags = {}
        for k in range(n):
            ent = buf[ifd + 2 + 12 * k: ifd + 14 + 12 * k]
            tag, typ, count = struct.unpack(e + 'HHI', ent[:8])
            size = _TYPE_SIZE.get(typ, 0) * count
            if typ in (3, 4) and size:
                raw = ent[8:8 + size] if size <= 4 else \
                    buf[struct.unpack(e + 'I', ent[8:12])[0]:][:size]
                tags[tag] = list(struct.unpack(e + _TYPE_FMT[typ] * count, raw))
        out.append({'ifd_offset': ifd, 'width': tags[TAG_WIDTH][0], 'height': tags[TAG_LENGTH][0],
                    'tile': (tags.get(TAG_TILE_W, [0])[0], tags.get(TAG_TILE_L, [0])[0]),
                    'first_block': min(tags[TAG_TILE_OFFSETS]) if TAG_TILE_OFFSETS in tags else 0,
                    'reduced': bool(tags.get(TAG_NEW_SUBFILE_TYPE, [0])[0] & 1)})
        (ifd,) = struct.unpack(e + 'I', buf[ifd + 2 + 12 * n: ifd + 6 + 12 * n])
    return out


def validate_cog(path):
    """Returns the list of layout errors (empty = valid), the rules of the reference's
    validator (extern/validate_cloud_optimized_geotiff.py): main IFD at byte 8 (:176-213),
    tiled when larger than 512 (:163-168), overviews by descending size with ascending
    IFD offsets (:218-258), block data from the smallest overview to the main image,
    all after the last IFD (:281-297)."""
    lv = cog_layout(path)
    errors = []
    main, ovr = lv[0], lv[1:]
    if main['ifd_offset'] != 8:
        errors.append(f"The offset of the main IFD should be 8. It is {main['ifd_offset']} instead")
    if (main['width'] > 512 or main['height'] > 512) and not main['tile'][0]:
        errors.append('The file is greater than 512xH or Wx512, but is not tiled')
    prev = main
    for i, o in enumerate(ovr):
        if not o['reduced']:
            errors.append(f'IFD {i + 1} is not a reduced-resolution image')
        if o['width'] > prev['width'] or o['height'] > prev['height']:
            errors.append(f'Overview of index {i} has larger dimension than the previous level')
        if o['ifd_offset'] < prev['ifd_offset']:
            errors.append(f'The offset of the IFD for overview of index {i} is not ascending')
        prev = o
    if lv[-1]['first_block'] < lv[-1]['ifd_offset']:
        errors.append('The offset of the first block of the smallest level should be after its IFD')
    for i in range(len(lv) - 1):
        if lv[i]['first_block'] < lv[i + 1]['first_block']:
            errors.append(f'The offset of the first block of level {i} should be after the one of level {i + 1}')
    return errors


def write_png_palette(path, indices, colormap, transparent_index=None):
    """8-bit palette PNG (what `gdal.Translate(format='PNG')` produces from a paletted
    Byte GeoTIFF): `indices` uint8 [H,W], `colormap` {value: (r,g,b)} or [256][3]."""
    idx = np.ascontiguousarray(indices, dtype=np.uint8)
    if idx.ndim != 2:
        raise GeoTiffError('PNG writer needs a 2-D uint8 array')
    h, w = idx.shape
    pal = np.zeros((256, 3), dtype=np.uint8)
    if isinstance(colormap, dict):
        for v, rgb in colormap.items():
            pal[int(v)] = rgb[:3]
    elif colormap is not None:
        pal[:] = np.asarray(colormap)[:, :3]

    def chunk(tag, data):
        body = tag + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)

    raw = np.empty((h, w + 1), dtype=np.uint8)
    raw[:, 0] = 0                      # filter type 0 on every scanline
    raw[:, 1:] = idx
    out = [b'\x89PNG\r\n\x1a\n', chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 3, 0, 0, 0)),
           chunk(b'PLTE', pal.tobytes())]
    if transparent_index is not None:
        alpha = bytearray([255] * 256)
        alpha[int(transparent_index)] = 0
        out.append(chunk(b'tRNS', bytes(alpha)))
    out.append(chunk(b'IDAT', zlib.compress(raw.tobytes(), 6)))
    out.append(chunk(b'IEND', b''))
    tmp_path = f'{path}.{os.getpid()}.{id(idx):x}.tmp'
    with open(tmp_path, 'wb') as fh:
        fh.write(b''.join(out))
    os.replace(tmp_path, path)


def resample_nearest(arr, out_height, out_width):
    """Nearest-neighbour decimation the way GDAL RasterIO picks source pixels:
    src = floor((dst + 0.5) * src_size / dst_size)."""
    ys, xs = resample_nearest_indices(arr.shape[-2], arr.shape[-1], out_height, out_width)
    return arr[..., ys[:, None], xs[None, :]]


def resample_nearest_indices(h, w, out_height, out_width):
    """(source rows [out_height], source columns [out_width]) of resample_nearest (the device gather takes them too)."""
    ys = np.minimum(((np.arange(out_height) + 0.5) * h / out_height).astype(np.int64), h - 1)
    xs = np.minimum(((np.arange(out_width) + 0.5) * w / out_width).astype(np.int64), w - 1)
    return ys, xs
