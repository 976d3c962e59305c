"""The product run on the device between the two codecs.

    band files --inflate (host threads, proteus_amd.codec)--> blocks in page-locked memory
        --copy engine--> HBM --dswx_untile_device--> planes
        --dswx_classify_device_2d--> layers (HBM)
        --dswx_cog_blocks_device (blocks + NEAREST overviews + predictor) / dswx_rgb_planes_device--> HBM
        --copy engine--> page-locked memory --deflate (host threads)--> GeoTIFF / COG

The host touches pixels only inside the DEFLATE codec; everything else the reference's GDAL calls do to a raster
(`ReadAsArray`, `save_as_cog`: core.py:7-91) is byte shuffling that runs in HBM.  One TileEngine per context; its
lock serialises the GPU part (milliseconds per tile), so that several tiles can be IN FLIGHT in one process -- tile
k + 1 inflating and tile k - 1 deflating on host threads while tile k is on the device (proteus_amd.batch) -- with one
HIP context per GPU.

No fallback hides here: a file layout the device kernels do not take (floating-point predictor, several samples per
pixel, big endian) is untiled by the host reader (geotiff.TiffDirectory.untile, the same bytes) and uploaded; the
classification itself has no host form anywhere.
"""
import threading
import time

import numpy as np

from . import _capi, geotiff, stages


class DevicePlane:
    """A raster in HBM: device pointer + shape + dtype.  The memory goes back to the engine's pool when the object dies."""

    def __init__(self, engine, buf, shape, dtype):
        self.engine, self.buf = engine, buf
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._host = None

    @property
    def ptr(self):
        return self.buf.ptr

    def numpy(self):
        """The raster on the host (copied once, then kept)."""
        if self._host is None:
            self._host = self.engine.download(self)
        return self._host

    def release(self):
        if self.buf is not None:
            self.engine._give(self.buf)
            self.buf = None

    def __del__(self):
        try:
            self.release()
        except Exception:           # noqa: BLE001
            pass


class _PlaneView:
    """A raster inside another device buffer (no ownership): what the writer-side helpers need of a plane."""

    def __init__(self, ptr, shape, dtype, keep=None):
        self.ptr, self.shape, self.dtype, self._keep = int(ptr), tuple(shape), np.dtype(dtype), keep


class TileEngine:
    POOL_CAP = 6 << 30          # bytes of released device buffers kept for the next tile

    def __init__(self, ctx):
        self.ctx = ctx
        self.lock = threading.RLock()
        self._free = {}                     # nbytes -> [DeviceBuffer]
        self._free_bytes = 0
        self._weights = {}                  # (n_in, n_out) -> (first DeviceBuffer, weights DeviceBuffer, taps): CUBICSPLINE passes
        self._pool_lock = threading.RLock()   # re-entrant: a plane's finalizer (_give) may run -- cyclic GC -- on a thread that is inside _take

    # ---- device memory ------------------------------------------------------------------------------------
    @staticmethod
    def _round(nbytes):
        return max((int(nbytes) + 0xffff) & ~0xffff, 1 << 16)

    def _take(self, nbytes):
        n = self._round(nbytes)
        with self._pool_lock:
            spare = self._free.get(n)
            if spare:
                self._free_bytes -= n
                return spare.pop()
        with self.lock:
            return self.ctx.malloc(n)

    def _give(self, buf):
        with self._pool_lock:
            if self.ctx.handle and self._free_bytes + buf.nbytes <= self.POOL_CAP:
                self._free.setdefault(buf.nbytes, []).append(buf)
                self._free_bytes += buf.nbytes
                return
        buf.free()

    def close(self):
        with self._pool_lock:
            spare, self._free, self._free_bytes = self._free, {}, 0
            weights, self._weights = self._weights, {}
        for d_first, d_w, _ in weights.values():
            d_first.free()
            d_w.free()
        for bufs in spare.values():
            for b in bufs:
                b.free()

    def plane(self, shape, dtype):
        dtype = np.dtype(dtype)
        return DevicePlane(self, self._take(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize), shape, dtype)

    def upload(self, arr):
        """numpy raster -> DevicePlane (through page-locked memory)."""
        arr = np.ascontiguousarray(arr)
        if arr.dtype == np.bool_:
            arr = arr.view(np.uint8)
        p = self.plane(arr.shape, arr.dtype)
        pinned = arr if self.ctx.is_pinned(arr) else None
        if pinned is None:
            pinned = self.ctx.pinned_empty(arr.shape, arr.dtype)
            np.copyto(pinned, arr)
        with self.lock, stages.span('gpu: upload'):
            self.ctx.h2d_async(p.ptr, pinned)
            self.ctx.synchronize()
        return p

    def download(self, plane):
        out = self.ctx.pinned_empty(plane.shape, plane.dtype)
        with self.lock, stages.span('gpu: download'):
            self.ctx.d2h_async(out, plane.ptr, plane.nbytes)
            self.ctx.synchronize()
        return out

    # ---- reader side ----------------------------------------------------------------------------------------
    def device_untile_ok(self, d):
        """Layouts dswx_untile_device takes: one sample per pixel, little endian; 1- / 2- / 4-byte integers with PREDICTOR
        1 / 2, Float32 with PREDICTOR 1 / 3 (a GDAL-written DEM)."""
        if d.spp != 1 or d.dt.byteorder not in ('=', '|', '<') or d.n_blocks * d.block_bytes >= (1 << 32):
            return False
        if d.dt.kind in 'iu':
            return d.dt.itemsize in (1, 2, 4) and d.predictor in (1, 2)
        return d.dt.kind == 'f' and d.dt.itemsize == 4 and d.predictor in (1, 3)

    def read_plane(self, path):
        return self.read_directory(geotiff.open_geotiff(path))

    def read_directory(self, d):
        """One single-band GeoTIFF (geotiff.TiffDirectory) -> (DevicePlane [H, W], GeoTiffInfo).  Inflate on host threads
        into page-locked memory; predictor + untile on the device when its kernel takes the layout, else by the host
        reader (same bytes)."""
        info = d.info
        if d.spp != 1:
            raise geotiff.GeoTiffError(f'{d.path}: {d.spp} samples per pixel, expected a single band')
        if not self.device_untile_ok(d):
            return self.upload(d.untile(d.inflate())[0]), info
        need = d.n_blocks * d.block_bytes
        staging = self.ctx.pinned_empty((need,), np.uint8)
        d.inflate(staging)
        dev_blocks = self._take(need)
        plane = self.plane((info.height, info.width), info.dtype)
        try:
            with self.lock, stages.span('gpu: blocks -> plane (h2d + untile)'):
                self.ctx.h2d_async(dev_blocks.ptr, staging, need)
                self.ctx.untile_device(dev_blocks.ptr, d.dt.itemsize, info.height, info.width, d.bw, d.bh, d.predictor, plane.ptr)
                self.ctx.synchronize()
        finally:
            self._give(dev_blocks)
        return plane, info

    # ---- the classifier on resident planes --------------------------------------------------------------------
    def classify(self, bands, fmask, params, land=None, shad=None, ocean=None, layers=()):
        """bands: six DevicePlanes (int16 [H, W]); fmask / land / shad / ocean: DevicePlane or ndarray or None.
        Returns {layer: DevicePlane} + 'counters' (ndarray int64 [1, 3])."""
        H, W = bands[0].shape
        keep = []

        def dev(a, name):
            if a is None:
                return None
            if not isinstance(a, DevicePlane):
                a = np.ascontiguousarray(a, dtype=np.uint8)
                if a.shape != (H, W):
                    raise ValueError(f'{name} shape {a.shape} != bands shape {(H, W)}')
                a = self.upload(a)
            elif a.shape != (H, W):
                raise ValueError(f'{name} shape {a.shape} != bands shape {(H, W)}')
            keep.append(a)
            return a

        pin, pout = _capi.PlanesIn(), _capi.PlanesOut()
        for i, b in enumerate(bands):
            if b.shape != (H, W) or b.dtype != np.int16:
                raise ValueError('bands must be int16 planes of one shape')
            pin.band[i] = b.ptr
        pin.fmask = dev(fmask, 'fmask').ptr
        for name, a in (('land', land), ('shad', shad), ('ocean', ocean)):
            p = dev(a, name)
            setattr(pin, name, p.ptr if p is not None else None)
        res = {}
        for name in layers:
            if name != 'diag' and name not in _capi.U8_LAYERS:
                raise KeyError(name)
            res[name] = self.plane((H, W), np.uint16 if name == 'diag' else np.uint8)
            setattr(pout, name, res[name].ptr)
        d_cnt = self._take(256)
        cnt = self.ctx.pinned_empty((1, 3), np.int64)
        try:
            with self.lock:
                t0 = time.perf_counter()
                self.ctx.classify_device_2d(params, 1, H, W, pin, pout, d_cnt.ptr)
                self.ctx.d2h_async(cnt, d_cnt.ptr, 24)
                self.ctx.synchronize()
                stages.add('gpu: classify (resident planes)', t0, time.perf_counter())
                self.kernel_info = self.ctx.last_kernel_info()
        finally:
            self._give(d_cnt)
        res['counters'] = np.array(cnt, dtype=np.int64)
        del keep
        return res

    # ---- the layers next to the path, resident --------------------------------------------------------------------
    def crop(self, plane, margin):
        """_crop_2d_array_all_sides (dswx_hls.py:4320) on the device: plane[margin:-margin, margin:-margin]."""
        if margin == 0:
            return plane
        H, W = plane.shape
        out = self.plane((H - 2 * margin, W - 2 * margin), plane.dtype)
        es = plane.dtype.itemsize
        with self.lock, stages.span('gpu: crop'):
            self.ctx.copy_2d_device(out.ptr, out.shape[1] * es, plane.ptr + (margin * W + margin) * es, W * es,
                                    out.shape[1] * es, out.shape[0])
            self.ctx.synchronize()
        return out

    def shadow_layer(self, dem, sun_vector, sin_azimuth, cos_azimuth, min_slope_angle, max_sun_local_inc_angle, margin,
                     float32, pixel_spacing_x=30, pixel_spacing_y=30):
        """dem: DevicePlane float32 [H, W] (with its margin) -> SHAD DevicePlane u8 [H - 2 m, W - 2 m] (1 = not shadow)."""
        H, W = dem.shape
        out = self.plane((H - 2 * margin, W - 2 * margin), np.uint8)
        with self.lock, stages.span('gpu: shadow layer'):
            self.ctx.shadow_layer_device(dem.ptr, 1, H, W, margin, sun_vector, sin_azimuth, cos_azimuth, min_slope_angle,
                                         max_sun_local_inc_angle, out.ptr, pixel_spacing_x, pixel_spacing_y, float32=float32)
            self.ctx.synchronize()
        return out

    def landcover_mask(self, worldcover_up3, copernicus, forest_classes, thresholds, year_offset):
        """WorldCover [3H, 3W] + CGLS [H, W] (DevicePlanes, u8) -> LAND DevicePlane [H, W] (create_landcover_mask's
        per-pixel part, :994-1115)."""
        H, W = copernicus.shape
        if worldcover_up3.shape != (3 * H, 3 * W):
            raise ValueError(f'WorldCover raster {worldcover_up3.shape} is not three times the CGLS grid {(H, W)}')
        out = self.plane((H, W), np.uint8)
        with self.lock, stages.span('gpu: LAND aggregation'):
            self.ctx.landcover_mask_device(worldcover_up3.ptr, copernicus.ptr, 1, H, W, forest_classes, out.ptr,
                                           thresholds=thresholds, year_offset=year_offset)
            self.ctx.synchronize()
        return out

    # ---- writer side ----------------------------------------------------------------------------------------
    WEIGHTS_CAP = 64            # (n_in, n_out) pairs kept: a worker sees one tile size, i.e. about ten pairs

    def _conv_weights(self, n_in, n_out):
        """Device copies of geotiff.convolve_weights(n_in, n_out), cached.  Called with the engine's lock HELD."""
        key = (int(n_in), int(n_out))
        hit = self._weights.get(key)
        if hit is None:
            first, w = geotiff.convolve_weights(*key)
            d_first, d_w = self.ctx.malloc(max(first.size * 4, 16)), self.ctx.malloc(max(w.nbytes, 16))
            d_first.upload(first.astype(np.int32))
            d_w.upload(np.ascontiguousarray(w.T))               # [taps][n_out]: the layout the device pass reads
            hit = self._weights[key] = (d_first, d_w, w.shape[1])
        return hit

    def cubicspline_overview(self, plane, factor):
        """One CUBICSPLINE overview level of a Float32 plane on the device (geotiff.overview_cubicspline: horizontal pass
        into float64, vertical pass, float32): what `save_as_cog` asks GDAL for on non-integer layers (core.py:41-46)."""
        H, W = plane.shape
        oh, ow = -(-H // factor), -(-W // factor)
        tmp = self._take(H * ow * 8)
        out = self.plane((oh, ow), np.float32)
        try:
            with self.lock, stages.span('gpu: CUBICSPLINE overview'):
                if len(self._weights) + 2 > self.WEIGHTS_CAP:   # rasters of ever-changing sizes: start over -- here, under the
                    for d_first, d_w, _ in self._weights.values():      # lock and before this call's own lookups, so that no
                        d_first.free()                                  # launch ever reads a freed entry
                        d_w.free()
                    self._weights.clear()
                fx, wx, tx = self._conv_weights(W, ow)
                fy, wy, ty = self._conv_weights(H, oh)
                self.ctx.convolve_axis_device(plane.ptr, False, H, W, W, 1, ow, tx, fx.ptr, wx.ptr, tmp.ptr, True, ow, 1)
                self.ctx.convolve_axis_device(tmp.ptr, True, ow, H, 1, ow, oh, ty, fy.ptr, wy.ptr, out.ptr, False, 1, ow)
                self.ctx.synchronize()
        finally:
            self._give(tmp)
        return out

    def _float_pyramid(self, plane, factors):
        """The levels write_geotiff builds for a floating-point layer: cascaded (each from the previous one when the factor
        divides, GDAL's GDALRegenerateCascadingOverviews), else from the full-resolution raster."""
        H, W = plane.shape
        rasters, prev_f = [plane], 1
        for f in factors:
            f = int(f)
            if f > 1 and (H > 1 or W > 1):
                lv = self.cubicspline_overview(rasters[-1], f // prev_f) if (prev_f > 1 and f % prev_f == 0) \
                    else self.cubicspline_overview(plane, f)
                if lv.shape != (-(-H // f), -(-W // f)):            # ceil of a ceil can differ by one: from the full image
                    lv = self.cubicspline_overview(plane, f)
                rasters.append(lv)
                prev_f = f
        return rasters

    def layer_levels(self, plane, factors=(), tile=512):
        """DevicePlane [H, W] -> [geotiff.BlockedLevel]: the blocks of the full-resolution image and of every overview,
        predictor-encoded, in page-locked memory.  Integer layers (u8 / u16 / i16): NEAREST overviews + PREDICTOR=2, one
        launch; Float32: CUBICSPLINE overviews (cascaded) + PREDICTOR=3 (core.py:37-46, :66-69)."""
        if plane.dtype.kind == 'f':
            return self.float_levels([plane], factors, tile)
        H, W = plane.shape
        dt = plane.dtype
        predictor = 2
        lay = _capi.cog_layout(H, W, dt.itemsize, factors, tile)
        total = lay['total_bytes']
        dev_blocks = self._take(total)
        host = self.ctx.pinned_empty((total,), np.uint8)
        try:
            with self.lock, stages.span('gpu: plane -> COG blocks (+ d2h)'):
                self.ctx.cog_blocks_device(plane.ptr, dt.itemsize, H, W, dev_blocks.ptr, factors, tile, predictor)
                self.ctx.d2h_async(host, dev_blocks.ptr, total)
                self.ctx.synchronize()
        finally:
            self._give(dev_blocks)
        out = []
        for k, lv in enumerate(lay['levels']):
            n = lv['blocks_down'] * lv['blocks_across'] * tile * tile * dt.itemsize
            out.append(geotiff.BlockedLevel(lv['height'], lv['width'], 1, dt, tile, predictor,
                                            host[lv['offset_bytes']: lv['offset_bytes'] + n]))
        return out

    def float_levels(self, bands, factors=(), tile=512):
        """Float32 planes of ONE file (1 band: the DEM layer; 3: an RGB composite) -> [geotiff.BlockedLevel] (planar:
        band-major inside every level), floating-point predictor, CUBICSPLINE overviews per band."""
        if any(b.dtype != np.float32 for b in bands):
            raise ValueError('float planes: float32 only')
        pyramids = [self._float_pyramid(b, factors) for b in bands]
        n_levels = len(pyramids[0])
        sizes = [_capi.cog_layout(pyramids[0][k].shape[0], pyramids[0][k].shape[1], 4, (), tile)['total_bytes'] for k in range(n_levels)]
        total = len(bands) * sum(sizes)
        dev_blocks = self._take(total)
        host = self.ctx.pinned_empty((total,), np.uint8)
        offs, cur = [], 0
        try:
            with self.lock, stages.span('gpu: Float32 planes -> blocks (+ d2h)'):
                for k in range(n_levels):
                    offs.append(cur)
                    for c in range(len(bands)):
                        lv = pyramids[c][k]
                        self.ctx.cog_blocks_device(lv.ptr, 4, lv.shape[0], lv.shape[1], dev_blocks.ptr + cur, (), tile, 3)
                        cur += sizes[k]
                self.ctx.d2h_async(host, dev_blocks.ptr, total)
                self.ctx.synchronize()
        finally:
            self._give(dev_blocks)
        return [geotiff.BlockedLevel(pyramids[0][k].shape[0], pyramids[0][k].shape[1], len(bands), np.float32, tile, 3,
                                     host[offs[k]: offs[k] + len(bands) * sizes[k]]) for k in range(n_levels)]

    def resample_nearest(self, plane, out_height, out_width):
        """geotiff.resample_nearest of a resident plane -> numpy [out_height, out_width] (the browse PNG's pixels): the gather
        runs in HBM, the small image is what crosses PCIe."""
        H, W = plane.shape
        ys, xs = geotiff.resample_nearest_indices(H, W, out_height, out_width)
        idx = np.concatenate([ys, xs]).astype(np.int32)
        n_out = int(out_height) * int(out_width) * plane.dtype.itemsize
        d_idx, d_out = self._take(max(idx.nbytes, 16)), self._take(max(n_out, 16))
        host = self.ctx.pinned_empty((int(out_height), int(out_width)), plane.dtype)
        pin_idx = self.ctx.pinned_empty(idx.shape, np.int32)
        np.copyto(pin_idx, idx)
        try:
            with self.lock, stages.span('gpu: resample (browse)'):
                self.ctx.h2d_async(d_idx.ptr, pin_idx)
                self.ctx.gather_2d_device(plane.ptr, plane.dtype.itemsize, H, W, d_idx.ptr, out_height, d_idx.ptr + 4 * int(out_height),
                                          out_width, d_out.ptr)
                if n_out:
                    self.ctx.d2h_async(host, d_out.ptr, n_out)
                self.ctx.synchronize()
        finally:
            self._give(d_idx)
            self._give(d_out)
        return host

    def byte_plane(self, plane):
        """A layer as GDAL stores it in a Byte band of the multi-band file (dswx_hls._gdal_byte; dswx_to_byte_device)."""
        if plane.dtype == np.uint8:
            return plane
        out = self.plane(plane.shape, np.uint8)
        with self.lock, stages.span('gpu: layer -> Byte band'):
            self.ctx.to_byte_device(plane.ptr, plane.dtype, plane.shape[0] * plane.shape[1], out.ptr)
            self.ctx.synchronize()
        return out

    def constant_plane(self, shape, value):
        out = self.plane(shape, np.uint8)
        with self.lock:
            _capi._check(self.ctx.lib.dswx_memset_d(self.ctx.handle, out.ptr, int(value) & 0xff, out.nbytes))
        return out

    def band_stack_levels(self, bands, factors=(), tile=512):
        """Byte planes of ONE multi-band file (save_dswx_product: ten bands, planar) -> [geotiff.BlockedLevel] (band-major
        inside every level): blocks + NEAREST overviews + PREDICTOR=2 of every band on the device, each (band, level) piece
        copied to its place in page-locked memory."""
        H, W = bands[0].shape
        if any(b.dtype != np.uint8 or b.shape != (H, W) for b in bands):
            raise ValueError('band stack: uint8 planes of one shape')
        lay = _capi.cog_layout(H, W, 1, factors, tile)
        per_band = lay['total_bytes']
        sizes = [lv['blocks_down'] * lv['blocks_across'] * tile * tile for lv in lay['levels']]
        starts = np.concatenate([[0], np.cumsum([len(bands) * n for n in sizes])]).astype(np.int64)
        dev_blocks = self._take(per_band)
        host = self.ctx.pinned_empty((len(bands) * per_band,), np.uint8)
        try:
            with self.lock, stages.span('gpu: plane -> COG blocks (+ d2h)'):
                for c, b in enumerate(bands):
                    self.ctx.cog_blocks_device(b.ptr, 1, H, W, dev_blocks.ptr, factors, tile, 2)
                    for k, lv in enumerate(lay['levels']):
                        at = int(starts[k]) + c * sizes[k]
                        self.ctx.d2h_async(host[at: at + sizes[k]], dev_blocks.ptr + lv['offset_bytes'], sizes[k])
                self.ctx.synchronize()
        finally:
            self._give(dev_blocks)
        return [geotiff.BlockedLevel(lv['height'], lv['width'], len(bands), np.uint8, tile, 2,
                                     host[int(starts[k]): int(starts[k + 1])]) for k, lv in enumerate(lay['levels'])]

    def rgb_levels(self, red, green, blue, diag, scale, offset, clip, tile=512, factors=()):
        """The three-band Float32 composite of _save_output_rgb_file (dswx_hls.py:3013-3036) as planar BlockedLevels
        (floating-point predictor, CUBICSPLINE overviews for `factors`): scaling on the device, NaN where `diag` carries the
        fill code."""
        H, W = red.shape
        n = H * W
        rgb = self._take(3 * n * 4)
        try:
            with self.lock, stages.span('gpu: RGB planes'):
                self.ctx.rgb_planes_device(red.ptr, green.ptr, blue.ptr, diag.ptr if diag is not None else None, n,
                                           scale, offset, clip, rgb.ptr)
                self.ctx.synchronize()
            views = [_PlaneView(rgb.ptr + c * n * 4, (H, W), np.float32, keep=rgb) for c in range(3)]
            return self.float_levels(views, factors, tile)
        finally:
            self._give(rgb)


_engines = {}
_engines_lock = threading.Lock()


def engine_of(ctx):
    with _engines_lock:
        e = _engines.get(id(ctx))
        if e is None or e.ctx is not ctx:
            e = _engines[id(ctx)] = TileEngine(ctx)
        return e
