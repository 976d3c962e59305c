"""ctypes binding of include/dswx_hip.h (the C-ABI of the HIP classifier).

There is exactly one compute path: the HIP library.  If it is missing, or no
MI355X is visible, every entry point here raises -- nothing falls back to numpy.
"""
import ctypes
import threading
import weakref
import os

import numpy as np

from . import build as _build

DSWX_ABI_VERSION = 6
OK, ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_UNSUPPORTED, ERR_ALIGN = 0, -1, -2, -3, -4, -5
ADJ_MODES = {'mask': 0, 'ignore': 1, 'cover': 2}
BAND_NAMES = ('blue', 'green', 'red', 'nir', 'swir1', 'swir2')
THRESHOLD_NAMES = ('wigt', 'awgt', 'pswt_1_mndwi', 'pswt_1_nir', 'pswt_1_swir1',
                   'pswt_1_ndvi', 'pswt_2_mndwi', 'pswt_2_blue', 'pswt_2_nir',
                   'pswt_2_swir1', 'pswt_2_swir2', 'lcmask_nir')
U8_LAYERS = ('wtr1', 'wtr1_aerosol', 'wtr2', 'wtr', 'bwtr', 'conf', 'cloud', 'browse')
F64_LAYERS = ('mndwi', 'ndvi', 'awesh')
# every symbol include/dswx_hip.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = (
    'dswx_abi_version', 'dswx_last_error', 'dswx_device_count', 'dswx_ctx_create',
    'dswx_ctx_destroy', 'dswx_params_default', 'dswx_classify_host',
    'dswx_classify_device', 'dswx_classify_device_2d', 'dswx_classify_batch', 'dswx_synth_batch', 'dswx_interpret_layer_host', 'dswx_shadow_layer_host', 'dswx_shadow_layer_device', 'dswx_shadow_thresholds', 'dswx_shadow_layer_host_q', 'dswx_shadow_layer_device_q', 'dswx_shadow_layer_host_q32', 'dswx_shadow_layer_device_q32', 'dswx_landcover_mask_host', 'dswx_landcover_mask_device',
    'dswx_synth_fill', 'dswx_device_malloc',
    'dswx_device_free', 'dswx_host_alloc', 'dswx_host_free', 'dswx_memcpy_h2d', 'dswx_memcpy_d2h', 'dswx_memset_d',
    'dswx_stream_synchronize', 'dswx_event_create', 'dswx_event_destroy',
    'dswx_event_record', 'dswx_event_elapsed_ms', 'dswx_last_kernel_info',
    'dswx_batch_layout', 'dswx_batch_create', 'dswx_batch_destroy', 'dswx_batch_planes', 'dswx_batch_info',
    'dswx_batch_classify', 'dswx_batch_synth', 'dswx_batch_place_search', 'dswx_batch_place_slide',
    'dswx_batch_va_budget', 'dswx_batch_pool_trim',
    'dswx_shadow_layer_batch', 'dswx_landcover_mask_batch',
    'dswx_cog_layout', 'dswx_cog_blocks_device', 'dswx_untile_device', 'dswx_rgb_planes_device', 'dswx_copy_2d_device', 'dswx_convolve_axis_device',
    'dswx_to_byte_device', 'dswx_gather_2d_device',
    'dswx_memcpy_h2d_async', 'dswx_memcpy_d2h_async')


class DswxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f'dswx_hip error {code}: {message}')
        self.code = code


class Params(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_double) for n in THRESHOLD_NAMES] +
                [('band_fill', ctypes.c_double * 6),
                 ('fmask_fill', ctypes.c_double),
                 ('aerosol_max_nir', ctypes.c_double),
                 ('clip_negative_reflectance', ctypes.c_int32),
                 ('mask_adjacent_to_cloud_mode', ctypes.c_int32),
                 ('apply_aerosol_class_remapping', ctypes.c_int32),
                 ('collapse_wtr_classes', ctypes.c_int32),
                 ('browse_exclude_psw_aggressive', ctypes.c_int32),
                 ('browse_not_water_to_nodata', ctypes.c_int32),
                 ('browse_cloud_to_nodata', ctypes.c_int32),
                 ('browse_snow_to_nodata', ctypes.c_int32),
                 ('browse_ocean_masked_to_nodata', ctypes.c_int32),
                 ('offset_and_scale_inputs', ctypes.c_int32),
                 ('aerosol_fmask_lut', (ctypes.c_uint8 * 256) * 4),
                 ('band_scale', ctypes.c_double * 6),
                 ('band_offset', ctypes.c_double * 6)])


class BatchGeom(ctypes.Structure):
    _fields_ = [('n_tiles', ctypes.c_int64), ('height', ctypes.c_int64), ('width', ctypes.c_int64),
                ('tile_stride', ctypes.c_int64)]


class PlanesIn(ctypes.Structure):
    _fields_ = [('band', ctypes.c_void_p * 6), ('fmask', ctypes.c_void_p),
                ('land', ctypes.c_void_p), ('shad', ctypes.c_void_p),
                ('ocean', ctypes.c_void_p)]


class PlanesOut(ctypes.Structure):
    _fields_ = ([('diag', ctypes.c_void_p)] +
                [(n, ctypes.c_void_p) for n in U8_LAYERS] +
                [(n, ctypes.c_void_p) for n in F64_LAYERS])


# resident batches (include/dswx_hip.h, ABI v4; v5: address-space accounting, DSWX_BATCH_ALL_TILES)
BATCH_ALL_TILES = -1
BATCH_MASKS, BATCH_WTR1_AEROSOL, BATCH_BROWSE, BATCH_SEPARATE_OUTPUTS, BATCH_SLIDING_OUTPUTS = 1, 2, 4, 1 << 10, 1 << 11
BATCH_MAX_PLANES = 20
PLANE_INDEX = dict(blue=0, green=1, red=2, nir=3, swir1=4, swir2=5, fmask=6, land=7, shad=8, ocean=9, diag=10,
                   wtr1=11, wtr1_aerosol=12, wtr2=13, wtr=14, bwtr=15, conf=16, cloud=17, browse=18, counters=19)


class BatchLayout(ctypes.Structure):
    _fields_ = [('tile_stride', ctypes.c_int64), ('arena_bytes', ctypes.c_uint64),
                ('plane_bytes', ctypes.c_uint64 * BATCH_MAX_PLANES),
                ('plane_offset', ctypes.c_uint64 * BATCH_MAX_PLANES),
                ('write_span_bytes', ctypes.c_uint64)]


class BatchInfo(ctypes.Structure):
    _fields_ = [('geom', BatchGeom), ('flags', ctypes.c_uint32), ('n_allocations', ctypes.c_int32),
                ('bytes_allocated', ctypes.c_uint64), ('search_candidates', ctypes.c_int32),
                ('search_probes', ctypes.c_int32), ('first_come_launch_ms', ctypes.c_float),
                ('kept_launch_ms', ctypes.c_float), ('va_reserved_bytes', ctypes.c_uint64),
                ('va_retired_bytes', ctypes.c_uint64), ('va_budget_bytes', ctypes.c_uint64),
                ('va_pooled_bytes', ctypes.c_uint64), ('note', ctypes.c_char * 256)]


COG_MAX_LEVELS = 8


class CogLayout(ctypes.Structure):
    _fields_ = [('n_levels', ctypes.c_int32), ('tile', ctypes.c_int32), ('factor', ctypes.c_int32 * COG_MAX_LEVELS),
                ('height', ctypes.c_int64 * COG_MAX_LEVELS), ('width', ctypes.c_int64 * COG_MAX_LEVELS),
                ('blocks_down', ctypes.c_int32 * COG_MAX_LEVELS), ('blocks_across', ctypes.c_int32 * COG_MAX_LEVELS),
                ('offset_bytes', ctypes.c_uint64 * COG_MAX_LEVELS), ('total_bytes', ctypes.c_uint64)]


_lib = None


def library_path():
    """The in-tree product library -- or another build of the same sources named by DSWX_HIP_LIB (tests: the build with
    the host code under UBSan, proteus_amd.build.build_ubsan)."""
    return os.environ.get('DSWX_HIP_LIB') or _build.LIB_PATH


_alt_libs = {}


def va_budget(new_budget_bytes=0):
    """dswx_batch_va_budget: the library's process-wide account of the address space its sliding ranges hold."""
    v = [ctypes.c_uint64() for _ in range(5)]
    _check(load_library().dswx_batch_va_budget(int(new_budget_bytes), *[ctypes.byref(x) for x in v]))
    return dict(zip(('budget_bytes', 'live_bytes', 'retired_bytes', 'loose_bytes', 'pooled_bytes'), (int(x.value) for x in v)))


def pool_trim():
    """dswx_batch_pool_trim: the pooled chunks of dropped sliding ranges back to the device (call when no other thread of
    the process allocates: the retired reservations are freed and re-reserved empty).  Returns the bytes released."""
    v = ctypes.c_uint64()
    _check(load_library().dswx_batch_pool_trim(ctypes.byref(v)))
    return int(v.value)


def load_library(path=None):
    """dlopen the in-tree HIP library; raises if it has not been built.  `path` loads another
    build of the same ABI next to it (tools/ab_variants.py: A/B of two builds in one process)."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    if path is not None and path in _alt_libs:
        return _alt_libs[path]
    alt = path is not None
    if not alt and not os.environ.get('DSWX_HIP_LIB'):
        _build.build()        # no-op unless the sources changed since the library was built (never a stale binary)
    path = path or library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} is missing: the DSWx HIP extension has not been built '
            '(run `python -m proteus_amd.build`); there is no CPU fallback')
    lib = ctypes.CDLL(path)
    vp, i64, cp = ctypes.c_void_p, ctypes.c_int64, ctypes.c_char_p
    pp = ctypes.POINTER(ctypes.c_void_p)
    sig = {
        'dswx_abi_version': (ctypes.c_int, []),
        'dswx_last_error': (cp, []),
        'dswx_device_count': (ctypes.c_int, []),
        'dswx_ctx_create': (ctypes.c_int, [ctypes.c_int, pp]),
        'dswx_ctx_destroy': (ctypes.c_int, [vp]),
        'dswx_params_default': (ctypes.c_int, [ctypes.POINTER(Params)]),
        'dswx_classify_host': (ctypes.c_int, [vp, ctypes.POINTER(Params), i64, i64, i64,
                                              ctypes.POINTER(PlanesIn),
                                              ctypes.POINTER(PlanesOut), vp]),
        'dswx_classify_device': (ctypes.c_int, [vp, ctypes.POINTER(Params), i64, i64,
                                                ctypes.POINTER(PlanesIn),
                                                ctypes.POINTER(PlanesOut), vp, vp]),
        'dswx_classify_device_2d': (ctypes.c_int, [vp, ctypes.POINTER(Params), i64, i64, i64,
                                                   ctypes.POINTER(PlanesIn),
                                                   ctypes.POINTER(PlanesOut), vp, vp]),
        'dswx_classify_batch': (ctypes.c_int, [vp, ctypes.POINTER(Params), ctypes.POINTER(BatchGeom),
                                               ctypes.POINTER(PlanesIn), ctypes.POINTER(PlanesOut),
                                               vp, vp]),
        'dswx_synth_batch': (ctypes.c_int, [vp, ctypes.c_uint64, i64, ctypes.POINTER(BatchGeom),
                                            ctypes.POINTER(PlanesIn), vp]),
        'dswx_interpret_layer_host': (ctypes.c_int, [vp, vp, i64, vp]),
        'dswx_shadow_layer_host': (ctypes.c_int, [vp, vp, i64, i64, i64,
                                                  ctypes.POINTER(ctypes.c_double * 3)] +
                                   [ctypes.c_double] * 6 + [vp]),
        'dswx_shadow_layer_device': (ctypes.c_int, [vp, vp, i64, i64, i64, i64,
                                                    ctypes.POINTER(ctypes.c_double * 3)] +
                                     [ctypes.c_double] * 6 + [vp, vp]),
        'dswx_shadow_thresholds': (ctypes.c_int, [ctypes.c_double, ctypes.c_double,
                                                  ctypes.POINTER(ctypes.c_double),
                                                  ctypes.POINTER(ctypes.c_double)]),
        'dswx_shadow_layer_host_q': (ctypes.c_int, [vp, vp, i64, i64, i64,
                                                    ctypes.POINTER(ctypes.c_double * 3)] +
                                     [ctypes.c_double] * 6 + [vp]),
        'dswx_shadow_layer_device_q': (ctypes.c_int, [vp, vp, i64, i64, i64, i64,
                                                      ctypes.POINTER(ctypes.c_double * 3)] +
                                       [ctypes.c_double] * 6 + [vp, vp]),
        'dswx_shadow_layer_host_q32': (ctypes.c_int, [vp, vp, i64, i64, i64,
                                                      ctypes.POINTER(ctypes.c_double * 3),
                                                      ctypes.c_double, ctypes.c_double, ctypes.c_float,
                                                      ctypes.c_float, ctypes.c_double, ctypes.c_double, vp]),
        'dswx_shadow_layer_device_q32': (ctypes.c_int, [vp, vp, i64, i64, i64, i64,
                                                        ctypes.POINTER(ctypes.c_double * 3),
                                                        ctypes.c_double, ctypes.c_double, ctypes.c_float,
                                                        ctypes.c_float, ctypes.c_double, ctypes.c_double,
                                                        vp, vp]),
        'dswx_landcover_mask_host': (ctypes.c_int, [vp, vp, vp, i64, i64, vp, ctypes.c_int32, vp,
                                                    ctypes.c_int32, vp]),
        'dswx_landcover_mask_device': (ctypes.c_int, [vp, vp, vp, i64, i64, i64, vp, ctypes.c_int32,
                                                      vp, ctypes.c_int32, vp, vp]),
        'dswx_shadow_layer_batch': (ctypes.c_int, [vp, vp, i64, i64, i64, i64, ctypes.POINTER(ctypes.c_double * 3)] +
                                    [ctypes.c_double] * 4 + [ctypes.c_int32, ctypes.c_double, ctypes.c_double, vp, i64, vp]),
        'dswx_landcover_mask_batch': (ctypes.c_int, [vp, vp, vp, i64, i64, i64, vp, ctypes.c_int32, vp,
                                                     ctypes.c_int32, vp, i64, vp]),
        'dswx_synth_fill': (ctypes.c_int, [vp, ctypes.c_uint64, i64, i64, i64, i64,
                                           ctypes.POINTER(PlanesIn), vp]),
        'dswx_device_malloc': (ctypes.c_int, [vp, ctypes.c_size_t, pp]),
        'dswx_device_free': (ctypes.c_int, [vp, vp]),
        'dswx_host_alloc': (ctypes.c_int, [vp, ctypes.c_size_t, pp]),
        'dswx_host_free': (ctypes.c_int, [vp, vp]),
        'dswx_memcpy_h2d': (ctypes.c_int, [vp, vp, vp, ctypes.c_size_t]),
        'dswx_memcpy_d2h': (ctypes.c_int, [vp, vp, vp, ctypes.c_size_t]),
        'dswx_memset_d': (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_size_t]),
        'dswx_stream_synchronize': (ctypes.c_int, [vp, vp]),
        'dswx_event_create': (ctypes.c_int, [vp, pp]),
        'dswx_event_destroy': (ctypes.c_int, [vp, vp]),
        'dswx_event_record': (ctypes.c_int, [vp, vp, vp]),
        'dswx_event_elapsed_ms': (ctypes.c_int, [vp, vp, vp,
                                                 ctypes.POINTER(ctypes.c_float)]),
        'dswx_last_kernel_info': (ctypes.c_int, [vp, ctypes.c_char_p, ctypes.c_size_t]),
        'dswx_batch_layout': (ctypes.c_int, [ctypes.POINTER(BatchGeom), ctypes.c_uint32,
                                             ctypes.POINTER(BatchLayout)]),
        'dswx_batch_create': (ctypes.c_int, [vp, ctypes.POINTER(BatchGeom), ctypes.c_uint32, pp]),
        'dswx_batch_destroy': (ctypes.c_int, [vp]),
        'dswx_batch_planes': (ctypes.c_int, [vp, ctypes.POINTER(BatchGeom), ctypes.POINTER(PlanesIn),
                                             ctypes.POINTER(PlanesOut), pp]),
        'dswx_batch_info': (ctypes.c_int, [vp, ctypes.POINTER(BatchInfo)]),
        'dswx_batch_classify': (ctypes.c_int, [vp, ctypes.POINTER(Params), i64, vp]),
        'dswx_batch_synth': (ctypes.c_int, [vp, ctypes.c_uint64, i64, vp]),
        'dswx_batch_place_search': (ctypes.c_int, [vp, ctypes.POINTER(Params), ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_uint64]),
        'dswx_batch_place_slide': (ctypes.c_int, [vp, ctypes.POINTER(Params), ctypes.c_uint64, ctypes.c_uint64,
                                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64]),
        'dswx_batch_va_budget': (ctypes.c_int, [ctypes.c_uint64] + [ctypes.POINTER(ctypes.c_uint64)] * 5),
        'dswx_batch_pool_trim': (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64)]),
        'dswx_cog_layout': (ctypes.c_int, [i64, i64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32),
                                           ctypes.c_int32, ctypes.POINTER(CogLayout)]),
        'dswx_cog_blocks_device': (ctypes.c_int, [vp, vp, ctypes.c_int32, i64, i64, ctypes.c_int32,
                                                  ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_int32, vp, vp]),
        'dswx_untile_device': (ctypes.c_int, [vp, vp, ctypes.c_int32, i64, i64, ctypes.c_int32, ctypes.c_int32,
                                              ctypes.c_int32, vp, vp]),
        'dswx_rgb_planes_device': (ctypes.c_int, [vp, vp, vp, vp, vp, i64, ctypes.POINTER(ctypes.c_double * 3),
                                                  ctypes.POINTER(ctypes.c_double * 3), ctypes.c_int32, vp, vp]),
        'dswx_convolve_axis_device': (ctypes.c_int, [vp, vp, ctypes.c_int32, i64, i64, i64, i64, i64, ctypes.c_int32, vp, vp, vp,
                                                     ctypes.c_int32, i64, i64, vp]),
        'dswx_to_byte_device': (ctypes.c_int, [vp, vp, ctypes.c_int32, i64, vp, vp]),
        'dswx_gather_2d_device': (ctypes.c_int, [vp, vp, ctypes.c_int32, i64, i64, vp, ctypes.c_int32, vp, ctypes.c_int32, vp, vp]),
        'dswx_copy_2d_device': (ctypes.c_int, [vp, vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vp]),
        'dswx_memcpy_h2d_async': (ctypes.c_int, [vp, vp, vp, ctypes.c_size_t, vp]),
        'dswx_memcpy_d2h_async': (ctypes.c_int, [vp, vp, vp, ctypes.c_size_t, vp]),
    }
    for name, (res, args) in sig.items():
        if alt and not hasattr(lib, name):
            continue                      # an older build of the same ABI version
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.dswx_abi_version() != DSWX_ABI_VERSION and not (alt and lib.dswx_abi_version() == DSWX_ABI_VERSION - 1):
        # (an A/B partner may be the previous ABI: tools/ab_variants.py drives it through the classify entries only,
        # whose signatures have not changed)
        raise RuntimeError('libdswx_hip.so ABI version mismatch; rebuild it')
    if alt:
        _alt_libs[path] = lib
    else:
        _lib = lib
    return lib


_lab = None
# every symbol tools/lab/csrc/dswx_lab.h declares
LAB_SYMBOLS = ('dswx_lab_attach', 'dswx_lab_configure', 'dswx_stream_probe')


def load_lab():
    """dlopen libdswx_lab.so (experiments: kernel structures that lost, roofline probes, A/B switches).
    Only tools/ and the variant tests call this; nothing in the product path does."""
    global _lab
    if _lab is not None:
        return _lab
    load_library()
    path = _build.build_lab()
    lab = ctypes.CDLL(path)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    lab.dswx_lab_attach.restype = ctypes.c_int
    lab.dswx_lab_attach.argtypes = [vp]
    lab.dswx_lab_configure.restype = ctypes.c_int
    lab.dswx_lab_configure.argtypes = [vp, ctypes.c_char_p, ctypes.c_int]
    lab.dswx_stream_probe.restype = ctypes.c_int
    lab.dswx_stream_probe.argtypes = [vp, i64, i64, i64, ctypes.POINTER(PlanesIn), ctypes.POINTER(PlanesOut),
                                      ctypes.c_int, vp]
    _lab = lab
    return lab


def _check(rc):
    if rc != OK:
        raise DswxError(rc, load_library().dswx_last_error().decode('utf-8', 'replace'))


def device_count():
    return load_library().dswx_device_count()


def default_params():
    p = Params()
    _check(load_library().dswx_params_default(ctypes.byref(p)))
    return p


def make_params(thresholds=None, *, band_fills=None, fmask_fill=255.0,
                clip_negative_reflectance=True, mask_adjacent_to_cloud_mode='mask',
                apply_aerosol_class_remapping=True, aerosol_fmask_values=None,
                collapse_wtr_classes=True, aerosol_max_nir=None,
                exclude_psw_aggressive_in_browse=True, not_water_in_browse='white',
                cloud_in_browse='gray', snow_in_browse='cyan',
                set_ocean_masked_to_nodata=True, offset_and_scale=None):
    """Build a dswx_params_t.

    thresholds: object with the HlsThresholds attributes, or dict, or None
    (defaults).  aerosol_fmask_values: {class: [fmask values]} for the WTR-1
    classes 0, 2, 3, 4, or None (defaults).  offset_and_scale: six (scale_factor, add_offset) pairs =
    flag_offset_and_scale_inputs (dswx_hls.py:2300-2302): the chain on float32 reflectances.  Unknown modes raise the same
    Exception text as the reference (dswx_hls.py:1977-1981).
    """
    p = default_params()
    if thresholds is not None:
        for name in THRESHOLD_NAMES:
            v = thresholds[name] if isinstance(thresholds, dict) else \
                getattr(thresholds, name)
            if v is None:
                raise ValueError(f'HLS threshold {name} is not set')
            setattr(p, name, float(v))
    if band_fills is not None:
        for i, f in enumerate(band_fills):
            p.band_fill[i] = float('nan') if f is None else float(f)
    p.fmask_fill = float('nan') if fmask_fill is None else float(fmask_fill)
    if mask_adjacent_to_cloud_mode not in ADJ_MODES:
        raise Exception('ERROR mask adjacent to cloud/cloud-shadow mode:'
                        f' {mask_adjacent_to_cloud_mode}')
    p.mask_adjacent_to_cloud_mode = ADJ_MODES[mask_adjacent_to_cloud_mode]
    p.clip_negative_reflectance = int(bool(clip_negative_reflectance))
    p.apply_aerosol_class_remapping = int(bool(apply_aerosol_class_remapping))
    p.collapse_wtr_classes = int(bool(collapse_wtr_classes))
    if aerosol_max_nir is not None:
        p.aerosol_max_nir = float(aerosol_max_nir)
    # browse options as generate_dswx_layers maps them (dswx_hls.py:5309-5316)
    p.browse_exclude_psw_aggressive = int(bool(exclude_psw_aggressive_in_browse))
    p.browse_not_water_to_nodata = int(not_water_in_browse == 'nodata')
    p.browse_cloud_to_nodata = int(cloud_in_browse == 'nodata')
    p.browse_snow_to_nodata = int(snow_in_browse == 'nodata')
    p.browse_ocean_masked_to_nodata = int(bool(set_ocean_masked_to_nodata))
    if offset_and_scale is not None:
        if len(offset_and_scale) != 6:
            raise ValueError('offset_and_scale needs six (scale_factor, add_offset) pairs')
        p.offset_and_scale_inputs = 1
        for i, (sf, off) in enumerate(offset_and_scale):
            p.band_scale[i], p.band_offset[i] = float(sf), float(off)
    if aerosol_fmask_values is not None:
        for row, cls in enumerate((0, 2, 3, 4)):
            for v in range(256):
                p.aerosol_fmask_lut[row][v] = 0
            for v in aerosol_fmask_values[cls]:
                if 0 <= int(v) <= 255 and int(v) == v:
                    p.aerosol_fmask_lut[row][int(v)] = 1
    return p


def _bisect_floats(pred, lo, hi, dtype=np.float64):
    """pred(lo) is False and pred(hi) True (or the reverse) for a predicate that is monotonic in
    the float `x` of `dtype`; returns the two adjacent floats (a, b), a < b, where it flips.
    `pred` is evaluated on ARRAYS (64 copies of the candidate) so that numpy takes the same SIMD
    loop it takes for a raster."""
    dtype = np.dtype(dtype)
    itype, sign = (np.int64, 0x7fffffffffffffff) if dtype == np.float64 else (np.int32, 0x7fffffff)

    def ordered(d):
        i = int(dtype.type(d).view(itype))
        return i if i >= 0 else -(i & sign)

    def value(k):
        return float(itype(k if k >= 0 else (-k) | (-sign - 1)).view(dtype))

    def test(d):
        with np.errstate(all='ignore'):
            return bool(pred(np.full(64, d, dtype=dtype))[17])
    a, b = ordered(lo), ordered(hi)
    fa = test(lo)
    while b - a > 1:
        m = a + (b - a) // 2
        if test(value(m)) == fa:
            a = m
        else:
            b = m
    return value(a), value(b)


_shadow_threshold_cache = {}


def shadow_thresholds(min_slope_angle, max_sun_local_inc_angle, float32=False):
    """(slope_arg_max, inc_q_min) for dswx_shadow_layer_*_q: the reference's two tests
    `degrees(arctan(t)) <= min_slope_angle` and `degrees(arccos(q)) <= max_sun_local_inc_angle`
    (dswx_hls.py:4264-4281) as bounds on t and q, located with numpy's own array functions.
    float32=True: the same in float32 arithmetic (numpy < 2 value-based casting) for the _q32 forms."""
    key = (float(min_slope_angle), float(max_sun_local_inc_angle), bool(float32))
    if key in _shadow_threshold_cache:
        return _shadow_threshold_cache[key]
    if np.isnan(key[0]) or np.isnan(key[1]):
        raise ValueError('shadow angle threshold is NaN')
    dt = np.float32 if float32 else np.float64
    # a Python-float threshold is a weak scalar: the comparison runs in the array's dtype
    inc_ok = lambda q: np.degrees(np.arccos(q)) <= key[1]          # noqa: E731
    slope_ok = lambda t: np.degrees(np.arctan(t)) <= key[0]        # noqa: E731
    one = lambda f, v: bool(f(np.full(64, v, dtype=dt))[17])       # noqa: E731
    with np.errstate(all='ignore'):
        if not one(inc_ok, 1.0):
            inc_q_min = 2.0                      # never
        elif one(inc_ok, -1.0):
            inc_q_min = -1.0                     # whenever arccos is defined
        else:
            inc_q_min = _bisect_floats(inc_ok, -1.0, 1.0, dt)[1]
        if one(slope_ok, np.inf):
            slope_arg_max = float('inf')
        elif not one(slope_ok, -np.inf):
            slope_arg_max = float('-inf')
        else:
            slope_arg_max = _bisect_floats(slope_ok, -np.inf, np.inf, dt)[0]
    _shadow_threshold_cache[key] = (slope_arg_max, inc_q_min)
    return slope_arg_max, inc_q_min


def _host_ptr(arr):
    return ctypes.c_void_p(arr.ctypes.data)


class DeviceBuffer:
    """A hipMalloc'ed span owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        ptr = ctypes.c_void_p()
        _check(ctx.lib.dswx_device_malloc(ctx.handle, self.nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value

    def free(self):
        if self.ptr:
            self.ctx.lib.dswx_device_free(self.ctx.handle, ctypes.c_void_p(self.ptr))
            self.ptr = None

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        _check(self.ctx.lib.dswx_memcpy_h2d(self.ctx.handle,
                                            ctypes.c_void_p(self.ptr + offset),
                                            _host_ptr(arr), arr.nbytes))

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        _check(self.ctx.lib.dswx_memcpy_d2h(self.ctx.handle, _host_ptr(out),
                                            ctypes.c_void_p(self.ptr + offset),
                                            out.nbytes))
        return out

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per process and device (dswx_ctx_t)."""

    def __init__(self, device=0, lib_path=None):
        self.lib = load_library(lib_path)
        h = ctypes.c_void_p()
        _check(self.lib.dswx_ctx_create(int(device), ctypes.byref(h)))
        self.handle = h
        self.device = int(device)
        self._pinned = {}          # page-locked host spans: address -> bytes
        self._pinned_pool = {}     # released spans by size
        self._pinned_pool_bytes = 0
        self._pinned_lock = threading.RLock()  # reader threads allocate; finalizers release on ANY thread, also (cyclic GC) on
                                               # one that is inside the lock already: re-entrant

    def close(self):
        if self.handle:
            with self._pinned_lock:
                # the pool's spans belong to no array any more: hand them back before the context goes.  Spans of
                # arrays that are still alive are freed by their finalizers (dswx_host_free takes a NULL context).
                for nbytes, spans in self._pinned_pool.items():
                    for addr in spans:
                        self._pinned.pop(addr, None)
                        self.lib.dswx_host_free(self.handle, ctypes.c_void_p(addr))
                self._pinned_pool = {}
                self._pinned_pool_bytes = 0
                handle, self.handle = self.handle, None
            self.lib.dswx_ctx_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- page-locked host arrays -----------------------------------------------
    PINNED_POOL_CAP = 4 << 30      # bytes of released page-locked spans kept for reuse

    def pinned_empty(self, shape, dtype):
        """numpy array over page-locked memory (dswx_host_alloc).  When every plane handed to
        classify_host() is such an array the library works on them in place (zero copy: the kernels
        read and write the host planes across PCIe); outputs are then allocated page-locked as well.  Page-locking is
        slow (~0.1 ms per MB), so released spans go to a per-context pool and are reused."""
        dtype = np.dtype(dtype)
        nbytes = max(int(np.prod(shape, dtype=np.int64)) * dtype.itemsize, 1)
        addr = None
        with self._pinned_lock:
            spans = self._pinned_pool.get(nbytes)
            if spans:
                addr = spans.pop()
                self._pinned_pool_bytes -= nbytes
        if addr is None:
            ptr = ctypes.c_void_p()
            _check(self.lib.dswx_host_alloc(self.handle, nbytes, ctypes.byref(ptr)))
            addr = ptr.value
            with self._pinned_lock:
                self._pinned[addr] = nbytes
        raw = (ctypes.c_char * nbytes).from_address(addr)
        arr = np.frombuffer(raw, dtype=dtype, count=nbytes // dtype.itemsize).reshape(shape) \
            if int(np.prod(shape, dtype=np.int64)) else np.empty(shape, dtype)

        def release(ctx_ref=weakref.ref(self), lib=self.lib, addr=addr, nbytes=nbytes):
            ctx = ctx_ref()
            if ctx is None:
                lib.dswx_host_free(None, ctypes.c_void_p(addr))      # the context is gone: the span still is ours to free
                return
            with ctx._pinned_lock:
                keep = bool(ctx.handle) and ctx._pinned_pool_bytes + nbytes <= ctx.PINNED_POOL_CAP
                if keep:
                    ctx._pinned_pool.setdefault(nbytes, []).append(addr)
                    ctx._pinned_pool_bytes += nbytes
                else:
                    ctx._pinned.pop(addr, None)
                    lib.dswx_host_free(ctx.handle, ctypes.c_void_p(addr))
        weakref.finalize(raw, release)
        return arr

    def is_pinned(self, arr):
        a = arr.ctypes.data
        with self._pinned_lock:
            spans = list(self._pinned.items())
        return any(base <= a and a + arr.nbytes <= base + size for base, size in spans)

    # ---- host-pointer path ----------------------------------------------------
    def classify_host(self, bands, fmask, params, *, land=None, shad=None, ocean=None,
                      layers=('diag', 'wtr1', 'wtr1_aerosol', 'wtr2', 'wtr', 'bwtr',
                              'conf', 'cloud'),
                      counters=True):
        """bands: six int16 arrays of identical shape [H,W] or [T,H,W].

        Returns {layer: ndarray} (+ 'counters': int64 [T,3]) with the input's shape.
        """
        bands = [np.ascontiguousarray(b, dtype=np.int16) for b in bands]
        if len(bands) != 6:
            raise ValueError('need six reflectance bands')
        shape = bands[0].shape
        if len(shape) == 2:
            n_tiles, (h, w) = 1, shape
        elif len(shape) == 3:
            n_tiles, h, w = shape
        else:
            raise ValueError('bands must be [H,W] or [T,H,W]')

        def prep(a, name):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.uint8)   # bool SHAD -> 0/1 u8
            if a.shape != shape:
                raise ValueError(f'{name} shape {a.shape} != bands shape {shape}')
            return a

        for b in bands:
            if b.shape != shape:
                raise ValueError('band shapes differ')
        fmask = prep(fmask, 'fmask')
        land, shad, ocean = prep(land, 'land'), prep(shad, 'shad'), prep(ocean, 'ocean')
        pin = PlanesIn()
        for i, b in enumerate(bands):
            pin.band[i] = b.ctypes.data
        pin.fmask = fmask.ctypes.data
        pin.land = land.ctypes.data if land is not None else None
        pin.shad = shad.ctypes.data if shad is not None else None
        pin.ocean = ocean.ctypes.data if ocean is not None else None
        pinned_in = all(self.is_pinned(a) for a in bands + [fmask, land, shad, ocean] if a is not None)
        pout = PlanesOut()
        res = {}
        for name in layers:
            dt = np.uint16 if name == 'diag' else (np.float64 if name in F64_LAYERS
                                                   else np.uint8)
            if name != 'diag' and name not in U8_LAYERS and name not in F64_LAYERS:
                raise KeyError(name)
            res[name] = self.pinned_empty(shape, dt) if pinned_in else np.empty(shape, dtype=dt)
            setattr(pout, name, res[name].ctypes.data)
        cnt = np.zeros((n_tiles, 3), dtype=np.int64) if counters else None
        _check(self.lib.dswx_classify_host(
            self.handle, ctypes.byref(params), n_tiles, h, w, ctypes.byref(pin),
            ctypes.byref(pout), _host_ptr(cnt) if counters else None))
        if counters:
            res['counters'] = cnt
        return res

    # ---- device-pointer path --------------------------------------------------
    def classify_device(self, params, n_tiles, n_pixels, pin, pout, counters_ptr=None,
                        stream=None):
        _check(self.lib.dswx_classify_device(
            self.handle, ctypes.byref(params), int(n_tiles), int(n_pixels),
            ctypes.byref(pin), ctypes.byref(pout),
            ctypes.c_void_p(counters_ptr) if counters_ptr else None,
            ctypes.c_void_p(stream) if stream else None))

    def interpret_layer(self, diag_decimal):
        """generate_interpreted_layer on the device; any integer array in, uint8 out."""
        d = np.ascontiguousarray(diag_decimal, dtype=np.int64)
        out = np.empty(d.shape, dtype=np.uint8)
        _check(self.lib.dswx_interpret_layer_host(self.handle, _host_ptr(d), d.size,
                                                  _host_ptr(out)))
        return out

    def shadow_layer(self, dem, sun_vector, sin_azimuth, cos_azimuth, min_slope_angle,
                     max_sun_local_inc_angle, pixel_spacing_x=30, pixel_spacing_y=30, margin=0,
                     float32=False):
        """Terrain shadow layer of one float32 DEM [H,W]; returns bool [H-2m, W-2m].  The two angle
        thresholds are pulled back through numpy's arccos / arctan (shadow_thresholds).
        float32=True reproduces numpy < 2 value-based casting (all-float32 arithmetic)."""
        dem = np.ascontiguousarray(dem, dtype=np.float32)
        if dem.ndim != 2:
            raise ValueError('dem must be 2-D')
        h, w = dem.shape
        out = np.empty((max(h - 2 * margin, 0), max(w - 2 * margin, 0)), dtype=np.uint8)
        vec = (ctypes.c_double * 3)(*[float(v) for v in sun_vector])
        slope_arg_max, inc_q_min = shadow_thresholds(min_slope_angle, max_sun_local_inc_angle, float32)
        fn = self.lib.dswx_shadow_layer_host_q32 if float32 else self.lib.dswx_shadow_layer_host_q
        _check(fn(self.handle, _host_ptr(dem), h, w, int(margin), ctypes.byref(vec),
                  float(sin_azimuth), float(cos_azimuth), slope_arg_max, inc_q_min,
                  float(pixel_spacing_x), float(pixel_spacing_y), _host_ptr(out)))
        return out.astype(bool)

    def landcover_mask(self, worldcover_up3, copernicus, forest_classes, thresholds=(6, 3, 7, 3),
                       year_offset=0):
        """LAND layer from the warped WorldCover (3x grid) and CGLS (HLS grid) maps."""
        wc = np.ascontiguousarray(worldcover_up3, dtype=np.uint8)
        cg = np.ascontiguousarray(copernicus, dtype=np.uint8)
        h, w = cg.shape
        if wc.shape != (3 * h, 3 * w):
            raise ValueError(f'worldcover_up3 shape {wc.shape} != {(3 * h, 3 * w)}')
        fc = np.ascontiguousarray(list(forest_classes or []), dtype=np.int32)
        thr = np.ascontiguousarray(thresholds, dtype=np.int32)
        out = np.empty((h, w), dtype=np.uint8)
        _check(self.lib.dswx_landcover_mask_host(
            self.handle, _host_ptr(wc), _host_ptr(cg), h, w,
            _host_ptr(fc) if fc.size else None, int(fc.size), _host_ptr(thr), int(year_offset),
            _host_ptr(out)))
        return out

    def landcover_mask_device(self, wc_ptr, cg_ptr, n_tiles, height, width, forest_classes, out_ptr,
                              thresholds=(6, 3, 7, 3), year_offset=0, stream=None, out_tile_stride=0):
        """Device-pointer form: [n_tiles][3H][3W] + [n_tiles][H][W] -> [n_tiles][H][W], asynchronous.
        out_tile_stride: pixels between the LAND rasters of consecutive tiles (dswx_landcover_mask_batch: the LAND plane
        of a resident batch); 0 = packed."""
        fc = np.ascontiguousarray(list(forest_classes or []), dtype=np.int32)
        thr = np.ascontiguousarray(thresholds, dtype=np.int32)
        _check(self.lib.dswx_landcover_mask_batch(
            self.handle, ctypes.c_void_p(wc_ptr), ctypes.c_void_p(cg_ptr), int(n_tiles), int(height),
            int(width), _host_ptr(fc) if fc.size else None, int(fc.size), _host_ptr(thr),
            int(year_offset), ctypes.c_void_p(out_ptr), int(out_tile_stride), ctypes.c_void_p(stream) if stream else None))

    def shadow_layer_device(self, dem_ptr, n_tiles, height, width, margin, sun_vector, sin_azimuth,
                            cos_azimuth, min_slope_angle, max_sun_local_inc_angle, out_ptr,
                            pixel_spacing_x=30, pixel_spacing_y=30, stream=None, float32=False, out_tile_stride=0):
        """Device-pointer form: [n_tiles][H][W] float32 DEMs -> [n_tiles][H-2m][W-2m] u8, asynchronous.
        float32=True: numpy < 2 value-based casting (the _q32 entry point).  out_tile_stride != 0: the shadow rasters
        go that many pixels apart (dswx_shadow_layer_batch: the SHAD plane of a resident batch)."""
        vec = (ctypes.c_double * 3)(*[float(v) for v in sun_vector])
        slope_arg_max, inc_q_min = shadow_thresholds(min_slope_angle, max_sun_local_inc_angle, float32)
        if out_tile_stride:
            _check(self.lib.dswx_shadow_layer_batch(
                self.handle, ctypes.c_void_p(dem_ptr), int(n_tiles), int(height), int(width), int(margin), ctypes.byref(vec),
                float(sin_azimuth), float(cos_azimuth), float(slope_arg_max), float(inc_q_min), int(bool(float32)),
                float(pixel_spacing_x), float(pixel_spacing_y), ctypes.c_void_p(out_ptr), int(out_tile_stride),
                ctypes.c_void_p(stream) if stream else None))
            return
        fn = self.lib.dswx_shadow_layer_device_q32 if float32 else self.lib.dswx_shadow_layer_device_q
        _check(fn(
            self.handle, ctypes.c_void_p(dem_ptr), int(n_tiles), int(height), int(width), int(margin),
            ctypes.byref(vec), float(sin_azimuth), float(cos_azimuth), slope_arg_max, inc_q_min,
            float(pixel_spacing_x), float(pixel_spacing_y),
            ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream) if stream else None))

    # ---- libdswx_lab.so (experiments; tools/ and the variant tests only) -----------
    def lab_configure(self, **settings):
        """A/B switches of this context, e.g. lab_configure(fused_variant=3) or (host_chunks=3);
        loads libdswx_lab.so and attaches its kernel structures on first use."""
        lab = load_lab()
        if not getattr(self, '_lab_attached', False):
            _check(lab.dswx_lab_attach(self.handle))
            self._lab_attached = True
        for key, value in settings.items():
            _check(lab.dswx_lab_configure(self.handle, key.encode(), int(value)))

    def stream_probe(self, n_tiles, n_pixels, pin, pout, variant=0, stream=None, tile_stride=0):
        _check(load_lab().dswx_stream_probe(
            self.handle, int(n_tiles), int(n_pixels), int(tile_stride), ctypes.byref(pin),
            ctypes.byref(pout), int(variant), ctypes.c_void_p(stream) if stream else None))

    def classify_batch(self, params, geom, pin, pout, counters_ptr=None, stream=None):
        _check(self.lib.dswx_classify_batch(
            self.handle, ctypes.byref(params), ctypes.byref(geom), ctypes.byref(pin),
            ctypes.byref(pout), ctypes.c_void_p(counters_ptr) if counters_ptr else None,
            ctypes.c_void_p(stream) if stream else None))

    def synth_batch(self, seed, tile0, geom, pin, stream=None):
        _check(self.lib.dswx_synth_batch(
            self.handle, int(seed), int(tile0), ctypes.byref(geom), ctypes.byref(pin),
            ctypes.c_void_p(stream) if stream else None))

    def classify_device_2d(self, params, n_tiles, height, width, pin, pout, counters_ptr=None,
                           stream=None):
        _check(self.lib.dswx_classify_device_2d(
            self.handle, ctypes.byref(params), int(n_tiles), int(height), int(width),
            ctypes.byref(pin), ctypes.byref(pout),
            ctypes.c_void_p(counters_ptr) if counters_ptr else None,
            ctypes.c_void_p(stream) if stream else None))

    def synth_fill(self, seed, tile0, n_tiles, height, width, pin, stream=None):
        _check(self.lib.dswx_synth_fill(
            self.handle, int(seed), int(tile0), int(n_tiles), int(height), int(width),
            ctypes.byref(pin), ctypes.c_void_p(stream) if stream else None))

    def malloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ---- the raster formats either side of the path (ABI v6; device pointers) ---------------
    def cog_blocks_device(self, plane_ptr, elem_bytes, height, width, blocks_ptr, factors=(), tile=512, predictor=2,
                          stream=None):
        f = (ctypes.c_int32 * max(len(factors), 1))(*[int(x) for x in factors])
        _check(self.lib.dswx_cog_blocks_device(self.handle, ctypes.c_void_p(plane_ptr), int(elem_bytes), int(height), int(width),
                                               int(tile), f, len(factors), int(predictor), ctypes.c_void_p(blocks_ptr),
                                               ctypes.c_void_p(stream) if stream else None))

    def untile_device(self, blocks_ptr, elem_bytes, height, width, block_width, block_height, predictor, plane_ptr, stream=None):
        _check(self.lib.dswx_untile_device(self.handle, ctypes.c_void_p(blocks_ptr), int(elem_bytes), int(height), int(width),
                                           int(block_width), int(block_height), int(predictor), ctypes.c_void_p(plane_ptr),
                                           ctypes.c_void_p(stream) if stream else None))

    def rgb_planes_device(self, red_ptr, green_ptr, blue_ptr, diag_ptr, n_pixels, scale, offset, clip, out_ptr, stream=None):
        sc = (ctypes.c_double * 3)(*[float(v) for v in scale])
        of = (ctypes.c_double * 3)(*[float(v) for v in offset])
        _check(self.lib.dswx_rgb_planes_device(self.handle, ctypes.c_void_p(red_ptr), ctypes.c_void_p(green_ptr),
                                               ctypes.c_void_p(blue_ptr), ctypes.c_void_p(diag_ptr) if diag_ptr else None,
                                               int(n_pixels), ctypes.byref(sc), ctypes.byref(of), int(bool(clip)),
                                               ctypes.c_void_p(out_ptr), ctypes.c_void_p(stream) if stream else None))

    def convolve_axis_device(self, src_ptr, src_is_f64, n_lines, n_in, src_line_stride, src_elem_stride, n_out, taps, first_ptr,
                             weights_ptr, dst_ptr, dst_is_f64, dst_line_stride, dst_elem_stride, stream=None):
        _check(self.lib.dswx_convolve_axis_device(
            self.handle, ctypes.c_void_p(src_ptr), int(bool(src_is_f64)), int(n_lines), int(n_in), int(src_line_stride),
            int(src_elem_stride), int(n_out), int(taps), ctypes.c_void_p(first_ptr), ctypes.c_void_p(weights_ptr),
            ctypes.c_void_p(dst_ptr), int(bool(dst_is_f64)), int(dst_line_stride), int(dst_elem_stride),
            ctypes.c_void_p(stream) if stream else None))

    def to_byte_device(self, src_ptr, src_dtype, n, dst_ptr, stream=None):
        kind = {'uint16': 1, 'int16': 2, 'float32': 3}.get(np.dtype(src_dtype).name)
        if kind is None:
            raise ValueError(f'to_byte_device: {np.dtype(src_dtype)} planes are not taken')
        _check(self.lib.dswx_to_byte_device(self.handle, ctypes.c_void_p(src_ptr), kind, int(n), ctypes.c_void_p(dst_ptr),
                                            ctypes.c_void_p(stream) if stream else None))

    def gather_2d_device(self, src_ptr, elem_bytes, src_height, src_width, rows_ptr, n_rows, cols_ptr, n_cols, dst_ptr, stream=None):
        _check(self.lib.dswx_gather_2d_device(self.handle, ctypes.c_void_p(src_ptr), int(elem_bytes), int(src_height), int(src_width),
                                              ctypes.c_void_p(rows_ptr), int(n_rows), ctypes.c_void_p(cols_ptr), int(n_cols),
                                              ctypes.c_void_p(dst_ptr), ctypes.c_void_p(stream) if stream else None))

    def copy_2d_device(self, dst_ptr, dst_pitch, src_ptr, src_pitch, width_bytes, height, stream=None):
        _check(self.lib.dswx_copy_2d_device(self.handle, ctypes.c_void_p(dst_ptr), int(dst_pitch), ctypes.c_void_p(src_ptr),
                                            int(src_pitch), int(width_bytes), int(height),
                                            ctypes.c_void_p(stream) if stream else None))

    def h2d_async(self, dst_ptr, host_arr, nbytes=None, stream=None):
        _check(self.lib.dswx_memcpy_h2d_async(self.handle, ctypes.c_void_p(dst_ptr), _host_ptr(host_arr),
                                              int(host_arr.nbytes if nbytes is None else nbytes),
                                              ctypes.c_void_p(stream) if stream else None))

    def d2h_async(self, host_arr, src_ptr, nbytes=None, stream=None):
        _check(self.lib.dswx_memcpy_d2h_async(self.handle, _host_ptr(host_arr), ctypes.c_void_p(src_ptr),
                                              int(host_arr.nbytes if nbytes is None else nbytes),
                                              ctypes.c_void_p(stream) if stream else None))

    def synchronize(self, stream=None):
        _check(self.lib.dswx_stream_synchronize(
            self.handle, ctypes.c_void_p(stream) if stream else None))

    def event(self):
        e = ctypes.c_void_p()
        _check(self.lib.dswx_event_create(self.handle, ctypes.byref(e)))
        return e

    def record(self, event, stream=None):
        _check(self.lib.dswx_event_record(
            self.handle, event, ctypes.c_void_p(stream) if stream else None))

    def elapsed_ms(self, start, stop):
        ms = ctypes.c_float()
        _check(self.lib.dswx_event_elapsed_ms(self.handle, start, stop, ctypes.byref(ms)))
        return ms.value

    def destroy_event(self, event):
        self.lib.dswx_event_destroy(self.handle, event)

    def last_kernel_info(self):
        buf = ctypes.create_string_buffer(256)
        _check(self.lib.dswx_last_kernel_info(self.handle, buf, 256))
        return buf.value.decode()


def cog_layout(height, width, elem_bytes, factors=(), tile=512):
    """dswx_cog_layout (no device needed): {'n_levels', 'total_bytes', 'levels': [{'factor', 'height', 'width',
    'blocks_down', 'blocks_across', 'offset_bytes'}]}."""
    lay = CogLayout()
    f = (ctypes.c_int32 * max(len(factors), 1))(*[int(x) for x in factors])
    _check(load_library().dswx_cog_layout(int(height), int(width), int(elem_bytes), int(tile), f, len(factors), ctypes.byref(lay)))
    return {'n_levels': lay.n_levels, 'total_bytes': int(lay.total_bytes), 'tile': lay.tile,
            'levels': [{'factor': lay.factor[k], 'height': int(lay.height[k]), 'width': int(lay.width[k]),
                        'blocks_down': lay.blocks_down[k], 'blocks_across': lay.blocks_across[k],
                        'offset_bytes': int(lay.offset_bytes[k])} for k in range(lay.n_levels)]}


def batch_layout(n_tiles, height, width, masks=False, extra_layers=(), tile_stride=0, separate_outputs=False,
                 sliding_outputs=False):
    """dswx_batch_layout: where dswx_batch_create puts every plane (pure function, no device).  Returns
    {'tile_stride', 'arena_bytes', 'write_span_bytes', 'planes': {name: (offset, nbytes)}}."""
    flags = _batch_flags(masks, extra_layers, separate_outputs, sliding_outputs)
    lay = BatchLayout()
    geom = BatchGeom(n_tiles, height, width, tile_stride)
    _check(load_library().dswx_batch_layout(ctypes.byref(geom), flags, ctypes.byref(lay)))
    planes = {name: (int(lay.plane_offset[k]), int(lay.plane_bytes[k])) for name, k in PLANE_INDEX.items()
              if lay.plane_bytes[k]}
    return {'tile_stride': int(lay.tile_stride), 'arena_bytes': int(lay.arena_bytes),
            'write_span_bytes': int(lay.write_span_bytes), 'planes': planes}


def _batch_flags(masks, extra_layers, separate_outputs, sliding_outputs=False):
    unknown = [x for x in extra_layers if x not in ('wtr1_aerosol', 'browse')]
    if unknown:
        raise ValueError(f'a resident batch has no plane {unknown[0]!r}')
    return ((BATCH_MASKS if masks else 0) | (BATCH_WTR1_AEROSOL if 'wtr1_aerosol' in extra_layers else 0)
            | (BATCH_BROWSE if 'browse' in extra_layers else 0) | (BATCH_SEPARATE_OUTPUTS if separate_outputs else 0)
            | (BATCH_SLIDING_OUTPUTS if sliding_outputs else 0))


class DeviceBatch:
    """Band-planar batch resident in HBM (dswx_batch_t of include/dswx_hip.h): every plane is [n_tiles][tile_stride].

    The LIBRARY allocates and lays out the planes (dswx_batch_create: one allocation, 256-byte aligned plane
    offsets; with separate_outputs one for the inputs and one per output plane, which is what `place_search`
    needs).  By default the tile stride is H*W rounded up to a multiple of 256 pixels, so every tile starts on a
    256-byte boundary in every plane (contiguous tiles of 3660 x 3660 do not: 13,395,600 = 144 mod 256, which
    costs ~20 % of the HBM rate); `tile_align=1` gives contiguous tiles.
    Used by bench.py, the multi-GPU driver and the device-path parity tests.
    """

    def __init__(self, ctx, n_tiles, height, width, masks=False, extra_layers=(), tile_align=256,
                 separate_outputs=False, sliding_outputs=False):
        self.ctx, self.n_tiles, self.height, self.width = ctx, n_tiles, height, width
        self.n_pixels = height * width
        self.masks = masks
        stride = -(-self.n_pixels // tile_align) * tile_align
        flags = _batch_flags(masks, extra_layers, separate_outputs, sliding_outputs)
        h = ctypes.c_void_p()
        geom = BatchGeom(n_tiles, height, width, stride)
        if stride == 0:                 # an empty tile: let the library resolve the stride (0 stays 0)
            geom.tile_stride = 0
        self._lib = ctx.lib             # kept for free(): the batch may outlive ctx.handle
        self.handle = None
        _check(ctx.lib.dswx_batch_create(ctx.handle, ctypes.byref(geom), flags, ctypes.byref(h)))
        self.handle = h
        self.out_layers = ['wtr1'] + (['wtr1_aerosol'] if 'wtr1_aerosol' in extra_layers else []) + \
            ['wtr2', 'wtr', 'bwtr', 'conf', 'cloud'] + (['browse'] if 'browse' in extra_layers else [])
        self._rebind()

    def _rebind(self):
        """Fetch the plane pointers (they change when place_search re-binds output planes)."""
        self.geom, self.pin, self.pout = BatchGeom(), PlanesIn(), PlanesOut()
        cnt = ctypes.c_void_p()
        _check(self.ctx.lib.dswx_batch_planes(self.handle, ctypes.byref(self.geom), ctypes.byref(self.pin),
                                              ctypes.byref(self.pout), ctypes.byref(cnt)))
        self.tile_stride = int(self.geom.tile_stride)
        self.counters_ptr = cnt.value
        info = self.info()
        self.nbytes = info['bytes_allocated']

    def info(self):
        bi = BatchInfo()
        _check(self.ctx.lib.dswx_batch_info(self.handle, ctypes.byref(bi)))
        return {'bytes_allocated': int(bi.bytes_allocated), 'n_allocations': int(bi.n_allocations),
                'flags': int(bi.flags), 'search_candidates': int(bi.search_candidates),
                'search_probes': int(bi.search_probes),
                'first_come_launch_ms': float(bi.first_come_launch_ms), 'kept_launch_ms': float(bi.kept_launch_ms),
                'va_reserved_bytes': int(bi.va_reserved_bytes), 'va_retired_bytes': int(bi.va_retired_bytes),
                'va_budget_bytes': int(bi.va_budget_bytes), 'va_pooled_bytes': int(bi.va_pooled_bytes),
                'note': bi.note.decode()}

    def place_search(self, params, candidates=6, launches=3, keep_free_bytes=8 << 30):
        """dswx_batch_place_search: measured placement of the output planes (separate_outputs batches whose inputs
        are resident).  On MI355X the fused kernel's rate depends on the ranges its seven write streams land in --
        a stable property of the allocation -- so a long-lived batch is worth placing.
        Returns the record of the search."""
        _check(self.ctx.lib.dswx_batch_place_search(self.handle, ctypes.byref(params), int(candidates),
                                                    int(launches), int(keep_free_bytes)))
        self._rebind()
        i = self.info()
        return {'trials': i['search_candidates'], 'probes': i['search_probes'],
                'first_come_launch_ms': round(i['first_come_launch_ms'], 4),
                'kept_launch_ms': round(i['kept_launch_ms'], 4)}

    def place_slide(self, params, slack_bytes=48 << 30, step_bytes=2 << 30, spread_gaps=4, refine_passes=1,
                    launches=3, keep_free_bytes=8 << 30):
        """dswx_batch_place_slide: the output region (sliding_outputs batches) timed at offsets 0, step, 2 step, ...
        of a range `slack_bytes` longer than itself; the best position is kept and the rest of the range returned to the
        device.  Returns the record of the search."""
        _check(self.ctx.lib.dswx_batch_place_slide(self.handle, ctypes.byref(params), int(slack_bytes), int(step_bytes),
                                                   int(spread_gaps), int(refine_passes), int(launches),
                                                   int(keep_free_bytes)))
        self._rebind()
        i = self.info()
        rec = {'positions': i['search_candidates'], 'probes': i['search_probes'],
               'first_come_launch_ms': round(i['first_come_launch_ms'], 4),
               'kept_launch_ms': round(i['kept_launch_ms'], 4), 'va_retired_gib': round(i['va_retired_bytes'] / 2 ** 30, 2)}
        if i['note']:
            rec['note'] = i['note']
        return rec

    def synth(self, seed, tile0=0, stream=None):
        _check(self.ctx.lib.dswx_batch_synth(self.handle, int(seed), int(tile0),
                                             ctypes.c_void_p(stream) if stream else None))

    def classify(self, params, stream=None, counters=True, n_tiles=None):
        """The first `n_tiles` resident tiles (None = all, DSWX_BATCH_ALL_TILES; 0 = none: an empty chunk is no work).
        counters=False goes through dswx_classify_batch with a NULL counters pointer (dswx_batch_classify always counts)."""
        if n_tiles is None:
            n_tiles = BATCH_ALL_TILES
        if counters:
            _check(self.ctx.lib.dswx_batch_classify(self.handle, ctypes.byref(params), int(n_tiles),
                                                    ctypes.c_void_p(stream) if stream else None))
        elif n_tiles != 0:
            geom = BatchGeom(self.n_tiles if n_tiles == BATCH_ALL_TILES else n_tiles, self.height, self.width,
                             self.tile_stride)
            self.ctx.classify_batch(params, geom, self.pin, self.pout, None, stream)

    def _plane(self, name):
        if name in BAND_NAMES:
            ptr, dt = self.pin.band[BAND_NAMES.index(name)], np.int16
        elif name in ('fmask', 'land', 'shad', 'ocean'):
            ptr, dt = getattr(self.pin, name), np.uint8
        elif name == 'diag' or name in U8_LAYERS:
            ptr, dt = getattr(self.pout, name), (np.uint16 if name == 'diag' else np.uint8)
        else:
            raise ValueError(f'unknown plane {name!r}')
        if not ptr:
            raise ValueError(f'plane {name!r} is not part of this batch (masks / extra_layers of DeviceBatch)')
        return ptr, dt

    def read_tile(self, name, tile):
        """Download one plane of one tile as [H,W]."""
        ptr, dt = self._plane(name)
        out = np.empty(self.n_pixels, dtype=dt)
        if out.nbytes:
            _check(self.ctx.lib.dswx_memcpy_d2h(
                self.ctx.handle, _host_ptr(out),
                ctypes.c_void_p(ptr + tile * self.tile_stride * out.itemsize), out.nbytes))
        return out.reshape(self.height, self.width)

    def write_tile(self, name, tile, arr):
        ptr, dt = self._plane(name)
        arr = np.ascontiguousarray(arr, dtype=dt).ravel()
        assert arr.size == self.n_pixels
        if arr.nbytes:
            _check(self.ctx.lib.dswx_memcpy_h2d(
                self.ctx.handle, ctypes.c_void_p(ptr + tile * self.tile_stride * arr.itemsize), _host_ptr(arr),
                arr.nbytes))

    def read_counters(self):
        out = np.empty((self.n_tiles, 3), dtype=np.int64)
        if out.nbytes:
            _check(self.ctx.lib.dswx_memcpy_d2h(self.ctx.handle, _host_ptr(out),
                                                ctypes.c_void_p(self.counters_ptr), out.nbytes))
        return out

    def write_counters_sentinel(self, value):
        """Fill the counters plane with `value` (tests: which tiles did a launch write?)."""
        arr = np.full((self.n_tiles, 3), value, dtype=np.int64)
        if arr.nbytes:
            _check(self.ctx.lib.dswx_memcpy_h2d(self.ctx.handle, ctypes.c_void_p(self.counters_ptr), _host_ptr(arr),
                                                arr.nbytes))

    def free(self):
        """dswx_batch_destroy: allowed after the context is gone (the batch remembers its device), so the HBM of a
        batch that outlives its Context is still returned."""
        if self.handle:
            self._lib.dswx_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
