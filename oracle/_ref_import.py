"""TEST INFRASTRUCTURE (oracle/) -- never imported by the product path.

Imports the upstream PROTEUS module `proteus.dswx_hls` from /root/reference in
THIS container only, so that `gen_golden.py` can run the reference's own numpy
functions and capture golden vectors under tests/golden/.

The reference imports yamale / ruamel.yaml / osgeo at module top
(src/proteus/dswx_hls.py:8-13, src/proteus/core.py:5); none is installed here and
none is touched by the per-pixel functions, so inert placeholders are registered
for the import to succeed.  Nothing of this travels to the GPU box: the reference
tree does not exist there and only the generated .npz fixtures are committed.
"""
import os
import sys
import types

REFERENCE_SRC = '/root/reference/src'


class _Inert:
    """Attribute/callable sink standing in for GDAL & friends."""

    def __getattr__(self, name):
        return _Inert()

    def __call__(self, *args, **kwargs):
        return _Inert()


def _register(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


def import_reference():
    """Return the reference module object, or None if the tree is absent."""
    if not os.path.isdir(REFERENCE_SRC):
        return None
    # we run as root: never let Python drop __pycache__ into the read-only tree
    sys.dont_write_bytecode = True
    if 'osgeo' not in sys.modules:
        osgeo = _register('osgeo')
        for sub in ('gdal', 'osr', 'ogr'):
            m = _register('osgeo.' + sub)
            m.__getattr__ = lambda key: _Inert()
            setattr(osgeo, sub, m)
        _register('osgeo.gdalconst', GDT_Float32=6, GDT_Byte=1)
    if 'yamale' not in sys.modules:
        _register('yamale')
    if 'ruamel' not in sys.modules:
        ru = _register('ruamel')
        ru.yaml = _register('ruamel.yaml', YAML=_Inert())
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import proteus.dswx_hls as ref
    return ref
