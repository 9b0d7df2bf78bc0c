"""Import harness for the read-only reference checkout (THIS container only).

Used by tools/gen_golden.py to run the reference's own numpy/scipy functions and
capture golden vectors.  Nothing here travels to the GPU box: tests read the
committed .npz fixtures under tests/golden/, never /root/reference.

Run the generators with /opt/conda/bin/python3.9: that interpreter has the REAL
scikit-image 0.18.3 and bottleneck 1.3.2 the reference calls.  Modules that only
do IO / networking (sentinelhub, rasterio, hickle, boto3, TF) are replaced with
permissive stubs; `bottleneck` / `skimage.transform.resize` fall back to numpy /
scipy.ndimage stand-ins ONLY if the import fails (recorded in VERSIONS).
"""
import os
import sys
import types

import numpy as np
import pandas  # noqa: F401  (must be imported before bottleneck is stubbed)

REF = os.environ.get("TTC_REFERENCE", "/root/reference")


class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __getitem__(self, k):
        return _Any()


def _stub(name):
    m = types.ModuleType(name)
    m.__getattr__ = lambda k: _Any()
    sys.modules[name] = m
    return m


_loaded = {}
VERSIONS = {}          # what produced the fixtures: filled by load()


def load():
    """Returns (J, CR): the reference job module and cloud_removal module."""
    if _loaded:
        return _loaded["J"], _loaded["CR"]
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference checkout not found at {REF}")
    sys.dont_write_bytecode = True
    sys.path[:0] = [REF, os.path.join(REF, "src")]
    for n in ['sentinelhub', 'sentinelhub.geo_utils', 'sentinelhub.api', 'sentinelhub.config',
              'pyproj', 'shapely', 'shapely.geometry', 'reverse_geocoder', 'pycountry',
              'pycountry_convert', 'rasterio', 'rasterio.transform', 'hickle', 'boto3',
              'botocore', 'botocore.config', 'boto3.s3', 'boto3.s3.transfer',
              'botocore.errorfactory', 'botocore.exceptions']:
        _stub(n)
    _stub('tensorflow').__version__ = '1.15.4'
    # skimage.transform.resize and bottleneck: use the REAL packages when the interpreter has them
    # (/opt/conda/bin/python3.9 in this image: scikit-image 0.18.3, bottleneck 1.3.2); stand-ins only as a
    # last resort, and the generator records which one produced the fixtures (tests/golden/GENERATOR.json).
    try:
        import skimage
        import skimage.transform  # noqa: F401
        VERSIONS["skimage"] = skimage.__version__
    except ImportError:
        import scipy.ndimage as ndi
        skt = types.ModuleType('skimage.transform')

        def _resize(img, shape, order=1, **kw):
            return ndi.zoom(img.astype(np.float64), [o / i for o, i in zip(shape, img.shape)],
                            order=order, mode='mirror', grid_mode=True)
        skt.resize = _resize
        sk = types.ModuleType('skimage')
        sk.transform = skt
        sys.modules.update({'skimage': sk, 'skimage.transform': skt})
        VERSIONS["skimage"] = "STAND-IN (scipy.ndimage.zoom, no anti-aliasing)"
    try:
        import bottleneck
        VERSIONS["bottleneck"] = bottleneck.__version__
    except ImportError:
        bn = types.ModuleType('bottleneck')
        bn.__version__ = '1.3.7'
        for k in ['nanmean', 'nanstd', 'nanmedian', 'nanmax', 'nanmin', 'median', 'nansum']:
            setattr(bn, k, getattr(np, k))
        sys.modules['bottleneck'] = bn
        VERSIONS["bottleneck"] = "STAND-IN (numpy nan-functions)"
    import scipy
    import sklearn
    VERSIONS.update(python=sys.version.split()[0], numpy=np.__version__, scipy=scipy.__version__,
                    sklearn=sklearn.__version__)
    import download_and_predict_job as J
    from src.preprocessing import cloud_removal as CR
    _loaded["J"], _loaded["CR"] = J, CR
    return J, CR
