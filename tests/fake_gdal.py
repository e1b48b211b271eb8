"""An in-memory stand-in for `osgeo.gdal` / `osgeo.osr` — just what testing/s2_tiles_supres.py (and dsen2_amd.cli.GdalProduct,
which follows it) touch — plus the seeded "product" and the stand-in network the command-line tests use.  GDAL itself is not
installed in the build image.  Shared by tests/test_cli_host.py and tests/golden/make_golden_cli.py (which runs the REFERENCE's
own script against it and records what it prints and writes)."""
import os
import types

import numpy as np

DESC10 = ['B4, central wavelength 665 nm', 'B3, central wavelength 560 nm', 'B2, central wavelength 490 nm',
          'B8, central wavelength 842 nm']
DESC20 = ['B5, central wavelength 705 nm', 'B6, central wavelength 740 nm', 'B7, central wavelength 783 nm',
          'B8A, central wavelength 865 nm', 'B11, central wavelength 1610 nm', 'B12, central wavelength 2190 nm']
DESC60 = ['B1, central wavelength 443 nm', 'B9, central wavelength 945 nm', 'B10, central wavelength 1375 nm']



def nearest_up(lo, k):
    """The stand-in "network" of the command-line tests: nearest-neighbour up-sampling (what is tested is the flow around it)."""
    return np.repeat(np.repeat(np.asarray(lo, np.float32), k, axis=0), k, axis=1)


def arrays(n=48):
    rng = np.random.default_rng(0)
    d10 = rng.integers(1, 9000, size=(n, n, 4)).astype(np.uint16)
    d20 = rng.integers(1, 9000, size=(n // 2, n // 2, 6)).astype(np.uint16)
    d60 = rng.integers(1, 9000, size=(n // 6, n // 6, 3)).astype(np.uint16)
    return d10, d20, d60


# ---- an in-memory stand-in for osgeo.gdal: just what s2_tiles_supres.py (and cli.GdalProduct) touch ----
class _Band(object):
    def __init__(self, ds, i):
        self.ds, self.i = ds, i

    def GetDescription(self):
        return self.ds.desc[self.i]

    def SetDescription(self, d):
        self.ds.desc[self.i] = d

    def WriteArray(self, a):
        self.ds.data[self.i] = np.array(a, dtype=np.float64)


class _Dataset(object):
    def __init__(self, data=None, desc=None, subs=None):
        self.data, self.desc, self.subs = data, desc, subs or []
        self.geot, self.proj, self.flushed = (600000.0, 10.0, 0.0, 5000000.0, 0.0, -10.0), 'PROJCS["UTM 33N"]', False

    RasterCount = property(lambda self: len(self.data))
    RasterXSize = property(lambda self: self.data[0].shape[1])
    RasterYSize = property(lambda self: self.data[0].shape[0])

    def GetSubDatasets(self):
        return self.subs

    def GetRasterBand(self, i):
        return _Band(self, i - 1)

    def ReadAsArray(self, xoff, yoff, xsize, ysize, buf_xsize, buf_ysize):
        return np.stack([b[yoff:yoff + ysize, xoff:xoff + xsize] for b in self.data])

    def GetGeoTransform(self):
        return self.geot

    def SetGeoTransform(self, g):
        self.geot = tuple(g)

    def GetProjection(self):
        return self.proj

    def SetProjection(self, p):
        self.proj = p

    def FlushCache(self):
        self.flushed = True


def fake_gdal(d10, d20, d60, can_create=True):
    gdal = types.ModuleType('osgeo.gdal')
    store = {'SUB10': _Dataset([d10[:, :, i] for i in range(4)], list(DESC10)),
             'SUB20': _Dataset([d20[:, :, i] for i in range(6)], list(DESC20)),
             'SUB60': _Dataset([d60[:, :, i] for i in range(3)], list(DESC60)),
             # the true-colour sub-dataset of a SAFE product: not a resolution group, but the reference opens it too when it
             # sizes the ROI (s2_tiles_supres.py:123-124 walks tenMsets + unknownMsets)
             'SUBTCI': _Dataset([(d10[:, :, i] >> 6).astype(np.uint8) for i in range(3)], ['TCI R', 'TCI G', 'TCI B'])}
    store['S2A.zip'] = _Dataset(subs=[('SUB10', 'Bands B2, B3, B4, B8 with 10m resolution, UTM 33N'),
                                      ('SUB20', 'Bands B5, ... with 20m resolution, UTM 33N'),
                                      ('SUB60', 'Bands B1, B9, B10 with 60m resolution, UTM 33N'),
                                      ('SUBTCI', 'True color image, UTM 33N')])
    created = {}

    class Driver(object):
        def GetMetadata(self):
            return {gdal.DCAP_CREATE: 'YES'} if can_create else {}

        def Create(self, path, w, h, n, dtype):
            created[path] = _Dataset([np.zeros((h, w)) for _ in range(n)], [''] * n)
            return created[path]
    class ListedDriver(object):
        def __init__(self, name, meta):
            self.name, self.meta = name, meta

        def GetMetadata(self):
            return self.meta

        def GetDescription(self):
            return self.name
    gdal.DCAP_CREATE, gdal.DCAP_RASTER, gdal.GDT_Float64 = 'DCAP_CREATE', 'DCAP_RASTER', 7
    listed = [ListedDriver('GTiff', {'DCAP_CREATE': 'YES', 'DCAP_RASTER': 'YES', 'DMD_LONGNAME': 'GeoTIFF', 'DMD_EXTENSIONS': 'tif tiff'}),
              ListedDriver('ENVI', {'DCAP_CREATE': 'YES', 'DCAP_RASTER': 'YES', 'DMD_LONGNAME': 'ENVI .hdr Labelled'}),
              ListedDriver('JP2ECW', {'DCAP_RASTER': 'YES', 'DMD_LONGNAME': 'read only'}),
              ListedDriver('GPKG', {'DCAP_CREATE': 'YES', 'DMD_LONGNAME': 'vector only'})]
    gdal.GetDriverCount = lambda: len(listed)
    gdal.GetDriver = lambda i: listed[i]
    gdal.Open = lambda name: store.get(os.path.basename(name))
    gdal.GetDriverByName = lambda fmt: Driver() if fmt in ('GTiff', 'ENVI', 'PCIDSK') else None
    gdal.created = created
    return gdal


def fake_osr():
    """osgeo.osr stand-in whose "projection" is UTM-like metres = 600000 + 1000 * lon, 5000000 + 1000 * (lat - 45):
    enough to check the geo-transform inversion of s2_tiles_supres.py:141-163."""
    osr = types.ModuleType('osgeo.osr')

    class SpatialReference(object):
        def ImportFromWkt(self, wkt):
            self.wkt = wkt

        def SetWellKnownGeogCS(self, name):
            self.wkt = name

    class CoordinateTransformation(object):
        def __init__(self, src, dst):
            assert src.wkt == 'WGS84' and dst.wkt.startswith('PROJCS')

        def TransformPoint(self, lon, lat, h):
            return 600000.0 + 1000.0 * lon, 5000000.0 + 1000.0 * (lat - 45.0), h
    osr.SpatialReference, osr.CoordinateTransformation = SpatialReference, CoordinateTransformation
    return osr


