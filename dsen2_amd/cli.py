"""Command line for the drop-in path: the super-resolution flow of the reference's testing/s2_tiles_supres.py.

    python -m dsen2_amd.cli INPUT [OUTPUT] [--run_60] [--copy_original_bands] [--roi_x_y x1,y1,x2,y2]
                                  [--output_file_format GTiff|ENVI|...|npz] [--select_UTM ZONE] [--save_prefix P]
                                  [--roi_lon_lat lon1,lat1,lon2,lat2] [--list_bands] [--list_UTM]
                                  [--list_output_file_formats]
                                  [--bands10 B4,B3,B2,B8] [--models DIR] [--precision fp32|bf16|bf16x3] [--deep]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m dsen2_amd.cli INPUT [OUTPUT] ...                       one process per GPU: the patches of the tile are
        sharded over the ranks (dsen2_amd/dist.py: RCCL over xGMI), rank 0 receives the predictions, recomposes
        and is the only rank that prints or writes.  Every rank opens INPUT itself and holds the whole host arrays
        (2.7 GB as uint16 for a 10980^2 product) but uploads only the rows its patches read.

Two kinds of INPUT:
  * an array file — .npz with keys data10 [x,y,4], data20 [x/2,y/2,6], data60 [x/6,y/6,2] (aliases d10/d20/d60),
    or a MATLAB v7.3 .mat with im10/im20/im60 as in the reference's data/*.mat (read by hdf5_min.py; CHW, transposed like
    testing/demoDSen2.py:14-28).  Output: np.savez(output, bands={description: 2-D array}) exactly like the
    reference's npz fallback (s2_tiles_supres.py:419-420).
  * anything else (a Sentinel-2 .zip / SAFE .xml) is opened with GDAL when `osgeo` is importable: sub-dataset and
    band selection, ROI rounding, ReadAsArray into HWC (s2_tiles_supres.py:102-329) and the GTiff/ENVI writer with
    the geo-transform shifted to the ROI (:371-413); a format GDAL cannot create falls back to npz (:350-360).
    Without `osgeo` this branch says so and exits; the reference's own script also keeps working unchanged, it only
    needs `from supres import DSen2_20, DSen2_60` to resolve to dsen2_amd.supres.

CHANNEL ORDER.  The 10 m array is in the order of the SAFE product's 10 m sub-dataset, which is what
s2_tiles_supres.py reads and what the checkpoints were trained on: B4, B3, B2, B8 (red, green, blue, NIR).
`--copy_original_bands` labels the channels with that list; an array file in another order says so with
--bands10 (or a `bands10` entry in the .npz).  20 m: B5 B6 B7 B8A B11 B12; 60 m: B1 B9 (B10 is not super-resolved).
"""
from __future__ import division

import argparse
import os
import re
import sys

import numpy as np

BANDS10 = ['B4', 'B3', 'B2', 'B8']            # order of the 10 m sub-dataset in a SAFE product
BANDS20 = ['B5', 'B6', 'B7', 'B8A', 'B11', 'B12']
BANDS60 = ['B1', 'B9']
ARRAY_EXTENSIONS = ('.npz', '.mat')


def _npz_member_memmap(path, name):
    """Member `name` (.npy) of an UNCOMPRESSED .npz (what np.savez writes: ZIP_STORED) as a read-only np.memmap straight into the
    zip file, or None when the member is deflated / not a plain array.  numpy's own np.load(..., mmap_mode) does not reach
    into .npz files; a stored member is just a .npy file at a known offset."""
    import struct
    import zipfile
    with zipfile.ZipFile(path) as zf:
        try:
            info = zf.getinfo(name + '.npy')
        except KeyError:
            return None
        if info.compress_type != zipfile.ZIP_STORED:
            return None
    with open(path, 'rb') as f:
        f.seek(info.header_offset)
        local = f.read(30)                                   # local file header: signature .. extra-field length
        if local[:4] != b'PK\x03\x04':
            return None
        name_len, extra_len = struct.unpack('<HH', local[26:30])
        f.seek(info.header_offset + 30 + name_len + extra_len)
        try:
            major, minor = np.lib.format.read_magic(f)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if (major, minor) == (1, 0)
                                     else np.lib.format.read_array_header_2_0(f))
        except Exception:
            return None
        if fortran or dtype.hasobject:
            return None
        return np.memmap(path, dtype=dtype, mode='r', offset=f.tell(), shape=shape, order='C')


def _load(path, lazy=False):
    """(data10, data20, data60) of an array file.  lazy=True (the command line under torch.distributed, one process per GPU):
    members of an uncompressed .npz come back as read-only memory maps — DSen2_20 / DSen2_60 slice exactly the rows their
    share of the patches needs (supres._run), so a rank pages in 1/world of the tile and the ranks of a node share ONE copy
    in the page cache instead of holding the whole tile each; compressed members are read whole, as before."""
    ext = os.path.splitext(path)[1].lower()
    if ext == '.npz':
        z = np.load(path)
        def pick(*names):
            for n in names:
                if n in z:
                    if lazy:
                        m = _npz_member_memmap(path, n)
                        if m is not None:
                            return m
                    return z[n]
            return None
        return pick('data10', 'd10'), pick('data20', 'd20'), pick('data60', 'd60')
    if ext == '.mat':
        from . import hdf5_min                 # MATLAB v7.3 = HDF5 behind a 512-byte user block; no h5py needed
        # testing/demoDSen2.py:14-28 (readh5): CHW -> HWC
        return hdf5_min.read_with(path, lambda f: tuple(np.array(f[k]).transpose() if k in f else None
                                                        for k in ('im10', 'im20', 'im60')),
                                  'convert to .npz with keys data10/data20/data60')
    raise ValueError('unsupported input %r (use .npz or .mat)' % path)


def snap_roi(x1, y1, x2, y2, width, height):
    """s2_tiles_supres.py:111-120 — ROI clipped to the raster and grown to 60 m pixel boundaries (10 m pixels)."""
    xmin = max(min(x1, x2, width - 1), 0)
    xmax = min(max(x1, x2, 0), width - 1)
    ymin = max(min(y1, y2, height - 1), 0)
    ymax = min(max(y1, y2, 0), height - 1)
    return int(xmin / 6) * 6, int(ymin / 6) * 6, int((xmax + 1) / 6) * 6 - 1, int((ymax + 1) / 6) * 6 - 1


def lon_lat_to_pixel(ds, osr, lon, lat):
    """s2_tiles_supres.py:141-163 — WGS84 (lon, lat) -> (x, y) pixel of a geo-referenced dataset: transform into the
    dataset's projection, subtract the origin, apply the inverse of the geo-transform's 2x2 matrix."""
    x0, ax, bx, y0, ay, by = ds.GetGeoTransform()
    proj = osr.SpatialReference()
    proj.ImportFromWkt(ds.GetProjection())
    wgs = osr.SpatialReference()
    wgs.SetWellKnownGeogCS('WGS84')
    px, py, _ = osr.CoordinateTransformation(wgs, proj).TransformPoint(lon, lat, 0.)
    px, py = px - x0, py - y0
    det = ax * by - ay * bx
    return int((by * px - bx * py) / det), int((-ay * px + ax * py) / det)


def creatable_raster_formats(gdal):
    """s2_tiles_supres.py:64-80 (--list_output_file_formats): 'NAME: long name (extensions)' of every GDAL driver
    that can create raster files."""
    lines = []
    for i in range(gdal.GetDriverCount()):
        driver = gdal.GetDriver(i)
        meta = driver.GetMetadata() if driver else None
        if not meta or meta.get(gdal.DCAP_CREATE) != 'YES' or meta.get(gdal.DCAP_RASTER) != 'YES':
            continue
        line = driver.GetDescription()
        if 'DMD_LONGNAME' in meta:
            line += ': ' + meta['DMD_LONGNAME']
        if 'DMD_EXTENSIONS' in meta:
            line += ' (' + meta['DMD_EXTENSIONS'] + ')'
        lines.append(line)
    return lines


def short_band_name(description):
    """s2_tiles_supres.py:243-248 — 'B4, central wavelength 665 nm' -> 'B4'."""
    for sep in (',', ' '):
        if sep in description:
            return description[:description.find(sep)]
    return description[:3]


def tidy_description(description, fmt):
    """s2_tiles_supres.py:217-225 — band descriptions as the reference writes them."""
    m = re.match(r'(.*?), central wavelength (\d+) nm', description)
    if m:
        return m.group(1) + ' (' + m.group(2) + ' nm)'
    if fmt == 'ENVI' and ',' in description:       # ENVI band names must not contain commas
        pos = description.find(',')
        return description[:pos] + description[pos + 1:]
    return description


class LazyRows(object):
    """An HWC image whose rows are read when asked for: `img[r0:r1]` reads (decodes) those rows only.  Under
    torch.distributed every rank runs this command line, and DSen2_20 / DSen2_60 slice exactly the rows their share of the
    patches needs (supres._run: `d[r0:r1]`) — with a lazily read product a rank decodes 1/world of the JPEG2000 tile
    instead of all of it.  The last window is kept, with a margin, so that the second network's slightly different
    window (patches of 128 vs 192) is served from memory; `np.asarray(img)` reads everything."""
    MARGIN_10M = 192          # rows of the 10 m image kept either side of a requested window

    def __init__(self, read_rows, shape, dtype, margin=0):
        self._read, self.shape, self.dtype, self._margin = read_rows, tuple(int(v) for v in shape), np.dtype(dtype), int(margin)
        self.ndim = 3
        self._win = None          # (r0, r1, array)
        self.rows_read = 0        # rows decoded so far (tests, and the line the command prints per rank)

    def _window(self, r0, r1):
        if self._win is None or r0 < self._win[0] or r1 > self._win[1]:
            a0, a1 = max(0, r0 - self._margin), min(self.shape[0], r1 + self._margin)
            self._win = (a0, a1, self._read(a0, a1))
            self.rows_read += a1 - a0
        return self._win[2][r0 - self._win[0]:r1 - self._win[0]]

    def __getitem__(self, key):
        rows = key[0] if isinstance(key, tuple) else key
        if isinstance(rows, slice) and rows.step in (None, 1):
            r0, r1, _ = rows.indices(self.shape[0])
            part = self._window(r0, max(r0, r1))
            return part[(slice(None),) + tuple(key[1:])] if isinstance(key, tuple) else part
        return np.asarray(self)[key]

    def __array__(self, dtype=None, copy=None):
        a = self._window(0, self.shape[0])
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.shape[0]


class GdalProduct(object):
    """The GDAL side of s2_tiles_supres.py for one product: which sub-datasets and bands, the ROI, the arrays."""

    def __init__(self, gdal, path, want, roi_x_y=None, select_utm='', fmt='GTiff', roi_lon_lat=None, osr=None):
        self.gdal, self.fmt = gdal, fmt
        if roi_lon_lat and not roi_x_y and osr is None:
            raise ImportError('--roi_lon_lat needs osgeo.osr')
        raster = gdal.Open(path)
        if raster is None:
            raise OSError('GDAL cannot open %r' % path)
        groups = {'10m': [], '20m': [], '60m': [], 'other': []}
        for name, desc in raster.GetSubDatasets():
            key = next((k for k in ('10m', '20m', '60m') if (k + ' resolution') in desc), 'other')
            groups[key].append((name, desc))
        tens = groups['10m'] or groups['other']
        if not tens or not groups['20m']:
            raise ValueError('no 10 m / 20 m sub-dataset in %r' % path)
        # several UTM zones in one product: the requested one, else the one whose ROI covers most pixels (:102-187)
        best = None
        self.coverage = {}                         # UTM zone -> ROI coverage in 10 m pixels (--list_UTM, :173-176,189-193)
        for idx, (name, desc) in enumerate(tens):
            ds = gdal.Open(name)
            w, h = ds.RasterXSize, ds.RasterYSize
            if roi_x_y:                            # pixels win over lon/lat when both are given (:125-137)
                box = snap_roi(roi_x_y[0], roi_x_y[1], roi_x_y[2], roi_x_y[3], w, h)
            elif roi_lon_lat:
                xa, ya = lon_lat_to_pixel(ds, osr, roi_lon_lat[0], roi_lon_lat[1])
                xb, yb = lon_lat_to_pixel(ds, osr, roi_lon_lat[2], roi_lon_lat[3])
                box = snap_roi(xa, ya, xb, yb, w, h)
            else:
                box = (0, 0, w - 1, h - 1)
            utm = desc[desc.find('UTM'):] if 'UTM' in desc else ''
            area = (box[2] - box[0] + 1) * (box[3] - box[1] + 1)
            self.coverage[utm] = max(area, self.coverage.get(utm, 0))
            if select_utm and utm == select_utm:
                best = (area, idx, box, utm)
                break
            if best is None or area > best[0]:
                best = (area, idx, box, utm)
        _, idx, (self.xmin, self.ymin, self.xmax, self.ymax), self.utm = best
        self.valid = self.xmax >= self.xmin and self.ymax >= self.ymin
        if not self.valid:                         # main() prints the reference's message and exits 0 (:196-198)
            return

        def same_zone(cands):
            hit = [c for c in cands if self.utm and self.utm in c[1]]
            return (hit or cands[idx:idx + 1] or cands[:1] or [None])[0]
        s20 = same_zone(groups['20m'])
        self.ds = {'10m': gdal.Open(tens[idx][0]), '20m': gdal.Open(s20[0])}
        s60 = same_zone(groups['60m'])
        self.ds['60m'] = gdal.Open(s60[0]) if s60 else None
        # the descriptions of the selected sub-datasets: the reference names them when it loads the data (:312,318,325)
        self.sub_desc = {'10m': tens[idx][1], '20m': s20[1], '60m': s60[1] if s60 else ''}
        # bands by short name, in sub-dataset order; a name is consumed once (:257-292)
        want = list(want)
        self.names, self.index, self.descriptions = {}, {}, {}
        for key in ('10m', '20m', '60m'):
            self.names[key], self.index[key] = [], []
            ds = self.ds[key]
            for b in range(ds.RasterCount if ds is not None else 0):
                desc = tidy_description(ds.GetRasterBand(b + 1).GetDescription(), fmt)
                sn = short_band_name(desc)
                if sn in want:
                    want.remove(sn)
                    self.names[key].append(sn)
                    self.index[key].append(b)
                    self.descriptions[sn] = desc

    def band_listing(self):
        """--list_bands (s2_tiles_supres.py:229-239): every band of the three selected sub-datasets."""
        lines = []
        for key in ('10m', '20m', '60m'):
            lines += ['', '%s bands:' % key]
            ds = self.ds[key]
            lines += ['- ' + tidy_description(ds.GetRasterBand(b + 1).GetDescription(), self.fmt)
                      for b in range(ds.RasterCount if ds is not None else 0)]
        return lines + ['']

    def read(self, key, row0=0, row1=None):
        """HWC array of the selected bands of one resolution, ROI applied (:311-329); rows [row0, row1) of it only when
        given (rows of THAT resolution, counted from the ROI's first row)."""
        if not self.index[key]:
            return None
        div = {'10m': 1, '20m': 2, '60m': 6}[key]
        xs, ys = (self.xmax - self.xmin + 1) // div, (self.ymax - self.ymin + 1) // div
        row1 = ys if row1 is None else row1
        a = self.ds[key].ReadAsArray(xoff=self.xmin // div, yoff=self.ymin // div + row0, xsize=xs, ysize=row1 - row0,
                                     buf_xsize=xs, buf_ysize=row1 - row0)
        a = np.asarray(a)
        if a.ndim == 2:
            a = a[None]
        return np.moveaxis(a, 0, 2)[:, :, self.index[key]]

    def rows(self, key):
        """The same image as `read(key)`, read on demand: see LazyRows."""
        if not self.index[key]:
            return None
        div = {'10m': 1, '20m': 2, '60m': 6}[key]
        probe = self.read(key, 0, 1)
        shape = ((self.ymax - self.ymin + 1) // div, (self.xmax - self.xmin + 1) // div, len(self.index[key]))
        return LazyRows(lambda r0, r1: self.read(key, r0, r1), shape, probe.dtype, margin=LazyRows.MARGIN_10M // div)

    def writer(self, path, width, height, nbands):
        """A GDAL dataset to write `nbands` float64 bands into, geo-referenced to the ROI (:371-382); None when the
        driver cannot create files (the caller then falls back to npz, :350-360)."""
        gdal = self.gdal
        driver = gdal.GetDriverByName(self.fmt)
        meta = driver.GetMetadata() if driver else {}
        if not driver or meta.get(gdal.DCAP_CREATE) != 'YES':
            return None
        out = driver.Create(path, width, height, nbands, gdal.GDT_Float64)
        geot = list(self.ds['10m'].GetGeoTransform())
        geot[0] += self.xmin * 10          # upper-left corner moves with the ROI: 10 m per pixel
        geot[3] -= self.ymin * 10
        out.SetGeoTransform(tuple(geot))
        out.SetProjection(self.ds['10m'].GetProjection())
        return out


def main(argv=None):
    ap = argparse.ArgumentParser(description='Perform super-resolution on Sentinel-2 with DSen2 on MI355X.')
    ap.add_argument('data_file', nargs='?')
    ap.add_argument('output_file', nargs='?')
    ap.add_argument('--roi_x_y', default='', help='x_1,y_1,x_2,y_2 on the 10m bands; extended to 60m pixel boundaries')
    ap.add_argument('--roi_lon_lat', default='', help='GDAL input: lon_1,lat_1,lon_2,lat_2 (WGS84, decimal); needs osgeo.osr')
    ap.add_argument('--list_bands', action='store_true', help='GDAL input: list the bands of the selected UTM zone and exit')
    ap.add_argument('--list_UTM', action='store_true', help='GDAL input: list the UTM zones with their ROI coverage and exit')
    ap.add_argument('--list_output_file_formats', action='store_true', help='list the raster formats GDAL can create and exit')
    ap.add_argument('--run_60', action='store_true', help='also super-resolve the 60m bands (B1,B9)')
    ap.add_argument('--copy_original_bands', action='store_true')
    ap.add_argument('--save_prefix', default='')
    ap.add_argument('--output_file_format', default=None, help='GDAL driver name (GDAL input: default GTiff) or npz')
    ap.add_argument('--select_UTM', default='', help='GDAL input: UTM zone to use (default: largest ROI coverage)')
    ap.add_argument('--bands10', default=None,
                    help='array input: names of the 10 m channels in the order they are stored (default B4,B3,B2,B8)')
    ap.add_argument('--models', default=None, help='directory with the checkpoints (default: supres.MDL_PATH)')
    ap.add_argument('--deep', action='store_true', help='VDSen2 (d=32, F=256)')
    ap.add_argument('--precision', default=None, choices=['fp32', 'bf16', 'bf16x3'])
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='under torch.distributed.run: nccl = RCCL, one GPU per rank; gloo = rehearsal with shared GPUs')
    args = ap.parse_args(argv)

    if args.list_output_file_formats:                              # s2_tiles_supres.py:64-80: before anything is opened
        try:
            from osgeo import gdal
        except ImportError:
            print('GDAL (osgeo) is not importable: the only output format is npz')
            return 2
        for line in creatable_raster_formats(gdal):
            print(line)
        return 0
    if not args.data_file:
        ap.error('the following arguments are required: data_file')

    # One process per GPU under torch.distributed.run: only rank 0 talks and writes.  The process group is entered
    # before the first HIP call (dist.init_from_env) and left on every way out.
    from . import dist
    rank, _, world = dist.launched_world()
    keep_stdout = sys.stdout
    try:
        if rank != 0:
            sys.stdout = open(os.devnull, 'w')
        if world > 1:
            dist.init_from_env(args.backend)
        rc = _run(args)
        if world > 1:
            dist.finalize()                # barrier, then leave the group (a rank that raised skips the barrier:
        return rc                          # torch.distributed.run then ends its peers)
    finally:
        if sys.stdout is not keep_stdout:
            sys.stdout.close()
            sys.stdout = keep_stdout


def _run(args):
    from . import supres
    if args.models:
        supres.MDL_PATH = os.path.join(args.models, '')
    if args.precision:
        supres.PRECISION = args.precision

    roi = [float(v) for v in re.split(',', args.roi_x_y)] if args.roi_x_y else None
    product = None
    is_array = os.path.splitext(args.data_file)[1].lower() in ARRAY_EXTENSIONS
    if is_array:
        from . import dist as _dist
        data10, data20, data60 = _load(args.data_file, lazy=_dist.rank_world()[1] > 1)
        if data10 is None or data20 is None:
            print('No super-resolution performed, exiting')          # s2_tiles_supres.py:346-348
            return 0
        names10 = list(BANDS10)
        if args.bands10:
            names10 = [b.strip() for b in args.bands10.split(',')]
        elif args.data_file.lower().endswith('.npz') and 'bands10' in np.load(args.data_file):
            names10 = [str(b) for b in np.load(args.data_file)['bands10']]
        if len(names10) < data10.shape[2]:
            raise ValueError('%d names for %d 10 m channels' % (len(names10), data10.shape[2]))
        names10, names20, names60 = names10[:data10.shape[2]], list(BANDS20), list(BANDS60)
        descriptions = dict((b, b) for b in names10 + names20 + names60)
        if args.roi_lon_lat or args.list_bands or args.list_UTM or args.select_UTM:
            print('--roi_lon_lat / --list_bands / --list_UTM / --select_UTM need a geo-referenced product; '
                  '%s is an array file' % args.data_file)
            return 2
        if args.output_file_format not in (None, 'npz'):
            print('array input carries no geo-reference: --output_file_format %s is ignored, writing npz'
                  % args.output_file_format)
        if roi:
            xmin, ymin, xmax, ymax = snap_roi(roi[0], roi[1], roi[2], roi[3], data10.shape[1], data10.shape[0])
            print('Selected pixel region: xmin=%d, ymin=%d, xmax=%d, ymax=%d:' % (xmin, ymin, xmax, ymax))
            if xmax < xmin or ymax < ymin:                         # smaller than one 60 m cell (s2_tiles_supres.py:196-198)
                print('Invalid region of interest / UTM Zone combination')
                return 0
            data10 = data10[ymin:ymax + 1, xmin:xmax + 1]
            data20 = data20[ymin // 2:(ymax + 1) // 2, xmin // 2:(xmax + 1) // 2]
            if data60 is not None:
                data60 = data60[ymin // 6:(ymax + 1) // 6, xmin // 6:(xmax + 1) // 6]
        fmt = 'npz'
        default_out = os.path.split(args.data_file)[1] + '.npz'
    else:
        try:
            from osgeo import gdal
        except ImportError:
            print('%s is not an array file (.npz / .mat) and GDAL (osgeo) is not importable: convert the product to '
                  '.npz (keys data10, data20, data60), or run the reference script with the supres shim of '
                  'INTEGRATION.md' % args.data_file)
            return 2
        try:
            from osgeo import osr
        except ImportError:
            osr = None
        fmt = args.output_file_format or 'GTiff'
        lon_lat = [float(v) for v in re.split(',', args.roi_lon_lat)] if args.roi_lon_lat else None
        want = 'B1,B2,B3,B4,B5,B6,B7,B8,B8A,B9,B11,B12' if args.run_60 else 'B2,B3,B4,B5,B6,B7,B8,B8A,B11,B12'
        try:
            product = GdalProduct(gdal, args.data_file, want.split(','), roi, args.select_UTM, fmt, lon_lat, osr)
        except (ValueError, ImportError) as e:
            print(e)
            return 2
        if args.list_UTM:                                          # :189-193
            print('List of UTM zones (with ROI coverage in pixels):')
            for zone, area in product.coverage.items():
                print('%s (%d)' % (zone, area))
            return 0
        print('Selected UTM Zone:', product.utm)
        print('Selected pixel region: xmin=%d, ymin=%d, xmax=%d, ymax=%d:' % (product.xmin, product.ymin, product.xmax, product.ymax))
        print('Image size: width=%d x height=%d' % (product.xmax - product.xmin + 1, product.ymax - product.ymin + 1))
        if not product.valid:
            print('Invalid region of interest / UTM Zone combination')
            return 0                                               # the reference exits 0 here too (:196-198)
        if args.list_bands:                                        # :229-239, then the selection lines, then exit (:295-296)
            for line in product.band_listing():
                print(line)
        for key in ('10m', '20m', '60m'):                          # (:257-292: "Selected 10m bands:" + " name" per band)
            print('Selected %s bands:%s' % (key, ''.join(' ' + n for n in product.names[key])))
        if args.list_bands:
            return 0
        if not args.output_file:                                   # :298-301 (the reference goes on, with the input's name)
            print('Error: you must provide the name of an output file. I will set it identical to the input...')
        for key in ('10m', '20m', '60m'):                          # :311-329
            if product.index[key]:
                print('Loading selected data from: %s' % product.sub_desc[key])
        from . import dist as _dist
        if _dist.rank_world()[1] > 1:
            # one process per GPU: each reads the rows its patches need, when DSen2_20 / DSen2_60 ask for them
            data10, data20, data60 = product.rows('10m'), product.rows('20m'), product.rows('60m')
        else:
            data10, data20, data60 = product.read('10m'), product.read('20m'), product.read('60m')
        names10, names20, names60 = product.names['10m'], product.names['20m'], product.names['60m']
        descriptions = product.descriptions
        if data10 is None or data20 is None:
            print('No super-resolution performed, exiting')
            return 0
        default_out = os.path.split(args.data_file)[1] + '.tif'
    if args.output_file_format == 'npz':
        fmt = 'npz'

    output_file = args.save_prefix + (args.output_file or default_out)
    if fmt == 'ENVI' and output_file[-4:].lower() == '.hdr':
        output_file = output_file[:-4] + '.bin'                    # ENVI wants the .bin name (:305-307)

    sr60 = None
    if args.run_60 and data60 is not None:
        print('Super-resolving the 60m data into 10m bands')
        sr60 = supres.DSen2_60(data10, data20, data60, deep=args.deep)
    print('Super-resolving the 20m data into 10m bands')
    sr20 = supres.DSen2_20(data10, data20, deep=args.deep)
    if isinstance(data10, LazyRows):
        sys.stderr.write('rank %d read %d of %d rows of the 10 m bands\n' % (_dist.rank_world()[0], data10.rows_read, data10.shape[0]))
    if sr20 is None:                                               # a rank other than 0 of a multi-GPU run
        return 0
    if args.copy_original_bands:
        data10 = np.asarray(data10)                                # rank 0 writes them: all rows after all

    if sr60 is not None:
        sr, sr_names = np.concatenate((sr20, sr60), axis=2), names20 + names60
    else:
        sr, sr_names = sr20, names20
    planes = []                                                    # (description, 2-D array) in output order
    if args.copy_original_bands:
        planes += [(descriptions[bn], data10[:, :, bi]) for bi, bn in enumerate(names10)]
    planes += [('SR' + descriptions[bn], sr[:, :, bi]) for bi, bn in enumerate(sr_names)]

    dataset = None
    if fmt != 'npz':
        dataset = product.writer(output_file, data10.shape[1], data10.shape[0], len(planes))
        if dataset is None:
            print("Gdal doesn't support creating %s files" % fmt)
            print('Writing to npz as a fallback')
            fmt = 'npz'
    sys.stdout.write('Writing')
    if args.copy_original_bands:
        sys.stdout.write(' the original 10m bands and')
    print(' the super-resolved bands in %s' % output_file)
    if fmt == 'npz':
        bands = dict()
        for desc, plane in planes:
            bands[desc] = plane
        np.savez(output_file, bands=bands)
    else:
        for i, (desc, plane) in enumerate(planes):
            band = dataset.GetRasterBand(i + 1)
            band.SetDescription(desc)
            band.WriteArray(plane)
        dataset.FlushCache()
    for desc, _ in planes:
        print(desc)
    return 0


if __name__ == '__main__':
    sys.exit(main())
