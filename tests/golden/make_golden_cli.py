#!/usr/bin/env python3
"""Capture what the REFERENCE's own command line — /root/reference/testing/s2_tiles_supres.py, unmodified, run as __main__ —
prints and writes for a set of argument lists, so that tests/test_cli_vs_reference_runs.py can hold dsen2_amd.cli to it on any
machine (the reference does not travel; this script and its output do).

    python tests/golden/make_golden_cli.py          -> tests/golden/cli_reference_runs.json (+ .npz)

The script's two imports that the image lacks are supplied in-process: `osgeo` = the in-memory stand-in of tests/fake_gdal.py
(a seeded 96 x 96 "product" with the band descriptions of a Sentinel-2 L1C SAFE), `supres` = a stand-in network (nearest-
neighbour up-sampling) behind the reference's names DSen2_20 / DSen2_60.  Neither is an oracle for arithmetic: what is
recorded is the FLOW of s2_tiles_supres.py:61-420 — sub-dataset and band selection, ROI snapping, what is read, which planes
go to the writer under which descriptions and geo-transform, every line it prints, how it exits.
"""
import contextlib
import io
import json
import os
import runpy
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fake_gdal as fg      # noqa: E402

REFERENCE_SCRIPT = '/root/reference/testing/s2_tiles_supres.py'
N = 96
# argument lists (the output path is a plain name: the stand-in driver keeps datasets in memory; npz files go to a scratch dir)
CASES = {
    'default': ['S2A.zip', 'sr.tif'],
    'run_60_copy': ['S2A.zip', 'all.tif', '--run_60', '--copy_original_bands'],
    'roi_x_y': ['S2A.zip', 'roi.tif', '--run_60', '--copy_original_bands', '--roi_x_y', '13,7,40,30'],
    'roi_swapped_points': ['S2A.zip', 'roi2.tif', '--roi_x_y', '70,66,20,31'],
    'roi_lon_lat': ['S2A.zip', 'll.tif', '--roi_lon_lat', '0.13,44.93,0.40,44.70'],
    'envi_hdr': ['S2A.zip', 'out.hdr', '--output_file_format', 'ENVI', '--copy_original_bands'],
    'save_prefix': ['S2A.zip', 'p.tif', '--save_prefix', 'res_'],
    'npz_format': ['S2A.zip', 'bands_out', '--output_file_format', 'npz', '--run_60'],
    'npz_fallback': ['S2A.zip', 'fallback_out', '--output_file_format', 'NoSuchDriver'],
    'list_bands': ['S2A.zip', '--list_bands', '--run_60'],
    'list_UTM': ['S2A.zip', '--list_UTM', '--roi_x_y', '13,7,40,30'],
    'list_formats': ['--list_output_file_formats'],
    'list_formats_with_file': ['S2A.zip', '--list_output_file_formats'],
    'select_utm': ['S2A.zip', 'u.tif', '--select_UTM', 'UTM 33N'],
    'roi_too_small': ['S2A.zip', 'tiny.tif', '--roi_x_y', '13,13,15,15'],
    'no_output_name': ['S2A.zip'],
}


def run_reference(argv, scratch):
    d10, d20, d60 = fg.arrays(N)
    gdal = fg.fake_gdal(d10, d20, d60)
    osgeo = types.ModuleType('osgeo')
    osgeo.gdal, osgeo.osr = gdal, fg.fake_osr()
    supres = types.ModuleType('supres')
    supres.DSen2_20 = lambda a10, a20, deep=False: fg.nearest_up(a20, 2)
    supres.DSen2_60 = lambda a10, a20, a60, deep=False: fg.nearest_up(a60, 6)
    saved = {k: sys.modules.get(k) for k in ('osgeo', 'osgeo.gdal', 'osgeo.osr', 'supres')}
    sys.modules.update({'osgeo': osgeo, 'osgeo.gdal': gdal, 'osgeo.osr': osgeo.osr, 'supres': supres})
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [REFERENCE_SCRIPT] + list(argv)
    os.chdir(scratch)
    out, code = io.StringIO(), 0
    try:
        with contextlib.redirect_stdout(out):
            try:
                runpy.run_path(REFERENCE_SCRIPT, run_name='__main__')
            except SystemExit as e:
                code = 0 if e.code is None else e.code
            except Exception as e:          # the reference's own bugs are part of what is recorded (its npz paths crash)
                code = '%s: %s' % (type(e).__name__, e)
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return code, out.getvalue(), gdal.created


def main():
    import tempfile
    text, arrays = {}, {}
    with tempfile.TemporaryDirectory() as scratch:
        for name, argv in CASES.items():
            code, printed, created = run_reference(argv, scratch)
            rec = {'argv': argv, 'exit': code, 'stdout': printed, 'datasets': {}, 'npz': {}}
            for path, ds in created.items():
                rec['datasets'][path] = {'desc': list(ds.desc), 'geot': list(ds.geot), 'proj': ds.proj, 'bands': len(ds.data)}
                for i, plane in enumerate(ds.data):
                    arrays['%s|%s|%d' % (name, path, i)] = np.asarray(plane)
            for f in sorted(os.listdir(scratch)):
                if f.endswith('.npz'):
                    bands = np.load(os.path.join(scratch, f), allow_pickle=True)['bands'].item()
                    rec['npz'][f] = list(bands)
                    for k, v in bands.items():
                        arrays['%s|%s|%s' % (name, f, k)] = np.asarray(v)
                    os.unlink(os.path.join(scratch, f))
            text[name] = rec
    with open(os.path.join(HERE, 'cli_reference_runs.json'), 'w') as f:
        json.dump({'generator': 'tests/golden/make_golden_cli.py', 'reference': 'testing/s2_tiles_supres.py (run unmodified)',
                   'product_size': N, 'cases': text}, f, indent=1, sort_keys=True)
        f.write('\n')
    np.savez_compressed(os.path.join(HERE, 'cli_reference_runs.npz'), **arrays)
    for name, rec in text.items():
        print('%-20s exit %s, %d dataset(s), %d npz, %d lines printed' % (name, rec['exit'], len(rec['datasets']), len(rec['npz']),
                                                                        len(rec['stdout'].splitlines())))


if __name__ == '__main__':
    main()
