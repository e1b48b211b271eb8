"""dsen2_amd.cli on the CPU: band labelling, ROI rounding, and the import-guarded GDAL branch against an in-memory
stand-in for `osgeo.gdal` (GDAL itself is not installed in the build image).  The network is replaced by a stand-in
(nearest-neighbour up-sampling) — what is tested is the flow of testing/s2_tiles_supres.py around it."""
import os
import sys
import types

import numpy as np
import pytest

from fake_gdal import DESC10, DESC20, DESC60, arrays as _arrays, fake_gdal as _fake_gdal, fake_osr as _fake_osr, nearest_up   # noqa: F401

@pytest.fixture
def fake_supres(monkeypatch):
    from dsen2_amd import supres

    monkeypatch.setattr(supres, 'DSen2_20', lambda d10, d20, deep=False: nearest_up(d20, 2))
    monkeypatch.setattr(supres, 'DSen2_60', lambda d10, d20, d60, deep=False: nearest_up(d60, 6))
    return supres


def test_copy_original_bands_labels_follow_the_channel_order(fake_supres, tmp_path):
    """The 10 m channels are labelled in the SAFE sub-dataset order B4,B3,B2,B8 by default and by --bands10 / a
    `bands10` entry otherwise: channel i always ends up under ITS name."""
    from dsen2_amd import cli
    d10, d20, d60 = _arrays()
    inp = str(tmp_path / 'tile.npz')
    np.savez(inp, data10=d10, data20=d20, data60=d60[:, :, :2])
    out = str(tmp_path / 'o1.npz')
    assert cli.main([inp, out, '--copy_original_bands']) == 0
    bands = np.load(out, allow_pickle=True)['bands'].item()
    assert list(bands)[:4] == ['B4', 'B3', 'B2', 'B8']
    for i, name in enumerate(['B4', 'B3', 'B2', 'B8']):
        assert np.array_equal(bands[name], d10[:, :, i])
    out = str(tmp_path / 'o2.npz')
    assert cli.main([inp, out, '--copy_original_bands', '--bands10', 'B2,B3,B4,B8', '--run_60']) == 0
    bands = np.load(out, allow_pickle=True)['bands'].item()
    assert list(bands) == ['B2', 'B3', 'B4', 'B8', 'SRB5', 'SRB6', 'SRB7', 'SRB8A', 'SRB11', 'SRB12', 'SRB1', 'SRB9']
    assert np.array_equal(bands['B2'], d10[:, :, 0]) and np.array_equal(bands['B4'], d10[:, :, 2])
    assert np.array_equal(bands['SRB9'], np.repeat(np.repeat(d60[:, :, 1].astype(np.float32), 6, 0), 6, 1))
    np.savez(inp, data10=d10, data20=d20, bands10=np.array(['B8', 'B2', 'B3', 'B4']))
    out = str(tmp_path / 'o3.npz')
    assert cli.main([inp, out, '--copy_original_bands']) == 0
    bands = np.load(out, allow_pickle=True)['bands'].item()
    assert np.array_equal(bands['B8'], d10[:, :, 0]) and np.array_equal(bands['B4'], d10[:, :, 3])
    with pytest.raises(ValueError):
        cli.main([inp, out, '--copy_original_bands', '--bands10', 'B2,B3'])


def test_roi_is_snapped_to_60m_pixels():
    from dsen2_amd import cli
    assert cli.snap_roi(5, 7, 250, 245, 264, 264) == (0, 6, 245, 245)          # s2_tiles_supres.py:111-120
    assert cli.snap_roi(250, 245, 5, 7, 264, 264) == (0, 6, 245, 245)          # point order does not matter
    assert cli.snap_roi(-10, -10, 10000, 10000, 120, 60) == (0, 0, 119, 59)
    assert cli.short_band_name('B8A, central wavelength 865 nm') == 'B8A'
    assert cli.tidy_description('B4, central wavelength 665 nm', 'GTiff') == 'B4 (665 nm)'
    assert cli.tidy_description('a, b', 'ENVI') == 'a b'


@pytest.fixture
def with_gdal(monkeypatch):
    def install(d10, d20, d60, **kw):
        gdal = _fake_gdal(d10, d20, d60, **kw)
        osgeo = types.ModuleType('osgeo')
        osgeo.gdal, osgeo.osr = gdal, _fake_osr()
        monkeypatch.setitem(sys.modules, 'osgeo', osgeo)
        monkeypatch.setitem(sys.modules, 'osgeo.gdal', gdal)
        monkeypatch.setitem(sys.modules, 'osgeo.osr', osgeo.osr)
        return gdal
    return install


def test_gdal_branch_reads_selects_and_writes_like_the_reference(fake_supres, with_gdal, tmp_path, capsys):
    """s2_tiles_supres.py:102-329 (sub-datasets, band selection by name, ROI, ReadAsArray -> HWC) and :371-413
    (GTiff writer, geo-transform moved to the ROI, descriptions)."""
    from dsen2_amd import cli
    d10, d20, d60 = _arrays(48)
    gdal = with_gdal(d10, d20, d60)
    out = str(tmp_path / 'sr.tif')
    assert cli.main(['S2A.zip', out, '--run_60', '--copy_original_bands', '--roi_x_y', '13,7,40,30']) == 0
    printed = capsys.readouterr().out
    assert 'Selected pixel region: xmin=12, ymin=6, xmax=35, ymax=29' in printed
    ds = gdal.created[out]
    assert ds.flushed and ds.RasterCount == 12 and ds.data[0].shape == (24, 24)
    assert ds.desc == ['B4 (665 nm)', 'B3 (560 nm)', 'B2 (490 nm)', 'B8 (842 nm)', 'SRB5 (705 nm)', 'SRB6 (740 nm)',
                       'SRB7 (783 nm)', 'SRB8A (865 nm)', 'SRB11 (1610 nm)', 'SRB12 (2190 nm)', 'SRB1 (443 nm)', 'SRB9 (945 nm)']
    assert np.array_equal(ds.data[0], d10[6:30, 12:36, 0])                                  # original B4, ROI applied
    assert np.array_equal(ds.data[4], np.repeat(np.repeat(d20[3:15, 6:18, 0], 2, 0), 2, 1))    # SRB5 from the stand-in
    assert np.array_equal(ds.data[11], np.repeat(np.repeat(d60[1:5, 2:6, 1], 6, 0), 6, 1))     # B10 is never selected
    assert ds.geot == (600000.0 + 120, 10.0, 0.0, 5000000.0 - 60, 0.0, -10.0) and ds.proj == 'PROJCS["UTM 33N"]'
    # without --run_60 the 60 m sub-dataset is not read and only the 20 m bands are written
    out2 = str(tmp_path / 'sr20.tif')
    assert cli.main(['S2A.zip', out2]) == 0
    assert gdal.created[out2].RasterCount == 6 and gdal.created[out2].data[0].shape == (48, 48)


def test_gdal_branch_falls_back_to_npz_and_without_gdal_says_so(fake_supres, with_gdal, tmp_path, capsys, monkeypatch):
    from dsen2_amd import cli
    d10, d20, d60 = _arrays(24)
    with_gdal(d10, d20, d60, can_create=False)
    out = str(tmp_path / 'sr.dat')
    assert cli.main(['S2A.zip', out, '--output_file_format', 'GTiff']) == 0            # s2_tiles_supres.py:350-360
    printed = capsys.readouterr().out
    assert "Gdal doesn't support creating GTiff files" in printed and 'Writing to npz as a fallback' in printed
    bands = np.load(out + '.npz', allow_pickle=True)['bands'].item()
    assert list(bands) == ['SRB5 (705 nm)', 'SRB6 (740 nm)', 'SRB7 (783 nm)', 'SRB8A (865 nm)', 'SRB11 (1610 nm)', 'SRB12 (2190 nm)']
    monkeypatch.setitem(sys.modules, 'osgeo', None)                                     # import osgeo -> ImportError
    assert cli.main(['S2A.zip', out]) == 2
    assert 'GDAL (osgeo) is not importable' in capsys.readouterr().out


def test_gdal_listing_options_and_lon_lat_roi(fake_supres, with_gdal, tmp_path, capsys):
    """--list_output_file_formats (s2_tiles_supres.py:64-80), --list_UTM (:189-193), --list_bands (:229-239,295-296)
    and --roi_lon_lat (:141-170): query options print and exit without an output file; the lon/lat box goes through
    the dataset's projection and the inverse geo-transform, then to 60 m boundaries like a pixel box."""
    from dsen2_amd import cli
    d10, d20, d60 = _arrays(48)
    gdal = with_gdal(d10, d20, d60)
    assert cli.main(['--list_output_file_formats']) == 0
    assert capsys.readouterr().out.splitlines() == ['GTiff: GeoTIFF (tif tiff)', 'ENVI: ENVI .hdr Labelled']
    assert cli.main(['S2A.zip', '--list_UTM', '--roi_x_y', '13,7,40,30']) == 0
    assert capsys.readouterr().out.splitlines() == ['List of UTM zones (with ROI coverage in pixels):', 'UTM 33N (576)']
    assert cli.main(['S2A.zip', '--list_bands', '--run_60']) == 0
    printed = capsys.readouterr().out
    assert '\n10m bands:\n- B4 (665 nm)\n- B3 (560 nm)' in printed and '\n60m bands:\n- B1 (443 nm)\n- B9 (945 nm)\n- B10 (1375 nm)\n' in printed
    assert 'Selected 10m bands: B4 B3 B2 B8' in printed and 'Selected 60m bands: B1 B9' in printed
    assert not gdal.created                                        # nothing was super-resolved or written
    # geo-transform (600000, 10, 0, 5000000, 0, -10): x = 100 * lon, y = -100 * (lat - 45)
    out = str(tmp_path / 'll.tif')
    assert cli.main(['S2A.zip', out, '--roi_lon_lat', '0.13,44.93,0.40,44.70']) == 0
    assert 'Selected pixel region: xmin=12, ymin=6, xmax=35, ymax=29' in capsys.readouterr().out
    assert gdal.created[out].data[0].shape == (24, 24)
    assert np.array_equal(gdal.created[out].data[0], np.repeat(np.repeat(d20[3:15, 6:18, 0], 2, 0), 2, 1))
    # a region smaller than one 60 m cell: the reference's message, exit 0, nothing written
    out2 = str(tmp_path / 'tiny.tif')
    assert cli.main(['S2A.zip', out2, '--roi_x_y', '13,13,15,15']) == 0
    assert 'Invalid region of interest / UTM Zone combination' in capsys.readouterr().out and out2 not in gdal.created


def test_array_input_validates_the_roi_and_says_what_it_ignores(fake_supres, tmp_path, capsys):
    """an ROI smaller than one 60 m cell prints the reference's message and exits 0 (s2_tiles_supres.py:196-198)
    instead of failing inside the tiling; a GDAL format / geo option on an array file is reported, not silently dropped."""
    from dsen2_amd import cli
    d10, d20, d60 = _arrays(48)
    inp = str(tmp_path / 'tile.npz')
    np.savez(inp, data10=d10, data20=d20, data60=d60[:, :, :2])
    out = str(tmp_path / 'o.npz')
    assert cli.main([inp, out, '--roi_x_y', '13,13,15,15']) == 0
    assert 'Invalid region of interest' in capsys.readouterr().out and not os.path.exists(out)
    assert cli.main([inp, out, '--output_file_format', 'GTiff']) == 0
    assert '--output_file_format GTiff is ignored, writing npz' in capsys.readouterr().out and os.path.exists(out)
    assert cli.main([inp, out, '--list_bands']) == 2
    assert 'need a geo-referenced product' in capsys.readouterr().out
    with pytest.raises(SystemExit):
        cli.main([])


def test_lazy_rows_read_what_is_asked_for_once():
    """cli.LazyRows: a row window is read when asked for, kept with its margin, re-read only when a request leaves it."""
    from dsen2_amd.cli import LazyRows
    full = np.arange(100 * 7 * 3, dtype=np.uint16).reshape(100, 7, 3)
    calls = []

    def read(r0, r1):
        calls.append((r0, r1))
        return full[r0:r1]
    z = LazyRows(read, full.shape, full.dtype, margin=5)
    assert z.shape == (100, 7, 3) and z.dtype == np.uint16 and len(z) == 100 and z.rows_read == 0
    assert np.array_equal(z[20:40], full[20:40]) and calls == [(15, 45)] and z.rows_read == 30
    assert np.array_equal(z[17:44], full[17:44]) and len(calls) == 1                    # inside the kept window
    assert np.array_equal(z[22:30, :4], full[22:30, :4]) and len(calls) == 1            # tuple keys: rows first
    assert np.array_equal(z[90:100], full[90:100]) and calls[-1] == (85, 100)           # clipped at the image's end
    assert z[50:50].shape == (0, 7, 3)
    assert np.array_equal(np.asarray(z), full) and calls[-1] == (0, 100)
    assert np.array_equal(z[:, :, 1], full[:, :, 1])


def test_gdal_branch_reads_rows_on_demand_under_torch_distributed(with_gdal, tmp_path, capsys, monkeypatch):
    """Several ranks: the command line hands DSen2_20 / DSen2_60 lazily read images, so a rank decodes only the rows its
    patches need (here: a stand-in that, like supres._run, slices the rows of the upper half)."""
    from dsen2_amd import cli, dist, supres
    d10, d20, d60 = _arrays(96)
    with_gdal(d10, d20, d60)
    seen = {}

    def fake20(a10, a20, deep=False):
        seen['types'] = (type(a10).__name__, type(a20).__name__)
        part10, part20 = a10[0:56], a20[0:28]                     # what a rank holding the upper patches uploads
        seen['rows'] = (a10.rows_read, a20.rows_read)
        assert part10.shape == (56, 96, 4) and part20.shape == (28, 48, 6)
        assert np.array_equal(part10, np.asarray(a10)[0:56]) and np.array_equal(part20, np.asarray(a20)[0:28])
        assert sorted(np.asarray(a10)[5, 7].tolist()) == sorted(d10[5, 7].tolist())
        return np.repeat(np.repeat(np.asarray(a20, np.float32), 2, axis=0), 2, axis=1)
    monkeypatch.setattr(supres, 'DSen2_20', fake20)
    monkeypatch.setattr(dist, 'rank_world', lambda: (0, 2))
    monkeypatch.setattr(cli.LazyRows, 'MARGIN_10M', 0)
    assert cli.main(['S2A.zip', str(tmp_path / 'o.tif'), '--copy_original_bands']) == 0
    assert seen['types'] == ('LazyRows', 'LazyRows') and seen['rows'] == (56, 28)
    assert 'rank 0 read' in capsys.readouterr().err


def test_array_file_is_memory_mapped_under_torch_distributed(fake_supres, tmp_path, monkeypatch):
    """One process per GPU: the members of an uncompressed .npz reach DSen2_20 / DSen2_60 as read-only memory maps into the
    zip file (a rank pages in only the rows it slices; the ranks of a node share one copy in the page cache), with the same
    values and the same output as a single-process run; a compressed .npz is read whole, as before; so is any input of a
    single-process run."""
    from dsen2_amd import cli, dist, supres
    d10, d20, d60 = _arrays(96)
    inp, inc = str(tmp_path / 'tile.npz'), str(tmp_path / 'tile_c.npz')
    np.savez(inp, data10=d10, data20=d20, data60=d60[:, :, :2])
    np.savez_compressed(inc, data10=d10, data20=d20, data60=d60[:, :, :2])
    for p, want in ((inp, 'memmap'), (inc, 'ndarray')):
        a10, a20, a60 = cli._load(p, lazy=True)
        assert (type(a10).__name__, type(a20).__name__, type(a60).__name__) == (want,) * 3
        assert np.array_equal(a10, d10) and np.array_equal(a20, d20) and np.array_equal(a60, d60[:, :, :2])
        assert a10.dtype == np.uint16 and np.array_equal(a10[17:40], d10[17:40])
    assert type(cli._load(inp)[0]).__name__ == 'ndarray'
    seen = {}
    real20 = supres.DSen2_20

    def spy20(a10, a20, deep=False):
        seen['types'] = (type(a10).__name__, type(a20).__name__)
        return real20(a10, a20, deep)
    monkeypatch.setattr(supres, 'DSen2_20', spy20)
    out1, out2 = str(tmp_path / 'one.npz'), str(tmp_path / 'two.npz')
    assert cli.main([inp, out1, '--copy_original_bands', '--run_60']) == 0
    assert seen['types'] == ('ndarray', 'ndarray')
    monkeypatch.setattr(dist, 'rank_world', lambda: (0, 2))
    assert cli.main([inp, out2, '--copy_original_bands', '--run_60', '--roi_x_y', '6,12,77,83']) == 0
    assert seen['types'] == ('memmap', 'memmap')
    monkeypatch.setattr(dist, 'rank_world', lambda: (0, 1))
    out3 = str(tmp_path / 'three.npz')
    assert cli.main([inp, out3, '--copy_original_bands', '--run_60', '--roi_x_y', '6,12,77,83']) == 0
    b2, b3 = (np.load(o, allow_pickle=True)['bands'].item() for o in (out2, out3))
    assert list(b2) == list(b3) and all(np.array_equal(b2[k], b3[k]) for k in b2)
