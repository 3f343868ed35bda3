"""Pins the CPU oracle (oracle/) against every known-answer test and data fixture the
reference holds for the regridding hot path.  CPU only.

Each test names the reference test it replays (paths relative to the reference tree).
"""
import math
import os

import numpy as np
import pytest

import oracle
from oracle import proj_oracle as po
import cases

EMEP = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +x_0=7 +y_0=109"
LATLONG = "+ellps=sphere +a=6370 +e=0 +proj=latlong"


# ---- test/testInterpolation.cc:49-58
def test_points2position_ascending():
    got = oracle.points2position([-3., 5., 1.3, 2., 6.], [1., 2., 3., 4., 5.])
    np.testing.assert_allclose(got, [-4., 4., 0.3, 1., 5.], atol=1e-10, rtol=0)


# ---- test/testInterpolation.cc:61-70
def test_points2position_descending():
    got = oracle.points2position([-3., 5., 1.3, 2., 6.], [5., 4., 3., 2., 1.])
    np.testing.assert_allclose(got, [8., 0., 3.7, 3., -1.], atol=1e-10, rtol=0)


# ---- test/testInterpolation.cc:73-80
def test_nearest_kat():
    v = oracle.get_values(oracle.NEAREST, [1., 2., 1., 2.], 0.3, 0.3, 2, 2)
    assert abs(v[0] - 1) < 1e-10


# ---- test/testInterpolation.cc:83-112
def test_bilinear_kat():
    f = np.array([1., 2., 2., 1 + np.sqrt(np.float32(2.0))], dtype=np.float32)
    g = lambda x, y: oracle.get_values(oracle.BILINEAR, f, x, y, 2, 2)[0]
    assert abs(g(0.3, 0.) - 1.3) < 1e-6
    assert abs(g(0.3, 0.0001) - 1.3) < 1e-4
    assert abs(g(0., 0.3) - 1.3) < 1e-6
    assert abs(g(0.0001, 0.3) - 1.3) < 1e-4
    assert not math.isnan(g(0, 0))
    assert not math.isnan(g(1, 1))
    assert math.isnan(g(1.5, 0.5))
    assert math.isnan(g(0.5, 1.5))
    assert math.isnan(g(0.5, -0.5))
    assert math.isnan(g(-0.5, 0.5))


# ---- test/testInterpolation.cc:115-155
def test_bicubic_kat():
    f = np.array([1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1], dtype=np.float32)
    ft = f.reshape(4, 4).T.copy().ravel()
    g = lambda a, x, y: oracle.get_values(oracle.BICUBIC, a, x, y, 4, 4)[0]
    close = lambda want, got: abs(want - got) <= abs(want) * 1e-5  # BOOST_CHECK_CLOSE 1e-3 percent
    assert close(2.0, g(f, 1, 1))
    assert close(2.0, g(f, 1, 1.99999))
    assert close(2.125, g(f, 1, 1.5))
    assert close(2.0, g(f, 1.5, 1))
    assert close(2.0, g(ft, 1, 1))
    assert close(2.0, g(ft, 1.99999, 1))
    assert close(2.125, g(ft, 1.5, 1))
    assert close(2.0, g(ft, 1, 1.5))
    assert math.isnan(g(f, .5, 1))
    assert math.isnan(g(f, 1, .5))
    assert math.isnan(g(f, 2.5, 1))
    assert math.isnan(g(f, 1, 2.5))


def _emep_case(golden_dir):
    iS, jS, lonS, latS = 170, 150, 180, 90
    field = np.full((jS, iS), np.nan, dtype=np.float32)
    raw = np.loadtxt(os.path.join(golden_dir, "inData.txt"))
    field[raw[:, 1].astype(int) - 1, raw[:, 0].astype(int) - 1] = raw[:, 2]
    assert abs(field[50, 93] - 4) < 1e-5  # testInterpolation.cc:340
    iaxis = np.arange(iS) + 1.
    jaxis = np.arange(jS) + 1.
    lon = (np.arange(lonS) + 1) / 2. - 30
    lat = (np.arange(latS) + 1) / 2. + 30
    # mifi_interpolate_f_functional (src/interpolation.c:231-279): degrees -> radians, project
    # the OUTPUT mesh into the input projection, then axis positions
    px, py = po.project_axes(LATLONG, EMEP, np.radians(lon), np.radians(lat))
    px = oracle.points2position(px, iaxis, oracle.PROJ_AXIS)
    py = oracle.points2position(py, jaxis, oracle.PROJ_AXIS)
    return field, px, py, (iS, jS, lonS, latS), lon, lat


# ---- test/testInterpolation.cc:280-393: cell (lon -25, lat 43) == 32 for all three methods
@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
def test_emep_chain_pinned_cell(golden_dir, method):
    field, px, py, (iS, jS, lonS, latS), lon, lat = _emep_case(golden_dir)
    out = oracle.interpolate_values(method, px, py, field, iS, jS, lonS, latS)[0]
    assert lon[9] == -25 and lat[25] == 43
    assert abs(out[25, 9] - 32) < 1e-6


# ---- test/outData.txt (read only by test/longlatcemap.ncl:13, never asserted by the reference):
# 16200 "lon lat value" lines of the EMEP chain.  It is NOT the output of the current
# test_mifi_interpolate_f: it reproduces to print precision (6 digits) as the BILINEAR result with the
# EMEP axes numbered from 0 instead of 1 (an older revision of that test), and predates the border
# branches of src/interpolation.c:903-948 (NEWS: "bilinear: better border handling").  Used as such it
# pins projection + points2position + the interior bilinear arithmetic on 14.7k cells.
def test_emep_chain_outdata_fixture(golden_dir):
    field, px, py, (iS, jS, lonS, latS), lon, lat = _emep_case(golden_dir)
    px, py = px + 1, py + 1  # axes 0..169 / 0..149 instead of 1..170 / 1..150
    out = oracle.interpolate_values(oracle.BILINEAR, px, py, field, iS, jS, lonS, latS)[0].astype(np.float64)
    want = np.loadtxt(os.path.join(golden_dir, "outData.txt"))  # lon-major
    assert want.shape == (lonS * latS, 3)
    np.testing.assert_array_equal(want[:, 0].reshape(lonS, latS)[:, 0], lon)
    np.testing.assert_array_equal(want[:, 1].reshape(lonS, latS)[0, :], lat)
    wv = want[:, 2].reshape(lonS, latS).T
    assert wv[25, 9] == 32  # the "-25 43 32" line the reference's test comment quotes
    both = ~np.isnan(out) & ~np.isnan(wv)
    assert both.sum() > 14000
    rel = np.abs(out[both] - wv[both]) / np.abs(wv[both])
    assert rel.max() < 1e-5
    # cells defined here but not in the fixture are exactly border-branch cells (x or y within half a
    # cell outside the linear range); nothing defined in the fixture is undefined here
    assert not np.any(np.isnan(out) & ~np.isnan(wv))
    extra = ~np.isnan(out) & np.isnan(wv)
    PX, PY = px.reshape(latS, lonS), py.reshape(latS, lonS)
    interior = (np.floor(PX) >= 0) & (np.floor(PX) + 1 < iS) & (np.floor(PY) >= 0) & (np.floor(PY) + 1 < jS)
    assert extra.sum() < 50 and not np.any(extra & interior)


# ---- tests/golden/coordTest.nc: stored 2-D longitude/latitude pin the stere inverse (121 points)
def test_stere_inverse_against_coordtest(golden_dir):
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        x = f.variables["x"].data.astype(np.float64)
        y = f.variables["y"].data.astype(np.float64)
        lon = f.variables["longitude"].data.astype(np.float64)
        lat = f.variables["latitude"].data.astype(np.float64)
        projstr = f.variables["projection_1"].proj4.decode()
    lo, la = po.project_axes(projstr, "+proj=latlong +R=6.371e6", x, y)
    # the file stores the coordinates as float-precision values widened to double
    np.testing.assert_allclose(np.degrees(la).reshape(11, 11), lat, atol=2e-5, rtol=0)
    dlon = (np.degrees(lo).reshape(11, 11) - lon + 180) % 360 - 180
    np.testing.assert_allclose(dlon, 0, atol=2e-5)
    # and the forward projection returns to the axes
    fx, fy = po.transform("+proj=latlong +R=6.371e6", projstr, np.radians(lon.ravel()), np.radians(lat.ravel()))
    xx, yy = np.meshgrid(x, y)
    np.testing.assert_allclose(fx, xx.ravel(), atol=20.0)  # float-precision lon/lat -> tens of metres
    np.testing.assert_allclose(fy, yy.ravel(), atol=20.0)


def _rotation_matrix(proj_in, proj_out, out_x_axis, out_y_axis, out_x_type, out_y_type):
    """mifi_get_vector_reproject_matrix (src/interpolation.c:719-788) + _proj (:441-521), projections via po."""
    ox, oy = len(out_x_axis), len(out_y_axis)
    xa = np.radians(out_x_axis) if out_x_type != oracle.PROJ_AXIS else np.asarray(out_x_axis, float)
    ya = np.radians(out_y_axis) if out_y_type != oracle.PROJ_AXIS else np.asarray(out_y_axis, float)
    xx, yy = np.meshgrid(xa, ya)
    out_x, out_y = xx.ravel(), yy.ravel()
    in_x, in_y = po.transform(proj_out, proj_in, out_x, out_y)
    return _rotation_matrix_field(proj_in, proj_out, in_x, in_y, out_x, out_y, ox, oy)


def _rotation_matrix_field(proj_in, proj_out, in_x, in_y, out_x, out_y, ox, oy):
    d = 1e-3
    assert ox > 1 and oy > 1
    ox2, oy2 = ox // 2, oy // 2
    delta = d * (in_x[ox + 1] - in_x[0])                      # :465 / :490 (both use in_x_field)
    delta2 = d * (in_x[(oy2 + 1) * ox + ox2 + 1] - in_x[oy2 * ox + ox2])
    delta = (delta + delta2) / 2
    if abs(delta) < 1e-9:
        delta = d
    xdx = po.transform(proj_in, proj_out, in_x + delta, in_y)
    ydy = po.transform(proj_in, proj_out, in_x, in_y + delta)
    latlon = po.is_latlong(po.parse(proj_out))
    return oracle.vector_matrix_from_deltas(out_x, out_y, xdx, ydy, delta, delta, latlon)


def _nearest_regrid(proj_in, proj_out, f, in_x, in_y, out_x, out_y, out_types):
    xa = np.radians(out_x) if out_types[0] != oracle.PROJ_AXIS else np.asarray(out_x, float)
    ya = np.radians(out_y) if out_types[1] != oracle.PROJ_AXIS else np.asarray(out_y, float)
    px, py = po.project_axes(proj_out, proj_in, xa, ya)
    px = oracle.points2position(px, in_x, oracle.PROJ_AXIS)
    py = oracle.points2position(py, in_y, oracle.PROJ_AXIS)
    return oracle.interpolate_values(oracle.NEAREST, px, py, f, len(in_x), len(in_y), len(out_x), len(out_y))[0]


# ---- test/testInterpolation.cc:396-453 (rotate 90) and :455-512 (rotate 180)
@pytest.mark.parametrize("lon0,check", [(90, "r90"), (180, "r180")])
def test_vector_rotation_polar(lon0, check):
    p1 = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=0 +lat_ts=60"
    p2 = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=%d +lat_ts=60" % lon0
    ax = np.arange(5) - 2.
    u = np.arange(25, dtype=np.float32)
    v = (25 - np.arange(25)).astype(np.float32)
    T = (oracle.PROJ_AXIS, oracle.PROJ_AXIS)
    uo = _nearest_regrid(p1, p2, u, ax, ax, ax, ax, T)
    vo = _nearest_regrid(p1, p2, v, ax, ax, ax, ax, T)
    m = _rotation_matrix(p1, p2, ax, ax, *T)
    ur, vr = oracle.vector_reproject_values(m, uo, vo, 5, 5)
    ur, vr = ur[0], vr[0]
    ok = ~(np.isnan(uo) | np.isnan(vo))
    assert ok.sum() >= 20
    if check == "r90":   # u -> -v, v -> u  (:448-449)
        assert np.all(np.abs(vo[ok] - ur[ok]) < 1e-4)
        assert np.all(np.abs(uo[ok] + vr[ok]) < 1e-4)
    else:                # (:507-508)
        assert np.all(np.abs(vo[ok] + vr[ok]) < 1e-5)
        assert np.all(np.abs(uo[ok] + ur[ok]) < 1e-5)


# ---- test/testInterpolation.cc:515-583
def test_vector_rotation_keeps_length():
    ia = np.arange(4) + 6.
    ja = np.arange(4) + 108.
    lon = np.arange(4) * 60.
    lat = np.arange(4) / 2. + 88.5
    u = np.arange(16, dtype=np.float32)
    v = (-16 + np.arange(16)).astype(np.float32)
    T = (oracle.LONGITUDE, oracle.LATITUDE)
    uo = _nearest_regrid(EMEP, LATLONG, u, ia, ja, lon, lat, T)
    vo = _nearest_regrid(EMEP, LATLONG, v, ia, ja, lon, lat, T)
    m = _rotation_matrix(EMEP, LATLONG, lon, lat, *T)
    ur, vr = oracle.vector_reproject_values(m, uo, vo, 4, 4)
    d2 = (ur[0].astype(np.float64) ** 2 + vr[0].astype(np.float64) ** 2
          - uo.astype(np.float64) ** 2 - vo.astype(np.float64) ** 2)
    ok = ~np.isnan(d2)
    assert ok.any()
    assert np.all(np.abs(d2[ok]) < 1e-3)


# ---- test/testInterpolation.cc:586-654
def test_vector_direction_angles():
    p1 = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=0 +lat_ts=60"
    ax = (np.arange(5) - 2) * 1000.
    xx, yy = np.meshgrid(ax, ax)
    in_x, in_y = xx.ravel(), yy.ravel()
    out_x, out_y = po.transform(p1, LATLONG, in_x, in_y)
    m = _rotation_matrix_field(p1, LATLONG, in_x, in_y, out_x, out_y, 5, 5)
    a = oracle.vector_reproject_direction(m, np.zeros(25, np.float32), 5, 5)[0]
    close = lambda want, got: abs(want - got) <= abs(want) * 1e-2  # BOOST_CHECK_CLOSE 1 percent
    assert close(315, a[0, 0]) and close(270, a[2, 0]) and close(225, a[4, 0])
    for j in (0, 1):
        o = a[j, 2] - 360 if a[j, 2] > 300 else a[j, 2]
        assert close(10, 10 + o)
    assert close(180, a[3, 2]) and close(180, a[4, 2])
    assert close(45, a[0, 4]) and close(90, a[2, 4]) and close(135, a[4, 4])


# ---- test/testInterpolation.cc:265-278
def test_project_axes_near_pole():
    ox, oy = po.project_axes(EMEP, LATLONG, [6, 7, 8], [108, 109, 110])
    assert np.all(np.degrees(oy) > 89)


# ---- test/testInterpolation.cc:158-262: 1-D blends between two fields
def test_linear_blend_kats():
    A = np.array([0, 1, -1, 1], np.float32)
    B = np.array([1, -1, 0, 1], np.float32)
    out, rc = oracle.get_values_1d(oracle.BLEND_LINEAR, A, B, 1., 1., .5)      # a == b: field A
    assert rc == oracle.OK and np.allclose(out, A, rtol=1e-7)
    out, rc = oracle.get_values_1d(oracle.BLEND_LINEAR, A, B, 1., 2., 1.5)
    assert np.allclose(out, np.float32(.5) * (A + B), rtol=1e-7)
    out, rc = oracle.get_values_1d(oracle.BLEND_LINEAR, A, B, 0., 1., 2.)      # extrapolation
    assert np.allclose(out, A + 2 * (B - A), rtol=1e-7)
    d = oracle.get_values_linear_d(A, B, 1., 2., 1.5)
    assert np.allclose(d, .5 * (A.astype(float) + B), rtol=1e-12)
    assert np.allclose(oracle.get_values_linear_d(A, B, 0., 1., 2.), A.astype(float) + 2 * (B.astype(float) - A), rtol=1e-12)


def test_log_blend_kats():
    A, B = np.array([1000.], np.float32), np.array([100.], np.float32)
    for x, want in ((100., 100.), (1000., 1000.), (500., 729.073), (1500., 1158.482), (200., 370.927), (800., 912.781)):
        out, rc = oracle.get_values_1d(oracle.BLEND_LOG, A, B, 1000., 100., x)
        assert rc == oracle.OK and abs(out[0] - want) / want < 1e-5, (x, out[0])
    for x, want in ((500., 763.1873), (200., 408.0904), (800., 926.384)):    # "results from NCLs vintp2p_ecmwf"
        out, rc = oracle.get_values_1d(oracle.BLEND_LOG_LOG, A, B, 1000., 100., x)
        assert rc == oracle.OK and abs(out[0] - want) / want < 1e-5, (x, out[0])
    assert oracle.get_values_1d(oracle.BLEND_LOG, A, B, 1000., 100., -1.)[1] == oracle.ERROR
    assert oracle.get_values_1d(oracle.BLEND_LOG_LOG, A, B, 0., 100., 5.)[1] == oracle.ERROR


def test_extrapolation_variants_of_the_linear_blend():
    A, B = np.array([1, 2, np.nan], np.float32), np.array([3, 6, 1], np.float32)
    weak = lambda x: oracle.get_values_1d(oracle.BLEND_LINEAR_WEAK_EXTRAPOL, A, B, 0., 1., x)[0]
    none = lambda x: oracle.get_values_1d(oracle.BLEND_LINEAR_NO_EXTRAPOL, A, B, 0., 1., x)[0]
    const = lambda x: oracle.get_values_1d(oracle.BLEND_LINEAR_CONST_EXTRAPOL, A, B, 0., 1., x)[0]
    assert cases.same(weak(1.5), A + np.float32(1.5) * (B - A)) and np.isnan(weak(2.5)).all() and np.isnan(weak(-1.5)).all()
    assert cases.same(none(.25), A + np.float32(.25) * (B - A)) and np.isnan(none(1.25)).all()
    assert cases.same(none(0.), A) and cases.same(none(1.), B)   # copies: no 0 * NaN side effects
    assert cases.same(const(7.), B) and cases.same(const(-3.), A) and cases.same(const(.5), A + np.float32(.5) * (B - A))
    assert cases.same(oracle.get_values_1d(oracle.BLEND_NEAREST, A, B, 0., 1., .9)[0], A)


def test_reference_kats_of_the_linear_family():
    """test/testInterpolation.cc:685-731 literally: in0 = 200 at a = 2, in1 = 300 at b = 3, x = 0.5 .. 4.5."""
    A, B = np.array([200], np.float32), np.array([300], np.float32)
    xs = (0.5, 1.5, 2.5, 3.5, 4.5)
    want = {oracle.BLEND_LINEAR_NO_EXTRAPOL: (np.nan, np.nan, 250, np.nan, np.nan),      # :690-694
            oracle.BLEND_LINEAR_CONST_EXTRAPOL: (200, 200, 250, 300, 300),               # :702-706
            oracle.BLEND_LINEAR_WEAK_EXTRAPOL: (np.nan, 150, 250, 350, np.nan),          # :714-718
            oracle.BLEND_LINEAR: (50, 150, 250, 350, 450)}                               # :726-730
    for kind, expect in want.items():
        for x, w in zip(xs, expect):
            out, rc = oracle.get_values_1d(kind, A, B, 2., 3., x)
            assert rc == oracle.OK
            assert (np.isnan(out[0]) and np.isnan(w)) or abs(out[0] - w) < 0.01, (kind, x, out, w)


def _erai(golden_dir):
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "erai.sfc.40N.0.75d.200301011200.nc"), "r", mmap=False) as f:
        v = f.variables
        e = dict(lon=v["longitude"].data.astype(np.float64), lat=v["latitude"].data.astype(np.float64),
                 proj=v["projection_regular_ll"].proj4.decode(), skt=v["ga_skt"].data.astype(np.float32).reshape(8, 11, 6))
    with netcdf_file(os.path.join(golden_dir, "template_noaa17.nc"), "r", mmap=False) as f:
        e["tlon"], e["tlat"] = f.variables["longitude"].data.astype(np.float64), f.variables["latitude"].data.astype(np.float64)
    return e


def _erai_positions(e, lon_deg, lat_deg):
    """src/CDMInterpolator.cc:1761-1793: template degrees -> radians -> the file's projection -> positions on its axes."""
    x, y = po.transform("+proj=latlong +datum=WGS84 +towgs84=0,0,0 +no_defs", e["proj"], np.radians(lon_deg), np.radians(lat_deg))
    return (oracle.points2position(x, np.radians(e["lon"]), oracle.LONGITUDE), oracle.points2position(y, np.radians(e["lat"]), oracle.LATITUDE))


def test_reference_file_tests_on_the_erai_fixture(golden_dir):
    """test/testInterpolator.cc:220-264 with the oracle alone: ERA-Interim skin temperature (test/erai.sfc.40N...nc) bicubic
    onto the NOAA-17 swath of test/template_noaa17.nc -- the first seven values defined, 270..280 K -- and bilinear to ten
    stations -- all defined, 266..281.1 K, the first 270..280 K."""
    e = _erai(golden_dir)
    px, py = _erai_positions(e, e["tlon"].ravel(), e["tlat"].ravel())
    out = oracle.interpolate_values(oracle.BICUBIC, px, py, e["skt"][0:1], 6, 11, 29, 31).ravel()
    assert out.size == 29 * 31 and np.all(np.isfinite(out[:7]) & (out[:7] > 270) & (out[:7] < 280)), out[:8]
    lat = np.array([59.109, 59.052, 58.994, 58.934, 58.874, 58.812, 58.749, 58.685, 58.62, 64.])
    lon = np.array([4.965, 5.13, 5.296, 5.465, 5.637, 5.81, 5.986, 6.164001, 6.344, 3.])
    px, py = _erai_positions(e, lon.astype(np.float32).astype(np.float64), lat.astype(np.float32).astype(np.float64))
    pts = oracle.interpolate_values(oracle.BILINEAR, px, py, e["skt"], 6, 11, 10, 1).reshape(8, 10)
    assert 270 < pts[0, 0] < 280 and np.all(np.isfinite(pts) & (pts > 266) & (pts < 281.1)), pts


SNYDER_MORE = [   # (geographic side, projection, (lon, lat), (x, y), tolerance): the worked examples of the remaining projections
    ("+proj=latlong +R=1", "+proj=sinu +R=1 +lon_0=-90", (-75., -50.), (0.1682814, -0.8726646), 1e-7),
    ("+proj=latlong +ellps=clrk66", "+proj=sinu +ellps=clrk66 +lon_0=-90", (-75., -50.), (1075471.5, -5540628.0), 0.1),
    ("+proj=latlong +R=1", "+proj=cea +R=1 +lat_ts=30 +lon_0=-75", (80., 35.), (2.3428242, 0.6623090), 1e-7),
    ("+proj=latlong +ellps=clrk66", "+proj=cea +ellps=clrk66 +lat_ts=5 +lon_0=-75", (-78., 5.), (-332699.8, 554248.5), 0.1),
    ("+proj=latlong +R=1", "+proj=ortho +R=1 +lat_0=40 +lon_0=-100", (-110., 30.), (-0.1503837, -0.1651911), 1e-7),
    ("+proj=latlong +R=3", "+proj=aeqd +R=3 +lat_0=40 +lon_0=-100", (100., -20.), (-5.8311398, 5.5444634), 1e-7),
    ("+proj=latlong +R=6371000", "+proj=nsper +R=6371000 +h=500000 +lat_0=39 +lon_0=-77", (-74., 41.), (247194.09, 222485.96), 0.01),
]


@pytest.mark.parametrize("geo,proj,lonlat,xy,tol", SNYDER_MORE)
def test_remaining_projections_reproduce_snyders_worked_examples(geo, proj, lonlat, xy, tol):
    lon, lat = np.radians([lonlat[0]]), np.radians([lonlat[1]])
    x, y = po.transform(geo, proj, lon, lat)
    assert abs(x[0] - xy[0]) < tol and abs(y[0] - xy[1]) < tol, (x, y)
    bl, bp = po.transform(proj, geo, x, y)
    np.testing.assert_allclose([bl[0], bp[0]], [lon[0], lat[0]], atol=1e-9)


def test_datum_shifts_against_the_epsg_worked_examples():
    """EPSG Guidance Note 7-2: geographic <-> geocentric (WGS84: 53 48 33.820 N, 2 07 46.380 E, 73.0 m = X 3771793.968,
    Y 140253.342, Z 5124304.349), geocentric translations WGS84 -> ED50 (+84.87, +96.49, +116.95: 53 48 36.565 N,
    2 07 51.477 E, 28.02 m) and the position-vector transformation WGS72 -> WGS84."""
    lat, lon = math.radians(53 + 48 / 60 + 33.820 / 3600), math.radians(2 + 7 / 60 + 46.380 / 3600)
    es84 = 0.0066943799901413165
    X, Y, Z = po.geodetic_to_geocentric(lon, lat, 73.0, 6378137.0, es84)
    assert abs(X - 3771793.968) < 1e-3 and abs(Y - 140253.342) < 1e-3 and abs(Z - 5124304.349) < 1e-3
    f = 1 / 297.
    lo, la, h = po.geocentric_to_geodetic(np.array([X + 84.87]), np.array([Y + 96.49]), np.array([Z + 116.95]), 6378388.0, f * (2 - f))
    assert abs(math.degrees(la[0]) - (53 + 48 / 60 + 36.565 / 3600)) < 2e-7 and abs(math.degrees(lo[0]) - (2 + 7 / 60 + 51.477 / 3600)) < 2e-7
    assert abs(h[0] - 28.02) < 0.01
    v = [0., 0., 4.5, 0., 0., 0.554 * po._SEC_TO_RAD, 1 + 0.219e-6]
    x, y, z = po.to_wgs84(2, v, 3657660.66, 255768.55, 5201382.11)
    assert abs(x - 3657660.78) < 0.01 and abs(y - 255778.43) < 0.01 and abs(z - 5201387.75) < 0.01
    # pj_transform: only when both sides name a datum; the way back undoes it (to the 1e-8 degree the approximate inverse leaves)
    lon2, lat2 = np.radians([10., 13.4]), np.radians([50., 52.5])
    a = po.transform("+proj=latlong +datum=WGS84", "+proj=latlong +datum=potsdam", lon2, lat2)
    assert 1e-5 < abs(a[0][0] - lon2[0]) < 1e-4 and 1e-5 < abs(a[1][0] - lat2[0]) < 1e-4     # a hundred metres or so
    b = po.transform("+proj=latlong +datum=potsdam", "+proj=latlong +datum=WGS84", *a)
    np.testing.assert_allclose(b[0], lon2, atol=1e-9); np.testing.assert_allclose(b[1], lat2, atol=1e-9)
    c = po.transform("+proj=latlong +datum=WGS84", "+proj=latlong +ellps=bessel", lon2, lat2)       # no datum on one side: untouched
    assert np.array_equal(c[0], lon2) and np.array_equal(c[1], lat2)
    d = po.transform("+proj=latlong +datum=WGS84", "+proj=latlong +datum=NAD83", lon2, lat2)        # zero shift between two ellipsoids
    assert np.abs(d[1] - lat2).max() < 1e-10 and np.abs(d[0] - lon2).max() < 1e-15


REF_CONVERSIONS = [   # test/testProjections.cc:84-208: a projection, its 10 x 10 mesh at 50 km, there and back within 1e-5
    "+proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +ellps=sphere +a=6371000 +e=0",
    "+proj=geos +lon_0=0 +h=3.57858e+07  +a=6.37817e+06  +b=6.35658e+06 +no_defs +x_0=-2.2098e+06 +y_0=-3.50297e+06",
    "+proj=omerc +lonc=5.34065 +lat_0=60.742 +alpha=19.0198 +no_rot   +a=6.37814e+06  +b=6.35675e+06 +no_defs +x_0=-3.86098e+06 +y_0=1.5594e+06",
]


def test_hotine_oblique_mercator_epsg_worked_example():
    """EPSG Guidance Note 7-2, Hotine oblique Mercator, Timbalai 1948 / RSO Borneo: 5 23 14.1129 N, 115 48 19.8196 E maps
    to E 679245.73, N 596562.78 -- variant A (natural origin: PROJ.4's +no_uoff, FE = FN = 0) and variant B (centre of the
    projection, Ec 590476.87, Nc 442857.65) alike."""
    common = "+proj=omerc +lat_0=4 +lonc=115 +alpha=53.31582047222222 +gamma=53.13010236111111 +k=0.99984 +a=6377298.556 +rf=300.8017"
    lon, lat = np.radians([115 + 48 / 60 + 19.8196 / 3600]), np.radians([5 + 23 / 60 + 14.1129 / 3600])
    for proj in (common + " +no_uoff", common + " +x_0=590476.87 +y_0=442857.65"):
        x, y = po.transform("+proj=latlong +a=6377298.556 +rf=300.8017", proj, lon, lat)
        assert abs(x[0] - 679245.73) < 0.01 and abs(y[0] - 596562.78) < 0.01, (proj, x, y)
        bl, bp = po.transform(proj, "+proj=latlong +a=6377298.556 +rf=300.8017", x, y)
        assert abs(bl[0] - lon[0]) < 1e-11 and abs(bp[0] - lat[0]) < 1e-11


@pytest.mark.parametrize("proj", REF_CONVERSIONS)
def test_reference_conversion_round_trips(proj):
    """test/testProjections.cc:84-208 (test_conversion, test_conversion_oblique_mercator, test_conversion_geostationary): x, y = 0 .. 450 km to
    longitude / latitude (inside +-180.001 / +-90.001 degrees) and back to within 1e-5 m.  """
    ll = "+proj=lonlat +ellps=sphere +a=6371000 +e=0"
    x, y = np.meshgrid(np.arange(10) * 50000., np.arange(10) * 50000., indexing="ij")
    lon, lat = po.transform(proj, ll, x.ravel(), y.ravel())
    assert np.all(np.abs(np.degrees(lat)) <= 90.001) and np.all(np.abs(np.degrees(lon)) <= 180.001)
    bx, by = po.transform(ll, proj, lon, lat)
    assert np.abs(bx - x.ravel()).max() < 1e-5 and np.abs(by - y.ravel()).max() < 1e-5


# ---------------------------------------------------------------- ellipsoidal projections (testInterpolator.cc:422)
# Worked numerical examples of Snyder, "Map Projections - A Working Manual" (USGS PP 1395), appendix A: the published
# known answers the PROJ.4 series are checked against in the absence of the library (values in metres, one decimal).
SNYDER = [
    ("+proj=merc +ellps=clrk66 +lon_0=-180", (-75., 35.), (11688673.7, 4139145.6)),
    ("+proj=tmerc +ellps=clrk66 +lon_0=-75 +lat_0=0 +k=0.9996", (-73.5, 40.5), (127106.5, 4484124.4)),
    ("+proj=lcc +ellps=clrk66 +lat_1=33 +lat_2=45 +lat_0=23 +lon_0=-96", (-75., 35.), (1894410.9, 1564649.5)),
    ("+proj=stere +ellps=intl +lat_0=-90 +lat_ts=-71 +lon_0=-100", (150., -75.), (-1540033.6, -560526.4)),
    ("+proj=stere +ellps=clrk66 +lat_0=40 +lon_0=-100 +k=0.9999", (-90., 30.), (971630.8, -1063049.3)),
    ("+proj=laea +ellps=clrk66 +lat_0=40 +lon_0=-100", (-110., 30.), (-965932.1, -1056814.9)),
    ("+proj=aea +ellps=clrk66 +lat_1=29.5 +lat_2=45.5 +lat_0=23 +lon_0=-96", (-75., 35.), (1885472.7, 1535925.0)),
]


def test_albers_on_the_unit_sphere():
    """Snyder's spherical example: standard parallels 29.5 and 45.5, origin 23 N 96 W, the point 35 N 75 W."""
    x, y = po.transform("+proj=latlong +R=1", "+proj=aea +R=1 +lat_1=29.5 +lat_2=45.5 +lat_0=23 +lon_0=-96", np.radians([-75.]), np.radians([35.]))
    assert abs(x[0] - 0.2952720) < 1e-7 and abs(y[0] - 0.2416774) < 1e-7


def test_lambert_azimuthal_equal_area_worked_examples():
    """Snyder's other two examples for the projection: the sphere of radius 3, and the polar aspect on the International ellipsoid."""
    x, y = po.transform("+proj=latlong +R=3", "+proj=laea +R=3 +lat_0=40 +lon_0=-100", np.radians([100.]), np.radians([-20.]))
    assert abs(x[0] + 4.2339303) < 1e-7 and abs(y[0] - 4.0257775) < 1e-7
    x, y = po.transform("+proj=latlong +ellps=intl", "+proj=laea +ellps=intl +lat_0=90 +lon_0=-100", np.radians([5.]), np.radians([80.]))
    assert abs(x[0] - 1077459.7) < 0.1 and abs(y[0] - 288704.5) < 0.1
    rng = np.random.default_rng(2)
    lon, lat = np.radians(rng.uniform(-40, 60, 500)), np.radians(rng.uniform(15, 85, 500))
    for proj in ("+proj=laea +lat_0=52 +lon_0=10 +x_0=4321000 +y_0=3210000 +ellps=GRS80", "+proj=laea +lat_0=90 +ellps=WGS84",
                 "+proj=laea +lat_0=0 +lon_0=20 +ellps=WGS84", "+proj=laea +lat_0=52 +lon_0=10 +R=6371000", "+proj=laea +lat_0=90 +R=6371000"):
        bl, bp = po.transform(proj, "+proj=latlong +ellps=WGS84", *po.transform("+proj=latlong +ellps=WGS84", proj, lon, lat))
        np.testing.assert_allclose(bl, lon, atol=1e-12); np.testing.assert_allclose(bp, lat, atol=5e-10)  # three-term authalic series


@pytest.mark.parametrize("proj,lonlat,xy", SNYDER)
def test_ellipsoidal_forms_reproduce_snyders_worked_examples(proj, lonlat, xy):
    lon, lat = np.radians([lonlat[0]]), np.radians([lonlat[1]])
    x, y = po.transform("+proj=latlong +ellps=clrk66", proj, lon, lat)
    assert abs(x[0] - xy[0]) < 0.1 and abs(y[0] - xy[1]) < 0.1, (x, y)  # printed to one decimal, from a rounded e^2
    bl, bp = po.transform(proj, "+proj=latlong +ellps=clrk66", x, y)
    np.testing.assert_allclose([bl[0], bp[0]], [lon[0], lat[0]], atol=2e-10)


def test_utm_is_etmerc_of_its_zone_and_central_meridian_arc():
    """UTM 33 (test/testInterpolator.cc:422): central meridian 15 E, k 0.9996, false easting 500 km; on the central
    meridian the northing is k times the meridian arc (WGS84 quarter meridian 10 001 965.729 m).  Since PROJ.4 4.9.3
    utm runs the extended series (etmerc); +proj=tmerc stays the truncated Gauss-Krueger series, and the two meet
    near the meridian."""
    geo = "+proj=latlong +datum=WGS84"
    utm = "+proj=utm +zone=33 +datum=WGS84 +no_defs"
    etm = "+proj=etmerc +lon_0=15 +k=0.9996 +x_0=500000 +ellps=WGS84"
    tm = "+proj=tmerc +lon_0=15 +k=0.9996 +x_0=500000 +ellps=WGS84"
    rng = np.random.default_rng(33)
    lon, lat = np.radians(rng.uniform(-25, 55, 2000)), np.radians(rng.uniform(-85, 85, 2000))
    a = po.transform(geo, utm, lon, lat)
    b = po.transform(geo, etm, lon, lat)
    np.testing.assert_allclose(a[0], b[0], atol=1e-7); np.testing.assert_allclose(a[1], b[1], atol=1e-7)  # lon_0 = 32.5 pi/30 - pi rounds differently
    bl, bp = po.transform(utm, geo, a[0], a[1])   # inverts itself 40 degrees from the meridian
    np.testing.assert_allclose(bl, lon, atol=1e-13); np.testing.assert_allclose(bp, lat, atol=1e-13)
    for width, tol in ((3, 2e-5), (6, 1e-3)):     # metres between the two series
        near = np.abs(lon - np.radians(15)) < np.radians(width)
        c = po.transform(geo, tm, lon[near], lat[near])
        assert np.hypot(a[0][near] - c[0], a[1][near] - c[1]).max() < tol
    x, y = po.transform(geo, utm, np.radians([15.]), np.radians([90. - 1e-9]))
    assert abs(x[0] - 500000.) < 1e-3 and abs(y[0] - 0.9996 * 10001965.729) < 2e-2
    south = po.transform(geo, "+proj=utm +zone=33 +south +ellps=WGS84", np.radians([15.]), np.radians([0.]))
    assert abs(south[1][0] - 1e7) < 1e-6
    x, y = po.transform("+proj=latlong +ellps=clrk66", "+proj=etmerc +ellps=clrk66 +lon_0=-75 +k=0.9996", np.radians([-73.5]), np.radians([40.5]))
    assert abs(x[0] - 127106.5) < 0.1 and abs(y[0] - 4484124.4) < 0.1   # Snyder's worked example again, through the other series
    # the truncated series inverts itself only near the meridian
    tl, tp = po.transform(tm, geo, *po.transform(geo, tm, lon, lat))
    narrow = np.abs(lon - np.radians(15)) < np.radians(3)
    np.testing.assert_allclose(tl[narrow], lon[narrow], atol=1e-11); np.testing.assert_allclose(tp[narrow], lat[narrow], atol=1e-11)


def test_etmerc_series_for_the_conformal_latitude_on_a_flatter_ellipsoid():
    """The n^6 series against the closed conformal latitude the oracle uses: they meet to n^7 (here rf = 150)."""
    geo, etm = "+proj=latlong +a=6378137 +rf=150", "+proj=etmerc +lon_0=0 +a=6378137 +rf=150"
    rng = np.random.default_rng(5)
    lon, lat = np.radians(rng.uniform(-30, 30, 500)), np.radians(rng.uniform(-85, 85, 500))
    x, y = po.transform(geo, etm, lon, lat)
    bl, bp = po.transform(etm, geo, x, y)
    np.testing.assert_allclose(bl, lon, atol=5e-13); np.testing.assert_allclose(bp, lat, atol=5e-13)


@pytest.mark.parametrize("proj", ["+proj=merc +lon_0=5 +lat_ts=30", "+proj=lcc +lat_0=48 +lon_0=8 +lat_1=30 +lat_2=60",
                                  "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60", "+proj=stere +lat_0=52 +lon_0=10", "+proj=tmerc +lon_0=12 +lat_0=20"])
def test_ellipsoidal_forms_tend_to_the_spherical_ones(proj):
    """e -> 0: the ellipsoidal series and the closed spherical forms are different code and must meet."""
    rng = np.random.default_rng(7)
    lon, lat = np.radians(rng.uniform(2, 22, 300) if "tmerc" in proj else rng.uniform(-15, 35, 300)), np.radians(rng.uniform(25, 80, 300))
    xs, ys = po.transform("+proj=latlong +R=6371000", proj + " +R=6371000", lon, lat)
    xe, ye = po.transform("+proj=latlong +R=6371000", proj + " +a=6371000 +es=1e-14", lon, lat)
    tol = 0.05 if "tmerc" in proj else 1e-3  # tmerc: a truncated series against the closed form, within 10 degrees of the meridian
    np.testing.assert_allclose(xe, xs, atol=tol); np.testing.assert_allclose(ye, ys, atol=tol)


def test_ellipsoid_parameters_follow_pj_ell_set():
    P = po._Proj
    assert P("+proj=merc +ellps=WGS84").es == pytest.approx(0.0066943799901413165, rel=1e-14)
    assert P("+proj=merc +datum=WGS84").a == 6378137.0
    assert P("+proj=merc +ellps=WGS84 +a=6000000").a == 6000000.0          # an explicit +a wins
    assert P("+proj=merc +ellps=WGS84 +a=6000000").es == P("+proj=merc +ellps=WGS84").es
    assert P("+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90").es == 0.0  # the reference's EMEP strings stay spherical
    assert P("+proj=merc +a=6378137 +b=6356752.314245").es == pytest.approx(0.0066943799901413165, rel=1e-9)
    assert P("+proj=merc +a=6378137 +f=0.0033528106647474805").es == pytest.approx(0.0066943799901413165, rel=1e-12)
    assert P("+proj=merc +R=6371000 +ellps=WGS84").es == 0.0                 # +R wins over everything
    for bad in ("+proj=utm +zone=33 +R=6371000", "+proj=utm +zone=61 +ellps=WGS84", "+proj=stere +lat_0=0 +ellps=WGS84",
                "+proj=merc +ellps=WGS84 +units=parsec", "+proj=merc +datum=NAD27"):
        with pytest.raises((NotImplementedError, ValueError)):
            P(bad)
    with pytest.raises((NotImplementedError, KeyError)):   # grid shifts are not restated
        po.transform("+proj=latlong +datum=WGS84", "+proj=latlong +ellps=clrk66 +nadgrids=conus", np.zeros(1), np.zeros(1))


def test_units_and_prime_meridians_follow_pj_fwd_pj_inv_and_pj_transform():
    """+units / +to_meter scale projected coordinates (pj_fwd.c: fr_meter * (a x + x_0); pj_inv.c: (x to_meter - x_0) / a), +pm shifts
    longitudes where the two sides meet (pj_transform.c).  Known answers: the US survey foot is 1200/3937 m and the Paris
    meridian lies 2.5969213 grads east of Greenwich (EPSG 8903); PROJ.4 is not in the reference tree: **parity unpinned** beyond these."""
    assert po.to_meter_of(po.parse("+proj=merc +units=us-ft")) == pytest.approx(1200.0 / 3937.0, rel=1e-15)
    assert po.to_meter_of(po.parse("+proj=merc +units=us-in")) == pytest.approx(100.0 / 3937.0, rel=1e-15)
    assert po.to_meter_of(po.parse("+proj=merc +units=km +to_meter=1/3.2808")) == pytest.approx(1 / 3.2808, rel=1e-15)  # +to_meter wins
    assert math.degrees(po.from_greenwich_of(po.parse("+proj=merc +pm=paris"))) == pytest.approx(2.5969213 * 0.9, abs=2e-8)
    assert math.degrees(po.from_greenwich_of(po.parse("+proj=merc +pm=ferro"))) == pytest.approx(-(17 + 40 / 60.0), abs=1e-12)
    stere = "+proj=stere +lat_0=90 +lon_0=10 +lat_ts=60 +R=6371000 +x_0=1000 +y_0=-500"
    geo = "+proj=latlong +R=6371000"
    rng = np.random.default_rng(3)
    lon, lat = np.radians(rng.uniform(-40, 60, 500)), np.radians(rng.uniform(40, 89, 500))
    x, y = po.transform(geo, stere, lon, lat)
    for unit, f in (("km", 1000.0), ("ft", 0.3048), ("us-ft", 1200.0 / 3937.0)):
        xk, yk = po.transform(geo, stere + " +units=" + unit, lon, lat)
        np.testing.assert_allclose(xk * f, x, rtol=1e-15, atol=1e-9)
        np.testing.assert_allclose(yk * f, y, rtol=1e-15, atol=1e-9)
        bl, bt = po.transform(stere + " +units=" + unit, geo, xk, yk)
        np.testing.assert_allclose(bl, lon, atol=1e-12); np.testing.assert_allclose(bt, lat, atol=1e-12)
    # a grid referred to Paris: its longitudes are smaller by the meridian's offset, the same point on the ground
    xp, yp = po.transform(geo + " +pm=paris", stere, lon - po.from_greenwich_of({"pm": "paris"}), lat)
    np.testing.assert_allclose(xp, x, atol=1e-6); np.testing.assert_allclose(yp, y, atol=1e-6)
    lo, la = po.transform(stere, geo + " +pm=-17.5", x, y)
    np.testing.assert_allclose(lo, lon + math.radians(17.5), atol=1e-12); np.testing.assert_allclose(la, lat, atol=1e-12)
    with pytest.raises(NotImplementedError):
        po.transform(geo, stere + " +units=parsec", lon, lat)
    with pytest.raises(NotImplementedError):
        po.transform(geo, stere + " +axis=wsu", lon, lat)
