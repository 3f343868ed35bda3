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
