"""The C++ host mirror of CDMInterpolator (fimex_amd/host/, over the C ABI) end to end on the GPU:
BASELINE configs[0] -- coordTest.nc reprojected to a 200x200 lat/lon grid -- and a forward-interpolation case.

Parity is checked in two layers: the plan positions / rotation matrix the C++ host computes against the oracle's
restatement of the same chain (projection code differs in libm rounding: tolerance), and the regridded data against
the oracle fed with exactly those positions (same arithmetic: bit-exact)."""
import os
import subprocess

import numpy as np
import pytest

import cases
import oracle
from oracle import proj_oracle as po

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "fimex_amd", "host_cli")
GEO = "+proj=latlong +R=6371000"
FILL_FLOAT = 9.9692099683868690e+36  # MIFI_FILL_FLOAT: CDM::getFillValue of a float variable without _FillValue (src/CDM.cc:490-518)


def _run(tmp, spec_lines):
    spec = tmp / "spec.txt"
    spec.write_text("\n".join(spec_lines) + "\n")
    out = tmp / "out"
    out.mkdir(exist_ok=True)
    r = subprocess.run([CLI, str(spec), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out, r.stdout


@pytest.fixture(scope="module")
def coordtest(golden_dir):
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        v = f.variables
        d = dict(x=v["x"].data.astype(np.float64), y=v["y"].data.astype(np.float64), proj=v["projection_1"].proj4.decode())
        for n in ("air_temperature", "x_wind_10m", "y_wind_10m", "precipitation_amount", "cloud_area_fraction_in_atmosphere_layer"):
            a = v[n].data.astype(np.float32)  # Data::asFloat(): raw stored values as float
            d[n] = a.reshape(a.shape[0], -1, 11, 11)
            d[n + "_fill"] = getattr(v[n], "_FillValue", None)
    return d


def _spec_common(tmp, ct, method, extra=()):
    ct["x"].tofile(tmp / "x.f64")
    ct["y"].tofile(tmp / "y.f64")
    lon = np.linspace(-15.2, -9.7, 200)
    lat = np.linspace(28.3, 33.0, 200)
    lon.tofile(tmp / "ox.f64")
    lat.tofile(tmp / "oy.f64")
    lines = ["proj " + ct["proj"], "xaxis %s" % (tmp / "x.f64"), "yaxis %s" % (tmp / "y.f64"), "method " + method,
             "outproj " + GEO, "outx %s degrees_east" % (tmp / "ox.f64"), "outy %s degrees_north" % (tmp / "oy.f64")]
    for n, vec in (("air_temperature", ""), ("x_wind_10m", " vector y_wind_10m x"), ("y_wind_10m", " vector x_wind_10m y"),
                   ("cloud_area_fraction_in_atmosphere_layer", "")):
        ct[n].tofile(tmp / (n + ".f32"))
        fill = ct[n + "_fill"]
        lines.append("var %s %d %s %s%s" % (n, ct[n].shape[1], tmp / (n + ".f32"), "nan" if fill is None else repr(float(fill)), vec))
    return lines + list(extra), lon, lat


def _oracle_positions(ct, lon, lat):
    px, py = po.project_axes(GEO, ct["proj"], np.radians(lon), np.radians(lat))
    return oracle.points2position(px, ct["x"]), oracle.points2position(py, ct["y"])


@pytest.mark.parametrize("method,code", [("bilinear", oracle.BILINEAR), ("nearestneighbor", oracle.NEAREST), ("bicubic", oracle.BICUBIC)])
def test_coordtest_to_latlon_200x200(tmp_path, coordtest, method, code):
    ct = coordtest
    gets = ["get air_temperature 0", "get air_temperature 3", "get cloud_area_fraction_in_atmosphere_layer 1",
            "get x_wind_10m 2", "get y_wind_10m 2"]
    lines, lon, lat = _spec_common(tmp_path, ct, method, gets)
    out, stdout = _run(tmp_path, lines)
    assert "outX 200 outY 200" in stdout and "reduced" in stdout
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    wx, wy = _oracle_positions(ct, lon, lat)
    np.testing.assert_allclose(px, wx, atol=1e-7)
    np.testing.assert_allclose(py, wy, atol=1e-7)
    assert (px < -0.5).any() and (px > 10.5).any()  # the target overshoots the 11x11 source

    def fill_of(name):
        return FILL_FLOAT if ct[name + "_fill"] is None else float(ct[name + "_fill"])

    def regrid(name, step):
        return oracle.interpolate_values(code, px, py, oracle.bad2nan(ct[name][step], fill_of(name)), 11, 11, 200, 200)

    def back(a, name):  # interpolationArray2Data for a float variable
        return oracle.interpolation_array2data(a, oracle.CDM_FLOAT, fill_of(name))

    for name, step in (("air_temperature", 0), ("air_temperature", 3), ("cloud_area_fraction_in_atmosphere_layer", 1)):
        got = np.fromfile(out / ("%s_%d.f32" % (name, step)), dtype=np.float32).reshape(-1, 200, 200)
        want = back(regrid(name, step), name)
        assert got.shape == want.shape
        assert cases.same(got, want), "%s: %s" % (name, cases.describe_mismatch(got, want))
    # x/y wind pair: both components regridded, then rotated with the matrix the host built
    m = np.fromfile(out / "matrix.f64")
    assert m.size == 4 * 200 * 200
    u, v = regrid("x_wind_10m", 2), regrid("y_wind_10m", 2)
    ru, rv = oracle.vector_reproject_values(m, u, v, 200, 200)
    gu = np.fromfile(out / "x_wind_10m_2.f32", dtype=np.float32).reshape(-1, 200, 200)
    gv = np.fromfile(out / "y_wind_10m_2.f32", dtype=np.float32).reshape(-1, 200, 200)
    assert cases.same(gu, back(ru, "x_wind_10m")), cases.describe_mismatch(gu, back(ru, "x_wind_10m"))
    assert cases.same(gv, back(rv, "y_wind_10m")), cases.describe_mismatch(gv, back(rv, "y_wind_10m"))
    # the rotation matrix itself against the oracle's restatement of mifi_get_vector_reproject_matrix
    from test_oracle_kats import _rotation_matrix
    wm = _rotation_matrix(ct["proj"], GEO, lon, lat, oracle.LONGITUDE, oracle.LATITUDE)
    np.testing.assert_allclose(m.reshape(-1, 4)[:, :2], wm.reshape(-1, 4)[:, :2], atol=1e-6)
    # wind speed survives the rotation where both components are defined (test/testInterpolation.cc:575-578)
    ok = ~np.isnan(u) & ~np.isnan(v)
    np.testing.assert_allclose(np.hypot(gu[ok], gv[ok]), np.hypot(u[ok], v[ok]), rtol=1e-5)


def test_coordtest_with_pre_and_postprocess(tmp_path, coordtest):
    """--interpolate.preprocess creepfill2d / .postprocess fill2d as in test/testInterpolation2DataFillValue.sh and
    testInterpolatorFill.sh: fills run per z slice before and after the regrid, all on the GPU."""
    ct = dict(coordtest)
    t = ct["air_temperature"].copy()
    rng = np.random.default_rng(3)
    t.reshape(-1)[rng.choice(t.size, 60, replace=False)] = ct["air_temperature_fill"]  # punch holes
    ct["air_temperature"] = t
    extra = ["pre creepfill2d 5 2", "post fill2d 4.0 1.6 100", "get air_temperature 1"]
    lines, lon, lat = _spec_common(tmp_path, ct, "bilinear", extra)
    out, _ = _run(tmp_path, lines)
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    f = oracle.bad2nan(t[1], ct["air_temperature_fill"])
    f = np.stack([oracle.creepfill2d(s, 5, 2)[0] for s in f])
    r = oracle.interpolate_values(oracle.BILINEAR, px, py, f, 11, 11, 200, 200)
    r = np.stack([oracle.fill2d(s, 4.0, 1.6, 100)[0] for s in r])
    want = oracle.interpolation_array2data(r, oracle.CDM_FLOAT, float(ct["air_temperature_fill"]))
    got = np.fromfile(out / "air_temperature_1.f32", dtype=np.float32).reshape(-1, 200, 200)
    assert not np.isnan(got).any() and not (got == ct["air_temperature_fill"]).any()  # fill2d closed the outside too
    assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("method,code", [("forward_mean", oracle.FWD_MEAN), ("forward_max", oracle.FWD_MAX), ("forward_median", oracle.FWD_MEDIAN)])
def test_forward_interpolation_latlon_to_lambert(tmp_path, method, code):
    """changeProjectionByForwardInterpolation: a 0.25-degree lat/lon source scattered onto a 40x30 Lambert grid."""
    lon = np.arange(0, 30, 0.25)
    lat = np.arange(55, 72, 0.25)
    lcc = "+proj=lcc +lat_0=63 +lon_0=15 +lat_1=63 +lat_2=63 +no_defs +R=6.371e+06"
    ox = (np.arange(40) - 19.5) * 25000.0
    oy = (np.arange(30) - 14.5) * 25000.0
    f = cases.field(2 * 3, lat.size, lon.size, seed=5, nan_frac=0.05, extremes=False).reshape(2, 3, lat.size, lon.size)
    for name, a in (("x", lon), ("y", lat), ("ox", ox), ("oy", oy)):
        a.astype(np.float64).tofile(tmp_path / (name + ".f64"))
    f.tofile(tmp_path / "f.f32")
    lines = ["proj " + GEO, "xaxis %s" % (tmp_path / "x.f64"), "yaxis %s" % (tmp_path / "y.f64"), "method " + method,
             "outproj " + lcc, "outx %s m" % (tmp_path / "ox.f64"), "outy %s m" % (tmp_path / "oy.f64"),
             "var field 3 %s nan" % (tmp_path / "f.f32"), "get field 1"]
    out, _ = _run(tmp_path, lines)
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    assert px.size == lon.size * lat.size
    wx, wy = po.project_axes(GEO, lcc, np.radians(lon), np.radians(lat))
    np.testing.assert_allclose(px, oracle.points2position(wx, ox), atol=1e-6)
    np.testing.assert_allclose(py, oracle.points2position(wy, oy), atol=1e-6)
    want = oracle.forward_interpolate_values(code, px, py, f[1], lon.size, lat.size, 40, 30)
    assert np.isnan(want).any() and np.isfinite(want).mean() > 0.5
    want = oracle.interpolation_array2data(want, oracle.CDM_FLOAT, FILL_FLOAT)  # no _FillValue: the type's default fill value
    got = np.fromfile(out / "field_1.f32", dtype=np.float32).reshape(3, 30, 40)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


def test_coordtest_packed_shorts_stay_shorts(tmp_path, golden_dir):
    """The stored type end to end: coordTest.nc's short variables go to the GPU as shorts, become float/NaN there
    (data2InterpolationArray), are regridded and come back as shorts (interpolationArray2Data: NaN -> _FillValue or the
    type's default fill value, values rounded half away from zero)."""
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        v = f.variables
        x, y = v["x"].data.astype(np.float64), v["y"].data.astype(np.float64)
        proj = v["projection_1"].proj4.decode()
        raw = {n: np.ascontiguousarray(v[n].data.astype(np.int16)).reshape(4, -1, 11, 11)
               for n in ("air_temperature", "sea_surface_temperature", "altitude")}
        fills = {n: (float(v[n]._FillValue) if hasattr(v[n], "_FillValue") else None) for n in raw}
    assert fills["sea_surface_temperature"] is None and fills["air_temperature"] == -32767.0
    x.tofile(tmp_path / "x.f64"); y.tofile(tmp_path / "y.f64")
    lon = np.linspace(-15.2, -9.7, 200); lat = np.linspace(28.3, 33.0, 200)
    lon.tofile(tmp_path / "ox.f64"); lat.tofile(tmp_path / "oy.f64")
    lines = ["proj " + proj, "xaxis %s" % (tmp_path / "x.f64"), "yaxis %s" % (tmp_path / "y.f64"), "method bilinear", "outproj " + GEO,
             "outx %s degrees_east" % (tmp_path / "ox.f64"), "outy %s degrees_north" % (tmp_path / "oy.f64"), "post creepfill2d 3 1"]
    for n, a in raw.items():
        a.tofile(tmp_path / (n + ".i16"))
        lines.append("var %s %d %s %s type short" % (n, a.shape[1], tmp_path / (n + ".i16"), "nan" if fills[n] is None else repr(fills[n])))
        lines.append("get %s 2" % n)
    out, _ = _run(tmp_path, lines)
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    for n, a in raw.items():
        bad = -32767.0 if fills[n] is None else fills[n]  # MIFI_FILL_SHORT
        f = oracle.data2interpolation_array(a[2], bad)
        r = oracle.interpolate_values(oracle.BILINEAR, px, py, f, 11, 11, 200, 200)
        r = np.stack([oracle.creepfill2d(s, 3, 1)[0] for s in r])
        want = oracle.interpolation_array2data(r, oracle.CDM_SHORT, bad)
        got = np.fromfile(out / ("%s_2.raw" % n), dtype=np.int16).reshape(want.shape)
        assert np.array_equal(got, want), n
        if n == "air_temperature":
            assert (got != np.int16(bad)).mean() > 0.1 and (a[2] != np.int16(bad)).all()


@pytest.mark.parametrize("method,code,none", [("coord_nearestneighbor", oracle.COORD_NN, -1), ("coord_kdtree", oracle.COORD_NN_KD, -1000)])
def test_coordtest_by_its_stored_coordinates(tmp_path, golden_dir, method, code, none):
    """changeProjectionByCoordinates (src/CDMInterpolator.cc:1335-1420): the source grid is described by coordTest.nc's 2-D
    longitude / latitude variables, the target cells take the value of the closest source cell.  The reference's tests of
    these methods need files that are not shipped (test/testInterpolator.cc:77-102); pinned here by the restatement."""
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        v = f.variables
        x, y = v["x"].data.astype(np.float64), v["y"].data.astype(np.float64)
        lon2d, lat2d = v["longitude"].data.astype(np.float64), v["latitude"].data.astype(np.float64)
        proj = v["projection_1"].proj4.decode()
        t = np.ascontiguousarray(v["air_temperature"].data.astype(np.int16)).reshape(4, 1, 11, 11)
    for name, a in (("x", x), ("y", y), ("lon2d", lon2d), ("lat2d", lat2d)):
        a.tofile(tmp_path / (name + ".f64"))
    olon = np.linspace(-14.5, -10.0, 120); olat = np.linspace(28.8, 32.6, 100)
    olon.tofile(tmp_path / "ox.f64"); olat.tofile(tmp_path / "oy.f64")
    t.tofile(tmp_path / "t.i16")
    lines = ["proj " + proj, "xaxis %s" % (tmp_path / "x.f64"), "yaxis %s" % (tmp_path / "y.f64"), "lon2d %s" % (tmp_path / "lon2d.f64"),
             "lat2d %s" % (tmp_path / "lat2d.f64"), "method " + method, "outproj " + GEO, "outx %s degrees_east" % (tmp_path / "ox.f64"),
             "outy %s degrees_north" % (tmp_path / "oy.f64"), "var air_temperature 1 %s -32767 type short" % (tmp_path / "t.i16"),
             "get air_temperature 3"]
    if method == "coord_kdtree":
        lines.insert(5, "maxdist 40000")
    out, stdout = _run(tmp_path, lines)
    assert "outX 120 outY 100" in stdout and "reduced" not in stdout
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    qlon, qlat = np.meshgrid(np.radians(olon), np.radians(olat))
    if code == oracle.COORD_NN:
        wx, wy = oracle.fast_translate_points(qlon.ravel(), qlat.ravel(), np.radians(lon2d), np.radians(lat2d))
    else:
        wx, wy = oracle.flann_translate_points(40000.0, qlon.ravel(), qlat.ravel(), np.radians(lon2d), np.radians(lat2d))
    assert ((px == wx) & (py == wy)).mean() > 0.995  # equidistant cells may be taken either way
    assert (px >= 0).mean() > 0.3 and (px == none).any()
    f = oracle.data2interpolation_array(t[3], -32767.0)
    want = oracle.interpolation_array2data(oracle.interpolate_values(code, px, py, f, 11, 11, 120, 100), oracle.CDM_SHORT, -32767.0)
    got = np.fromfile(out / "air_temperature_3.raw", dtype=np.int16).reshape(want.shape)
    assert np.array_equal(got, want)
    assert set(np.unique(got)) <= set(np.unique(t[3])) | {-32767}  # nearest neighbour: only values of the source


@pytest.mark.parametrize("method,code", [("bilinear", oracle.BILINEAR), ("nearestneighbor", oracle.NEAREST), ("bicubic", oracle.BICUBIC)])
def test_coordtest_to_a_list_of_points(tmp_path, coordtest, method, code):
    """changeProjection(method, lonVals, latVals) (src/CDMInterpolator.cc:460-510 -> :1706-1824): the target is a list of
    stations; the values travel as float, the plan is outX = n, outY = 1, x/y winds are turned to east / north with
    mifi_get_vector_reproject_matrix_points."""
    ct = coordtest
    rng = np.random.default_rng(12)
    lon, lat = rng.uniform(-15.0, -9.9, 57), rng.uniform(28.5, 32.8, 57)
    lon[:3], lat[:3] = (-40., 0., 170.), (10., 89., -30.)   # far outside the 11x11 grid
    lon.tofile(tmp_path / "plon.f64"); lat.tofile(tmp_path / "plat.f64")
    lines, _, _ = _spec_common(tmp_path, ct, method, ["points %s %s" % (tmp_path / "plon.f64", tmp_path / "plat.f64"),
                                                     "get air_temperature 1", "get x_wind_10m 0", "get y_wind_10m 0"])
    out, stdout = _run(tmp_path, lines)
    assert "outX 57 outY 1" in stdout
    lonf, latf = lon.astype(np.float32).astype(np.float64), lat.astype(np.float32).astype(np.float64)
    geo = "+proj=latlong +datum=WGS84 +towgs84=0,0,0 +no_defs"
    wx, wy = po.transform(geo, ct["proj"], np.radians(lonf), np.radians(latf))
    wx, wy = oracle.points2position(wx, ct["x"]), oracle.points2position(wy, ct["y"])
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    np.testing.assert_allclose(px, wx, atol=1e-7); np.testing.assert_allclose(py, wy, atol=1e-7)

    def regrid(name, step):
        fill = FILL_FLOAT if ct[name + "_fill"] is None else float(ct[name + "_fill"])
        return oracle.interpolate_values(code, px, py, oracle.bad2nan(ct[name][step], fill), 11, 11, 57, 1), fill

    got = np.fromfile(out / "air_temperature_1.f32", dtype=np.float32).reshape(-1, 1, 57)
    want, fill = regrid("air_temperature", 1)
    want = oracle.interpolation_array2data(want, oracle.CDM_FLOAT, fill)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert (got[:, 0, :3] == np.float32(fill)).all() and (got[:, 0, 3:] != np.float32(fill)).any()
    # winds: the matrix at the points against the restatement, the data through the matrix the host built
    m = np.fromfile(out / "matrix.f64")
    ix, iy = po.transform(geo, ct["proj"], np.radians(lonf), np.radians(latf))
    xdx = po.transform(ct["proj"], geo, ix + 100, iy)
    ydy = po.transform(ct["proj"], geo, ix, iy + 100)
    wm = oracle.vector_matrix_from_deltas(np.radians(lonf), np.radians(latf), xdx, ydy, 100., 100., True).reshape(-1, 4)
    np.testing.assert_allclose(m.reshape(-1, 4)[3:, :3], wm[3:, :3], atol=1e-6)
    (u, fu), (v, fv) = regrid("x_wind_10m", 0), regrid("y_wind_10m", 0)
    ru, rv = oracle.vector_reproject_values(m, u, v, 57, 1)
    gu = np.fromfile(out / "x_wind_10m_0.f32", dtype=np.float32).reshape(-1, 1, 57)
    gv = np.fromfile(out / "y_wind_10m_0.f32", dtype=np.float32).reshape(-1, 1, 57)
    assert cases.same(gu, oracle.interpolation_array2data(ru, oracle.CDM_FLOAT, fu))
    assert cases.same(gv, oracle.interpolation_array2data(rv, oracle.CDM_FLOAT, fv))


def test_coordtest_to_a_template_grid(tmp_path, coordtest):
    """changeProjection(method, template) (src/CDMInterpolator.cc:651-712 -> :1706-1824): the target grid is given by the 2-D
    longitude / latitude of a template file -- here a curvilinear 40 x 25 mesh."""
    ct = coordtest
    jj, ii = np.meshgrid(np.arange(25), np.arange(40), indexing="ij")
    lon2 = (-14.6 + 0.11 * ii + 0.02 * jj + 0.3 * np.sin(jj / 6.0)).astype(np.float32)
    lat2 = (28.9 + 0.14 * jj - 0.015 * ii + 0.2 * np.cos(ii / 9.0)).astype(np.float32)
    lon2.tofile(tmp_path / "tlon.f32"); lat2.tofile(tmp_path / "tlat.f32")
    lines, _, _ = _spec_common(tmp_path, ct, "bilinear", ["template %s %s 40 25" % (tmp_path / "tlon.f32", tmp_path / "tlat.f32"),
                                                         "get air_temperature 2"])
    out, stdout = _run(tmp_path, lines)
    assert "outX 40 outY 25" in stdout
    geo = "+proj=latlong +datum=WGS84 +towgs84=0,0,0 +no_defs"
    wx, wy = po.transform(geo, ct["proj"], np.radians(lon2.astype(np.float64).ravel()), np.radians(lat2.astype(np.float64).ravel()))
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    np.testing.assert_allclose(px, oracle.points2position(wx, ct["x"]), atol=1e-7)
    np.testing.assert_allclose(py, oracle.points2position(wy, ct["y"]), atol=1e-7)
    fill = FILL_FLOAT if ct["air_temperature_fill"] is None else float(ct["air_temperature_fill"])
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, oracle.bad2nan(ct["air_temperature"][2], fill), 11, 11, 40, 25)
    want = oracle.interpolation_array2data(want, oracle.CDM_FLOAT, fill)
    got = np.fromfile(out / "air_temperature_2.f32", dtype=np.float32).reshape(-1, 25, 40)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.fromfile(out / "matrix.f64").size == 4 * 40 * 25


def _earth(proj):
    return " ".join(t for t in proj.split() if t.split("=")[0] in ("+a", "+b", "+e", "+es", "+f", "+rf", "+R", "+ellps", "+datum", "+towgs84"))


def _cross_section_points(ct, sections):
    """numpy restatement of src/CDMInterpolator.cc:512-585 with the oracle's projection code."""
    geo = "+proj=latlong " + _earth(ct["proj"])
    dx, dy = ct["x"][1] - ct["x"][0], ct["y"][1] - ct["y"][0]
    lons, lats, starts = [], [], []
    for pts in sections:
        starts.append(len(lons))
        if len(pts) == 1:
            lons.append(pts[0][0]); lats.append(pts[0][1])
            continue
        for i in range(1, len(pts)):
            x, y = po.transform(geo, ct["proj"], np.radians([pts[i - 1][0], pts[i][0]]), np.radians([pts[i - 1][1], pts[i][1]]))
            xd, yd = x[1] - x[0], y[1] - y[0]
            num = int(np.floor(max(abs(xd / dx), abs(yd / dy))))
            xs = ([x[0]] if i == 1 else []) + [x[0] + j * xd / num for j in range(1, num)] + [x[1]]
            ys = ([y[0]] if i == 1 else []) + [y[0] + j * yd / num for j in range(1, num)] + [y[1]]
            lo, la = po.transform(ct["proj"], geo, np.array(xs), np.array(ys))
            lons += list(np.degrees(lo)); lats += list(np.degrees(la))
    return np.array(lons), np.array(lats), starts


def test_coordtest_cross_sections(tmp_path, coordtest):
    """changeProjectionToCrossSections (src/CDMInterpolator.cc:512-633; test/testInterpolator.cc:474-497 checks two named
    sections and more than five points): waypoints joined by straight lines in the grid's projection, one point per
    grid step, then the point-list plan."""
    ct = coordtest
    sections = [[(-14.2, 29.1), (-12.0, 30.6), (-10.4, 32.3)], [(-13.0, 31.9)], [(-11.1, 29.0), (-14.0, 32.0)]]
    extra = ["crosssection %s %s" % (n, " ".join("%r %r" % p for p in pts)) for n, pts in zip(("ABC", "single", "back"), sections)]
    lines, _, _ = _spec_common(tmp_path, ct, "bilinear", extra + ["get air_temperature 0"])
    out, stdout = _run(tmp_path, lines)
    assert "vcross ABC single back" in stdout
    wlon, wlat, starts = _cross_section_points(ct, sections)
    glon, glat = np.fromfile(out / "target_lon.f64"), np.fromfile(out / "target_lat.f64")
    assert glon.size == wlon.size > 5
    np.testing.assert_allclose(glon, wlon, atol=1e-9); np.testing.assert_allclose(glat, wlat, atol=1e-9)
    bnds = np.fromfile(out / "vcross_bnds.i32", dtype=np.int32).reshape(-1, 2)
    assert bnds[:, 0].tolist() == starts and bnds[:, 1].tolist() == [s - 1 for s in starts[1:]] + [wlon.size - 1]
    assert bnds[1, 0] == bnds[1, 1]   # the single-point section
    assert "outX %d outY 1" % wlon.size in stdout
    # waypoints themselves are part of the list
    assert abs(glon[0] + 14.2) < 1e-9 and abs(glat[bnds[0, 1]] - 32.3) < 1e-9 and abs(glon[bnds[1, 0]] + 13.0) < 1e-12
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    fill = FILL_FLOAT if ct["air_temperature_fill"] is None else float(ct["air_temperature_fill"])
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, oracle.bad2nan(ct["air_temperature"][0], fill), 11, 11, wlon.size, 1)
    want = oracle.interpolation_array2data(want, oracle.CDM_FLOAT, fill)
    got = np.fromfile(out / "air_temperature_0.f32", dtype=np.float32).reshape(-1, 1, wlon.size)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


def test_get_data_slice_with_a_slicebuilder(tmp_path, coordtest):
    """getDataSlice(varName, SliceBuilder) (src/CDMInterpolator.cc:162-233): a range of levels is read, whole horizontal
    slices are regridded (and rotated for the wind pair), the result is cut to the requested x / y window."""
    ct = coordtest
    nlev = ct["cloud_area_fraction_in_atmosphere_layer"].shape[1]
    assert nlev >= 3
    lines, lon, lat = _spec_common(tmp_path, ct, "bilinear", [
        "get cloud_area_fraction_in_atmosphere_layer 1", "get x_wind_10m 2",
        "slice cloud_area_fraction_in_atmosphere_layer 1 1 2 17 50 120 33",
        "slice x_wind_10m 2 0 1 0 200 199 1"])
    out, _ = _run(tmp_path, lines)
    full = np.fromfile(out / "cloud_area_fraction_in_atmosphere_layer_1.f32", dtype=np.float32).reshape(nlev, 200, 200)
    part = np.fromfile(out / "cloud_area_fraction_in_atmosphere_layer_1_slice.f32", dtype=np.float32).reshape(2, 33, 50)
    assert cases.same(part, full[1:3, 120:153, 17:67])
    wind = np.fromfile(out / "x_wind_10m_2.f32", dtype=np.float32).reshape(-1, 200, 200)
    row = np.fromfile(out / "x_wind_10m_2_slice.f32", dtype=np.float32).reshape(1, 1, 200)
    assert cases.same(row, wind[:1, 199:, :])


# ---------------------------------------------------------------- the reference's own file-based tests on the ERA-Interim fixture
@pytest.fixture(scope="module")
def erai(golden_dir):
    """test/erai.sfc.40N.0.75d.200301011200.nc (NetCDF-3 classic): skin temperature as double on a 6 x 11 lat/lon grid with
    descending latitudes, 8 time steps; test/template_noaa17.nc: 2-D longitude / latitude of a 29 x 31 satellite swath."""
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "erai.sfc.40N.0.75d.200301011200.nc"), "r", mmap=False) as f:
        v = f.variables
        d = dict(lon=v["longitude"].data.astype(np.float64), lat=v["latitude"].data.astype(np.float64),
                 proj=v["projection_regular_ll"].proj4.decode(), skt=v["ga_skt"].data.astype(np.float64).reshape(8, 1, 11, 6))
    with netcdf_file(os.path.join(golden_dir, "template_noaa17.nc"), "r", mmap=False) as f:
        d["tlon"] = f.variables["longitude"].data.astype(np.float32)
        d["tlat"] = f.variables["latitude"].data.astype(np.float32)
    return d


def _erai_spec(tmp, e, method, extra):
    e["lon"].tofile(tmp / "x.f64"); e["lat"].tofile(tmp / "y.f64"); e["skt"].tofile(tmp / "skt.f64")
    return ["proj " + e["proj"], "xaxis %s" % (tmp / "x.f64"), "yaxis %s" % (tmp / "y.f64"), "method " + method,
            "var ga_skt 1 %s nan type double" % (tmp / "skt.f64")] + list(extra)


def _erai_oracle(e, code, px, py, step, n_out_x, n_out_y):
    src = oracle.data2interpolation_array(e["skt"][step], 9.9692099683868690e+36)   # CDM::getFillValue default of a double variable
    return oracle.interpolate_values(code, px, py, src, 6, 11, n_out_x, n_out_y).astype(np.float64)


def test_erai_to_the_noaa17_template(tmp_path, erai):
    """test/testInterpolator.cc:220-239 (test_interpolator_template): bicubic onto the template's 29 x 31 swath; the first
    seven values of ga_skt are defined and lie between 270 and 280 K."""
    e = erai
    e["tlon"].tofile(tmp_path / "tlon.f32"); e["tlat"].tofile(tmp_path / "tlat.f32")
    gets = ["get ga_skt %d" % t for t in range(8)]
    out, stdout = _run(tmp_path, _erai_spec(tmp_path, e, "bicubic", ["template %s %s 29 31" % (tmp_path / "tlon.f32", tmp_path / "tlat.f32")] + gets))
    assert "outX 29 outY 31" in stdout
    first = np.fromfile(out / "ga_skt_0.raw", dtype=np.float64)
    assert first.size == 29 * 31                                               # :228-231
    assert np.all(np.isfinite(first[:7]) & (first[:7] > 270) & (first[:7] < 280)), first[:8]   # :235-237
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    for t in range(8):
        got = np.fromfile(out / ("ga_skt_%d.raw" % t), dtype=np.float64)
        want = _erai_oracle(e, oracle.BICUBIC, px, py, t, 29, 31).ravel()
        want = np.where(np.isnan(want), 9.9692099683868690e+36, want)          # interpolationArray2Data: the double variable's fill value
        assert np.array_equal(got, want), (t, np.abs(got - want).max())


def test_erai_to_ten_points(tmp_path, erai):
    """test/testInterpolator.cc:241-264 (test_interpolator_latlon): bilinear to ten stations; all values defined, the
    first between 270 and 280 K, all between 266 and 281.1 K, over the eight time steps."""
    e = erai
    lat = np.array([59.109, 59.052, 58.994, 58.934, 58.874, 58.812, 58.749, 58.685, 58.62, 64.])
    lon = np.array([4.965, 5.13, 5.296, 5.465, 5.637, 5.81, 5.986, 6.164001, 6.344, 3.])
    lon.tofile(tmp_path / "plon.f64"); lat.tofile(tmp_path / "plat.f64")
    gets = ["get ga_skt %d" % t for t in range(8)]
    out, stdout = _run(tmp_path, _erai_spec(tmp_path, e, "bilinear", ["points %s %s" % (tmp_path / "plon.f64", tmp_path / "plat.f64")] + gets))
    assert "outX 10 outY 1" in stdout
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    allv = np.stack([np.fromfile(out / ("ga_skt_%d.raw" % t), dtype=np.float64) for t in range(8)])
    assert allv.shape == (8, 10)
    assert 270 < allv[0, 0] < 280                                             # :258
    assert np.all(np.isfinite(allv) & (allv > 266) & (allv < 281.1)), allv     # :259-261
    for t in range(8):
        assert np.array_equal(allv[t], _erai_oracle(e, oracle.BILINEAR, px, py, t, 10, 1).ravel())


def test_erai_cross_sections(tmp_path, erai):
    """test/testInterpolator.cc:474-497 (test_interpolator_vcross): Oslo - Trondheim - Tromso and Bergen - Oslo on the
    0.75 degree grid: two named sections, more than five points."""
    out, stdout = _run(tmp_path, _erai_spec(tmp_path, erai, "bilinear", [
        "crosssection OsloTrondheimTromso 10.74 59.9 10.3951 63.4305 18.9551 69.6489", "crosssection BergenOslo 5.3290 60.3983 10.74 59.9",
        "get ga_skt 0"]))
    assert "vcross OsloTrondheimTromso BergenOslo" in stdout
    bnds = np.fromfile(out / "vcross_bnds.i32", dtype=np.int32).reshape(-1, 2)
    lon = np.fromfile(out / "target_lon.f64")
    assert bnds.shape == (2, 2) and lon.size > 5 and bnds[0, 0] == 0 and bnds[1, 1] == lon.size - 1 and bnds[1, 0] == bnds[0, 1] + 1
    assert abs(lon[0] - 10.74) < 1e-9 and abs(lon[-1] - 10.74) < 1e-9 and abs(lon[bnds[1, 0]] - 5.3290) < 1e-9
    got = np.fromfile(out / "ga_skt_0.raw", dtype=np.float64)
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    want = _erai_oracle(erai, oracle.BILINEAR, px, py, 0, lon.size, 1).ravel()
    assert np.array_equal(got, np.where(np.isnan(want), 9.9692099683868690e+36, want))


def test_two_coordinate_systems_fixture(tmp_path, golden_dir):
    """test/testInterpolator.cc:123-181 (test_interpolator2coords): temp2 of test/twoCoordsTest.nc lives on the file's second
    grid (x_c, y_c, longitude2 / latitude2, packed shorts); coord_kdtree and nearestneighbor onto a 12 x 12 polar-stereographic
    grid at 50 km leave more than 100 cells above 29000 in the first time step."""
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "twoCoordsTest.nc"), "r", mmap=False) as f:
        v = f.variables
        xc, yc = v["x_c"].data.astype(np.float64), v["y_c"].data.astype(np.float64)
        lon2, lat2 = v["longitude2"].data.astype(np.float64), v["latitude2"].data.astype(np.float64)
        temp2 = v["temp2"].data.astype(np.int16)
        proj = v["projection_1"].proj4.decode()
    xc.tofile(tmp_path / "x.f64"); yc.tofile(tmp_path / "y.f64"); lon2.tofile(tmp_path / "lon.f64"); lat2.tofile(tmp_path / "lat.f64")
    temp2.tofile(tmp_path / "temp2.i16")
    ox = -1705516 + 50000. * np.arange(12)
    oy = -6872225 + 50000. * np.arange(12)
    ox.tofile(tmp_path / "ox.f64"); oy.tofile(tmp_path / "oy.f64")
    target = "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +ellps=sphere +a=6371000 +e=0"
    results = {}
    for method in ("coord_kdtree", "nearestneighbor"):
        lines = ["proj " + proj, "xaxis %s" % (tmp_path / "x.f64"), "yaxis %s" % (tmp_path / "y.f64"),
                 "lon2d %s" % (tmp_path / "lon.f64"), "lat2d %s" % (tmp_path / "lat.f64"), "method " + method, "outproj " + target,
                 "outx %s m" % (tmp_path / "ox.f64"), "outy %s m" % (tmp_path / "oy.f64"),
                 "var temp2 1 %s nan type short" % (tmp_path / "temp2.i16"), "get temp2 0"]
        out, _ = _run(tmp_path, lines)
        got = np.fromfile(out / "temp2_0.raw", dtype=np.int16)
        assert got.size == 144
        assert (got > 29000).sum() > 100, (method, (got > 29000).sum())       # :151, :176
        results[method] = got
        # the regrid itself: nearest cell of the plan the host built, on the raw shorts
        px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
        src = oracle.data2interpolation_array(temp2[0], -32767.0)               # CDM::getFillValue default of a short
        want = oracle.interpolation_array2data(oracle.interpolate_values(oracle.NEAREST, px, py, src, 11, 10, 12, 12), oracle.CDM_SHORT, -32767.0)
        assert np.array_equal(got, want.ravel())
    inside = (results["coord_kdtree"] != -32767) & (results["nearestneighbor"] != -32767)
    assert inside.sum() > 100 and np.array_equal(results["coord_kdtree"][inside], results["nearestneighbor"][inside])


@pytest.mark.parametrize("method,code", [("bilinear", oracle.BILINEAR), ("nearestneighbor", oracle.NEAREST), ("bicubic", oracle.BICUBIC)])
def test_reduced_domain_crop_through_the_staged_kernels(tmp_path, method, code):
    """CachedInterpolation::createReducedDomain + the cropped read of getInputDataSlice
    (src/CachedInterpolation.cc:44-90,159-200) with a target strictly inside a 1201 x 1003 source.  The crop starts at
    xMin, yMin > 0 and has an odd width, and the LDS-staged kernels run on it (row segments aligned per row); the result
    equals the oracle on the UNCROPPED source, the crop equals orc_create_reduced_domain."""
    import re
    NX, NY, OX, OY, NL = 1201, 1003, 300, 200, 5
    slon, slat = -30.0 + 0.05 * np.arange(NX), 40.0 + 0.03 * np.arange(NY)
    slon.tofile(tmp_path / "x.f64")
    slat.tofile(tmp_path / "y.f64")
    tlon = np.linspace(-30.0 + 0.05 * 400.3, -30.0 + 0.05 * 700.9, OX)
    tlat = np.linspace(40.0 + 0.03 * 250.7, 40.0 + 0.03 * 600.2, OY)
    tlon.tofile(tmp_path / "ox.f64")
    tlat.tofile(tmp_path / "oy.f64")
    f = cases.field(NL, NY, NX, seed=11)
    f.tofile(tmp_path / "t.f32")
    packed = np.random.default_rng(12).integers(-3000, 3000, (NL, NY, NX)).astype(np.int16)
    packed[:, ::37, ::41] = -32767
    packed.tofile(tmp_path / "p.i16")
    lines = ["proj " + GEO, "xaxis %s" % (tmp_path / "x.f64"), "yaxis %s" % (tmp_path / "y.f64"), "method " + method, "outproj " + GEO,
             "outx %s degrees_east" % (tmp_path / "ox.f64"), "outy %s degrees_north" % (tmp_path / "oy.f64"),
             "var t %d %s nan" % (NL, tmp_path / "t.f32"), "var p %d %s -32767 type short" % (NL, tmp_path / "p.i16"), "get t 0", "get p 0"]
    out, stdout = _run(tmp_path, lines)
    m = re.search(r"inX (\d+) inY (\d+) outX (\d+) outY (\d+) reduced xMin (\d+) yMin (\d+) stagedCells (\d+) tile (\d+)x(\d+)", stdout)
    assert m, stdout
    inX, inY, outX, outY, xMin, yMin, staged, tw, th = map(int, m.groups())
    # the positions on the full grid, then the reference's crop of them
    fx = oracle.points2position(np.radians(np.tile(tlon, OY)), np.radians(slon), oracle.LONGITUDE)
    fy = oracle.points2position(np.radians(np.repeat(tlat, OX)), np.radians(slat), oracle.LATITUDE)
    rd = oracle.create_reduced_domain(fx, fy, NX, NY)
    assert rd is not None
    assert (xMin, yMin, inX, inY) == (rd["xMin"], rd["yMin"], rd["inX"], rd["inY"])
    assert xMin > 300 and yMin > 200 and inX < 400 and inY < 400 and inX % 4 != 0, (xMin, yMin, inX, inY)
    assert (outX, outY) == (OX, OY)
    assert staged > 0 and tw > 0, "the cropped grid did not get an LDS-staged plan: " + stdout
    px, py = np.fromfile(out / "points_x.f64"), np.fromfile(out / "points_y.f64")
    # CDMInterpolator keeps the positions on the full grid; the plan holds them minus the crop's origin
    np.testing.assert_allclose(px, fx, atol=1e-7)
    np.testing.assert_allclose(py, fy, atol=1e-7)
    np.testing.assert_allclose(px - xMin, rd["px"], atol=1e-7)
    # subtracting the integer offset is exact in double, so floor and fraction of every position survive the crop:
    # the oracle on the uncropped source with the uncropped positions must give the same bits
    want = oracle.interpolate_values(code, px, py, f, NX, NY, OX, OY)
    got = np.fromfile(out / "t_0.f32", dtype=np.float32).reshape(NL, OY, OX)
    want = oracle.interpolation_array2data(want, oracle.CDM_FLOAT, FILL_FLOAT)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.isfinite(got).mean() > 0.9
    # the same through the stored type (packed shorts with a fill value; odd crop width: the gather form on the stored type)
    wf = oracle.interpolate_values(code, px, py, oracle.data2interpolation_array(packed, -32767.0), NX, NY, OX, OY)
    wantp = oracle.interpolation_array2data(wf, oracle.CDM_SHORT, -32767.0)
    gotp = np.fromfile(out / "p_0.raw", dtype=np.int16).reshape(NL, OY, OX)
    np.testing.assert_array_equal(gotp, wantp)
