"""Plan building across projections on the GPU: fimex_amd_project_values / _project_axes /
_get_vector_reproject_matrix against (a) the fixtures of the reference that cross the PROJ.4 boundary -- the EMEP chain
cell of test/testInterpolation.cc:280-393 with test/inData.txt, test/outData.txt, the stored lon/lat of test/coordTest.nc,
the rotation KATs :396-654 -- and (b) the numpy restatement of the same closed forms (oracle/proj_oracle.py).  Device libm
differs from the host's in the last bits: positions are compared with tolerances, data computed from them bit-exactly in
the other test files."""
import os

import numpy as np
import pytest

import cases
import oracle
from oracle import proj_oracle as po
from test_oracle_kats import EMEP, LATLONG, _emep_case, _rotation_matrix

pytestmark = pytest.mark.gpu

GEO = "+proj=latlong +R=6371000"
STERE = "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +a=6371000 +e=0"
STERE_OBL = "+proj=stere +lat_0=52 +lon_0=10 +R=6371000 +x_0=1000 +y_0=-2000"
STERE_EQ = "+proj=stere +lat_0=0 +lon_0=-20 +R=6371000"
STERE_S = "+proj=stere +lat_0=-90 +lon_0=30 +lat_ts=-71 +R=6371000"
LCC = "+proj=lcc +lat_0=63 +lon_0=15 +lat_1=63 +lat_2=63 +no_defs +R=6.371e+06"
LCC2 = "+proj=lcc +lat_0=48 +lon_0=8 +lat_1=30 +lat_2=60 +R=6371229"
MERC = "+proj=merc +lon_0=5 +lat_ts=30 +R=6371000"
ROT = "+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=22 +R=6.371e+06 +no_defs"
LAEA_S = "+proj=laea +lat_0=52 +lon_0=10 +R=6371000"
LAEA_SP = "+proj=laea +lat_0=90 +lon_0=-30 +R=6371000 +x_0=100"
AEA_S = "+proj=aea +lat_1=40 +lat_2=60 +lat_0=50 +lon_0=10 +R=6371000"
SINU_S = "+proj=sinu +lon_0=10 +R=6371000"
CEA_S = "+proj=cea +lat_ts=30 +lon_0=-20 +R=6371000"
ORTHO = "+proj=ortho +lat_0=40 +lon_0=10 +R=6371000"
ORTHO_P = "+proj=ortho +lat_0=90 +lon_0=-30 +R=6371000"
AEQD = "+proj=aeqd +lat_0=40 +lon_0=10 +R=6371000"
AEQD_P = "+proj=aeqd +lat_0=90 +lon_0=0 +R=6371000"
NSPER = "+proj=nsper +h=2e7 +lat_0=50 +lon_0=10 +R=6371000"
ALL = [GEO, STERE, STERE_OBL, STERE_EQ, STERE_S, LCC, LCC2, MERC, ROT, LAEA_S, LAEA_SP, AEA_S, SINU_S, CEA_S, ORTHO, ORTHO_P, AEQD, AEQD_P, NSPER]
# on an ellipsoid (the UTM string is the one of test/testInterpolator.cc:422)
GEO_W = "+proj=latlong +datum=WGS84"
UTM33 = "+proj=utm +zone=33 +datum=WGS84 +no_defs"
ETMERC = "+proj=etmerc +lon_0=-3 +lat_0=49 +k=0.9996012717 +x_0=400000 +y_0=-100000 +ellps=airy"
UTM17S = "+proj=utm +zone=17 +south +ellps=GRS80"
TMERC_S = "+proj=tmerc +lon_0=12 +lat_0=0 +k=0.9996 +R=6371000"   # lat_0 != 0 on the sphere: PROJ.4 releases pick the hemisphere differently
TMERC_B = "+proj=tmerc +lon_0=9 +lat_0=0 +k=1 +x_0=3500000 +ellps=bessel"
STERE_W = "+proj=stere +lat_0=90 +lon_0=-45 +lat_ts=70 +ellps=WGS84"
STERE_WS = "+proj=stere +lat_0=-90 +lon_0=0 +lat_ts=-71 +ellps=WGS84"
STERE_WP = "+proj=stere +lat_0=90 +lon_0=10 +k=0.994 +ellps=intl"
STERE_WO = "+proj=stere +lat_0=52.156 +lon_0=5.387 +k=0.9999079 +x_0=155000 +y_0=463000 +ellps=bessel"
LCC_W = "+proj=lcc +lat_0=52 +lon_0=10 +lat_1=35 +lat_2=65 +x_0=4000000 +y_0=2800000 +ellps=GRS80"
LCC_W1 = "+proj=lcc +lat_0=63 +lon_0=15 +lat_1=63 +a=6378137 +rf=298.257223563"
MERC_W = "+proj=merc +lon_0=5 +lat_ts=30 +ellps=WGS84"
LAEA_W = "+proj=laea +lat_0=52 +lon_0=10 +x_0=4321000 +y_0=3210000 +ellps=GRS80"   # ETRS89-LAEA
LAEA_WP = "+proj=laea +lat_0=90 +lon_0=0 +ellps=WGS84"
LAEA_WE = "+proj=laea +lat_0=0 +lon_0=20 +ellps=WGS84"
AEA_W = "+proj=aea +lat_1=29.5 +lat_2=45.5 +lat_0=23 +lon_0=-96 +x_0=0 +y_0=0 +ellps=GRS80"
AEA_W1 = "+proj=aea +lat_1=55 +lat_2=55 +lat_0=50 +lon_0=10 +ellps=WGS84"
GEOS_MSG = "+proj=geos +lon_0=0 +h=3.57858e+07  +a=6.37817e+06  +b=6.35658e+06 +no_defs +x_0=-2.2098e+06 +y_0=-3.50297e+06"   # testProjections.cc:69
GEOS_X = "+proj=geos +lon_0=-75 +h=35786023 +sweep=x +ellps=GRS80"
OMERC_REF = "+proj=omerc +lonc=5.34065 +lat_0=60.742 +alpha=19.0198 +no_rot   +a=6.37814e+06  +b=6.35675e+06 +no_defs +x_0=-3.86098e+06 +y_0=1.5594e+06"   # testProjections.cc:56
OMERC_RSO = "+proj=omerc +lat_0=4 +lonc=115 +alpha=53.31582047222222 +gamma=53.13010236111111 +k=0.99984 +x_0=590476.87 +y_0=442857.65 +a=6377298.556 +rf=300.8017"
SINU_W = "+proj=sinu +lon_0=10 +ellps=WGS84"
CEA_W = "+proj=cea +lat_ts=30 +lon_0=0 +ellps=WGS84"
ELLIPSOIDAL = [SINU_W, CEA_W, GEOS_MSG, GEOS_X, AEA_W, AEA_W1, LAEA_W, LAEA_WP, LAEA_WE, UTM33, UTM17S, ETMERC, TMERC_S, TMERC_B, STERE_W, STERE_WS, STERE_WP, STERE_WO, LCC_W, LCC_W1, MERC_W]


@pytest.fixture(scope="module")
def fa():
    from fimex_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


def _is_degree(proj):
    return po.parse(proj)["proj"] in ("latlong", "longlat", "latlon", "lonlat", "ob_tran")


def _close(a, b, proj):
    scale = 1.0 if _is_degree(proj) else 6.4e6  # radians or metres
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-13 * scale * 10)


@pytest.mark.parametrize("dst", ALL)
def test_project_values_from_geographic_and_back(fa, dst):
    rng = np.random.default_rng(len(dst))
    lon = np.radians(rng.uniform(-60, 80, 20000))
    lat = np.radians(rng.uniform(-75 if dst in (STERE_S, MERC, STERE_EQ) else 20, 85 if dst != STERE_S else -30, 20000))
    x, y = fa.project_values_host(GEO, dst, lon, lat)
    wx, wy = po.transform(GEO, dst, lon, lat)
    _close(x, wx, dst); _close(y, wy, dst)
    bx, by = fa.project_values_host(dst, GEO, x, y)  # round trip through the inverse
    np.testing.assert_allclose(bx, lon, atol=1e-11); np.testing.assert_allclose(by, lat, atol=1e-11)
    assert fa.projection_is_degree(dst) == _is_degree(dst)


@pytest.mark.parametrize("src,dst", [(ROT, STERE), (LCC, ROT), (STERE_OBL, MERC), (LCC2, STERE_S)])
def test_project_axes_between_projections(fa, src, dst):
    if _is_degree(src):
        ax, ay = np.radians(np.linspace(-12, 12, 301)), np.radians(np.linspace(15, 45, 200))
    else:
        ax, ay = np.linspace(-9e5, 9e5, 301), np.linspace(-7e5, 8e5, 200)
    gx, gy = fa.project_axes_host(src, dst, ax, ay)
    wx, wy = po.project_axes(src, dst, ax, ay)
    assert gx.shape == (200, 301)
    _close(gx.ravel(), wx, dst); _close(gy.ravel(), wy, dst)


@pytest.mark.parametrize("dst", ELLIPSOIDAL)
def test_ellipsoidal_projections_from_geographic_and_back(fa, dst):
    """The series PROJ.4 uses on an ellipsoid (oracle/proj_oracle.py, pinned by Snyder's worked examples in
    tests/test_oracle_kats.py); transverse Mercator within 12 degrees of its meridian."""
    rng = np.random.default_rng(len(dst))
    tm = "tmerc" in dst or "utm" in dst
    series = "proj=tmerc" in dst   # Gauss-Krueger truncated; utm and etmerc carry the series to n^6 and hold far out
    lon0 = np.degrees(po._Proj(dst).lam0)
    geos = "proj=geos" in dst
    lon = np.radians(rng.uniform(lon0 - 12, lon0 + 12, 20000) if series else (rng.uniform(lon0 - 40, lon0 + 40, 20000) if tm or geos else rng.uniform(-60, 80, 20000)))
    south = dst == STERE_WS
    lat = np.radians(rng.uniform(-85 if south else (-70 if tm or dst in (MERC_W, LAEA_WE) else 20), -30 if south else 85, 20000))
    if geos:
        lat = np.radians(rng.uniform(-60, 60, 20000))   # on the visible disc
    x, y = fa.project_values_host(GEO_W, dst, lon, lat)
    wx, wy = po.transform(GEO_W, dst, lon, lat)
    atol = 2e-6 if dst == TMERC_S else 2e-8   # metres; the spherical form takes acos of nearly 1 at the equator
    np.testing.assert_allclose(x, wx, rtol=2e-12, atol=atol); np.testing.assert_allclose(y, wy, rtol=2e-12, atol=atol)
    bx, by = fa.project_values_host(dst, GEO_W, x, y)
    wbx, wby = po.transform(dst, GEO_W, x, y)
    np.testing.assert_allclose(bx, wbx, atol=2e-13); np.testing.assert_allclose(by, wby, atol=2e-13)
    tol = 2e-7 if series else (1e-9 if "laea" in dst or "cea" in dst else 2e-10)  # truncated series do not invert themselves exactly
    np.testing.assert_allclose(bx, lon, atol=tol); np.testing.assert_allclose(by, lat, atol=tol)
    assert not fa.projection_is_degree(dst)


def test_snyders_worked_examples_on_the_gpu(fa):
    from test_oracle_kats import SNYDER
    for proj, lonlat, xy in SNYDER:
        x, y = fa.project_values_host("+proj=latlong +ellps=clrk66", proj, np.radians([lonlat[0]]), np.radians([lonlat[1]]))
        assert abs(x[0] - xy[0]) < 0.1 and abs(y[0] - xy[1]) < 0.1, (proj, x, y)


def test_transverse_mercator_refuses_the_far_side(fa):
    lon, lat = np.radians([15., 120., -100., 15.]), np.radians([60., 60., 10., -30.])
    tm33 = "+proj=tmerc +lon_0=15 +k=0.9996 +x_0=500000 +ellps=WGS84"
    x, y = fa.project_values_host(GEO_W, tm33, lon, lat)
    wx, wy = po.transform(GEO_W, tm33, lon, lat)
    assert np.array_equal(np.isnan(x), [False, True, True, False]) and np.array_equal(np.isnan(x), np.isnan(wx))
    assert abs(x[0] - 500000.) < 1e-6 and abs(y[0] - wy[0]) < 1e-7


def test_axes_between_an_ellipsoid_and_a_sphere(fa):
    """pj_transform without datums on either side passes geodetic coordinates unchanged (the arome LCC sphere to UTM 33)."""
    ax, ay = np.linspace(-2e5, 12e5, 281), np.linspace(62e5, 80e5, 181)
    gx, gy = fa.project_axes_host(UTM33, LCC, ax, ay)
    wx, wy = po.project_axes(UTM33, LCC, ax, ay)
    np.testing.assert_allclose(gx.ravel(), wx, atol=2e-8); np.testing.assert_allclose(gy.ravel(), wy, atol=2e-8)


def test_unsupported_projection_strings_fail_loudly(fa):
    for bad in ("+proj=utm +zone=33 +R=6371000", "+proj=utm +zone=0 +ellps=WGS84", "+proj=stere +lat_0=0 +ellps=WGS84", "+lat_0=3",
                "+proj=ob_tran +o_proj=stere +R=1", "+proj=ob_tran +o_proj=longlat +o_lat_p=30 +ellps=WGS84", "+proj=merc +ellps=WGS84 +units=parsec", "+proj=merc +R=6371000 +to_meter=0", "+proj=merc +R=6371000 +axis=wsu",
                "+proj=ob_tran +o_proj=longlat +o_lat_p=30 +R=6371000 +units=km", "+proj=merc +R=6371000 +pm=12d30",
                "+proj=merc +datum=NAD27", "+proj=merc +ellps=nonesuch", "+proj=stere +lat_0=90", "+proj=lcc +lat_1=30 +lat_2=-30 +R=1",
                "+proj=omerc +lat_0=60 +R=6371000", "+proj=ortho +lat_0=40 +ellps=WGS84", "+proj=nsper +lat_0=40 +R=6371000",
                "+proj=moll +R=6371000"):
        with pytest.raises(fa.FimexAmdError):
            fa.project_values_host(GEO, bad, np.zeros(3), np.zeros(3))
    with pytest.raises(fa.FimexAmdError):  # grid shifts
        fa.project_values_host(GEO_W, "+proj=latlong +ellps=clrk66 +nadgrids=conus", np.zeros(3), np.zeros(3))


@pytest.mark.parametrize("proj", [STERE, LCC, UTM33, MERC_W, LAEA_W])
@pytest.mark.parametrize("extra", ["+units=km", "+units=us-ft", "+to_meter=1/3.2808", "+pm=paris", "+pm=-17.5", "+units=km +pm=oslo", "+axis=enu +units=m"])
def test_units_and_prime_meridians_on_the_gpu(fa, proj, extra):
    """+units / +to_meter (pj_fwd.c, pj_inv.c) and +pm (pj_transform.c) on either side of a transformation, against the numpy oracle."""
    rng = np.random.default_rng(len(proj) + len(extra))
    lon = np.radians(rng.uniform(2, 28, 5000))
    lat = np.radians(rng.uniform(45, 80, 5000))
    geo = GEO_W if "ellps" in proj or "datum" in proj else GEO
    dst = proj + " " + extra
    x, y = fa.project_values_host(geo, dst, lon, lat)
    wx, wy = po.transform(geo, dst, lon, lat)
    _close(x, wx, dst); _close(y, wy, dst)
    bx, by = fa.project_values_host(dst, geo, x, y)
    np.testing.assert_allclose(bx, lon, atol=1e-11); np.testing.assert_allclose(by, lat, atol=1e-11)
    if "pm" in extra:  # the meridian on the geographic side
        gx, gy = fa.project_values_host(geo + " +pm=lisbon", proj, lon, lat)
        vx, vy = po.transform(geo + " +pm=lisbon", proj, lon, lat)
        _close(gx, vx, proj); _close(gy, vy, proj)
    ax, ay = np.linspace(2e5, 6e5, 40) / po.to_meter_of(po.parse(dst)), np.linspace(5.5e6, 6.5e6, 30) / po.to_meter_of(po.parse(dst))
    if proj == UTM33:
        gx, gy = fa.project_axes_host(dst, geo, ax, ay)
        vx, vy = po.project_axes(dst, geo, ax, ay)
        np.testing.assert_allclose(gx.ravel(), vx, atol=1e-12); np.testing.assert_allclose(gy.ravel(), vy, atol=1e-12)


def test_datum_shifts_on_the_gpu(fa):
    """pj_datum_transform between sides that both name a datum: three and seven parameters, through projections, both ways,
    against the oracle (whose geocentric steps reproduce the EPSG worked examples) -- and nothing when one side names none."""
    rng = np.random.default_rng(3)
    lon, lat = np.radians(rng.uniform(-10, 30, 5000)), np.radians(rng.uniform(35, 70, 5000))
    cases_ = [("+proj=latlong +datum=WGS84", "+proj=latlong +datum=potsdam"),
              ("+proj=latlong +datum=WGS84", "+proj=utm +zone=33 +ellps=intl +towgs84=-87,-98,-121"),                 # ED50
              ("+proj=latlong +ellps=bessel +towgs84=598.1,73.7,418.2,0.202,0.045,-2.455,6.7", "+proj=etmerc +lat_0=0 +lon_0=9 +k=1 +x_0=3500000 +datum=potsdam"),   # Gauss-Krueger zone 3
              ("+proj=latlong +datum=GGRS87", "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +datum=WGS84"),
              ("+proj=latlong +datum=WGS84", "+proj=latlong +datum=NAD83")]
    for src, dst in cases_:
        x, y = fa.project_values_host(src, dst, lon, lat)
        wx, wy = po.transform(src, dst, lon, lat)
        scale = 1.0 if "latlong" in dst else 6.4e6
        np.testing.assert_allclose(x, wx, rtol=0, atol=2e-12 * scale); np.testing.assert_allclose(y, wy, rtol=0, atol=2e-12 * scale)
        bx, by = fa.project_values_host(dst, src, x, y)
        np.testing.assert_allclose(bx, lon, atol=2e-9); np.testing.assert_allclose(by, lat, atol=2e-9)
    moved = fa.project_values_host("+proj=latlong +datum=WGS84", "+proj=latlong +datum=potsdam", lon, lat)
    assert 1e-5 < np.abs(moved[0] - lon).max() < 3e-4
    same = fa.project_values_host("+proj=latlong +datum=WGS84", "+proj=latlong +ellps=bessel", lon, lat)
    assert np.array_equal(same[0], lon) and np.array_equal(same[1], lat)
    # the mesh form shifts as well
    ax, ay = np.linspace(3e5, 7e5, 41), np.linspace(55e5, 60e5, 31)
    gx, gy = fa.project_axes_host("+proj=utm +zone=32 +datum=potsdam", "+proj=latlong +datum=WGS84", ax, ay)
    wx, wy = po.project_axes("+proj=utm +zone=32 +datum=potsdam", "+proj=latlong +datum=WGS84", ax, ay)
    np.testing.assert_allclose(gx.ravel(), wx, atol=2e-12); np.testing.assert_allclose(gy.ravel(), wy, atol=2e-12)


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
def test_emep_chain_on_the_gpu(fa, golden_dir, method):
    """test/testInterpolation.cc:280-393 with plan building and regridding on the device: cell (-25, 43) == 32."""
    import torch
    field, wpx, wpy, (iS, jS, lonS, latS), lon, lat = _emep_case(golden_dir)
    n = lonS * latS
    d = torch.empty(2 * n, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    fa.project_axes_device(LATLONG, EMEP, np.radians(lon), np.radians(lat), d.data_ptr(), d.data_ptr() + 8 * n, st)
    fa.points2position_device(d.data_ptr(), n, np.arange(iS) + 1., fa.PROJ_AXIS, st)
    fa.points2position_device(d.data_ptr() + 8 * n, n, np.arange(jS) + 1., fa.PROJ_AXIS, st)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d[:n].cpu().numpy(), wpx, atol=1e-9)
    np.testing.assert_allclose(d[n:].cpu().numpy(), wpy, atol=1e-9)
    plan = fa.RegridPlan.from_device(method, d.data_ptr(), d.data_ptr() + 8 * n, n, iS, jS, lonS, latS, st)
    out = plan.apply_host(field[None])[0]
    assert lon[9] == -25 and lat[25] == 43 and abs(out[25, 9] - 32) < 1e-6


def test_emep_outdata_fixture_on_the_gpu(fa, golden_dir):
    """test/outData.txt (bilinear, axes numbered from 0) through device-built positions."""
    field, _, _, (iS, jS, lonS, latS), lon, lat = _emep_case(golden_dir)
    gx, gy = fa.project_axes_host(LATLONG, EMEP, np.radians(lon), np.radians(lat))
    px = fa.points2position_host(gx.ravel(), np.arange(iS) + 0.)
    py = fa.points2position_host(gy.ravel(), np.arange(jS) + 0.)
    out = fa.RegridPlan(oracle.BILINEAR, px, py, iS, jS, lonS, latS).apply_host(field[None])[0].astype(np.float64)
    want = np.loadtxt(os.path.join(golden_dir, "outData.txt"))[:, 2].reshape(lonS, latS).T
    both = ~np.isnan(out) & ~np.isnan(want)
    assert both.sum() > 14000
    assert (np.abs(out[both] - want[both]) / np.abs(want[both])).max() < 1e-5


def test_stere_inverse_against_coordtest_on_the_gpu(fa, golden_dir):
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        x, y = f.variables["x"].data.astype(np.float64), f.variables["y"].data.astype(np.float64)
        lon, lat = f.variables["longitude"].data.astype(np.float64), f.variables["latitude"].data.astype(np.float64)
        projstr = f.variables["projection_1"].proj4.decode()
    lo, la = fa.project_axes_host(projstr, "+proj=latlong +R=6.371e6", x, y)
    np.testing.assert_allclose(np.degrees(lo), lon, atol=2e-5)
    np.testing.assert_allclose(np.degrees(la), lat, atol=2e-5)


@pytest.mark.parametrize("pin,pout,xa,ya,types", [
    (STERE, GEO, np.linspace(-25, 25, 60), np.linspace(52, 78, 40), (1, 2)),
    (GEO, LCC, (np.arange(40) - 19.5) * 25000.0, (np.arange(30) - 14.5) * 25000.0, (0, 0)),
    (ROT, STERE_OBL, np.linspace(-8e5, 8e5, 33), np.linspace(-6e5, 6e5, 21), (0, 0)),
    ("+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=0 +lat_ts=60", "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=90 +lat_ts=60",
     np.arange(5) - 2., np.arange(5) - 2., (0, 0)),
])
def test_vector_reproject_matrix_on_the_gpu(fa, pin, pout, xa, ya, types):
    """mifi_get_vector_reproject_matrix against the restatement that passes the reference's rotation KATs
    (tests/test_oracle_kats.py: 90 / 180 degree turns, length preservation, direction angles)."""
    got = fa.get_vector_reproject_matrix_host(pin, pout, xa, ya, types[0], types[1]).reshape(-1, 4)
    want = _rotation_matrix(pin, pout, xa, ya, types[0], types[1]).reshape(-1, 4)
    np.testing.assert_allclose(got[:, :3], want[:, :3], atol=2e-6)  # the angle is a difference quotient over 0.1 % of a cell
    np.testing.assert_allclose(np.hypot(got[:, 0], got[:, 1]), 1.0, atol=1e-14)
    np.testing.assert_array_equal(got[:, 2], -got[:, 1])
    np.testing.assert_allclose(np.cos(got[:, 3]), got[:, 0], atol=1e-15)


def test_quarter_turn_rotates_u_into_v(fa):
    """test/testInterpolation.cc:396-453 with the matrix built on the device: lon_0 turned by 90 degrees maps u -> -v, v -> u."""
    p1 = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=0 +lat_ts=60"
    p2 = "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=90 +lat_ts=60"
    ax = np.arange(5) - 2.
    m = fa.get_vector_reproject_matrix_host(p1, p2, ax, ax)
    u = np.arange(25, dtype=np.float32)
    v = (25 - np.arange(25)).astype(np.float32)
    gu, gv = fa.VectorPlan(m, 5, 5).reproject_values_host(u, v)
    wu, wv = oracle.vector_reproject_values(_rotation_matrix(p1, p2, ax, ax, 0, 0), u, v, 5, 5)
    np.testing.assert_allclose(np.ravel(gu), np.ravel(wu), atol=1e-4); np.testing.assert_allclose(np.ravel(gv), np.ravel(wv), atol=1e-4)
    np.testing.assert_allclose(np.hypot(np.ravel(gu), np.ravel(gv)), np.hypot(u, v), rtol=1e-5)
    np.testing.assert_allclose(np.ravel(gu), v, atol=1e-4)  # u' = v, v' = -u for this turn (testInterpolation.cc:441-447)


def test_matrix_from_a_field_in_the_input_projection(fa):
    """mifi_get_vector_reproject_matrix_field (CDMProcessor's rotation to lat/lon): the mesh of the grid's own axes."""
    from test_oracle_kats import _rotation_matrix_field
    xa, ya = np.linspace(-8e5, 8e5, 41), np.linspace(-6e5, 7e5, 29)
    xx, yy = np.meshgrid(xa, ya)
    got = fa.get_vector_reproject_matrix_field_host(STERE_OBL, GEO, xx, yy).reshape(-1, 4)
    ox, oy = po.transform(STERE_OBL, GEO, xx.ravel(), yy.ravel())
    want = _rotation_matrix_field(STERE_OBL, GEO, xx.ravel(), yy.ravel(), ox, oy, xa.size, ya.size).reshape(-1, 4)
    np.testing.assert_allclose(got[:, :3], want[:, :3], atol=2e-6)


def test_matrix_at_points(fa):
    """mifi_get_vector_reproject_matrix_points: fixed 100 m / 1e-5 rad differences."""
    rng = np.random.default_rng(8)
    lon, lat = np.radians(rng.uniform(-20, 40, 500)), np.radians(rng.uniform(45, 80, 500))
    got = fa.get_vector_reproject_matrix_points_host(STERE, GEO, True, lon, lat).reshape(-1, 4)
    ix, iy = po.transform(GEO, STERE, lon, lat)
    xdx = po.transform(STERE, GEO, ix + 100, iy)
    ydy = po.transform(STERE, GEO, ix, iy + 100)
    want = oracle.vector_matrix_from_deltas(lon, lat, xdx, ydy, 100., 100., True).reshape(-1, 4)
    np.testing.assert_allclose(got[:, :3], want[:, :3], atol=1e-7)


def test_rotate_vector_on_stored_types_and_packed_directions(fa):
    """src/CDMProcessor.cc:590-636: both components from their stored types, the requested one back in its type; directions
    stored as short with scale_factor 0.1 unpacked, rotated and packed again."""
    ox, oy, oz = 37, 23, 3
    m = cases.rotation_matrix(ox, oy, seed=3)
    vec = fa.VectorPlan(m, ox, oy)
    rng = np.random.default_rng(4)
    x = rng.integers(-3000, 3000, (oz, oy, ox)).astype(np.int16)
    y = rng.normal(0, 12, (oz, oy, ox)).astype(np.float32)
    x.reshape(-1)[rng.choice(x.size, 40, replace=False)] = -32767
    y.reshape(-1)[rng.choice(y.size, 40, replace=False)] = np.float32(9.96921e36)
    fx, fy = oracle.data2interpolation_array(x, -32767.0), oracle.data2interpolation_array(y, 9.96921e36)
    ru, rv = oracle.vector_reproject_values(m, fx, fy, ox, oy)
    gx = fa.rotate_vector_typed_host(vec, x, -32767.0, y, 9.96921e36, returnX=True)
    gy = fa.rotate_vector_typed_host(vec, x, -32767.0, y, 9.96921e36, returnX=False)
    assert np.array_equal(gx, oracle.interpolation_array2data(ru, oracle.CDM_SHORT, -32767.0).reshape(gx.shape))
    wy = oracle.interpolation_array2data(rv, oracle.CDM_FLOAT, 9.96921e36).reshape(gy.shape)
    assert np.array_equal(gy.view(np.uint32), wy.view(np.uint32))
    # packed directions
    ang = rng.integers(0, 3600, (oz, oy, ox)).astype(np.float32)
    ang[0, 0, :5] = np.nan
    scale, offset = 0.1, 5.0
    unpacked = (scale * ang.astype(np.float64) + offset).astype(np.float32)
    rotated = oracle.vector_reproject_direction(m, unpacked, ox, oy)
    want = ((1 / scale) * (rotated.astype(np.float64) - offset)).astype(np.float32)
    got = vec.reproject_direction_scaled_host(ang, scale, offset)
    assert cases.same(got, want.reshape(got.shape)), cases.describe_mismatch(got, want.reshape(got.shape))


BACKFORTH = [  # test/testInterpolator.cc:401-424: projection, its axes, their unit, lon / lat axes of the way back, tolerance
    ("+proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +ellps=sphere +R=6371000", np.arange(-3e7, 3e7 + 1, 50000.), np.arange(-3e7, 3e7 + 1, 50000.), "m",
     np.arange(-180., 180.), np.arange(55., 88.), 8e-2),
    ("+proj=stere +lat_0=-90 +lon_0=0 +lat_ts=-90 +ellps=sphere +R=6371000", np.arange(-3e7, 3e7 + 1, 50000.), np.arange(-3e7, 3e7 + 1, 50000.), "m",
     np.array([0., 1., 359.]), -np.arange(55., 88.), 8e-2),   # the reference's "0,1,359" is these three meridians
    ("+proj=lcc +lat_0=63 +lon_0=15 +lat_1=63 +lat_2=63 +no_defs +R=6.371e+06", np.arange(-922000., 922001, 20000.), np.arange(-1130000., 1230001, 20000.), "m",
     np.arange(-30., 41.), np.arange(50., 86.), 1e-2),
    ("+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=22 +R=6.371e+06 +no_defs", np.arange(16.5, 24.25, .1), np.arange(-3.8, 14.95, .1), "degree",
     np.arange(-30., 41.), np.arange(50., 86.), 5e-3),
    ("+proj=latlon +R=6.371e+06 +no_defs", np.arange(-179., 180.), np.arange(-89.5, 90., .5), "degree",
     np.arange(-180., 180.), np.arange(-90., 91.), 1e-3),
    ("+proj=utm +zone=33 +datum=WGS84 +no_defs", np.arange(-4e5, 14e5 + 1, 2000.), np.arange(55e5, 90e5 + 1, 2000.), "m",
     np.arange(-30., 41.), np.arange(50., 86.), 1e-2),
]


@pytest.mark.parametrize("wind", [(0., 1.), (1., 0.)], ids=["north", "east"])
@pytest.mark.parametrize("case", BACKFORTH, ids=["stere", "stere_south", "lcc", "hirlam8", "latlon", "utm33"])
def test_wind_to_a_projection_and_back(fa, case, wind):
    """test/testInterpolator.cc:398-472 (test_interpolator_vector_backforth): a constant northward (eastward) wind on a
    lat/lon grid, regridded (nearest) and rotated into a projection, then regridded and rotated back to lat/lon, is the
    same wind again within the reference's tolerance.  test/data/north.nc, east.nc are NetCDF-4 (not readable here):
    the constant fields are re-created on a 1-degree global grid.  Plans and both rotation matrices are built on the device."""
    proj, xa, ya, unit, lonb, latb, delta = case
    geo = "+proj=latlon +R=6371000"
    deg = unit == "degree"
    slon, slat = np.arange(-180., 181.), np.arange(-90., 91.)
    u0 = np.full((slat.size, slon.size), wind[0], np.float32)
    v0 = np.full((slat.size, slon.size), wind[1], np.float32)
    xr, yr = (np.radians(xa), np.radians(ya)) if deg else (xa, ya)   # src/CDMInterpolator.cc:1443-1451
    types = (fa.LONGITUDE, fa.LATITUDE) if deg else (fa.PROJ_AXIS, fa.PROJ_AXIS)
    # there
    lon, lat = fa.project_axes_host(proj, geo, xr, yr)
    px = fa.points2position_host(lon.ravel(), np.radians(slon), fa.LONGITUDE)
    py = fa.points2position_host(lat.ravel(), np.radians(slat), fa.LATITUDE)
    plan = fa.RegridPlan(oracle.NEAREST, px, py, slon.size, slat.size, xa.size, ya.size)
    u1, v1 = plan.apply_host(u0), plan.apply_host(v0)
    m1 = fa.get_vector_reproject_matrix_host(geo, proj, xa, ya, *types)
    u1, v1 = fa.VectorPlan(m1, xa.size, ya.size).reproject_values_host(u1, v1)
    assert np.isfinite(np.asarray(u1)).mean() > 0.5
    # and back
    qx, qy = fa.project_axes_host(geo, proj, np.radians(lonb), np.radians(latb))
    bx = fa.points2position_host(qx.ravel(), xr, types[0])
    by = fa.points2position_host(qy.ravel(), yr, types[1])
    back = fa.RegridPlan(oracle.NEAREST, bx, by, xa.size, ya.size, lonb.size, latb.size)
    u2, v2 = back.apply_host(np.asarray(u1, np.float32)), back.apply_host(np.asarray(v1, np.float32))
    m2 = fa.get_vector_reproject_matrix_host(proj, geo, lonb, latb, fa.LONGITUDE, fa.LATITUDE)
    u2, v2 = fa.VectorPlan(m2, lonb.size, latb.size).reproject_values_host(u2, v2)
    u2, v2 = np.ravel(u2), np.ravel(v2)
    ok = ~(np.isnan(u2) | np.isnan(v2))
    assert ok.mean() > 0.05, ok.mean()
    assert np.abs(u2[ok] - wind[0]).max() < delta and np.abs(v2[ok] - wind[1]).max() < delta, \
        (np.abs(u2[ok] - wind[0]).max(), np.abs(v2[ok] - wind[1]).max())


def test_snyders_examples_of_the_remaining_projections_on_the_gpu(fa):
    from test_oracle_kats import SNYDER_MORE
    for geo, proj, lonlat, xy, tol in SNYDER_MORE:
        x, y = fa.project_values_host(geo, proj, np.radians([lonlat[0]]), np.radians([lonlat[1]]))
        assert abs(x[0] - xy[0]) < tol and abs(y[0] - xy[1]) < tol, (proj, x, y)
    # what lies behind the globe is not a number
    for proj in (ORTHO, NSPER):
        x, y = fa.project_values_host(GEO, proj, np.radians([-170., 10.]), np.radians([-40., 50.]))
        assert np.isnan(x[0]) and np.isnan(y[0]) and np.isfinite(x[1])


def test_oblique_mercator_on_the_gpu(fa):
    """The EPSG worked example (Timbalai 1948 / RSO Borneo) and the oracle on a swath of points around each projection's line."""
    lon, lat = np.radians([115 + 48 / 60 + 19.8196 / 3600]), np.radians([5 + 23 / 60 + 14.1129 / 3600])
    x, y = fa.project_values_host("+proj=latlong +a=6377298.556 +rf=300.8017", OMERC_RSO, lon, lat)
    assert abs(x[0] - 679245.73) < 0.01 and abs(y[0] - 596562.78) < 0.01
    rng = np.random.default_rng(9)
    for proj, lo, la in ((OMERC_RSO, (108, 120), (0, 8)), (OMERC_REF, (-25, 15), (30, 70))):
        lon, lat = np.radians(rng.uniform(*lo, 5000)), np.radians(rng.uniform(*la, 5000))
        x, y = fa.project_values_host(GEO_W, proj, lon, lat)
        wx, wy = po.transform(GEO_W, proj, lon, lat)
        np.testing.assert_allclose(x, wx, rtol=2e-12, atol=2e-8); np.testing.assert_allclose(y, wy, rtol=2e-12, atol=2e-8)
        bx, by = fa.project_values_host(proj, GEO_W, x, y)
        np.testing.assert_allclose(bx, lon, atol=2e-10); np.testing.assert_allclose(by, lat, atol=2e-10)
    with pytest.raises(fa.FimexAmdError):   # the two-point form is not implemented
        fa.project_values_host(GEO_W, "+proj=omerc +lat_0=40 +lat_1=47.5 +lon_1=-122.3 +lat_2=39 +lon_2=-104.5 +ellps=clrk66", lon, lat)


@pytest.mark.parametrize("proj", ["+proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +ellps=sphere +a=6371000 +e=0", GEOS_MSG, OMERC_REF])
def test_reference_conversion_round_trips_on_the_gpu(fa, proj):
    """test/testProjections.cc:84-208: the projection's 10 x 10 mesh at 50 km to longitude / latitude and back
    within 1e-5 m, here through fimex_amd_project_values; beyond the limb the satellite view yields NaN."""
    ll = "+proj=lonlat +ellps=sphere +a=6371000 +e=0"
    x, y = np.meshgrid(np.arange(10) * 50000., np.arange(10) * 50000., indexing="ij")
    lon, lat = fa.project_values_host(proj, ll, x.ravel(), y.ravel())
    assert np.all(np.abs(np.degrees(lat)) <= 90.001) and np.all(np.abs(np.degrees(lon)) <= 180.001)
    bx, by = fa.project_values_host(ll, proj, lon, lat)
    assert np.abs(bx - x.ravel()).max() < 1e-5 and np.abs(by - y.ravel()).max() < 1e-5
    if "geos" in proj:
        hx, hy = fa.project_values_host(ll, proj, np.radians([150., 0.]), np.radians([0., 0.]))
        assert np.isnan(hx[0]) and np.isnan(hy[0]) and abs(hx[1] + 2.2098e6) < 1e-6
