"""The C++ host mirror's own projection code (fimex_amd/host/Projection.cc) against the numpy restatement in
oracle/proj_oracle.py and against the coordTest.nc fixture; method-name mapping.  CPU only (host_cli host modes)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import proj_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "fimex_amd", "host_cli.so")


def _project(tmp_path, src, dst, x, y):
    np.asarray(x, np.float64).tofile(tmp_path / "x.f64")
    np.asarray(y, np.float64).tofile(tmp_path / "y.f64")
    subprocess.check_call([CLI, "--project", src, dst, str(tmp_path / "x.f64"), str(tmp_path / "y.f64"), str(tmp_path)])
    return np.fromfile(tmp_path / "px.f64"), np.fromfile(tmp_path / "py.f64")


PROJS = [
    "+proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +ellps=sphere +R=6371000",
    "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=90 +R=6371000",
    "+proj=stere +lat_0=-90 +lon_0=0 +lat_ts=-90 +R=6371000",
    "+proj=stere +lat_0=52 +lon_0=10 +R=6371000",
    "+proj=lcc +lat_0=63 +lon_0=15 +lat_1=63 +lat_2=63 +no_defs +R=6.371e+06",
    "+proj=lcc +lat_0=45 +lon_0=5 +lat_1=40 +lat_2=50 +R=6.371e+06",
    "+proj=ob_tran +o_proj=longlat +lon_0=-40 +o_lat_p=22 +R=6.371e+06 +no_defs",
    "+proj=ob_tran +o_proj=longlat +lon_0=0 +o_lat_p=60 +R=6.371e6",
    "+proj=merc +lon_0=0 +lat_ts=30 +R=6371000",
    "+ellps=sphere +a=127.4 +e=0 +proj=stere +lat_0=90 +lon_0=-32 +lat_ts=60 +x_0=7 +y_0=109",
]
GEO = "+proj=latlong +R=6371000"


@pytest.mark.parametrize("proj", PROJS)
def test_forward_inverse_match_numpy_restatement(tmp_path, proj):
    rng = np.random.default_rng(1)
    lon = np.radians(rng.uniform(-170, 170, 500))
    lat = np.radians(rng.uniform(-80, 85, 500) if "lat_0=-90" not in proj else rng.uniform(-85, -20, 500))
    if "lat_0=90" in proj:
        lat = np.abs(lat) * 0.9 + 0.1
    x, y = _project(tmp_path, GEO, proj, lon, lat)
    wx, wy = po.transform(GEO, proj, lon, lat)
    np.testing.assert_allclose(x, wx, rtol=1e-12, atol=1e-6)
    np.testing.assert_allclose(y, wy, rtol=1e-12, atol=1e-6)
    lo, la = _project(tmp_path, proj, GEO, x, y)
    np.testing.assert_allclose(la, lat, atol=1e-9)
    np.testing.assert_allclose(np.angle(np.exp(1j * (lo - lon))), 0, atol=1e-9)


def test_stere_inverse_against_coordtest(tmp_path, golden_dir):
    from scipy.io import netcdf_file
    with netcdf_file(os.path.join(golden_dir, "coordTest.nc"), "r", mmap=False) as f:
        x = f.variables["x"].data.astype(np.float64)
        y = f.variables["y"].data.astype(np.float64)
        lon = f.variables["longitude"].data.astype(np.float64)
        lat = f.variables["latitude"].data.astype(np.float64)
        projstr = f.variables["projection_1"].proj4.decode()
    xx, yy = np.meshgrid(x, y)
    lo, la = _project(tmp_path, projstr, GEO, xx.ravel(), yy.ravel())
    np.testing.assert_allclose(np.degrees(la).reshape(11, 11), lat, atol=2e-5)
    np.testing.assert_allclose((np.degrees(lo).reshape(11, 11) - lon + 180) % 360 - 180, 0, atol=2e-5)


def test_rotation_matrix_matches_oracle_restatement(tmp_path):
    """vectorReprojectMatrix (a15) vs orc_vector_matrix_from_deltas fed with numpy projections."""
    import oracle
    p1 = "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +R=6371000"
    for p2, ax, ay in [("+proj=stere +lat_0=90 +lon_0=90 +lat_ts=60 +R=6371000", (np.arange(7) - 3) * 1e5, (np.arange(5) - 2) * 1e5),
                       (GEO, np.radians(np.arange(-20, 21, 5.0)), np.radians(np.arange(60, 86, 5.0)))]:
        np.asarray(ax, np.float64).tofile(tmp_path / "ax.f64")
        np.asarray(ay, np.float64).tofile(tmp_path / "ay.f64")
        subprocess.check_call([CLI, "--matrix", p1, p2, str(tmp_path / "ax.f64"), str(tmp_path / "ay.f64"), str(tmp_path)])
        got = np.fromfile(tmp_path / "matrix.f64")
        ox, oy = len(ax), len(ay)
        xx, yy = np.meshgrid(ax, ay)
        out_x, out_y = xx.ravel(), yy.ravel()
        in_x, in_y = po.transform(p2, p1, out_x, out_y)
        d = 1e-3
        delta = (d * (in_x[ox + 1] - in_x[0]) + d * (in_x[(oy // 2 + 1) * ox + ox // 2 + 1] - in_x[(oy // 2) * ox + ox // 2])) / 2
        want = oracle.vector_matrix_from_deltas(out_x, out_y, po.transform(p1, p2, in_x + delta, in_y),
                                                po.transform(p1, p2, in_x, in_y + delta), delta, delta, po.is_latlong(po.parse(p2)))
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def test_method_names():
    code = lambda s: int(subprocess.check_output([CLI, "--method", s]).decode())
    assert code("bilinear") == 1 and code("nearestneighbor") == 0 and code("bicubic") == 2
    assert code("coord_nearestneighbor") == 3 and code("coord_kdtree") == 4
    assert code("forward_mean") == 6 and code("forward_undef_max") == 13
    assert code("forward_undef_min") == 9  # the reference's mapping, src/interpolation.c:97-98
    assert code("no_such_method") == -1


def test_unsupported_projection_is_an_error(tmp_path):
    np.zeros(1).tofile(tmp_path / "x.f64")
    r = subprocess.run([CLI, "--project", "+proj=utm +zone=33 +datum=WGS84", GEO, str(tmp_path / "x.f64"), str(tmp_path / "x.f64"),
                        str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 1 and "CDMException" in r.stderr
