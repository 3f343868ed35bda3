"""Properties of the CPU oracle beyond the reference's own KATs: the batch loops against independent
numpy / pure-Python restatements on small cases, and the branch structure of the point kernels.  CPU only."""
import math

import numpy as np
import pytest

import cases
import oracle


def _lround(v):
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def _bilinear_py(f, x, y):
    """src/interpolation.c:881-957 for one slice in plain Python with numpy float32 scalars."""
    iy, ix = f.shape
    F = np.float32
    if not (math.isfinite(x) and math.isfinite(y) and abs(x) < 2 ** 30 and abs(y) < 2 ** 30):
        return F(np.nan)
    x0, y0 = math.floor(x), math.floor(y)
    xf, yf = F(x - x0), F(y - y0)
    one = F(1)
    if 0 <= x0 and x0 + 1 < ix:
        if 0 <= y0 and y0 + 1 < iy:
            a = (one - xf) * f[y0, x0] + xf * f[y0, x0 + 1]
            b = (one - xf) * f[y0 + 1, x0] + xf * f[y0 + 1, x0 + 1]
            return (one - yf) * a + yf * b
        ry = _lround(y)
        if 0 <= ry < iy:
            return (one - xf) * f[ry, x0] + xf * f[ry, x0 + 1]
        return F(np.nan)
    rx = _lround(x)
    if 0 <= rx < ix:
        if 0 <= y0 and y0 + 1 < iy:
            return (one - yf) * f[y0, rx] + yf * f[y0 + 1, rx]
        ry = _lround(y)
        if 0 <= ry < iy:
            return f[ry, rx]
    return F(np.nan)


def test_bilinear_batch_loop_against_python():
    inX, inY, outX, outY = 13, 11, 17, 15
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=3)
    f = cases.field(2, inY, inX, seed=1)
    got = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
    with np.errstate(all="ignore"):
        want = np.array([[_bilinear_py(f[z], px[i], py[i]) for i in range(px.size)] for z in range(2)], dtype=np.float32)
    assert cases.same(got.reshape(2, -1), want), cases.describe_mismatch(got, want)


def test_nearest_batch_loop_against_numpy():
    inX, inY, outX, outY = 23, 19, 31, 29
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=4)
    f = cases.field(3, inY, inX, seed=2)
    got = oracle.interpolate_values(oracle.NEAREST, px, py, f, inX, inY, outX, outY).reshape(3, -1)
    for i in range(px.size):
        x, y = px[i], py[i]
        ok = math.isfinite(x) and math.isfinite(y) and abs(x) < 2 ** 30 and abs(y) < 2 ** 30
        rx, ry = (_lround(x), _lround(y)) if ok else (-1, -1)
        if 0 <= rx < inX and 0 <= ry < inY:
            assert cases.same(got[:, i], f[:, ry, rx])
        else:
            assert np.all(np.isnan(got[:, i]))


def test_bicubic_reproduces_cubic_polynomials():
    """Keys' kernel (a = -0.5) is exact for quadratics; double weights, float accumulation."""
    inX, inY = 20, 18
    y, x = np.meshgrid(np.arange(inY, dtype=np.float64), np.arange(inX, dtype=np.float64), indexing="ij")
    f = (3 + 0.5 * x - 0.25 * y + 0.125 * x * x + 0.0625 * y * y).astype(np.float32)
    rng = np.random.default_rng(0)
    px = rng.uniform(1, inX - 2.001, 500)
    py = rng.uniform(1, inY - 2.001, 500)
    got = oracle.interpolate_values(oracle.BICUBIC, px, py, f, inX, inY, 500, 1)[0, 0]
    want = 3 + 0.5 * px - 0.25 * py + 0.125 * px * px + 0.0625 * py * py
    np.testing.assert_allclose(got, want, rtol=2e-6)
    # outside the 4x4 support: undefined, no border fallback (interpolation.c:975-976)
    out = oracle.interpolate_values(oracle.BICUBIC, [0.99, 1.0, inX - 3.0, inX - 2.0], [5, 5, 5, 5], f, inX, inY, 4, 1)[0, 0]
    assert np.isnan(out[0]) and not np.isnan(out[1]) and not np.isnan(out[2]) and np.isnan(out[3])


def test_apply_loop_threads_do_not_change_results():
    inX, inY, outX, outY = 64, 48, 70, 50
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=8)
    f = cases.field(5, inY, inX, seed=8)
    for m in (oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC):
        a = oracle.interpolate_values(m, px, py, f, inX, inY, outX, outY, nthreads=1)
        b = oracle.interpolate_values(m, px, py, f, inX, inY, outX, outY, nthreads=4)
        assert cases.same(a, b)


def test_unknown_methods_raise():
    with pytest.raises(ValueError):
        oracle.interpolate_values(oracle.FWD_SUM, [0.], [0.], np.zeros((1, 2, 2), np.float32), 2, 2, 1, 1)
    with pytest.raises(ValueError):
        oracle.forward_interpolate_values(oracle.BILINEAR, np.zeros(4), np.zeros(4), np.zeros((1, 2, 2), np.float32), 2, 2, 1, 1)


def test_round_and_clamp():
    rc = oracle.round_and_clamp
    assert rc(0.5, 0, 9) == 1 and rc(-0.5, 0, 9) == -1 and rc(-0.49, 0, 9) == 0   # half away from zero
    assert rc(9.49, 0, 9) == 9 and rc(9.5, 0, 9) == -1
    assert rc(float("nan"), 0, 9) == -1 and rc(1e300, 0, 9) == -1 and rc(-999.0, 0, 9) == -1


@pytest.mark.parametrize("method", range(oracle.FWD_SUM, oracle.FWD_UNDEF_MIN + 1))
def test_forward_against_python_buckets(method):
    """src/CachedForwardInterpolation.cc:92-131 with Python lists as the std::vector buckets."""
    inX, inY, outX, outY, nz = 19, 17, 7, 6, 2
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=5, density=4.0)
    f = cases.field(nz, inY, inX, seed=method, nan_frac=0.1, extremes=False)
    got = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    kind = (method - oracle.FWD_SUM) % 5
    undef = method >= oracle.FWD_UNDEF_SUM
    F = np.float32
    for z in range(nz):
        buckets = [[] for _ in range(outX * outY)]
        for i, val in enumerate(f[z].ravel()):
            if undef or not np.isnan(val):
                tx = oracle.round_and_clamp(px[i], 0, outX - 1)
                ty = oracle.round_and_clamp(py[i], 0, outY - 1)
                if tx >= 0 and ty >= 0:
                    buckets[ty * outX + tx].append(val)
        want = np.full(outX * outY, np.nan, np.float32)
        with np.errstate(all="ignore"):
            for t, b in enumerate(buckets):
                if not b:
                    continue
                if kind in (0, 1):
                    s = F(0)
                    for v in b:
                        s = F(s + v)
                    want[t] = s if kind == 0 else F(s / F(len(b)))
                elif kind == 2:
                    want[t] = np.nan if any(np.isnan(v) for v in b) else sorted(b)[len(b) // 2]
                elif kind == 3:
                    m = b[0]
                    for v in b[1:]:
                        if m < v:
                            m = v
                    want[t] = m
                else:
                    m = b[0]
                    for v in b[1:]:
                        if v < m:
                            m = v
                    want[t] = m
        assert cases.same(got[z].ravel(), want), cases.describe_mismatch(got[z], want)


def test_rotation_against_numpy_double():
    ox, oy, oz = 9, 7, 3
    m = cases.rotation_matrix(ox, oy, seed=1)
    u = cases.field(oz, oy, ox, seed=1, extremes=False)
    v = cases.field(oz, oy, ox, seed=2, extremes=False)
    gu, gv = oracle.vector_reproject_values(m, u, v, ox, oy)
    c, s = m[0::4].reshape(oy, ox), m[1::4].reshape(oy, ox)
    wu = (u.astype(np.float64) * c - v.astype(np.float64) * s).astype(np.float32)
    wv = (u.astype(np.float64) * s + v.astype(np.float64) * c).astype(np.float32)
    assert cases.same(gu, wu) and cases.same(gv, wv)
    a = np.array([[[0., 10., 350., 359.5, -5., 365.]]], dtype=np.float32)
    mm = np.zeros(24)
    mm[3::4] = np.radians(20.0)
    ga = oracle.vector_reproject_direction(mm, a, 6, 1)[0, 0]
    np.testing.assert_allclose(ga, [340., 350., 330., 339.5, 335., 345.], rtol=1e-6)


def _fill2d_py(f, relaxCrit, corrEff, maxLoop):
    """src/interpolation.c:1246-1376 in plain Python (float32 / float64 scalars as in the C code)."""
    F, D = np.float32, np.float64
    f = f.copy()
    ny, nx = f.shape
    nan = np.isnan(f)
    nUndef = int(nan.sum())
    nDef = f.size - nUndef
    if nDef == 0 or nUndef == 0:
        return f
    s = D(0)
    for v in f.ravel():
        if not np.isnan(v):
            s = s + D(v)
    avg = s / D(nDef)
    dev = D(0)
    for v in f.ravel():
        if not np.isnan(v):
            dev = dev + abs(D(v) - avg)
    dev = dev / D(nDef)
    crit = D(F(relaxCrit)) * dev
    w = np.where(nan, F(1), F(0)).astype(F)
    f[nan] = F(avg)
    w[1:ny - 1, 1:nx - 1] *= F(corrEff)
    e = np.zeros_like(f)
    for n in range(maxLoop):
        for y in range(1, ny - 1):
            for x in range(1, nx - 1):
                ssum = F(F(F(f[y, x + 1] + f[y, x - 1]) + f[y + 1, x]) + f[y - 1, x])
                e[y, x] = F(D(ssum) * D(0.25) - D(f[y, x]))
                f[y, x] = F(f[y, x] + F(e[y, x] * w[y, x]))
        if n < ((maxLoop - 5) % 2 ** 64) and n % 10 == 0:
            crtest = F(crit * D(F(corrEff)))
            if not np.any(np.abs((e * w)[1:ny - 1, 1:nx - 1]) > crtest):
                return f
        for y in range(1, ny - 1):
            f[y, 0] = F(f[y, 0] + F(F(f[y, 1] - f[y, 0]) * w[y, 0]))
            f[y, nx - 1] = F(f[y, nx - 1] + F(F(f[y, nx - 2] - f[y, nx - 1]) * w[y, nx - 1]))
        for x in range(nx):
            f[0, x] = F(f[0, x] + F(F(f[1, x] - f[0, x]) * w[0, x]))
            f[ny - 1, x] = F(f[ny - 1, x] + F(F(f[ny - 2, x] - f[ny - 1, x]) * w[ny - 1, x]))
    return f


@pytest.mark.parametrize("params", [(4.0, 1.6, 25), (0.5, 1.0, 7), (4.0, 1.9, 3)])
def test_fill2d_against_python(params):
    f = cases.holes(1, 12, 15, seed=3)[0]
    got, n, rc = oracle.fill2d(f, *params)
    assert rc == oracle.OK and n == int(np.isnan(f).sum())
    with np.errstate(all="ignore"):
        want = _fill2d_py(f, *params)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert not np.isnan(got).any()


def test_fills_basic_properties():
    f = cases.holes(1, 40, 50, seed=9)[0]
    defined = ~np.isnan(f)
    for got in (oracle.fill2d(f, 4.0, 1.6, 100)[0], oracle.creepfill2d(f, 20, 2)[0], oracle.creepfillval2d(f, 250.0, 5, 2)[0]):
        assert not np.isnan(got).any()
        assert cases.same(got[defined], f[defined])  # defined cells keep their values
        assert got.min() >= np.nanmin(f) - 40 and got.max() <= np.nanmax(f) + 40
    # nothing to do: no NaN, or only NaN
    full = cases.field(1, 10, 10, seed=1, nan_frac=0, extremes=False)[0]
    assert cases.same(oracle.fill2d(full, 4.0, 1.6, 100)[0], full)
    allnan = np.full((10, 10), np.nan, np.float32)
    assert np.isnan(oracle.creepfill2d(allnan, 20, 2)[0]).all()


def test_reduced_domain():
    inX, inY, outX, outY = 200, 150, 20, 10
    rng = np.random.default_rng(1)
    px = rng.uniform(60.3, 90.7, outX * outY)
    py = rng.uniform(40.2, 55.9, outX * outY)
    rd = oracle.create_reduced_domain(px, py, inX, inY)
    assert rd["xMin"] == math.floor(px.min()) - 2 and rd["yMin"] == math.floor(py.min()) - 2
    assert rd["inX"] == math.ceil(px.max()) + 2 - rd["xMin"] + 1
    f = cases.field(2, inY, inX, seed=5)
    crop = f[:, rd["yMin"]:rd["yMin"] + rd["inY"], rd["xMin"]:rd["xMin"] + rd["inX"]]
    for m in (oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC):
        a = oracle.interpolate_values(m, px, py, f, inX, inY, outX, outY)
        b = oracle.interpolate_values(m, rd["px"], rd["py"], crop, rd["inX"], rd["inY"], outX, outY)
        assert cases.same(a, b)


def test_points2position_longitude_wrap():
    axis = np.radians(np.arange(-180, 180, 1.0))  # circular: next value would be 180 = -180
    p = np.radians([179.5, 179.9, -180.0, 190.0, -170.0])
    got = oracle.points2position(p, axis, oracle.LONGITUDE)
    np.testing.assert_allclose(got, [359.5, -0.1, 0.0, 10.0, 10.0], atol=1e-9)
    assert oracle.points2position([np.nan, np.inf], axis, oracle.LONGITUDE).tolist() == [-999.0, -999.0]


# ---------------------------------------------------------------- typed slice edges
def _typed_samples(dt, rng, n=4000):
    info = np.iinfo(dt) if np.issubdtype(dt, np.integer) else None
    if info is not None:
        a = rng.integers(max(info.min, -2 ** 62), min(info.max, 2 ** 62), n, dtype=np.int64 if info.min < 0 else np.uint64).astype(dt)
        a[:4] = [info.min, info.max, 0, info.max // 2]
    else:
        a = (rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 12, n)).astype(dt)
        a[:4] = [0.0, -0.0, np.nan, np.inf]
    return a


@pytest.mark.parametrize("code", sorted(oracle.CDM_DTYPES))
def test_data2interpolation_array_is_cast_then_bad2nan(code):
    """src/CDMInterpolator.cc:115-119 against numpy: astype(float32) is static_cast<float> (round to nearest even)."""
    dt = oracle.CDM_DTYPES[code]
    rng = np.random.default_rng(code)
    a = _typed_samples(dt, rng)
    bad = float(a[7])
    got = oracle.data2interpolation_array(a, bad)
    with np.errstate(over="ignore", invalid="ignore"):
        want = a.astype(np.float32)
        want[want == np.float32(bad)] = np.nan
    assert cases.same(got, want)
    assert np.isnan(got[7])
    assert cases.same(oracle.data2interpolation_array(a, float("nan")), a.astype(np.float32))  # NaN fill value: untouched


@pytest.mark.parametrize("code", sorted(oracle.CDM_DTYPES))
def test_interpolation_array2data_rounds_half_away_and_restores_fill(code):
    """src/CDMInterpolator.cc:121-124 -> ScaleValue<float, OUT> (include/fimex/Utils.h:444-464)."""
    dt = oracle.CDM_DTYPES[code]
    f = np.array([1.5, 2.5, -0.5, -1.5, -2.5, 0.49999997, np.nan, -0.0, 100.4, 126.5, -128.5], np.float32)
    got = oracle.interpolation_array2data(f, code, 42.0)
    assert got.dtype == dt and got[6] == dt(42)
    if np.issubdtype(dt, np.integer):
        want = np.array([2, 3, -1, -2, -3, 0, 42, 0, 100, 127, -129], np.int64)  # lround: half away from zero
        assert np.array_equal(got, want.astype(np.int32).astype(dt))             # through int, then the narrowing cast wraps
    else:
        keep = ~np.isnan(f)
        assert np.array_equal(got[keep], f[keep].astype(dt))
        assert not np.signbit(got[7])  # 1.0 * (-0.0) + 0.0 = +0.0
    # int64 / uint64 also pass through MetNoFimex::round's int: 3e9 wraps
    if code == oracle.CDM_INT64:
        assert oracle.interpolation_array2data(np.array([3e9], np.float32), code, 0.0)[0] == np.int64(3000000000 - 2 ** 32)


def test_typed_round_trip_of_packed_shorts():
    """short data with fill -32767 -> float/NaN -> short: identity (the common packed-variable case)."""
    rng = np.random.default_rng(5)
    a = rng.integers(-32000, 32000, 10000).astype(np.int16)
    a[rng.choice(a.size, 300, replace=False)] = -32767
    f = oracle.data2interpolation_array(a, -32767.0)
    assert np.isnan(f).sum() == (a == -32767).sum()
    assert np.array_equal(oracle.interpolation_array2data(f, oracle.CDM_SHORT, -32767.0), a)


# ---------------------------------------------------------------- coordinate-based nearest neighbour plans
def _curvilinear_grid(nx, ny, seed, jitter=0.3, nan=0):
    """lon / lat (rad) of a rotated, slightly irregular grid, [ny][nx]."""
    rng = np.random.default_rng(seed)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny))
    u = (i + jitter * rng.uniform(-1, 1, i.shape)) * 0.05
    v = (j + jitter * rng.uniform(-1, 1, i.shape)) * 0.04
    lon = np.radians(5 + 0.9 * u - 0.3 * v)
    lat = np.radians(58 + 0.35 * u + 0.8 * v)
    if nan:
        k = rng.choice(lon.size, nan, replace=False)
        lon.reshape(-1)[k] = np.nan
    return lon, lat


def _great_circle_cos(lon0, lat0, lon1, lat1):
    return np.cos(lat1) * np.cos(lat0) * np.cos(lon1 - lon0) + np.sin(lat1) * np.sin(lat0)


def test_fast_translate_finds_the_closest_cell_within_the_region_of_influence():
    lon, lat = _curvilinear_grid(60, 45, seed=1, nan=5)
    rng = np.random.default_rng(2)
    qlon = np.radians(rng.uniform(4.5, 8.5, 3000)); qlat = np.radians(rng.uniform(57.5, 61.5, 3000))
    px, py = oracle.fast_translate_points(qlon, qlat, lon, lat)
    roi = oracle.grid_distance(lon, lat)
    assert 0 < roi < np.radians(0.5)
    cosd = _great_circle_cos(qlon[:, None], qlat[:, None], lon.ravel()[None], lat.ravel()[None])
    cosd = np.where(np.isnan(cosd), -2, cosd)
    best = cosd.argmax(axis=1)
    inside = cosd.max(axis=1) > np.cos(roi)
    assert inside.any() and (~inside).any()
    assert np.all(px[~inside] == -1) and np.all(py[~inside] == -1)
    agree = (px[inside] == best[inside] % 60) & (py[inside] == best[inside] // 60)
    assert agree.mean() > 0.999  # numpy's cos differs from libm's in the last bit at exact near-ties only


def test_flann_translate_is_the_closest_cell_within_max_dist():
    lon, lat = _curvilinear_grid(50, 40, seed=3, nan=4)
    rng = np.random.default_rng(4)
    qlon = np.radians(rng.uniform(4.5, 8.0, 2000)); qlat = np.radians(rng.uniform(57.5, 60.5, 2000))
    maxDist = 4000.0
    px, py = oracle.flann_translate_points(maxDist, qlon, qlat, lon, lat)
    def xyz(lo, la):
        return np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], -1)
    d2 = ((xyz(qlon, qlat)[:, None, :] - xyz(lon.ravel(), lat.ravel())[None]) ** 2).sum(-1)
    d2 = np.where(np.isnan(d2), 9, d2)
    inside = d2.min(axis=1) < (maxDist / 6371000.) ** 2
    best = d2.argmin(axis=1)
    assert inside.any() and (~inside).any()
    assert np.all(px[~inside] == -1000)
    assert ((px[inside] == best[inside] % 50) & (py[inside] == best[inside] // 50)).mean() > 0.999


def _round_fraction_to_f32(q):
    """Correctly rounded (nearest, ties to even) float32 of an exact rational."""
    from fractions import Fraction
    if q == 0:
        return np.float32(0.0)
    sign = -1 if q < 0 else 1
    q = abs(q)
    e = q.numerator.bit_length() - q.denominator.bit_length()
    if Fraction(2) ** e > q:
        e -= 1
    e = max(e, -126)                                  # subnormals share the exponent of the smallest normal
    scaled = q / Fraction(2) ** (e - 23)              # integer part = 24-bit significand
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return np.float32(sign * float(n) * 2.0 ** (e - 23))   # n < 2^25 and the power of two: exact in double, then exact in float32


def test_sor_error_in_double_equals_one_fused_operation():
    """interpolation.c:1332 computes e = (float)((double)sum * 0.25 - (double)f).  The device takes fmaf(sum, 0.25f, -f): the
    exactly rounded value of sum / 4 - f.  The two agree for every pair of floats (fill.hip, sor_error); here the C
    expression, as numpy evaluates it, is compared with the exact rational result rounded once -- over random pairs at every
    exponent gap from -45 to +45 binades, near-cancellations, halfway cases and subnormal results."""
    from fractions import Fraction
    rng = np.random.default_rng(1332)
    pairs = []
    for gap in range(-45, 46):
        for _ in range(40):
            s = np.float32(rng.uniform(1, 2) * 2.0 ** int(rng.integers(-20, 20)) * rng.choice([-1, 1]))
            f = np.float32(rng.uniform(1, 2) * 2.0 ** (np.log2(abs(float(s))) // 1 - 2 + gap) * rng.choice([-1, 1]))
            pairs.append((s, f))
    for _ in range(2000):                              # sum / 4 next to f: cancellation down to single ulps
        f = np.float32(rng.normal(280, 20))
        s = np.float32(4 * float(f)) + np.float32(rng.integers(-8, 9)) * np.spacing(np.float32(4 * float(f)))
        pairs.append((np.float32(s), f))
    for _ in range(500):                               # results in the subnormal range
        f = np.float32(rng.uniform(1, 2) * 2.0 ** -130)
        pairs.append((np.float32(rng.uniform(1, 2) * 2.0 ** -128), f))
    for s, f in pairs:
        with np.errstate(all="ignore"):
            c_expression = np.float32(np.float64(s) * 0.25 - np.float64(f))
        exact = _round_fraction_to_f32(Fraction(float(s)) / 4 - Fraction(float(f)))
        assert c_expression.tobytes() == exact.tobytes() or (c_expression == 0 and exact == 0), (s, f, c_expression, exact)
