"""Parity of the HIP path (through the C ABI, include/fimex_amd.h) with the CPU oracle on the same
seeded inputs.  Bar (BASELINE.json north_star): bit-exact for nearest / index methods, <= 1e-5
relative for bilinear / bicubic / rotation.  The kernels are built with -ffp-contract=off and
follow the reference's operation order, so every comparison below is in fact bit-exact on defined
values with identical NaN positions; the 1e-5 bound is asserted as well where north_star names it.
"""
import functools

import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    from fimex_amd import capi
    capi.load()
    assert capi.device_count() >= 1, "no gfx950 device visible"
    return capi


def _rel_ok(got, want, tol=1e-5):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    ok = np.isfinite(want)
    return np.all(np.abs(got[ok] - want[ok]) <= tol * np.abs(want[ok]) + 1e-30)


BACKWARD = [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC, oracle.COORD_NN, oracle.COORD_NN_KD]


@pytest.mark.parametrize("method", BACKWARD)
@pytest.mark.parametrize("shape", [(37, 29, 41, 33, 3), (300, 200, 157, 211, 11), (64, 64, 256, 4, 1), (5, 4, 7, 6, 19)])
def test_backward_methods_match_oracle(fa, method, shape):
    inX, inY, outX, outY, nz = shape
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=100 + method)
    f = cases.field(nz, inY, inX, seed=7 + method)
    want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    got = plan.apply_host(f)
    assert got.shape == want.shape
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert _rel_ok(got, want)
    info = plan.info()
    assert info["undefinedCells"] > 0  # the case overshoots the source on purpose
    if method == oracle.BILINEAR:
        assert info["borderCells"] > 0


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
@pytest.mark.parametrize("nz", [1, 2, 7, 8, 9, 16, 17, 41, 83])
def test_backward_ragged_z_counts(fa, method, nz):
    """z counts around the in-kernel unroll factors and the z-chunking of the grid."""
    inX, inY, outX, outY = 120, 90, 100, 70
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=5)
    f = cases.field(nz, inY, inX, seed=nz)
    want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    got = fa.RegridPlan(method, px, py, inX, inY, outX, outY).apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
@pytest.mark.parametrize("shape", [(400, 300, 200, 200, 10), (1000, 700, 333, 257, 7), (128, 96, 640, 480, 3), (64, 2000, 50, 300, 5)])
@pytest.mark.parametrize("knobs", [{"STAGED": "0"}, {"STAGED": "1"}, {"STAGED": "1", "STAGE_TW": "64"},
                                   {"STAGED": "1", "STAGE_TW": "32", "STAGE_ZPB": "3"},
                                   {"STAGED": "1", "STAGE_TW": "256", "STAGE_ZPB": "1", "XCD": "1"},
                                   {"STAGED": "1", "XCD": "3", "STAGED_MIN_NZ": "1"}])
def test_gather_and_lds_staged_paths_agree_with_oracle(fa, monkeypatch, method, shape, knobs, tuning_build):
    """Both kernels of nearest, bilinear and bicubic (per-lane gather, LDS-staged tiles) on source widths that allow staging
    (inX % 4 == 0): shrinking (several source cells per target cell), magnifying, and strongly anisotropic geometries,
    different tile shapes, z chunkings and tile orders."""
    inX, inY, outX, outY, nz = shape
    for k, v in knobs.items():
        monkeypatch.setenv("FIMEX_AMD_" + k, v)
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=inX + outX)
    f = cases.field(nz, inY, inX, seed=3)
    want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    got = fa.RegridPlan(method, px, py, inX, inY, outX, outY).apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
@pytest.mark.parametrize("outY", [5, 24, 40, 63])
def test_wide_short_targets_use_every_xcd(fa, method, outY):
    """Fewer tile rows than XCDs (a cross-section: outY < 8 tiles of 8 rows): the staged plan deals the tiles themselves over the
    XCDs instead of putting all of them on one; results against the oracle, and the gather kernels agree on the device."""
    import torch
    inX, inY, outX, nz = 1600, 48, 2304, 9
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=outY, special=False)
    f = cases.field(nz, inY, inX, seed=outY + 1)
    want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    assert plan.info()["stagedCells"] > 0
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    d_in = torch.from_numpy(f).cuda()
    a = torch.empty((nz, outY, outX), dtype=torch.float32, device="cuda")
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    plan.apply_device(d_in.data_ptr(), nz, a.data_ptr(), st)
    plan.apply_gather_device(d_in.data_ptr(), nz, b.data_ptr(), st)
    torch.cuda.synchronize()
    assert cases.same(a.cpu().numpy(), want) and cases.same(b.cpu().numpy(), want)


def test_bilinear_scattered_positions_fall_back_to_gather(fa):
    """Positions without spatial coherence (every tile would need the whole source): the staged plan does not fit
    its LDS budget and the plan silently keeps the gather kernel; results are the same."""
    inX, inY, outX, outY, nz = 512, 384, 200, 100, 4
    rng = np.random.default_rng(5)
    px = rng.uniform(-3, inX + 2, outX * outY)
    py = rng.uniform(-3, inY + 2, outX * outY)
    f = cases.field(nz, inY, inX, seed=8)
    for method in (oracle.BILINEAR, oracle.NEAREST):
        want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
        plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
        assert plan.info()["stagedCells"] == 0
        got = plan.apply_host(f)
        assert cases.same(got, want), cases.describe_mismatch(got, want)


def test_backward_empty_and_query(fa):
    px, py = cases.backward_positions(10, 8, 6, 5, seed=1, special=False)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, 10, 8, 6, 5)
    out = plan.apply_host(np.zeros((0, 8, 10), np.float32))
    assert out.shape == (0, 5, 6)
    # size not a multiple of the slice: trailing floats are ignored like CachedInterpolation.cc:121 (integer division)
    f = cases.field(2, 8, 10, seed=3)
    flat = np.concatenate([f.ravel(), np.ones(13, np.float32)])
    got = plan.apply_host(flat)
    assert cases.same(got, oracle.interpolate_values(oracle.BILINEAR, px, py, f, 10, 8, 6, 5))


def test_unknown_method_fails_like_the_reference(fa):
    px, py = cases.backward_positions(10, 8, 6, 5, seed=1, special=False)
    with pytest.raises(fa.FimexAmdError, match="unknown interpolation function"):
        fa.RegridPlan(99, px, py, 10, 8, 6, 5)
    with pytest.raises(fa.FimexAmdError):
        fa.RegridPlan(oracle.BILINEAR, px[:-1], py[:-1], 10, 8, 6, 5)


def test_reference_kats_on_gpu(fa):
    """test/testInterpolation.cc:73-155 replayed through the HIP path."""
    g = lambda m, f, x, y, ix, iy: fa.RegridPlan(m, [x], [y], ix, iy, 1, 1).apply_host(np.asarray(f, np.float32))[0, 0, 0]
    assert g(oracle.NEAREST, [1., 2., 1., 2.], 0.3, 0.3, 2, 2) == 1
    f = np.array([1., 2., 2., 1 + np.sqrt(np.float32(2.0))], dtype=np.float32)
    assert abs(g(oracle.BILINEAR, f, 0.3, 0., 2, 2) - 1.3) < 1e-6
    assert abs(g(oracle.BILINEAR, f, 0., 0.3, 2, 2) - 1.3) < 1e-6
    assert not np.isnan(g(oracle.BILINEAR, f, 1, 1, 2, 2))
    for x, y in [(1.5, 0.5), (0.5, 1.5), (0.5, -0.5), (-0.5, 0.5)]:
        assert np.isnan(g(oracle.BILINEAR, f, x, y, 2, 2))
    c = np.array([1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 1, 1, 1, 1], dtype=np.float32)
    assert abs(g(oracle.BICUBIC, c, 1, 1.5, 4, 4) - 2.125) < 2.125e-5
    assert abs(g(oracle.BICUBIC, c, 1.5, 1, 4, 4) - 2.0) < 2e-5
    for x, y in [(.5, 1), (1, .5), (2.5, 1), (1, 2.5)]:
        assert np.isnan(g(oracle.BICUBIC, c, x, y, 4, 4))


FORWARD = list(range(oracle.FWD_SUM, oracle.FWD_UNDEF_MIN + 1))


@pytest.mark.parametrize("method", FORWARD)
@pytest.mark.parametrize("density", [0.3, 1.0, 9.0])
def test_forward_methods_match_oracle(fa, method, density):
    inX, inY, outX, outY, nz = 140, 110, 60, 50, 5
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=11, density=density)
    f = cases.field(nz, inY, inX, seed=method, nan_frac=0.05)
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    info = plan.info()
    assert info["mappedSourceCells"] > 0 and info["maxBucket"] >= 1


@pytest.mark.parametrize("method", [oracle.FWD_SUM, oracle.FWD_MEAN, oracle.FWD_MAX, oracle.FWD_MIN,
                                    oracle.FWD_UNDEF_SUM, oracle.FWD_UNDEF_MEAN, oracle.FWD_UNDEF_MAX, oracle.FWD_UNDEF_MIN])
def test_forward_long_buckets_wave_path(fa, method):
    """~130 source cells per target cell: the wave-per-bucket reduction, still in the reference's add order."""
    inX, inY, outX, outY, nz = 400, 300, 36, 26, 3
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=21, density=130.0)
    f = cases.field(nz, inY, inX, seed=90 + method, nan_frac=0.03, extremes=False)
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    assert plan.info()["maxBucket"] > 64
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("shape", [(64, 48, 4), (101, 37, 3), (7, 5, 2), (2000, 3, 1)])
def test_vector_rotation_matches_oracle(fa, shape):
    ox, oy, oz = shape
    m = cases.rotation_matrix(ox, oy, seed=ox)
    u = cases.field(oz, oy, ox, seed=1)
    v = cases.field(oz, oy, ox, seed=2) - 280
    wu, wv = oracle.vector_reproject_values(m, u, v, ox, oy)
    plan = fa.VectorPlan(m, ox, oy)
    gu, gv = plan.reproject_values_host(u, v)
    assert cases.same(gu, wu), cases.describe_mismatch(gu, wu)
    assert cases.same(gv, wv), cases.describe_mismatch(gv, wv)
    assert _rel_ok(gu, wu) and _rel_ok(gv, wv)
    # length is preserved by the rotation (test/testInterpolation.cc:575-578)
    ok = np.isfinite(u) & np.isfinite(v) & (np.abs(u) < 1e18) & (np.abs(v) < 1e18)
    l0 = u[ok].astype(np.float64) ** 2 + v[ok].astype(np.float64) ** 2
    l1 = gu[ok].astype(np.float64) ** 2 + gv[ok].astype(np.float64) ** 2
    assert np.all(np.abs(l1 - l0) <= 1e-5 * l0 + 1e-3)
    ang = (np.random.default_rng(3).uniform(-30, 400, (oz, oy, ox))).astype(np.float32)
    wa = oracle.vector_reproject_direction(m, ang, ox, oy)
    ga = plan.reproject_direction_host(ang)
    assert cases.same(ga, wa), cases.describe_mismatch(ga, wa)


def test_vector_rotation_90_degrees(fa):
    """test/testInterpolation.cc:396-453: a quarter turn maps u -> -v, v -> u."""
    n = 25
    phi = np.full(n, np.pi / 2)
    m = np.stack([np.cos(phi), np.sin(phi), -np.sin(phi), phi], axis=1).ravel()
    u = np.arange(n, dtype=np.float32)
    v = (25 - np.arange(n)).astype(np.float32)
    gu, gv = fa.VectorPlan(m, 5, 5).reproject_values_host(u, v)
    assert np.all(np.abs(gu + v) < 1e-4) and np.all(np.abs(gv - u) < 1e-4)


@pytest.mark.parametrize("shape", [(40, 30, 3), (97, 61, 2), (2, 2, 1), (3, 17, 1), (130, 5, 2), (4, 4, 1), (5, 70, 1), (300, 200, 2),
                                   (64, 66, 1), (33, 1200, 1), (1000, 131, 1), (5000, 1100, 1)])
@pytest.mark.parametrize("params", [(4.0, 1.6, 100), (0.5, 1.0, 23), (4.0, 1.9, 3), (1e-9, 1.6, 41)])
@pytest.mark.parametrize("geometry", ["1", "2"], ids=["16waves_x_16columns", "8waves_x_32columns"])
def test_fill2d_matches_oracle(fa, monkeypatch, shape, params, geometry, tuning_build):
    """Both band geometries of the systolic kernel (the launcher picks by batch size) against the CPU restatement."""
    nx, ny, nz = shape
    relaxCrit, corrEff, maxLoop = params
    monkeypatch.setenv("FIMEX_AMD_FILL_GEOMETRY", geometry)
    f = cases.holes(nz, ny, nx, seed=nx * 31 + ny)
    got, nch = fa.fill2d_host(f, relaxCrit, corrEff, maxLoop)
    for z in range(nz):
        want, wn, rc = _oracle_fill2d(nx, ny, nz, z, relaxCrit, corrEff, maxLoop)
        assert rc == oracle.OK
        assert nch[z] == wn
        assert cases.same(got[z], want), "slice %d: %s" % (z, cases.describe_mismatch(got[z], want))


@functools.lru_cache(maxsize=None)
def _oracle_fill2d(nx, ny, nz, z, relaxCrit, corrEff, maxLoop):
    return oracle.fill2d(cases.holes(nz, ny, nx, seed=nx * 31 + ny)[z], relaxCrit, corrEff, maxLoop)


def test_fill2d_geometry_follows_the_batch_size(fa, monkeypatch, tuning_build):
    """Eight slices and more run the 8-wave, 32-column geometry by default; the result does not depend on it."""
    f = np.stack([cases.holes(1, 90, 140, seed=5)[0]] * 50)
    want, wn, rc = oracle.fill2d(f[0], 4.0, 1.6, 40)
    got, nch = fa.fill2d_host(f, 4.0, 1.6, 40)
    assert all(n == wn for n in nch) and all(cases.same(got[z], want) for z in range(50))
    monkeypatch.setenv("FIMEX_AMD_FILL_GEOMETRY", "1")
    again, _ = fa.fill2d_host(f, 4.0, 1.6, 40)
    assert cases.same(again, got)


@pytest.mark.parametrize("shape", [(40, 30, 3), (97, 61, 2), (2, 2, 1), (3, 17, 1), (130, 5, 2), (4, 4, 1), (5, 70, 1), (300, 200, 2),
                                   (64, 66, 1), (33, 1200, 1), (1000, 131, 1), (5000, 1100, 1)])
@pytest.mark.parametrize("params", [(20, 2), (1, 1), (5, 2), (3, 0), (2, 7)])
def test_creepfill_matches_oracle(fa, shape, params):
    nx, ny, nz = shape
    repeat, weight = params
    f = cases.holes(nz, ny, nx, seed=nx * 17 + ny)
    got, nch = fa.creepfill2d_host(f, repeat, weight)
    gotv, nchv = fa.creepfillval2d_host(f, 271.25, repeat, weight)
    for z in range(nz):
        want, wn, rc = oracle.creepfill2d(f[z], repeat, weight)
        assert rc == oracle.OK and nch[z] == wn
        assert cases.same(got[z], want), "slice %d: %s" % (z, cases.describe_mismatch(got[z], want))
        wantv, wnv, rc = oracle.creepfillval2d(f[z], 271.25, repeat, weight)
        assert rc == oracle.OK and nchv[z] == wnv
        assert cases.same(gotv[z], wantv), "slice %d: %s" % (z, cases.describe_mismatch(gotv[z], wantv))


@pytest.mark.parametrize("algo", ["0", "1", "2"], ids=["add_chain", "one_workgroup", "whole_chip"])
def test_fills_with_each_evaluation_of_the_sums(fa, monkeypatch, algo, tuning_build):
    """The first-guess sums (mean, mean absolute deviation) as an add chain, binade-parallel in one workgroup per slice
    (batches of 100 slices and more) and binade-parallel over the whole chip (smaller batches): the same bits."""
    monkeypatch.setenv("FIMEX_AMD_SUM_ALGO", algo)
    f = cases.holes(3, 170, 260, seed=91)
    got, nch = fa.fill2d_host(f, 4.0, 1.6, 30)
    gotc, nchc = fa.creepfill2d_host(f, 3, 2)
    gotv, _ = fa.creepfillval2d_host(f, 271.25, 3, 2)
    for z in range(3):
        want, wn, _ = oracle.fill2d(f[z], 4.0, 1.6, 30)
        assert nch[z] == wn and cases.same(got[z], want), cases.describe_mismatch(got[z], want)
        wantc, wnc, _ = oracle.creepfill2d(f[z], 3, 2)
        assert nchc[z] == wnc and cases.same(gotc[z], wantc)
        wantv, _, _ = oracle.creepfillval2d(f[z], 271.25, 3, 2)
        assert cases.same(gotv[z], wantv)


def test_stored_type_regrid_on_random_shapes(fa):
    """The LDS-staged kernels on 1- and 2-byte elements over random grid sizes (row lengths that are and are not multiples
    of the 16-byte chunk of each element size), slice counts around the staging threshold, and an input that does not
    start on a 4-byte boundary (gather form): all equal the oracle's three steps."""
    import torch
    rng = np.random.default_rng(77)
    for case in range(24):
        dt = [np.int16, np.uint16, np.int8, np.uint8][case % 4]
        method = [oracle.BILINEAR, oracle.NEAREST, oracle.BICUBIC][case % 3]
        inX = int(rng.integers(3, 60)) * 4 if case % 6 else int(rng.integers(9, 200))
        inY, outX, outY, nz = int(rng.integers(5, 140)), int(rng.integers(1, 260)), int(rng.integers(1, 180)), int(rng.integers(1, 12))
        px, py = cases.backward_positions(inX, inY, outX, outY, seed=case)
        info = np.iinfo(dt)
        bad = float(info.max if case % 2 else info.min)
        f = rng.integers(info.min, int(info.max) + 1, (nz, inY, inX)).astype(dt)
        f.reshape(-1)[rng.random(f.size) < 0.04] = dt(bad)
        code = oracle.cdm_type_of(dt)
        want = oracle.interpolation_array2data(
            oracle.interpolate_values(method, px, py, oracle.data2interpolation_array(f, bad), inX, inY, outX, outY), code, bad)
        plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
        shift = 2 if case in (5, 11) else 0   # an odd element offset: the slices no longer start on 4-byte boundaries
        raw = torch.zeros(f.nbytes + 8, dtype=torch.uint8, device="cuda")
        raw[shift:shift + f.nbytes] = torch.from_numpy(f.view(np.uint8).ravel()).cuda()
        out = torch.zeros(want.nbytes, dtype=torch.uint8, device="cuda")
        fa.regrid_apply_typed_device(plan, raw.data_ptr() + shift, code, nz, bad, out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(dt).reshape(want.shape)
        assert np.array_equal(got, want), (case, np.dtype(dt).name, method, inX, inY, outX, outY, nz, int((got != want).sum()))
        plan.close()


@pytest.mark.parametrize("method", [oracle.BILINEAR, oracle.NEAREST])
@pytest.mark.parametrize("dt", [np.int16, np.uint16, np.int8, np.uint8])
@pytest.mark.parametrize("outX,bad", [(212, "min"), (213, "max"), (212, None), (77, "max")])
def test_stored_types_second_staged_form_edges(fa, method, dt, outX, bad):
    """The second staged form on stored types (staged2.hip: staged_apply2_typed), where its special cases lie: even row lengths
    (two results per store) and odd ones (one per store, pairs that straddle rows), no fill value at all (NaN: nothing is ever
    undefined on input), fill values at either end of the type's range, an output buffer that is not 4-byte aligned, and data that
    use the type's whole range (rounding at the ends).  Against the oracle's three steps, element for element."""
    import torch
    inX, inY, outY, nz = 336, 97, 45, 9
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=outX)
    info = np.iinfo(dt)
    badv = float("nan") if bad is None else float(info.min if bad == "min" else info.max)
    rng = np.random.default_rng(outX + int(np.dtype(dt).itemsize))
    f = rng.integers(info.min, int(info.max) + 1, (nz, inY, inX)).astype(dt)
    if bad is not None:
        f.reshape(-1)[rng.random(f.size) < 0.03] = dt(badv)
    code = oracle.cdm_type_of(dt)
    fl = oracle.interpolate_values(method, px, py, oracle.data2interpolation_array(f, badv), inX, inY, outX, outY)
    want = oracle.interpolation_array2data(fl, code, badv)
    # without a fill value an undefined result becomes (T)NaN, which C leaves undefined: those cells are not compared
    defined = np.ones(fl.shape, bool) if bad is not None else ~np.isnan(fl)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    assert plan.info()["stagedCells"] > 0
    t = torch.from_numpy(f.view(np.uint8)).cuda()
    for shift in (0, np.dtype(dt).itemsize):  # the second run writes to an address that is not a multiple of 4
        out = torch.zeros(want.nbytes + 8, dtype=torch.uint8, device="cuda")
        fa.regrid_apply_typed_device(plan, t.data_ptr(), code, nz, badv, out.data_ptr() + shift)
        torch.cuda.synchronize()
        got = out.cpu().numpy()[shift:shift + want.nbytes].view(dt).reshape(want.shape)
        assert np.array_equal(got[defined], want[defined]), (np.dtype(dt).name, method, outX, bad, shift, int((got != want)[defined].sum()))


def test_sor_error_is_the_reference_expression(fa):
    """fill2d's step takes one fused multiply-add where the reference goes through double (fill.hip, sor_error): fields whose
    neighbours lie up to 60 binades apart, subnormals and values next to the float range's end put every regime of the two
    roundings through the kernel; the oracle computes the reference's expression as written."""
    rng = np.random.default_rng(1333)
    for case, (lo, hi) in enumerate(((-30, 30), (-60, 0), (0, 38), (-149, -110), (-5, 5))):
        ny, nx = 70, 150
        f = (rng.uniform(1, 2, (ny, nx)) * 2.0 ** rng.integers(lo, hi, (ny, nx)) * rng.choice([-1, 1], (ny, nx))).astype(np.float32)
        f[rng.random((ny, nx)) < 0.2] = np.nan
        f = f[None]
        for corr in (1.0, 1.6):
            with np.errstate(all="ignore"):
                want, wn, rc = oracle.fill2d(f[0], 1e-9, corr, 7)
            got, nch = fa.fill2d_host(f, 1e-9, corr, 7)
            assert rc == oracle.OK and nch[0] == wn and cases.same(got[0], want), (case, corr, cases.describe_mismatch(got[0], want))


def test_fills_on_random_shapes(fa, monkeypatch, tuning_build):
    """Forty slices of random size, hole pattern and parameters through both band geometries and all three fills:
    every width class of the systolic kernels (narrower than a chunk, one band, ragged last band, several hand-off
    windows) meets the CPU restatement."""
    rng = np.random.default_rng(20261004)
    for case in range(40):
        nx, ny = int(rng.integers(4, 900)), int(rng.integers(4, 400))
        if case % 5 == 0:
            nx, ny = ny, nx
        f = cases.holes(1, ny, nx, seed=1000 + case, frac=float(rng.choice([0.02, 0.3, 0.8])), blobs=int(rng.integers(1, 9)))
        if case == 7:
            f[:] = np.nan          # nothing defined
        if case == 8:
            f = np.nan_to_num(f)   # nothing to fill
        if case == 9:
            f[0, :, 0] = np.nan; f[0, 0, :] = np.nan; f[0, -1, :] = np.nan; f[0, :, -1] = np.nan  # the whole border
        relax, corr, loops = float(rng.choice([4.0, 0.3, 1e-9])), float(rng.choice([1.0, 1.6, 1.9])), int(rng.integers(1, 45))
        repeat, weight = int(rng.integers(1, 25)), int(rng.integers(0, 4))
        monkeypatch.setenv("FIMEX_AMD_FILL_GEOMETRY", "1" if case % 2 else "2")
        got, nch = fa.fill2d_host(f, relax, corr, loops)
        want, wn, rc = oracle.fill2d(f[0], relax, corr, loops)
        assert rc == oracle.OK and nch[0] == wn and cases.same(got[0], want), (case, nx, ny, relax, corr, loops, cases.describe_mismatch(got[0], want))
        gotc, nchc = fa.creepfill2d_host(f, repeat, weight)
        wantc, wnc, rc = oracle.creepfill2d(f[0], repeat, weight)
        assert rc == oracle.OK and nchc[0] == wnc and cases.same(gotc[0], wantc), (case, nx, ny, repeat, weight, cases.describe_mismatch(gotc[0], wantc))


def test_fill2d_both_kernels_agree(fa, monkeypatch, tuning_build):
    """The systolic row-band kernel and the anti-diagonal wavefront kernel are two implementations of the same order."""
    f = cases.holes(2, 150, 210, seed=77)
    monkeypatch.setenv("FIMEX_AMD_FILL_V2", "1")
    a, na = fa.fill2d_host(f, 4.0, 1.6, 60)
    monkeypatch.setenv("FIMEX_AMD_FILL_V2", "0")
    b, nb = fa.fill2d_host(f, 4.0, 1.6, 60)
    assert na == nb and cases.same(a, b)


def test_creepfill_both_kernels_agree(fa, monkeypatch, tuning_build):
    """Row-band kernel with mask generations against the wavefront kernel with per-cell counters; a hole that reaches the
    lower right corner creeps one cell per sweep, so the sweep count goes well beyond repeat."""
    f = cases.holes(2, 170, 230, seed=78)
    f[:, 100:, 150:] = np.nan
    f[0, :40, :50] = np.nan
    for repeat, weight in ((3, 2), (20, 1), (300, 2)):
        monkeypatch.setenv("FIMEX_AMD_CREEP_V2", "1")
        a, na = fa.creepfill2d_host(f, repeat, weight)
        monkeypatch.setenv("FIMEX_AMD_CREEP_V2", "0")
        b, nb = fa.creepfill2d_host(f, repeat, weight)
        assert na == nb and cases.same(a, b), cases.describe_mismatch(a, b)
    want, wn, rc = oracle.creepfill2d(f[0], 3, 2)
    monkeypatch.setenv("FIMEX_AMD_CREEP_V2", "1")
    assert cases.same(fa.creepfill2d_host(f[:1], 3, 2)[0][0], want)


def _patchy_field(nz, ny, nx, seed):
    """Mostly defined slices with a few undefined regions: what a regridded field looks like (a corner outside the source domain, masked
    areas), not scattered holes.  Regions: a wedge in the upper left corner that reaches both borders (filled from below and from the
    right: one row / column per sweep), a block in the middle, two blocks side by side in the same rows, two blocks with exactly one
    defined row between them, single cells, undefined cells ON the right border column and the bottom row."""
    rng = np.random.default_rng(seed)
    f = (280 + 10 * rng.standard_normal((nz, ny, nx))).astype(np.float32)
    yy, xx = np.mgrid[0:ny, 0:nx]
    f[:, (yy * 2 + xx) < ny // 3] = np.nan                                  # wedge at the upper left corner
    f[:, ny // 2: ny // 2 + 23, nx // 3: nx // 3 + 31] = np.nan             # a block
    f[:, ny // 2 + 40: ny // 2 + 52, 5:17] = np.nan                         # two blocks in the same rows ...
    f[:, ny // 2 + 41: ny // 2 + 50, nx - 30: nx - 11] = np.nan             # ... far apart in x
    f[:, ny - 30: ny - 25, 40:60] = np.nan                                  # two blocks, one defined row between them
    f[:, ny - 24: ny - 20, 45:80] = np.nan
    f[:, 70, 90] = np.nan
    f[:, ny // 2 + 5: ny // 2 + 9, nx - 1] = np.nan                         # on the right border
    f[:, ny - 1, nx // 2: nx // 2 + 6] = np.nan                             # on the bottom row
    return f


@pytest.mark.parametrize("shape", [(230, 170, 3), (301, 260, 2), (96, 400, 2)])
@pytest.mark.parametrize("params", [(20, 2), (3, 1), (1, 0), (40, 2)])
def test_creepfill_by_rectangles(fa, monkeypatch, shape, params, tuning_build):
    """Fields that are defined in most rows and columns are cut along those into rectangles that are filled on their own (rows and
    columns defined throughout never change and weigh the same as borders or as interior cells).  Same bits as the oracle and as
    the sweeps over the whole field, for creepfill2d and creepfillval2d; slices with different masks, a slice without undefined
    cells and one without defined cells in the batch."""
    nx, ny, nz = shape
    repeat, weight = params
    f = _patchy_field(nz + 3, ny, nx, seed=nx + ny + repeat)
    f[1, 20:30, 100:140 if nx > 140 else 50:90] = np.nan   # slice 1: one region more than its neighbours
    f[nz] = 281.0                                           # nothing undefined
    f[nz + 1] = np.nan                                      # nothing defined
    monkeypatch.setenv("FIMEX_AMD_CREEP_RECTS", "2")   # 2: fails instead of falling back to the whole field
    got, nch = fa.creepfill2d_host(f, repeat, weight)
    gotv, nchv = fa.creepfillval2d_host(f, 271.25, repeat, weight)
    monkeypatch.setenv("FIMEX_AMD_CREEP_RECTS", "0")
    whole, nchw = fa.creepfill2d_host(f, repeat, weight)
    assert list(nch) == list(nchw)
    assert cases.same(got, whole), cases.describe_mismatch(got, whole)
    for z in range(f.shape[0]):
        want, wn, rc = oracle.creepfill2d(f[z], repeat, weight)
        assert rc == oracle.OK and nch[z] == wn
        assert cases.same(got[z], want), "slice %d: %s" % (z, cases.describe_mismatch(got[z], want))
        wantv, wnv, rc = oracle.creepfillval2d(f[z], 271.25, repeat, weight)
        assert rc == oracle.OK and nchv[z] == wnv
        assert cases.same(gotv[z], wantv), "slice %d: %s" % (z, cases.describe_mismatch(gotv[z], wantv))


@pytest.mark.parametrize("shape", [(230, 170, 3), (301, 260, 2), (96, 400, 2)])
@pytest.mark.parametrize("params", [(1e-9, 1.6, 40), (4.0, 1.6, 100), (0.05, 1.0, 23), (0.5, 1.9, 7)])
def test_fill2d_by_rectangles(fa, monkeypatch, shape, params, tuning_build):
    """fill2d on fields that are defined in most rows and columns: the rectangles of a slice, padded to one size, sweep as slices of
    one launch and end together by the criterion over all of them (regions that converge at different checks among them; criteria
    that are met after ten or twenty sweeps, or never; maxLoop values on either side of the "no check in the last five sweeps"
    rule).  Same bits as the oracle and as the sweeps over the whole field; a -0.0 among the defined cells sends the call over
    the whole field (the reference's sweep turns it into +0.0)."""
    nx, ny, nz = shape
    crit, cor, loops = params
    f = _patchy_field(nz + 2, ny, nx, seed=nx + ny + loops)
    f[1, 20:30, 100:140 if nx > 140 else 50:90] = np.nan
    f[nz] = 281.0                                           # nothing undefined
    f[nz + 1] = np.nan                                      # nothing defined
    # 2: fails instead of falling back to the whole field (the narrow shape's rectangles, padded to one size, cover too much of it: 1)
    monkeypatch.setenv("FIMEX_AMD_FILL_RECTS", "2" if nx > 100 else "1")
    got, nch = fa.fill2d_host(f, crit, cor, loops)
    monkeypatch.setenv("FIMEX_AMD_FILL_RECTS", "0")
    whole, nchw = fa.fill2d_host(f, crit, cor, loops)
    assert list(nch) == list(nchw)
    assert cases.same(got, whole), cases.describe_mismatch(got, whole)
    for z in range(f.shape[0]):
        want, wn, rc = oracle.fill2d(f[z], crit, cor, loops)
        assert rc == oracle.OK and nch[z] == wn
        assert cases.same(got[z], want), "slice %d: %s" % (z, cases.describe_mismatch(got[z], want))
    g = f.copy()
    g[0, 3, 200 if nx > 200 else 60] = np.float32(-0.0)
    monkeypatch.setenv("FIMEX_AMD_FILL_RECTS", "1")
    got, _ = fa.fill2d_host(g, crit, cor, loops)
    want, _, _ = oracle.fill2d(g[0], crit, cor, loops)
    assert cases.same(got[0], want) and np.array_equal(np.signbit(got[0]), np.signbit(want))


def test_fills_by_rectangles_long_batch(fa, monkeypatch, tuning_build):
    """More boxes than the chip holds workgroups: the coupled fill2d launches go in runs of slices, the creep fill's rectangles fall
    back to one workgroup per box.  Against the sweeps over the whole field for every slice, against the oracle for a few."""
    nx, ny, nz = 301, 260, 90
    f = _patchy_field(nz, ny, nx, seed=7)
    monkeypatch.setenv("FIMEX_AMD_FILL_RECTS", "2")
    monkeypatch.setenv("FIMEX_AMD_CREEP_RECTS", "2")
    got, nch = fa.fill2d_host(f, 0.3, 1.6, 60)
    gotc, nchc = fa.creepfill2d_host(f, 20, 2)
    monkeypatch.setenv("FIMEX_AMD_FILL_RECTS", "0")
    monkeypatch.setenv("FIMEX_AMD_CREEP_RECTS", "0")
    whole, nchw = fa.fill2d_host(f, 0.3, 1.6, 60)
    wholec, nchcw = fa.creepfill2d_host(f, 20, 2)
    assert list(nch) == list(nchw) and list(nchc) == list(nchcw)
    assert cases.same(got, whole), cases.describe_mismatch(got, whole)
    assert cases.same(gotc, wholec), cases.describe_mismatch(gotc, wholec)
    for z in (0, 41, 89):
        want, wn, rc = oracle.fill2d(f[z], 0.3, 1.6, 60)
        assert rc == oracle.OK and nch[z] == wn and cases.same(got[z], want)
        wantc, wnc, rc = oracle.creepfill2d(f[z], 20, 2)
        assert rc == oracle.OK and nchc[z] == wnc and cases.same(gotc[z], wantc)


def test_creepfill_negative_weight_takes_the_counter_kernel(fa):
    """a negative setWeight wraps in the reference's size_t sum (interpolation.c:1445); only the counter kernel mirrors that."""
    f = cases.holes(1, 40, 50, seed=5)
    got, n = fa.creepfill2d_host(f, 2, -1)
    want, wn, rc = oracle.creepfill2d(f[0], 2, -1)
    assert cases.same(got[0], want), cases.describe_mismatch(got[0], want)


def _seq_sum(d):
    """double sum = 0; for (...) sum += d[i];  -- cumsum adds strictly one after the other"""
    return np.cumsum(np.concatenate([[0.0], d]))[-1]


def _scan_field(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "kelvin":
        x = rng.normal(280, 12, n)
    elif kind == "wind":
        x = rng.normal(0, 8, n) * np.sin(np.arange(n) * 1e-3)
    elif kind == "tiny_and_huge":
        x = rng.normal(0, 1, n) * 10.0 ** rng.integers(-30, 30, n)
    elif kind == "alternating":
        x = np.where(np.arange(n) % 2 == 0, 1.0, -1.0) * rng.uniform(0.5, 2.0, n) * 1e6
    elif kind == "precip":
        x = np.where(rng.random(n) < 0.7, 0.0, rng.gamma(0.5, 2.0, n) * 1e-3)
    elif kind == "ties":  # small integers and halves: exact ties against large binades are common
        x = rng.integers(-4, 5, n) * 0.5 + np.where(np.arange(n) == 5, 2.0 ** 40, 0.0)
    elif kind == "cancel":  # the running sum returns to zero and changes sign again and again
        b = rng.normal(0, 100, n // 2 + 1)
        x = np.stack([b, -b], axis=1).ravel()[:n]
    else:
        raise ValueError(kind)
    x = x.astype(np.float32)
    x[rng.random(n) < 0.03] = np.nan
    return x


@pytest.mark.parametrize("kind", ["kelvin", "wind", "tiny_and_huge", "alternating", "precip", "ties", "cancel"])
@pytest.mark.parametrize("n", [1, 511, 513, 8191, 8193, 300007])
def test_scan_order_sums_equal_the_sequential_loop(fa, kind, n):
    """Both evaluations of the fills' double sums (chain, binade-parallel) against numpy's strictly sequential cumsum,
    which performs the additions of interpolation.c:1256-1264 / 1288-1299 one after the other."""
    import torch
    x = _scan_field(kind, n, seed=n % 97 + len(kind))
    ok = ~np.isnan(x)
    d = x[ok].astype(np.float64)
    want0 = _seq_sum(d)
    avg = want0 / d.size if d.size else 0.0
    want1 = _seq_sum(np.abs(d - avg))
    t = torch.from_numpy(x).cuda()
    for algo in (0, 1, 2):  # add chain, binade-parallel in one workgroup, the same over the whole chip
        s0, u0 = fa.scan_sum_device(t.data_ptr(), n, 0, 0.0, algo)
        s1, u1 = fa.scan_sum_device(t.data_ptr(), n, 1, avg, algo)
        s2, u2 = fa.scan_sum_device(t.data_ptr(), n, 2, 0.0, algo)
        assert u0 == u1 == u2 == n - d.size
        assert np.float64(s0).tobytes() == np.float64(want0).tobytes(), (algo, s0, want0)
        assert np.float64(s1).tobytes() == np.float64(want1).tobytes(), (algo, s1, want1)


def test_scan_order_sums_full_slice_and_nonfinite(fa):
    import torch
    for kind, n in (("kelvin", 9_000_000), ("wind", 9_000_000), ("cancel", 2_000_001)):
        x = _scan_field(kind, n, seed=3)
        d = x[~np.isnan(x)].astype(np.float64)
        want0 = _seq_sum(d)
        avg = want0 / d.size
        want1 = _seq_sum(np.abs(d - avg))
        t = torch.from_numpy(x).cuda()
        assert np.float64(fa.scan_sum_device(t.data_ptr(), n, 0, 0.0, 1)[0]).tobytes() == np.float64(want0).tobytes()
        assert np.float64(fa.scan_sum_device(t.data_ptr(), n, 1, avg, 1)[0]).tobytes() == np.float64(want1).tobytes()
    x = _scan_field("kelvin", 50000, seed=9)
    x[20000] = np.inf
    t = torch.from_numpy(x).cuda()
    assert fa.scan_sum_device(t.data_ptr(), x.size, 0, 0.0, 1)[0] == np.inf
    x[30000] = -np.inf
    t = torch.from_numpy(x).cuda()
    assert np.isnan(fa.scan_sum_device(t.data_ptr(), x.size, 0, 0.0, 1)[0])


def test_fills_leave_complete_and_empty_slices_alone(fa):
    full = cases.field(1, 20, 30, seed=1, nan_frac=0, extremes=False)
    empty = np.full((1, 20, 30), np.nan, np.float32)
    for f in (full, empty):
        got, n = fa.fill2d_host(f, 4.0, 1.6, 100)
        assert cases.same(got, f)
        got, n = fa.creepfill2d_host(f, 20, 2)
        assert cases.same(got, f)
    assert fa.fill2d_host(empty, 4.0, 1.6, 100)[1] == [600]


def test_bad2nan_nan2bad_and_points2position_device(fa):
    import torch
    rng = np.random.default_rng(0)
    a = rng.normal(0, 1, 100003).astype(np.float32)
    a[rng.choice(a.size, 500, replace=False)] = np.float32(9.96921e36)
    a[rng.choice(a.size, 300, replace=False)] = np.nan
    for off in (0, 1):  # 16-byte aligned and unaligned start
        t = torch.from_numpy(a.copy()).cuda()
        v = t[off:]
        fa.bad2nan_device(v.data_ptr(), v.numel(), 9.96921e36, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert cases.same(v.cpu().numpy(), oracle.bad2nan(a[off:], 9.96921e36))
        fa.nan2bad_device(v.data_ptr(), v.numel(), -32767.0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert cases.same(v.cpu().numpy(), oracle.nan2bad(oracle.bad2nan(a[off:], 9.96921e36), -32767.0))
    # points2position: ascending, descending, circular longitude
    for axis, typ in [(np.linspace(-5, 5, 41), oracle.PROJ_AXIS), (np.linspace(9, -3, 25), oracle.PROJ_AXIS),
                      (np.radians(np.arange(-180, 180, 1.0)), oracle.LONGITUDE), (np.radians(np.arange(0, 360, 0.5)), oracle.LONGITUDE),
                      (np.radians(np.linspace(80, -80, 321)), oracle.LATITUDE), (np.array([1., 2., 3., 4., 5.]), oracle.PROJ_AXIS)]:
        p = rng.uniform(-8, 8, 5000)
        p[::97] = axis[rng.integers(0, axis.size, p[::97].size)]  # exact hits
        p[5], p[6], p[7] = np.nan, np.inf, -np.inf
        t = torch.from_numpy(p.copy()).cuda()
        fa.points2position_device(t.data_ptr(), t.numel(), axis, typ, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(t.cpu().numpy(), oracle.points2position(p, axis, typ))


def test_device_resident_apply_matches_host_apply(fa):
    """The *_device entry point on torch-owned HBM buffers and torch's stream == the *_host round trip."""
    import torch
    inX, inY, outX, outY, nz = 200, 150, 120, 90, 13
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=9)
    f = cases.field(nz, inY, inX, seed=4)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, inX, inY, outX, outY)
    d_in = torch.from_numpy(f).cuda()
    d_out = torch.empty((nz, outY, outX), dtype=torch.float32, device="cuda")
    plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
    assert cases.same(d_out.cpu().numpy(), want)
    # plan built from positions already resident in HBM
    dpx, dpy = torch.from_numpy(px).cuda(), torch.from_numpy(py).cuda()
    plan2 = fa.RegridPlan.from_device(oracle.BICUBIC, dpx.data_ptr(), dpy.data_ptr(), px.size, inX, inY, outX, outY,
                                      torch.cuda.current_stream().cuda_stream)
    plan2.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert cases.same(d_out.cpu().numpy(), oracle.interpolate_values(oracle.BICUBIC, px, py, f, inX, inY, outX, outY))


def test_plan_tune_keeps_results_and_both_shapes_match_the_oracle(fa, monkeypatch, tuning_build):
    """fimex_amd_regrid_plan_tune_device: a bilinear plan holds two workgroup shapes of its LDS-staged form (1024 threads on
    512 x 8 tiles, 512 threads on 256 x 8); the call times both on the caller's buffers and keeps one.  Whatever it picks,
    and with either shape forced (tuning build), the result is the oracle's bit for bit; plans without a second shape
    (nearest, forward) and empty batches answer 0."""
    import torch
    inX, inY, outX, outY, nz = 1441, 721, 900, 500, 9
    px, py = cases.coherent_positions(inX, inY, outX, outY, seed=41, outliers=4)
    f = cases.field(nz, inY, inX, seed=42)
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(f).cuda()
    d_out = torch.full((nz, outY, outX), 7.0, device="cuda")
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, inX, inY, outX, outY)
    before = plan.info()
    assert before["stagedCells"] > 0 and before["tileW"] == 512
    chosen = plan.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
    assert chosen in (0, 1)
    assert cases.same(d_out.cpu().numpy(), want)  # the tuning launches leave the regridded slices behind
    after = plan.info()
    assert after["tileW"] == (256 if chosen else 512) and after["stagedCells"] > 0 and after["planBytes"] > 0
    d_out.fill_(7.0)
    plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    assert cases.same(d_out.cpu().numpy(), want)
    assert cases.same(plan.apply_host(f), want)
    for forced in ("0", "1"):
        monkeypatch.setenv("FIMEX_AMD_STAGE2_USE_ALT", forced)
        d_out.fill_(7.0)
        plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        assert cases.same(d_out.cpu().numpy(), want), "shape %s: %s" % (forced, cases.describe_mismatch(d_out.cpu().numpy(), want))
    monkeypatch.delenv("FIMEX_AMD_STAGE2_USE_ALT")
    assert plan.tune_device(d_in.data_ptr(), 0, d_out.data_ptr(), st) == chosen  # nothing to time: the choice stands
    # the 4 x 4 stencil in float arithmetic holds two shapes as well: the same bits from both
    cub = fa.RegridPlan(oracle.BICUBIC, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_FAST)
    first = cub.apply_host(f)
    assert cub.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), st) in (0, 1)
    assert cases.same(d_out.cpu().numpy(), first)
    for forced in ("0", "1"):
        monkeypatch.setenv("FIMEX_AMD_STAGE2_USE_ALT", forced)
        assert cases.same(cub.apply_host(f), first), "float bicubic, shape " + forced
    monkeypatch.delenv("FIMEX_AMD_STAGE2_USE_ALT")
    near = fa.RegridPlan(oracle.NEAREST, px, py, inX, inY, outX, outY)
    assert near.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), st) == 0
    assert cases.same(near.apply_host(f), oracle.interpolate_values(oracle.NEAREST, px, py, f, inX, inY, outX, outY))
    fpx, fpy = cases.forward_positions(inX, inY, outX, outY, seed=43)
    fwd = fa.RegridPlan(oracle.FWD_MEAN, fpx, fpy, inX, inY, outX, outY)
    assert fwd.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), st) == 0


def test_concurrent_host_applies_are_reentrant(fa):
    """interpolateValues is called from concurrent OpenMP tasks in the reference's writers
    (src/NetCDF_CDMWriter.cc:749-753): several threads, one plan."""
    import threading
    inX, inY, outX, outY = 160, 120, 140, 100
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=2)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, inX, inY, outX, outY)
    fields = [cases.field(3, inY, inX, seed=s) for s in range(6)]
    wants = [oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY) for f in fields]
    results = [None] * len(fields)

    def work(i):
        results[i] = plan.apply_host(fields[i])

    ts = [threading.Thread(target=work, args=(i,)) for i in range(len(fields))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for got, want in zip(results, wants):
        assert cases.same(got, want)


# ---------------------------------------------------------------- typed slice edges
def _typed_field(dt, n, seed):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        a = rng.integers(max(info.min, -2 ** 62), min(info.max, 2 ** 62), n, dtype=np.int64 if info.min < 0 else np.uint64).astype(dt)
        a[:3] = [info.min, info.max, 0]
    else:
        a = (rng.normal(0, 1, n) * 10.0 ** rng.integers(-5, 12, n)).astype(dt)
        k = min(6, n)
        a[:k] = [0.0, -0.0, np.nan, np.inf, -np.inf, 3.4e38][:k]
    a[rng.choice(n, n // 50, replace=False)] = a[min(10, n - 1)]
    return a


@pytest.mark.parametrize("code", sorted(oracle.CDM_DTYPES))
@pytest.mark.parametrize("n,off", [(100003, 0), (100003, 1), (5, 0), (4096, 3)])
def test_data2interpolation_matches_oracle(fa, code, n, off):
    import torch
    dt = oracle.CDM_DTYPES[code]
    a = _typed_field(dt, n + off, seed=code + n)
    bad = float(a[min(10, n + off - 1)])
    t = torch.from_numpy(a.view(np.uint8)).cuda()
    out = torch.empty(n + 4, dtype=torch.float32, device="cuda")
    for o_off in (0, 1):  # aligned (vector) and unaligned (scalar) destinations
        fa.data2interpolation_device(t.data_ptr() + off * a.itemsize, code, n, bad, out.data_ptr() + 4 * o_off)
        torch.cuda.synchronize()
        got = out[o_off:o_off + n].cpu().numpy()
        want = oracle.data2interpolation_array(a[off:], bad)
        assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("code", sorted(oracle.CDM_DTYPES))
@pytest.mark.parametrize("n,off", [(100003, 0), (100001, 1), (7, 0)])
def test_interpolation2data_matches_oracle(fa, code, n, off):
    import torch
    dt = oracle.CDM_DTYPES[code]
    rng = np.random.default_rng(code * 7 + n)
    f = (rng.normal(0, 1, n + off) * 10.0 ** rng.integers(-3, 6, n + off)).astype(np.float32)
    f[rng.choice(n, n // 20 + 1, replace=False)] = np.nan
    half = rng.integers(-300, 300, n // 10 + 1).astype(np.float32) + np.float32(0.5)  # exact ties: half away from zero
    f[off:off + half.size] = half[:f.size - off]
    f[-1] = -0.0
    bad = 77.0
    t = torch.from_numpy(f).cuda()
    out = torch.zeros((n + 4) * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
    for o_off in (0, 1):
        fa.interpolation2data_device(t.data_ptr() + 4 * off, n, code, bad, out.data_ptr() + o_off * np.dtype(dt).itemsize)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(dt)[o_off:o_off + n]
        want = oracle.interpolation_array2data(f[off:], code, bad)
        if np.issubdtype(dt, np.floating):
            assert np.array_equal(got.view(np.uint32 if dt == np.float32 else np.uint64), want.view(np.uint32 if dt == np.float32 else np.uint64))
        else:
            assert np.array_equal(got, want)


@pytest.mark.parametrize("dt,bad", [(np.int16, -32767.0), (np.float32, 9.96921e36), (np.float64, -999.0), (np.int32, -2147483647.0), (np.uint8, 255.0)])
def test_regrid_slice_typed_matches_the_reference_sequence(fa, dt, bad):
    """getDataSlice (src/CDMInterpolator.cc:251-285) on the stored type: asFloat + bad2nan, pre-process, regrid, rotation with
    a counterpart stored in another type, post-process, convertDataType back -- against the same sequence of oracle calls."""
    inX, inY, outX, outY, nz = 70, 50, 90, 60, 3
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=4)
    rng = np.random.default_rng(11)
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        u = rng.integers(max(info.min, -20000) // 2, min(info.max, 20000) // 2, (nz, inY, inX)).astype(dt)
    else:
        u = rng.normal(5, 40, (nz, inY, inX)).astype(dt)
    u.reshape(-1)[rng.choice(u.size, u.size // 30, replace=False)] = dt(bad)
    v = rng.normal(0, 30, (nz, inY, inX)).astype(np.float64)  # the counterpart lives in another type
    v.reshape(-1)[rng.choice(v.size, v.size // 40, replace=False)] = -1e30
    m = cases.rotation_matrix(outX, outY, seed=2)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, inX, inY, outX, outY)
    vec = fa.VectorPlan(m, outX, outY)
    pre = [fa.creepfill2d_process(2, 1)]
    post = [fa.fill2d_process(4.0, 1.6, 30)]
    got = fa.regrid_slice_typed_host(plan, u, bad, pre=pre, post=post, counterpart=v, badValueCounterpart=-1e30, vec=vec, isXComponent=True)
    assert got.dtype == dt

    def chain(a, b):
        f = oracle.data2interpolation_array(a, b)
        f = np.stack([oracle.creepfill2d(s, 2, 1)[0] for s in f])
        return oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)

    ru, rv = oracle.vector_reproject_values(m, chain(u, bad), chain(v, -1e30), outX, outY)
    ru = np.stack([oracle.fill2d(s, 4.0, 1.6, 30)[0] for s in ru])
    want = oracle.interpolation_array2data(ru, oracle.cdm_type_of(dt), bad)
    assert np.array_equal(got.view(np.uint8), want.reshape(got.shape).view(np.uint8))
    # without a counterpart and processes: the float entry point agrees on float data
    if dt == np.float32:
        a = fa.regrid_slice_typed_host(plan, u, bad)
        b = fa.regrid_slice_host(plan, u, bad)
        ok = ~np.isnan(b)
        assert cases.same(np.where(a == 0, 0 * a, a), np.where(b == 0, 0 * b, b))  # apart from -0.0 -> +0.0 (ScaleValue)


# ---------------------------------------------------------------- 1-D blends between two fields
@pytest.mark.parametrize("kind", range(7))
@pytest.mark.parametrize("abx", [(1., 2., 1.5), (1., 1., .5), (0., 1., 2.), (0., 1., 1.), (0., 1., 0.), (0., 1., -.5), (0., 1., -1.5), (0., 1., 2.5),
                                 (1000., 100., 500.), (1000., 100., 1500.), (1000., 100., 100.), (3., 7., 3.0000001)])
def test_blends_between_two_fields_match_oracle(fa, kind, abx):
    a, b, x = abx
    A = cases.field(1, 37, 53, seed=kind + 1, nan_frac=0.05)[0]
    B = cases.field(1, 37, 53, seed=kind + 50, nan_frac=0.05)[0]
    want, rc = oracle.get_values_1d(kind, A, B, a, b, x)
    if rc != oracle.OK:
        with pytest.raises(fa.FimexAmdError):
            fa.get_values_1d_host(kind, A, B, a, b, x)
        return
    got = fa.get_values_1d_host(kind, A, B, a, b, x)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


def test_log_blends_reject_non_positive_coordinates_and_kats(fa):
    A, B = np.array([1000.], np.float32), np.array([100.], np.float32)
    for kind in (fa.BLEND_LOG, fa.BLEND_LOG_LOG):
        for abx in ((0., 1., 2.), (1., -1., 2.), (1., 2., 0.)):
            with pytest.raises(fa.FimexAmdError):
                fa.get_values_1d_host(kind, A, B, *abx)
    # test/testInterpolation.cc:212-262
    for x, w in ((500., 729.073), (1500., 1158.482), (200., 370.927), (800., 912.781)):
        assert abs(fa.get_values_1d_host(fa.BLEND_LOG, A, B, 1000., 100., x)[0] - w) / w < 1e-5
    for x, w in ((500., 763.1873), (200., 408.0904), (800., 926.384)):
        assert abs(fa.get_values_1d_host(fa.BLEND_LOG_LOG, A, B, 1000., 100., x)[0] - w) / w < 1e-5


def test_reference_kats_of_the_linear_family_on_the_gpu(fa):
    """test/testInterpolation.cc:685-731: in0 = 200 at a = 2, in1 = 300 at b = 3, x = 0.5 .. 4.5."""
    A, B = np.array([200], np.float32), np.array([300], np.float32)
    want = {fa.BLEND_LINEAR_NO_EXTRAPOL: (np.nan, np.nan, 250, np.nan, np.nan), fa.BLEND_LINEAR_CONST_EXTRAPOL: (200, 200, 250, 300, 300),
            fa.BLEND_LINEAR_WEAK_EXTRAPOL: (np.nan, 150, 250, 350, np.nan), fa.BLEND_LINEAR: (50, 150, 250, 350, 450)}
    for kind, expect in want.items():
        for x, w in zip((0.5, 1.5, 2.5, 3.5, 4.5), expect):
            out = fa.get_values_1d_host(kind, A, B, 2., 3., x)
            assert (np.isnan(out[0]) and np.isnan(w)) or abs(out[0] - w) < 0.01, (kind, x, out, w)


def test_blend_device_in_place_and_double(fa):
    import torch
    rng = np.random.default_rng(2)
    A, B = rng.normal(0, 5, 100001), rng.normal(3, 5, 100001)
    tA, tB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    out = torch.empty_like(tA)
    fa.get_values_linear_d_device(tA.data_ptr(), tB.data_ptr(), out.data_ptr(), A.size, 2., 5., 3.)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), oracle.get_values_linear_d(A, B, 2., 5., 3.))
    fA, fB = torch.from_numpy(A.astype(np.float32)).cuda(), torch.from_numpy(B.astype(np.float32)).cuda()
    want = oracle.get_values_1d(oracle.BLEND_LINEAR, A.astype(np.float32), B.astype(np.float32), 2., 5., 3.)[0]
    fa.get_values_1d_device(fa.BLEND_LINEAR, fA.data_ptr(), fB.data_ptr(), fA.data_ptr(), A.size, 2., 5., 3.)  # out aliases A
    torch.cuda.synchronize()
    assert cases.same(fA.cpu().numpy(), want)


# ---------------------------------------------------------------- coordinate-based nearest neighbour plans
from test_oracle_props import _curvilinear_grid, _great_circle_cos


def _same_cells_or_ties(gx, gy, wx, wy, metric_of, none):
    """identical plans; where they differ, both candidates must be numerically equidistant (ties are unspecified)."""
    diff = (gx != wx) | (gy != wy)
    if diff.any():
        assert not np.any((gx[diff] == none) | (wx[diff] == none))
        mg, mw = metric_of(np.nonzero(diff)[0], gx[diff], gy[diff]), metric_of(np.nonzero(diff)[0], wx[diff], wy[diff])
        np.testing.assert_allclose(mg, mw, rtol=0, atol=1e-15)
    return diff.mean()


@pytest.mark.parametrize("shape,jitter", [((60, 45), 0.3), ((200, 150), 0.3), ((64, 64), 0.0), ((1, 7), 0.2), ((5, 1), 0.2)])
def test_coord_nearest_matches_oracle(fa, shape, jitter):
    nx, ny = shape
    lon, lat = _curvilinear_grid(nx, ny, seed=nx + ny, jitter=jitter, nan=min(5, nx * ny // 10))
    rng = np.random.default_rng(nx)
    qlon = np.radians(rng.uniform(4.0, 17.0, 20000)); qlat = np.radians(rng.uniform(57.0, 65.5, 20000))
    qlon[:3] = [np.nan, 0.1, 0.2]; qlat[:3] = [1.0, np.nan, 3.0]
    assert abs(fa.grid_distance_host(lon, lat) - oracle.grid_distance(lon, lat)) < 1e-12
    gx, gy = fa.coord_nearest_host(qlon, qlat, lon, lat)
    wx, wy = oracle.fast_translate_points(qlon, qlat, lon, lat)
    def metric(i, x, y):
        k = (y * nx + x).astype(int)
        return _great_circle_cos(qlon[i], qlat[i], lon.ravel()[k], lat.ravel()[k])
    assert _same_cells_or_ties(gx, gy, wx, wy, metric, -1) < 0.02
    assert (gx >= 0).any() and (gx == -1).any()
    # the plan applies as row a1
    f = cases.field(2, ny, nx, seed=3)
    got = fa.RegridPlan(oracle.COORD_NN, gx, gy, nx, ny, 200, 100).apply_host(f)
    want = oracle.interpolate_values(oracle.COORD_NN, gx, gy, f, nx, ny, 200, 100)
    assert cases.same(got, want)


@pytest.mark.parametrize("shape,maxDist", [((60, 45), 3000.0), ((200, 150), 2500.0), ((64, 64), 4000.0), ((30, 20), 2.0e7), ((30, 20), 1.0)])
def test_coord_kdtree_matches_oracle(fa, shape, maxDist):
    nx, ny = shape
    lon, lat = _curvilinear_grid(nx, ny, seed=nx * 3 + ny, jitter=0.3 if nx != 64 else 0.0, nan=5)
    rng = np.random.default_rng(ny)
    nq = 3000 if maxDist > 1e6 else 20000
    qlon = np.radians(rng.uniform(4.0, 17.0, nq)); qlat = np.radians(rng.uniform(57.0, 65.5, nq))
    gx, gy = fa.coord_kdtree_host(maxDist, qlon, qlat, lon, lat)
    wx, wy = oracle.flann_translate_points(maxDist, qlon, qlat, lon, lat)
    def xyz(lo, la):
        return np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], -1)
    def metric(i, x, y):
        k = (y * nx + x).astype(int)
        return ((xyz(qlon[i], qlat[i]) - xyz(lon.ravel()[k], lat.ravel()[k])) ** 2).sum(-1)
    assert _same_cells_or_ties(gx, gy, wx, wy, metric, -1000) < 0.02
    if maxDist > 1e6:
        assert (gx >= 0).all()       # everything is within 20 000 km
    elif maxDist < 10:
        assert (gx == -1000).all()   # nothing within a metre
    else:
        assert (gx >= 0).any() and (gx == -1000).any()


def test_coord_search_rejects_bad_arguments(fa):
    lon, lat = _curvilinear_grid(10, 8, seed=1)
    with pytest.raises(fa.FimexAmdError):
        fa.coord_kdtree_host(0.0, lon[0], lat[0], lon, lat)
    with pytest.raises(fa.FimexAmdError):
        fa.coord_nearest_host(lon[0], lat[0], np.full((8, 10), np.nan), lat)


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
@pytest.mark.parametrize("dt,bad", [(np.int16, -32767.0), (np.uint16, 65535.0), (np.int8, -127.0), (np.uint8, 255.0), (np.int32, -2147483647.0),
                                    (np.float32, 9.96921e36), (np.float64, -1e30), (np.int64, -9.0e18)])
def test_regrid_on_the_stored_type_device_resident(fa, monkeypatch, method, dt, bad, tuning_build):
    """fimex_amd_regrid_apply_typed_device: one kernel on the stored type for the small integer types, three passes for the
    others, both against the oracle's three steps; the fused and the unfused GPU paths agree as well."""
    import torch
    inX, inY, outX, outY, nz = 96, 70, 150, 110, 19
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=12)
    rng = np.random.default_rng(3)
    if np.issubdtype(dt, np.integer):
        info = np.iinfo(dt)
        f = rng.integers(max(info.min, -30000) // 2, min(info.max, 30000) // 2 + 1, (nz, inY, inX)).astype(dt)
    else:
        f = rng.normal(0, 50, (nz, inY, inX)).astype(dt)
    f.reshape(-1)[rng.choice(f.size, f.size // 25, replace=False)] = dt(bad)
    code = oracle.cdm_type_of(dt)
    want = oracle.interpolation_array2data(
        oracle.interpolate_values(method, px, py, oracle.data2interpolation_array(f, bad), inX, inY, outX, outY), code, bad)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    t = torch.from_numpy(f.view(np.uint8)).cuda()
    results = []
    # TYPED_STAGED 1: 1- and 2-byte types through the LDS-staged kernels; 0: the gather kernels on the stored type
    # TYPED_FUSED 2: also bicubic through the stored-type gather kernel; 0: three passes
    # TYPED_STAGED2 1: the second staged form (staged2.hip, nearest and bilinear) before the first
    for staged, fused, second in (("1", "1", "1"), ("1", "1", "0"), ("0", "2", "1"), ("0", "1", "1"), ("0", "0", "1")):
        monkeypatch.setenv("FIMEX_AMD_TYPED_STAGED", staged)
        monkeypatch.setenv("FIMEX_AMD_TYPED_FUSED", fused)
        monkeypatch.setenv("FIMEX_AMD_TYPED_STAGED2", second)
        out = torch.zeros(nz * outY * outX * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
        fa.regrid_apply_typed_device(plan, t.data_ptr(), code, nz, bad, out.data_ptr())
        torch.cuda.synchronize()
        results.append(out.cpu().numpy().view(dt).reshape(want.shape))
        assert np.array_equal(results[-1].view(np.uint8), want.view(np.uint8)), (staged, fused, np.dtype(dt).name)


@pytest.mark.parametrize("method", [oracle.FWD_MEAN, oracle.FWD_UNDEF_MEAN, oracle.FWD_SUM, oracle.FWD_MAX, oracle.FWD_UNDEF_MIN, oracle.FWD_MEDIAN])
@pytest.mark.parametrize("dt,bad", [(np.int16, -32767.0), (np.uint16, 65535.0), (np.int8, -127.0), (np.uint8, 255.0), (np.int16, float("nan")), (np.int32, -2147483647.0)])
@pytest.mark.parametrize("shape,density", [((400, 300, 37, 29), 1.5), ((333, 377, 32, 80), 1.3), ((257, 129, 64, 33), 0.9)])
def test_forward_on_the_stored_type(fa, monkeypatch, method, dt, bad, shape, density, tuning_build):
    """fimex_amd_regrid_apply_typed_device on forward plans with long buckets: the LDS-staged forward kernel reads the slices in
    their 1- or 2-byte stored type (a tiled form of the plan with 8 or 16 cells per 16-byte chunk, built on the first such call)
    and writes results in that type; wider types, medians and TYPED_FORWARD=0 take the three passes.  All against the oracle's three
    steps (data2InterpolationArray, interpolateValues, interpolationArray2Data); with and without a fill value; buckets that hold
    nothing but fill values; sums that leave the type's range (the reference's casts wrap)."""
    import torch
    inX, inY, outX, outY = shape
    nz = 11
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=41, density=density, special=False)
    rng = np.random.default_rng(5)
    info = np.iinfo(dt)
    f = rng.integers(max(info.min, -3000) // 2, min(info.max, 3000) // 2 + 1, (nz, inY, inX)).astype(dt)
    if bad == bad:
        f.reshape(-1)[rng.choice(f.size, f.size // 20, replace=False)] = dt(bad)
        f[:, 40:90, 100:180] = dt(bad)     # whole buckets of fill values
    code = oracle.cdm_type_of(dt)
    want = oracle.interpolation_array2data(
        oracle.forward_interpolate_values(method, px, py, oracle.data2interpolation_array(f, bad), inX, inY, outX, outY), code, bad)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    t = torch.from_numpy(f.view(np.uint8)).cuda()
    for fused in ("1", "0"):
        monkeypatch.setenv("FIMEX_AMD_TYPED_FORWARD", fused)
        out = torch.zeros(nz * outY * outX * np.dtype(dt).itemsize, dtype=torch.uint8, device="cuda")
        fa.regrid_apply_typed_device(plan, t.data_ptr(), code, nz, bad, out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(dt).reshape(want.shape)
        assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (fused, np.dtype(dt).name, int((got != want).sum()))


@pytest.mark.parametrize("dt,bad", [(np.int16, -32767.0), (np.uint8, 255.0)])
def test_forward_on_the_stored_type_through_host_buffers(fa, dt, bad):
    """fimex_amd_regrid_slice_typed_host with a forward plan of long buckets: the streamed host pipeline hands its chunks of slices to the
    stored-type forward kernel; against the oracle's three steps."""
    inX, inY, outX, outY, nz = 400, 300, 37, 29, 23
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=41, density=1.5, special=False)
    rng = np.random.default_rng(8)
    info = np.iinfo(dt)
    f = rng.integers(max(info.min, -3000) // 2, min(info.max, 3000) // 2 + 1, (nz, inY, inX)).astype(dt)
    f.reshape(-1)[rng.choice(f.size, f.size // 20, replace=False)] = dt(bad)
    code = oracle.cdm_type_of(dt)
    for method in (oracle.FWD_MEAN, oracle.FWD_UNDEF_MAX):
        want = oracle.interpolation_array2data(
            oracle.forward_interpolate_values(method, px, py, oracle.data2interpolation_array(f, bad), inX, inY, outX, outY), code, bad)
        plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
        got = fa.regrid_slice_typed_host(plan, f, bad)
        assert got.dtype == np.dtype(dt) and np.array_equal(got.reshape(want.shape).view(np.uint8), want.view(np.uint8))


@pytest.mark.parametrize("shape", [(400, 300, 200, 200), (403, 301, 130, 77), (1200, 900, 600, 500)])
def test_bicubic_fast_arithmetic_within_the_stated_tolerance(fa, shape):
    """FIMEX_AMD_BICUBIC_FAST (float fused multiply-adds, include/fimex_amd.h) against the oracle's reference arithmetic
    (src/interpolation.c:959-1028) on adversarial input: a field that crosses zero everywhere (cancellation in the stencil
    sums) with magnitudes over six decades.  Tolerance as BASELINE.json states it, 1e-5, relative to the largest magnitude
    in the cell's 4x4 stencil (a result near zero has no digits to be relative to: the reference's own float accumulation
    errs by 6e-8 of that magnitude there); NaN positions are identical; the reference arithmetic stays bit-exact."""
    inX, inY, outX, outY = shape
    nz = 6
    px, py = cases.coherent_positions(inX, inY, outX, outY, seed=21, outliers=3)
    rng = np.random.default_rng(22)
    f = rng.standard_normal((nz, inY, inX)).astype(np.float32)
    f *= np.float32(10.0) ** rng.integers(-3, 4, (nz, 1, 1)).astype(np.float32)
    f[:, ::53, ::47] = np.nan
    want = oracle.interpolate_values(oracle.BICUBIC, px, py, f, inX, inY, outX, outY)
    fast = fa.RegridPlan(fa.BICUBIC, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_FAST)
    assert fast.info()["stagedCells"] > 0
    got = fast.apply_host(f)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    # largest |f| in each output cell's stencil: 4x4 window whose corner is (floor(x) - 1, floor(y) - 1)
    from numpy.lib.stride_tricks import sliding_window_view
    x0 = np.floor(px).astype(np.int64) - 1
    y0 = np.floor(py).astype(np.int64) - 1
    ok = (x0 >= 0) & (x0 + 3 < inX) & (y0 >= 0) & (y0 + 3 < inY)
    err_rel = 0.0
    for z in range(nz):
        win = sliding_window_view(np.abs(np.nan_to_num(f[z])), (4, 4)).max(axis=(2, 3))
        mag = np.zeros(px.size, np.float32)
        mag[ok] = win[y0[ok], x0[ok]]
        d = np.abs(got[z].ravel().astype(np.float64) - want[z].ravel())
        fin = np.isfinite(want[z].ravel())
        assert fin.sum() > 0.5 * ok.sum()
        assert (d[fin] <= 1e-5 * mag[fin]).all(), float((d[fin] / mag[fin]).max())
        err_rel = max(err_rel, float((d[fin] / mag[fin]).max()))
    assert 0 < err_rel < 3e-6, err_rel  # typical: a few 1e-7; 0 would mean the reference kernel ran
    # the default and the explicit reference arithmetic are the bit-exact kernel
    ref = fa.RegridPlan(fa.BICUBIC, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_REFERENCE).apply_host(f)
    assert cases.same(ref, want)
    # other methods ignore the switch
    lin = fa.RegridPlan(fa.BILINEAR, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_FAST).apply_host(f)
    assert cases.same(lin, oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY))
    with pytest.raises(fa.FimexAmdError, match="bicubic arithmetic"):
        fa.RegridPlan(fa.BICUBIC, px, py, inX, inY, outX, outY, bicubic=7)


@pytest.mark.parametrize("method", [oracle.NEAREST, oracle.BILINEAR, oracle.BICUBIC])
@pytest.mark.parametrize("wrap,outliers", [(True, 0), (False, 5), (True, 4)])
def test_staged_plan_with_a_seam_and_isolated_points(fa, method, wrap, outliers):
    """A target that straddles the seam of a periodic source (positions jump by the row length inside an output row) and
    isolated cells that point elsewhere: the tiles that hold them cannot be staged through LDS and read their stencils
    from memory instead (gather tiles, staged2.hip); every other tile stays staged, the result is the oracle's bit for bit.
    Odd source width on purpose (row segments aligned per row)."""
    inX, inY, outX, outY, nz = 1441, 721, 900, 500, 6
    px, py = cases.coherent_positions(inX, inY, outX, outY, seed=31, wrap=wrap, outliers=outliers)
    f = cases.field(nz, inY, inX, seed=32)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    assert plan.info()["stagedCells"] > 0, "the plan lost its staged form"
    got = plan.apply_host(f)
    want = oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.isfinite(want).mean() > 0.8


@pytest.mark.parametrize("method,arith", [(oracle.NEAREST, 0), (oracle.BILINEAR, 0), (oracle.BICUBIC, 1)])
def test_staged_tiles_that_fill_their_slot_are_repeatable(fa, method, arith):
    """Tiles whose chunk list is not a whole number of workgroup-wide DMA rounds but nearly fills its LDS slot: the lanes
    without a chunk in the last round write zeros (LDS-DMA with an out-of-range offset does, scripts/calib/dma_oob.hip), and
    where that round ran past the slot they fell into the first 512 bytes of the slot being interpolated -- a race lost a
    few times in a hundred (round 2: staged_apply2 now points such a round at a spare KiB).  The same plan is rebuilt and
    applied many times between allocations of other sizes; every result is the oracle's (bicubic: the first one's)."""
    import torch
    inX, inY, outX, outY, nz = 403, 301, 130, 77, 6
    px, py = cases.coherent_positions(inX, inY, outX, outY, seed=21, outliers=3)
    f = cases.field(nz, inY, inX, seed=23)
    want = None if arith else oracle.interpolate_values(method, px, py, f, inX, inY, outX, outY)
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(f).cuda()
    rng = np.random.default_rng(5)
    junk = []
    for it in range(60):
        junk.append(torch.empty(int(rng.integers(1, 64)) * 262144, device="cuda").normal_())
        if len(junk) > 6:
            junk.pop(0)
        plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_FAST if arith else fa.BICUBIC_REFERENCE)
        assert plan.info()["stagedCells"] > 0
        d_out = torch.full((nz, outY, outX), 7.0, device="cuda")
        plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        if want is None:
            want = got
        assert cases.same(got, want), "iteration %d: %s" % (it, cases.describe_mismatch(got, want))
        plan.close()


@pytest.mark.parametrize("shape", [(257, 1300, 3), (500, 700, 9), (64, 2100, 2)])
@pytest.mark.parametrize("params", [(4.0, 1.6, 60), (1e-9, 1.9, 25)])
@pytest.mark.parametrize("geometry", ["1", "2"], ids=["16waves_x_16columns", "8waves_x_32columns"])
def test_fill2d_with_several_workgroups_per_slice(fa, monkeypatch, shape, params, geometry, tuning_build):
    """Small batches of tall slices: the bands of a slice are dealt to several workgroups (fill2d_kernel_v3: hand-off of
    every W-th band boundary through write-through stores and a global progress word, a barrier of the slice's workgroups
    per sweep, the convergence test reduced across them).  Same bits as the oracle and as one workgroup per slice."""
    nx, ny, nz = shape
    relaxCrit, corrEff, maxLoop = params
    monkeypatch.setenv("FIMEX_AMD_FILL_GEOMETRY", geometry)
    f = cases.holes(nz, ny, nx, seed=nx * 7 + ny)
    monkeypatch.setenv("FIMEX_AMD_FILL_MULTI", "1")
    got, nch = fa.fill2d_host(f, relaxCrit, corrEff, maxLoop)
    monkeypatch.setenv("FIMEX_AMD_FILL_MULTI", "0")
    one, nch1 = fa.fill2d_host(f, relaxCrit, corrEff, maxLoop)
    assert list(nch) == list(nch1)
    for z in range(nz):
        want, wn, rc = oracle.fill2d(f[z], relaxCrit, corrEff, maxLoop)
        assert rc == oracle.OK and nch[z] == wn
        assert cases.same(one[z], want), "one workgroup, slice %d: %s" % (z, cases.describe_mismatch(one[z], want))
        assert cases.same(got[z], want), "several workgroups, slice %d: %s" % (z, cases.describe_mismatch(got[z], want))


@pytest.mark.parametrize("shape", [(257, 1300, 3), (500, 700, 9), (64, 2100, 1)])
@pytest.mark.parametrize("params", [(20, 2), (3, 1), (40, 5)])
def test_creepfill_with_several_workgroups_per_slice(fa, monkeypatch, shape, params, tuning_build):
    """creepfill_kernel_v3: the bands of a slice on several workgroups (values and this sweep's U words of a band's last row
    cross workgroups write-through, "something changed" is reduced over the slice's workgroups every sweep).  Same bits as
    the oracle and as one workgroup per slice, for creepfill2d and creepfillval2d."""
    nx, ny, nz = shape
    repeat, weight = params
    f = cases.holes(nz, ny, nx, seed=nx * 5 + ny + repeat)
    monkeypatch.setenv("FIMEX_AMD_FILL_MULTI", "1")
    got, nch = fa.creepfill2d_host(f, repeat, weight)
    gotv, nchv = fa.creepfillval2d_host(f, 271.5, repeat, weight)
    monkeypatch.setenv("FIMEX_AMD_FILL_MULTI", "0")
    one, nch1 = fa.creepfill2d_host(f, repeat, weight)
    assert list(nch) == list(nch1)
    for z in range(nz):
        want, wn = oracle.creepfill2d(f[z], repeat, weight)[:2]
        assert nch[z] == wn
        assert cases.same(one[z], want), "one workgroup, slice %d: %s" % (z, cases.describe_mismatch(one[z], want))
        assert cases.same(got[z], want), "several workgroups, slice %d: %s" % (z, cases.describe_mismatch(got[z], want))
        wantv = oracle.creepfillval2d(f[z], 271.5, repeat, weight)[0]
        assert cases.same(gotv[z], wantv), "creepfillval2d, slice %d: %s" % (z, cases.describe_mismatch(gotv[z], wantv))


def test_release_caches_between_host_calls(fa):
    """fimex_amd_release_caches frees the pinned staging the *_host calls keep between calls; the next call builds it again
    (buffers are made on first use and only the kind a call needs)."""
    inX, inY, outX, outY, nz = 700, 500, 400, 300, 40  # 56 MB in: the streamed path
    px, py = cases.coherent_positions(inX, inY, outX, outY, seed=41)
    f = cases.field(nz, inY, inX, seed=42)
    plan = fa.RegridPlan(fa.BILINEAR, px, py, inX, inY, outX, outY)
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
    assert cases.same(plan.apply_host(f), want)
    fa.release_caches()
    fa.release_caches()
    assert cases.same(plan.apply_host(f), want)
    packed = np.clip(np.nan_to_num(f, nan=0.0, posinf=0.0, neginf=0.0) * 10, -30000, 30000).astype(np.int16)
    got = fa.regrid_slice_typed_host(plan, packed[:8], -32767.0)
    fa.release_caches()
    assert cases.same(plan.apply_host(f[:3]), want[:3])
    assert got.dtype == np.int16


@pytest.mark.parametrize("method", [oracle.FWD_MEDIAN, oracle.FWD_UNDEF_MEDIAN, oracle.FWD_MEAN, oracle.FWD_MAX, oracle.FWD_UNDEF_MIN])
@pytest.mark.parametrize("shape,density", [((240, 200, 40, 30), 1.2), ((240, 200, 40, 30), 7.0), ((300, 256, 64, 48), 3.0)])
def test_forward_dense_mappings(fa, monkeypatch, method, shape, density, tuning_build):
    """A source finer than the target (what forward interpolation is for): tens to a couple of hundred source cells per bucket.  The
    median goes through the wave kernels (selection on keys, one to four registers of bucket values per lane -- zeros of either sign
    as the median included; before it: ranks by broadcast), sums and extrema through the LDS-staged kernel or the lane kernels with
    eight cells of look-ahead; NaNs, signed zeros and many equal values in the data.  Same bits as the oracle from every one of them,
    and as the rank-counting median of the lane kernel."""
    inX, inY, outX, outY = shape
    nz = 5
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=23, density=density, special=False)
    f = cases.field(nz, inY, inX, seed=60 + method, nan_frac=0.03)
    f[:, ::3, ::2] = np.float32(0.0)
    f[:, 1::3, 1::2] = np.float32(-0.0)
    f[:, 2::7, ::3] = np.float32(281.5)   # many equal values inside a bucket: ties are broken by position
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    info = plan.info()
    assert 8 < info["maxBucket"] <= 256, info["maxBucket"]
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(want[~np.isnan(want)]))
    monkeypatch.setenv("FIMEX_AMD_FWD_TILED", "0")
    lanes = plan.apply_host(f)   # sums and extrema: the lane kernels with look-ahead instead of the LDS-staged kernel
    assert cases.same(lanes, want), cases.describe_mismatch(lanes, want)
    monkeypatch.setenv("FIMEX_AMD_FWD_MEDIAN_WAVE_MIN", "1")
    selected = plan.apply_host(f)  # the median by selection on keys also where the plan's buckets are short on average
    assert cases.same(selected, want), cases.describe_mismatch(selected, want)
    assert np.array_equal(np.signbit(selected[~np.isnan(selected)]), np.signbit(want[~np.isnan(want)]))
    monkeypatch.setenv("FIMEX_AMD_FWD_MEDIAN_SELECT", "0")
    ranked = plan.apply_host(f)  # the median by ranks (broadcasts) instead of by selection on keys
    assert cases.same(ranked, want), cases.describe_mismatch(ranked, want)
    assert np.array_equal(np.signbit(ranked[~np.isnan(ranked)]), np.signbit(want[~np.isnan(want)]))
    monkeypatch.setenv("FIMEX_AMD_FWD_MEDIAN_WAVE", "0")
    monkeypatch.setenv("FIMEX_AMD_FWD_WAVE", "1")
    other = plan.apply_host(f)   # the other kernels of the same method: rank counting per lane, wave per bucket
    assert cases.same(other, want), cases.describe_mismatch(other, want)


@pytest.mark.parametrize("method", [oracle.FWD_SUM, oracle.FWD_MEAN, oracle.FWD_UNDEF_MEAN, oracle.FWD_MAX, oracle.FWD_UNDEF_MAX, oracle.FWD_MIN, oracle.FWD_UNDEF_SUM, oracle.FWD_MEDIAN])
@pytest.mark.parametrize("shape,density,special", [((400, 300, 37, 29), 1.5, False), ((300, 200, 50, 33), 1.0, True), ((512, 200, 130, 13), 1.6, False),
                                                    ((333, 377, 32, 80), 1.3, True), ((1203, 60, 60, 9), 1.8, False)])
def test_forward_tiled_mappings(fa, monkeypatch, method, shape, density, special, tuning_build):
    """Long buckets (mean lengths 20 to 135 here): the LDS-staged forward kernel (forward_tiled.hip) -- tiles of 64 targets, the source rows of
    a tile's buckets streamed through LDS, the lanes walking their buckets through the step table in the reference's scan order.
    Rotated, wavy mappings, target grids that are no multiple of the tile, source rows of odd length (chunks are cut on multiples
    of four cells of the slice, not of the row), NaNs / signed zeros / equal values, `special`: source cells thrown to the corners
    and edges of the target (buckets whose cells lie far apart: tiles that cannot be staged read from memory).  Same bits as the
    oracle with every tile shape, one to eight slices of a tile in LDS per pass, and as the lane kernels."""
    inX, inY, outX, outY = shape
    nz = 7
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=31, density=density, special=special)
    f = cases.field(nz, inY, inX, seed=70 + method, nan_frac=0.02)
    f[:, ::3, ::2] = np.float32(0.0)
    f[:, 1::3, 1::2] = np.float32(-0.0)
    f[:, 2::7, ::3] = np.float32(281.5)
    f[:, 5::11, :] = np.nan  # whole rows undefined: buckets that start with NaNs
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    info = plan.info()
    median = method == oracle.FWD_MEDIAN  # ranks in registers, one wave per target: no tiled form
    assert median or (info["stagedCells"] > 0 and info["tileW"] * info["tileH"] == 64), info
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(want[~np.isnan(want)]))
    for slots in ("1", "3", "8"):
        monkeypatch.setenv("FIMEX_AMD_FWD_TILED_SLOTS", slots)
        other = plan.apply_host(f[:3])
        assert cases.same(other, want[:3]), (slots, cases.describe_mismatch(other, want[:3]))
    monkeypatch.delenv("FIMEX_AMD_FWD_TILED_SLOTS")
    for tw in ("8", "64", "4"):
        monkeypatch.setenv("FIMEX_AMD_FWD_TILE_W", tw)
        p2 = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
        assert p2.info()["tileW"] in (0, int(tw))  # 0: this shape cannot be staged with tiles of that width (lane kernels)
        other = p2.apply_host(f)
        assert cases.same(other, want), (tw, cases.describe_mismatch(other, want))
    monkeypatch.setenv("FIMEX_AMD_FWD_TILED", "0")
    p3 = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    assert p3.info()["stagedCells"] == 0
    lanes = p3.apply_host(f)
    assert cases.same(lanes, want), cases.describe_mismatch(lanes, want)


@pytest.mark.parametrize("method", [oracle.FWD_MEAN, oracle.FWD_UNDEF_SUM, oracle.FWD_UNDEF_MIN, oracle.FWD_MAX, oracle.FWD_MEDIAN, oracle.FWD_UNDEF_MEDIAN])
@pytest.mark.parametrize("shape,nz", [((40, 28, 3, 2), 1), ((40, 28, 3, 2), 2), ((64, 48, 5, 70), 3), ((40, 30, 4, 3), 1300), ((257, 129, 33, 17), 37)])
def test_forward_tiled_edges(fa, method, shape, nz):
    """The LDS-staged forward kernel and the selection median at the edges of their launch geometry: target grids smaller than one
    tile, one and two slices (a single z chunk), more slices than the z chunks of one launch hold at four each (the chunk length
    grows), a slice count that leaves a short last chunk; whole slices of NaN, of +0.0, of -0.0 and of +-inf among them."""
    inX, inY, outX, outY = shape
    j, i = np.meshgrid(np.arange(inY, dtype=np.float64), np.arange(inX, dtype=np.float64), indexing="ij")
    px = ((i + 0.5) * outX / inX - 0.5 + 0.2 * np.sin(j / 7.0)).ravel()   # every source cell maps: buckets of inX * inY / (outX * outY) cells
    py = ((j + 0.5) * outY / inY - 0.5 + 0.2 * np.cos(i / 5.0)).ravel()
    f = cases.field(nz, inY, inX, seed=90 + method, nan_frac=0.04)
    for k, v in enumerate((np.nan, 0.0, -0.0, np.inf, -np.inf)):
        if k < nz:
            f[(k * 7) % nz] = np.float32(v)
    if nz > 6:
        f[5, ::2, :] = np.float32(-0.0)   # zeros of both signs inside every bucket: the median's tie-break, the sums' sign
        f[5, 1::2, :] = np.float32(0.0)
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    info = plan.info()
    assert 8 <= info["maxBucket"] <= 256, info
    if method not in (oracle.FWD_MEDIAN, oracle.FWD_UNDEF_MEDIAN):
        assert info["stagedCells"] > 0, info
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(want[~np.isnan(want)]))


@pytest.mark.parametrize("method", FORWARD)
@pytest.mark.parametrize("shape,density", [((150, 120, 64, 50), 0.08), ((150, 120, 63, 51), 0.08), ((300, 200, 100, 80), 0.5), ((200, 150, 40, 36), 2.5)])
def test_forward_sparse_mappings_and_short_median(fa, monkeypatch, method, shape, density, tuning_build):
    """Sparse forward mappings (BASELINE configs[3]: most targets stay empty, buckets of one or two cells): the median of
    buckets of at most two cells runs without rank counting.  Same bits as the oracle and as the rank-counting median; signed
    zeros and equal values included (the median of two equal-comparing values is the second one)."""
    inX, inY, outX, outY = shape
    nz = 9
    px, py = cases.forward_positions(inX, inY, outX, outY, seed=17, density=density)
    f = cases.field(nz, inY, inX, seed=50 + method, nan_frac=0.05)
    f[:, ::3, ::2] = np.float32(0.0)
    f[:, 1::3, 1::2] = np.float32(-0.0)
    f[:, 2::3, ::5] = np.float32(7.25)
    want = oracle.forward_interpolate_values(method, px, py, f, inX, inY, outX, outY)
    plan = fa.RegridPlan(method, px, py, inX, inY, outX, outY)
    got = plan.apply_host(f)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    assert np.array_equal(np.signbit(got[~np.isnan(got)]), np.signbit(want[~np.isnan(want)]))
    for few in (1, 2, 3, 4):  # the reference's call pattern is one slice per call: kernel forms for one, two and four slices per pass
        part = plan.apply_host(f[9 - few:])
        assert cases.same(part, want[9 - few:]), (few, cases.describe_mismatch(part, want[9 - few:]))
    monkeypatch.setenv("FIMEX_AMD_FWD_MEDIAN_SHORT", "0")
    plain = plan.apply_host(f)
    assert cases.same(plain, want), cases.describe_mismatch(plain, want)
