"""BASELINE.json configurations at their full sizes on the GPU (a few slices each), through the C ABI.

Grids are the benchmark's own (workloads.py).  The CPU oracle is fast enough to check whole slices at these sizes
(OpenMP over output cells), so every case is compared in full; size-independent properties ride along: both bilinear
kernels agree, slices are independent (batching / order do not matter), a constant field stays constant, rotation
preserves vector length, a filled field has no holes and keeps its defined cells."""
import numpy as np
import pytest

import cases
import oracle
import workloads
from oracle import proj_oracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    from fimex_amd import capi
    capi.load()
    assert capi.device_count() >= 1
    return capi


@pytest.fixture(scope="module")
def c2(fa):
    """configs[1]/[2]: 4000x3000 lon/lat -> 2000x2000 rotated pole; positions through the product's points2position."""
    wl = workloads.BilinearRotatedPole()
    lon, lat = wl.target_lonlat()
    ax, ay = wl.source_axes_rad()
    px = fa.points2position_host(lon, ax, fa.LONGITUDE)
    py = fa.points2position_host(lat, ay, fa.LATITUDE)
    np.testing.assert_array_equal(px, oracle.points2position(lon, ax, oracle.LONGITUDE))
    np.testing.assert_array_equal(py, oracle.points2position(lat, ay, oracle.LATITUDE))
    base = wl.base_field()
    f = np.stack([base + np.float32(0.01 * k) for k in range(5)])
    return wl, px, py, f


@pytest.mark.parametrize("method", [oracle.BILINEAR, oracle.NEAREST, oracle.BICUBIC])
def test_c2_c3_full_grid_matches_oracle(fa, c2, method):
    wl, px, py, f = c2
    plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    got = plan.apply_host(f)
    want = oracle.interpolate_values(method, px, py, f, wl.inX, wl.inY, wl.outX, wl.outY, nthreads=16)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    info = plan.info()
    assert info["undefinedCells"] == int(np.isnan(oracle.interpolate_values(method, px, py, np.zeros((1, wl.inY, wl.inX), np.float32),
                                                                            wl.inX, wl.inY, wl.outX, wl.outY, nthreads=16)).sum())
    # slices are independent: one call over 5 slices == 5 calls, in any order
    perm = [3, 0, 4, 1, 2]
    assert cases.same(plan.apply_host(f[perm]), got[perm])
    assert cases.same(plan.apply_host(f[2:3]), got[2:3])


def test_c2_both_bilinear_kernels_agree_and_constants_survive(fa, c2, monkeypatch, tuning_build):
    wl, px, py, f = c2
    monkeypatch.setenv("FIMEX_AMD_STAGED", "1")
    staged = fa.RegridPlan(oracle.BILINEAR, px, py, wl.inX, wl.inY, wl.outX, wl.outY).apply_host(f)
    monkeypatch.setenv("FIMEX_AMD_STAGED", "0")
    gather = fa.RegridPlan(oracle.BILINEAR, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    assert cases.same(gather.apply_host(f), staged)
    const = np.full((4, wl.inY, wl.inX), 273.15, np.float32)
    out = gather.apply_host(const)
    ok = ~np.isnan(out)
    assert ok.mean() > 0.85
    np.testing.assert_allclose(out[ok], 273.15, rtol=3e-7)


@pytest.mark.parametrize("method", [oracle.BILINEAR, oracle.BICUBIC])
def test_north_star_batch_of_200_slices(fa, c2, method):
    """The headline launch itself: ONE device call over 200 resident slices (9.6 GB in, 3.2 GB out), slices 0, 99 and 199
    against the oracle bit for bit; the three-kernel agreement test above covers the other paths.  All three backward
    kernels also agree with each other on a cheap property: a slice that is another slice plus a constant differs from it."""
    import torch
    wl, px, py, f = c2
    nz = 200
    base = torch.from_numpy(f[0]).cuda()
    d_in = torch.empty((nz, wl.inY, wl.inX), dtype=torch.float32, device="cuda")
    for k0 in range(0, nz, 20):
        off = 0.01 * torch.arange(k0, k0 + 20, dtype=torch.float32, device="cuda")
        d_in[k0:k0 + 20] = base[None] + off[:, None, None]
    d_out = torch.full((nz, wl.outY, wl.outX), -1.0, dtype=torch.float32, device="cuda")
    plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k in (0, 99, 199):
        want = oracle.interpolate_values(method, px, py, d_in[k].cpu().numpy()[None], wl.inX, wl.inY, wl.outX, wl.outY, nthreads=16)[0]
        got = d_out[k].cpu().numpy()
        assert cases.same(got, want), "slice %d: %s" % (k, cases.describe_mismatch(got, want))
    # EVERY slice against the per-lane gather kernels' result for the same batch (a second buffer, compared on the device):
    # round 2's ring overflow corrupted 128 floats of rare tiles in slices the three oracle checks above never looked at
    chk = torch.full((nz, wl.outY, wl.outX), -2.0, dtype=torch.float32, device="cuda")
    plan.apply_gather_device(d_in.data_ptr(), nz, chk.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k0 in range(0, nz, 25):
        a, b = d_out[k0:k0 + 25], chk[k0:k0 + 25]
        same = (a.view(torch.int32) == b.view(torch.int32)) | (a.isnan() & b.isnan())
        assert bool(same.all()), "slices %d..%d differ from the gather kernels in %d cells" % (k0, k0 + 24, int((~same).sum()))
    del chk
    # every slice was written (no cell keeps the -1 it started with) and the NaN pattern is the plan's, slice after slice
    assert not bool((d_out == -1.0).any())
    nan0 = torch.isnan(d_out[0])
    assert all(bool((torch.isnan(d_out[k]) == nan0).all()) for k in (1, 57, 123, 198))
    del d_in, d_out
    torch.cuda.empty_cache()


def test_c4_forward_mean_global_to_lambert(fa):
    """configs[3]: 0.1-degree global lon/lat -> 1500x1500 Lambert, forward methods."""
    wl = workloads.ForwardLambert()
    x, y = wl.source_in_target_metres()
    px = fa.points2position_host(x, wl.x_axis)
    py = fa.points2position_host(y, wl.y_axis)
    np.testing.assert_array_equal(px, oracle.points2position(x, wl.x_axis))
    base = wl.base_field()
    f = np.stack([base, base * np.float32(1.5) - np.float32(3)])
    for method in (oracle.FWD_MEAN, oracle.FWD_SUM, oracle.FWD_MAX, oracle.FWD_MEDIAN, oracle.FWD_UNDEF_MIN):
        plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
        got = plan.apply_host(f)
        want = oracle.forward_interpolate_values(method, px, py, f, wl.inX, wl.inY, wl.outX, wl.outY)
        assert cases.same(got, want), "%d: %s" % (method, cases.describe_mismatch(got, want))
    info = plan.info()
    # the source is coarser than the target: most buckets hold one cell or none
    assert 0 < info["mappedSourceCells"] < wl.inX * wl.inY // 4 and 1 <= info["maxBucket"] <= 8
    assert info["undefinedCells"] > wl.outX * wl.outY // 2


def test_c5_wind_pair_rotation_and_creepfill(fa):
    """configs[4]: u/v on a 3000x3000 polar-stereographic grid -> lat/lon bilinear, rotated, creepfill2d(20, 2)."""
    n = 3000
    stere = "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +R=6371000"
    geo = "+proj=latlong +R=6371000"
    sx = (np.arange(n) - n / 2 + 0.5) * 1000.0           # 1 km grid around the pole
    sy = (np.arange(n) - n / 2 + 0.5) * 1000.0 - 2.6e6   # shifted south: 55-80 N
    ox, oy = 1200, 900
    lon = np.linspace(-25, 25, ox)
    lat = np.linspace(52, 78, oy)
    qx, qy = po.project_axes(geo, stere, np.radians(lon), np.radians(lat))
    px = fa.points2position_host(qx, sx)
    py = fa.points2position_host(qy, sy)
    yy, xx = np.meshgrid(sy, sx, indexing="ij")
    ang = 1e-6 * xx + 2e-6 * yy
    u = (10 * np.cos(ang)).astype(np.float32)[None]
    v = (10 * np.sin(ang)).astype(np.float32)[None]
    u[0, 500:900, 1000:1500] = np.nan  # a hole in both components (e.g. masked land)
    v[0, 500:900, 1000:1500] = np.nan
    from test_oracle_kats import _rotation_matrix
    m = _rotation_matrix(stere, geo, lon, lat, oracle.LONGITUDE, oracle.LATITUDE)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, n, n, ox, oy)
    vec = fa.VectorPlan(m, ox, oy)
    post = [fa.creepfill2d_process(20, 2)]
    gu = fa.regrid_slice_host(plan, u, counterpart=v, vec=vec, isXComponent=True, post=post)
    gv = fa.regrid_slice_host(plan, v, counterpart=u, vec=vec, isXComponent=False, post=post)
    iu = oracle.interpolate_values(oracle.BILINEAR, px, py, u, n, n, ox, oy, nthreads=16)
    iv = oracle.interpolate_values(oracle.BILINEAR, px, py, v, n, n, ox, oy, nthreads=16)
    ru, rv = oracle.vector_reproject_values(m, iu, iv, ox, oy)
    wu = oracle.creepfill2d(ru[0], 20, 2)[0]
    wv = oracle.creepfill2d(rv[0], 20, 2)[0]
    assert cases.same(gu[0], wu), cases.describe_mismatch(gu[0], wu)
    assert cases.same(gv[0], wv), cases.describe_mismatch(gv[0], wv)
    assert np.isnan(ru).any() and not np.isnan(gu).any()  # the fill closed the hole and the outside
    ok = ~np.isnan(ru[0])
    assert cases.same(gu[0][ok], ru[0][ok])               # defined cells are untouched by the fill
    np.testing.assert_allclose(np.hypot(gu[0][ok], gv[0][ok]), np.hypot(iu[0][ok], iv[0][ok]), rtol=1e-5)  # length kept


def test_host_calls_with_slices_beyond_the_staging_buffers(fa):
    """A slice larger than one pinned staging slot (64 MB) leaves the streamed path and is copied whole; smaller ones stream in
    several chunks through the ring of three slots.  Both against the oracle, both orders of magnitude of nz."""
    rng = np.random.default_rng(7)
    inX, inY, outX, outY = 4200, 4100, 300, 200          # 68.9 MB per source slice
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=9)
    f = rng.normal(0, 1, (2, inY, inX)).astype(np.float32)
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, inX, inY, outX, outY)
    got = plan.apply_host(f)
    want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY, nthreads=16)
    assert cases.same(got, want), cases.describe_mismatch(got, want)
    s16 = (f * 1000).astype(np.int16)
    got16 = fa.regrid_slice_typed_host(plan, s16, -32767.0)
    w16 = oracle.interpolation_array2data(oracle.interpolate_values(oracle.BILINEAR, px, py, oracle.data2interpolation_array(s16, -32767.0),
                                                                    inX, inY, outX, outY, nthreads=16), oracle.CDM_SHORT, -32767.0)
    assert np.array_equal(got16, w16.reshape(got16.shape))
    # 11 slices of 22 MB: chunks of two slices, six chunks through three slots
    inX, inY = 2400, 2300
    px, py = cases.backward_positions(inX, inY, outX, outY, seed=10)
    f = rng.normal(0, 1, (11, inY, inX)).astype(np.float32)
    plan = fa.RegridPlan(oracle.BICUBIC, px, py, inX, inY, outX, outY)
    got = plan.apply_host(f)
    want = oracle.interpolate_values(oracle.BICUBIC, px, py, f, inX, inY, outX, outY, nthreads=16)
    assert cases.same(got, want), cases.describe_mismatch(got, want)


@pytest.mark.parametrize("method,tol", [(oracle.BILINEAR, 2e-6), (oracle.BICUBIC, 4e-6), (oracle.NEAREST, 0.0)])
def test_c2_linearity_at_full_size(fa, c2, method, tol):
    """Size-independent property on the whole 2000x2000 target: regrid(a f + b g) = a regrid(f) + b regrid(g) up to float
    rounding (exactly for nearest, which only copies), with the same cells undefined."""
    wl, px, py, f = c2
    rng = np.random.default_rng(5)
    g = rng.normal(0, 3, f[:2].shape).astype(np.float32)
    a, b = np.float32(0.5), np.float32(2.0)   # powers of two: the combination itself is exact
    plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    lhs = plan.apply_host(a * f[:2] + b * g)
    rf, rg = plan.apply_host(f[:2]), plan.apply_host(g)
    rhs = a * rf + b * rg
    assert np.array_equal(np.isnan(lhs), np.isnan(rhs))
    ok = ~np.isnan(lhs)
    scale = np.abs(a * rf[ok]) + np.abs(b * rg[ok]) + 1e-30
    assert np.max(np.abs(lhs[ok] - rhs[ok]) / scale) <= tol


def test_fills_are_idempotent_at_full_size(fa):
    """A filled slice has nothing left to fill: the second call returns its input bit for bit and reports no undefined cell."""
    n = 3000
    f = cases.holes(2, n, n, seed=12, frac=0.2)
    once, n1 = fa.creepfill2d_host(f, 5, 2)
    twice, n2 = fa.creepfill2d_host(once, 5, 2)
    assert not np.isnan(once).any() and all(k > 0 for k in n1) and n2 == [0, 0]
    assert np.array_equal(once.view(np.uint32), twice.view(np.uint32))
    keep = ~np.isnan(f)
    assert np.array_equal(once[keep].view(np.uint32), f[keep].view(np.uint32))  # defined cells are never touched
    filled, m1 = fa.fill2d_host(f, 4.0, 1.6, 30)
    again, m2 = fa.fill2d_host(filled, 4.0, 1.6, 30)
    assert m2 == [0, 0] and np.array_equal(filled.view(np.uint32), again.view(np.uint32))


def test_library_placed_output_batch(fa, c2):
    """fimex_amd_regrid_batch_alloc_device / fimex_amd_regrid_source_batch_alloc_device: the batches the library places hold the same
    results as plain allocations, the probing is reported, and the candidates that were not kept are freed."""
    import torch
    wl, px, py, f = c2
    nz = 12
    st = torch.cuda.current_stream().cuda_stream
    d_in = torch.from_numpy(np.stack([f[k % 5] + np.float32(k) for k in range(nz)])).cuda()
    plan = fa.RegridPlan(oracle.BILINEAR, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    plain = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    plan.apply_device(d_in.data_ptr(), nz, plain.data_ptr(), st)
    free_before = torch.cuda.mem_get_info()[0]
    batch = plan.alloc_batch(d_in.data_ptr(), nz, positions=4, stream=st)
    bi = batch.info
    assert bi["positions"] == 4 and 0 <= bi["chosen"] < 4 and len(bi["msAtPosition"]) == 4
    assert all(t > 0 for t in bi["msAtPosition"]) and bi["msAtPosition"][bi["chosen"]] == min(bi["msAtPosition"])
    assert bi["bytes"] == nz * wl.outX * wl.outY * 4 == bi["bytesHeld"] and bi["bytesProbed"] == 4 * bi["bytes"]
    assert bi["probeSeconds"] > 0 and bi["trimmed"] == 1
    assert free_before - torch.cuda.mem_get_info()[0] < bi["bytes"] + (256 << 20)  # the other candidates were freed
    out = batch.as_tensor()
    out.fill_(-1.0)
    plan.apply_device(d_in.data_ptr(), nz, out.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool(((out.view(torch.int32) == plain.view(torch.int32)) | (out.isnan() & plain.isnan())).all())
    # the source batch from the library: whole allocations tried, the kept one zero-filled; the same results out of it
    src = plan.alloc_source_batch(nz, candidates=3, stream=st)
    si = src.info
    assert si["positions"] == 3 and len(si["msAtPosition"]) == 3 and si["msAtPosition"][si["chosen"]] == min(si["msAtPosition"])
    assert si["bytes"] == nz * wl.inX * wl.inY * 4 == si["bytesHeld"] and si["bytesProbed"] >= 3 * si["bytes"]
    t_src = src.as_tensor()
    assert t_src.shape == (nz, wl.inY, wl.inX) and not bool(t_src.any())
    t_src.copy_(d_in)
    out.fill_(-1.0)
    plan.apply_device(t_src.data_ptr(), nz, out.data_ptr(), st)
    torch.cuda.synchronize()
    assert bool(((out.view(torch.int32) == plain.view(torch.int32)) | (out.isnan() & plain.isnan())).all())
    del t_src
    src.close()
    one = plan.alloc_batch(0, nz, positions=1, stream=st)  # plain allocation: nothing timed, no source needed
    assert one.info["positions"] == 1 and one.info["msAtPosition"] == []
    one.as_tensor().zero_()
    torch.cuda.synchronize()
    del out
    batch.close()
    one.close()
    with pytest.raises(fa.FimexAmdError):
        plan.alloc_batch(d_in.data_ptr(), nz, positions=17, stream=st)


def test_bench_launcher_with_two_ranks_on_one_device():
    """bench.py --gpus 2 started as a fresh child process (it starts its own ranks before touching the GPU): the N > 1 launcher path,
    strong scaling with the chunked overlapped write-back, rehearsed over gloo on this one device -- so that a first RCCL run on an
    8-GPU node is not also the first run of that code on a GPU (the reference's counterpart: src/NetCDF_CDMWriter.cc:632-663)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--one-device", "--backend", "gloo", "--scaling", "strong",
           "--method", "bicubic", "--nz", "8", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["slices_total"] == 8 and d["config"]["slices_per_gpu"] == 4
    assert d["verified_slices"] == [0, 1, 3] and d["verified_all_slices_vs_gather"] is True  # rank 0's block; a failure on any rank exits 1
    assert d["gather"]["write_back_verified"] is True
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


@pytest.mark.parametrize("method", [oracle.BILINEAR, oracle.NEAREST])
@pytest.mark.parametrize("dt", [np.int16, np.uint8])
def test_c2_stored_types_full_grid(fa, c2, method, dt):
    """configs[1] geometry on a variable's stored type (packed shorts, bytes): fimex_amd_regrid_apply_typed_device -- one kernel that
    reads and writes the stored type through the second staged form -- against the oracle's three steps (data2InterpolationArray,
    interpolateValues, interpolationArray2Data; src/CDMInterpolator.cc:115-124, 251-285) on the whole 2000 x 2000 grid, 7 slices;
    with and without fill values in the data, and an odd slice count below and above the staging threshold."""
    import torch
    wl, px, py, f = c2
    info = np.iinfo(dt)
    bad = float(info.min)
    rng = np.random.default_rng(5)
    code = oracle.cdm_type_of(dt)
    plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
    for nz, holes in ((7, True), (3, True), (5, False)):
        scale = (info.max - 3) / 40.0
        g = np.clip(np.round((np.nan_to_num(f[:1], nan=280.0) - 280.0) * scale / 4 + rng.integers(-3, 4, (nz, wl.inY, wl.inX))), info.min + 1, info.max).astype(dt)
        if holes:
            g.reshape(-1)[rng.choice(g.size, g.size // 500, replace=False)] = dt(bad)
        want = oracle.interpolation_array2data(
            oracle.interpolate_values(method, px, py, oracle.data2interpolation_array(g, bad), wl.inX, wl.inY, wl.outX, wl.outY, nthreads=16), code, bad)
        t = torch.from_numpy(g.view(np.uint8)).cuda()
        out = torch.zeros(want.nbytes, dtype=torch.uint8, device="cuda")
        fa.regrid_apply_typed_device(plan, t.data_ptr(), code, nz, bad, out.data_ptr())
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(dt).reshape(want.shape)
        assert np.array_equal(got, want), (np.dtype(dt).name, method, nz, int((got != want).sum()))


def test_small_batch_fills_on_a_busy_device(fa):
    """The fills of small batches run several workgroups per slice that wait for each other (cooperative launch).  While a long
    one-workgroup-per-slice fill of another batch holds most CUs, some of those workgroups are queued behind it for seconds: the
    waits inside the kernels are bounded in WALL time (30 s), not in polls, so the call neither fails nor changes its result.
    A second stream from a second thread runs the long fill; the small batch is filled meanwhile and compared with the same fill
    on the idle device and, for one slice, with the oracle."""
    import threading
    import torch
    nyS, nxS, nzS = 1500, 1500, 8
    small = cases.holes(nzS, nyS, nxS, seed=31, frac=0.2)
    d_ref = torch.from_numpy(small).cuda()
    fa.fill2d_device(d_ref.data_ptr(), nxS, nyS, nzS, 1e-9, 1.6, 100, torch.cuda.current_stream().cuda_stream)
    ref = d_ref.cpu().numpy()
    want0 = oracle.fill2d(small[0], 1e-9, 1.6, 100)[0]
    assert cases.same(ref[0], want0), cases.describe_mismatch(ref[0], want0)
    # the long one: 220 slices of 2000 x 2000 (one workgroup per slice: nearly every CU), thirty times the sweeps
    import time
    big = torch.from_numpy(cases.holes(4, 2000, 2000, seed=32)).cuda().repeat(55, 1, 1).contiguous()
    side = torch.cuda.Stream()
    errors, span = [], {}

    def long_fill():
        try:
            torch.cuda.set_device(0)
            fa.set_device(0)
            span["start"] = time.perf_counter()
            fa.fill2d_device(big.data_ptr(), 2000, 2000, 220, 1e-12, 1.6, 3000, side.cuda_stream)
            span["end"] = time.perf_counter()
        except Exception as e:  # reported by the main thread
            errors.append(e)

    t = threading.Thread(target=long_fill)
    t.start()
    time.sleep(0.5)
    busy_results = []
    for rep in range(2):
        d = torch.from_numpy(small).cuda()
        t0 = time.perf_counter()
        fa.fill2d_device(d.data_ptr(), nxS, nyS, nzS, 1e-9, 1.6, 100, torch.cuda.current_stream().cuda_stream)
        busy_results.append((t0, time.perf_counter(), d.cpu().numpy()))
        d2 = torch.from_numpy(small).cuda()
        fa.creepfill2d_device(d2.data_ptr(), nxS, nyS, nzS, 20, 2, torch.cuda.current_stream().cuda_stream)
    t.join()
    assert not errors, errors
    for _, _, got in busy_results:
        assert cases.same(got, ref)
    # the first small fill was submitted while the long one was running (whether it then ran beside it or queued behind it is
    # the device's business; either way its waits must outlast the long fill)
    assert span["start"] < busy_results[0][0] < span["end"], (span, busy_results[0][:2])
