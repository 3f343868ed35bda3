#!/usr/bin/env python3
"""BASELINE configs[4] end to end on one GPU, device resident: a u/v wind pair on a 3000 x 3000 polar-stereographic grid ->
bilinear regrid of both components onto a lat/lon grid -> CachedVectorReprojection's rotation -> creepfill2d(20, 2) post-process
on both (src/CDMInterpolator.cc:261-284), NZ slices per component, per-stage times from HIP events.
Checks: u^2 + v^2 is kept by the rotation (test/testInterpolation.cc:575-578), the fills close every hole, defined cells stay.
usage: python scripts/bench_c5_chain.py [--nz 16] [--out 3000] [--reps 3]   -> one JSON line"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nz", type=int, default=16)
    ap.add_argument("--out", type=int, default=3000, help="target grid: OUT x OUT lat/lon cells over lon -25..25, lat 52..78")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--tuning-build", action="store_true")
    a = ap.parse_args()
    import torch
    from fimex_amd import capi as fa
    if a.tuning_build:
        fa.use_tuning_build(True)
    fa.load(); fa.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    n, ox, oy, nz = 3000, a.out, a.out, a.nz
    stere = "+proj=stere +lat_0=90 +lon_0=0 +lat_ts=60 +R=6371000"
    geo = "+proj=latlong +R=6371000"
    sx = (np.arange(n) - n / 2 + 0.5) * 1000.0
    sy = (np.arange(n) - n / 2 + 0.5) * 1000.0 - 2.6e6
    lon, lat = np.linspace(-25, 25, ox), np.linspace(52, 78, oy)
    ev = lambda: torch.cuda.Event(enable_timing=True)
    e0 = ev(); e0.record()
    d_px = torch.empty(ox * oy, dtype=torch.float64, device="cuda"); d_py = torch.empty_like(d_px)
    fa.project_axes_device(geo, stere, np.radians(lon), np.radians(lat), d_px.data_ptr(), d_py.data_ptr(), st)
    fa.points2position_device(d_px.data_ptr(), d_px.numel(), sx, fa.PROJ_AXIS, st)
    fa.points2position_device(d_py.data_ptr(), d_py.numel(), sy, fa.PROJ_AXIS, st)
    plan = fa.RegridPlan.from_device(fa.BILINEAR, d_px.data_ptr(), d_py.data_ptr(), d_px.numel(), n, n, ox, oy, st)
    m = fa.get_vector_reproject_matrix_host(stere, geo, lon, lat, fa.LONGITUDE, fa.LATITUDE)
    vec = fa.VectorPlan(m, ox, oy)
    e1 = ev(); e1.record(); torch.cuda.synchronize()
    t_plan = e0.elapsed_time(e1)
    yy, xx = torch.meshgrid(torch.from_numpy(sy).cuda(), torch.from_numpy(sx).cuda(), indexing="ij")
    ang = (1e-6 * xx + 2e-6 * yy).float()
    u0, v0 = 10 * torch.cos(ang), 10 * torch.sin(ang)
    u0[500:900, 1000:1500] = float("nan"); v0[500:900, 1000:1500] = float("nan")  # a hole in both components (masked land)
    d_u = torch.stack([u0 * (1 + 0.01 * k) for k in range(nz)]); d_v = torch.stack([v0 * (1 + 0.01 * k) for k in range(nz)])
    del xx, yy, ang
    d_ru = torch.empty((nz, oy, ox), dtype=torch.float32, device="cuda"); d_rv = torch.empty_like(d_ru)
    stages = {"regrid_u": [], "regrid_v": [], "rotate": [], "creepfill_u": [], "creepfill_v": [], "chain": []}
    checks = {}
    for rep in range(a.reps + 1):
        es = [ev() for _ in range(6)]
        es[0].record()
        plan.apply_device(d_u.data_ptr(), nz, d_ru.data_ptr(), st); es[1].record()
        plan.apply_device(d_v.data_ptr(), nz, d_rv.data_ptr(), st); es[2].record()
        if rep == 0:
            before = torch.hypot(d_ru, d_rv)
        vec.reproject_values_device(d_ru.data_ptr(), d_rv.data_ptr(), nz, st); es[3].record()
        if rep == 0:
            after = torch.hypot(d_ru, d_rv)
            ok = ~before.isnan()
            checks["length_kept_rel"] = float(((after[ok] - before[ok]).abs() / before[ok].clamp_min(1e-6)).max().item())
            defined = ok.clone(); keep_u = d_ru.clone()
        fa.creepfill2d_device(d_ru.data_ptr(), ox, oy, nz, 20, 2, st); es[4].record()
        fa.creepfill2d_device(d_rv.data_ptr(), ox, oy, nz, 20, 2, st); es[5].record()
        torch.cuda.synchronize()
        if rep == 0:
            checks["holes_before_fill"] = int((~defined).sum().item())
            checks["holes_after_fill"] = int(d_ru.isnan().sum().item() + d_rv.isnan().sum().item())
            checks["defined_cells_untouched"] = bool(torch.equal(d_ru[defined], keep_u[defined]))
            del before, after, defined, keep_u, ok
            continue
        for k, name in enumerate(["regrid_u", "regrid_v", "rotate", "creepfill_u", "creepfill_v"]):
            stages[name].append(es[k].elapsed_time(es[k + 1]))
        stages["chain"].append(es[0].elapsed_time(es[5]))
    cells = nz * ox * oy
    rec = {"case": "c5_chain", "workload": "BASELINE configs[4]: u/v %d slices, 3000x3000 polar stereographic -> %dx%d lat/lon, bilinear + rotation + creepfill2d(20, 2), device resident" % (nz, ox, oy),
           "plan_and_matrix_build_ms": t_plan, "ms": {k: float(np.median(v)) for k, v in stages.items()}, "reps": a.reps,
           "Mcells_per_s_chain_both_components": 2 * cells / (float(np.median(stages["chain"])) * 1e-3) / 1e6, "checks": checks,
           "plan": {k: plan.info()[k] for k in ("stagedCells", "tileW", "tileH", "undefinedCells")}}
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
