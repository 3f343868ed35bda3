#!/usr/bin/env python3
"""One secondary kernel of the path at its BASELINE.json configuration, device resident, a few timed calls: the program
that scripts/collect_case_profiles.py puts behind `rocprofv3 ... --` (and a stand-alone HIP-event measurement).
usage: python scripts/run_case.py <case> [--reps N]
cases: forward_mean_c4 forward_median_c4 rotate_values rotate_direction fill2d_nz16 creepfill_nz16 bilinear_short bicubic_short
       bicubicfast_short typed_short_bilinear typed_short_nearest typed_uchar_bilinear typed_short_bilinear_short"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tuning-build", action="store_true", help="libfimex_amd_tuning.so: reads the FIMEX_AMD_<NAME> experiment switches")
    a = ap.parse_args()
    import torch
    from fimex_amd import capi as fa
    import workloads, bench, cases
    if a.tuning_build:
        fa.use_tuning_build(True)
    fa.load(); fa.set_device(0)
    st = torch.cuda.current_stream().cuda_stream

    def timed(fn, setup=None, warm=2):
        ts = []
        for i in range(warm + a.reps):
            if setup: setup()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            if i >= warm: ts.append(e0.elapsed_time(e1))
        return ts

    r = {"case": a.case, "reps": a.reps, "warm": 2}
    if a.case.startswith("forward_"):
        fw = workloads.ForwardLambert()
        x, y = fw.source_in_target_metres()
        dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        fa.points2position_device(dx.data_ptr(), dx.numel(), fw.x_axis, fa.PROJ_AXIS, st)
        fa.points2position_device(dy.data_ptr(), dy.numel(), fw.y_axis, fa.PROJ_AXIS, st)
        rx = torch.copysign(torch.floor(dx.abs() + 0.5), dx).long()
        ry = torch.copysign(torch.floor(dy.abs() + 0.5), dy).long()
        inside = (rx >= 0) & (rx < fw.outX) & (ry >= 0) & (ry < fw.outY)
        granules = int(torch.unique(torch.nonzero(inside).flatten() // 16).numel())  # distinct 64-byte pieces of a source slice that hold a mapped cell
        m = {"forward_mean_c4": fa.FORWARD_MEAN, "forward_median_c4": fa.FORWARD_MEDIAN}[a.case]
        nz = 100
        d_in = bench.make_slices(torch, fw.base_field(), nz)
        d_out = torch.empty((nz, fw.outY, fw.outX), dtype=torch.float32, device="cuda")
        plan = fa.RegridPlan.from_device(m, dx.data_ptr(), dy.data_ptr(), dx.numel(), fw.inX, fw.inY, fw.outX, fw.outY, st)
        info = plan.info()
        ts = timed(lambda: plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st))
        out = fw.outX * fw.outY
        r.update(workload="BASELINE configs[3]: 3600x1800 global 0.1 deg -> 1500x1500 Lambert, %d slices" % nz, kernel_pattern="forward_apply",
                 bytes_survey_8d=nz * 4 * (fw.inX * fw.inY + out) + info["planBytes"],
                 bytes_must_move=nz * (64 * granules + 4 * out) + info["planBytes"],
                 mapped_source_cells=info["mappedSourceCells"], source_granules_64B=granules, plan_bytes=info["planBytes"], cells=nz * fw.inX * fw.inY,
                 note="bytes_must_move = per slice the 64-byte pieces of the source that hold a mapped cell + the output, + the CSR plan once; "
                      "SURVEY 8d's formula charges the whole source slice although 4 % of it maps into the target")
    elif a.case.startswith("forwarddense"):
        # a dense forward mapping: the same 0.1-degree global source onto a 1-degree global lat/lon grid, 100 source cells per bucket
        # (the wave-per-bucket kernels: coalesced index loads, ballot, ordered lane fold)
        # forwarddense_<aggregate>: onto 1 degree (100 cells per bucket); forwarddense<k>_<aggregate>: onto 1/k degree (k = 2: 25 cells, k = 4: 4-9 cells)
        fw = workloads.ForwardLambert()
        head, aggr = a.case.split("_")
        k = int(head[len("forwarddense"):] or 1)
        ox, oy = 360 * k, 180 * k
        tx = (np.arange(ox) + 0.5) / k - 180.0; ty = (np.arange(oy) + 0.5) / k - 90.0
        lon, lat = np.meshgrid(fw.src_lon, fw.src_lat)
        px = workloads.axis_positions_numpy(lon.ravel(), tx); py = workloads.axis_positions_numpy(lat.ravel(), ty)
        m = {"mean": fa.FORWARD_MEAN, "max": fa.FORWARD_MAX, "median": fa.FORWARD_MEDIAN, "sum": fa.FORWARD_SUM}[aggr]
        nz = 100
        d_in = bench.make_slices(torch, fw.base_field(), nz)
        d_out = torch.empty((nz, oy, ox), dtype=torch.float32, device="cuda")
        plan = fa.RegridPlan(m, px, py, fw.inX, fw.inY, ox, oy)
        info = plan.info()
        ts = timed(lambda: plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st))
        r.update(workload="0.1-degree global 3600x1800 -> 1/%d-degree global %dx%d, forward, %d slices, %d source cells per bucket at most" % (k, ox, oy, nz, info["maxBucket"]),
                 kernel_pattern="forward_apply", bytes_survey_8d=nz * 4 * (fw.inX * fw.inY + ox * oy) + info["planBytes"], cells=nz * fw.inX * fw.inY,
                 mapped_source_cells=info["mappedSourceCells"], plan_bytes=info["planBytes"], staged_cells_per_slice=info["stagedCells"], tile=[info["tileW"], info["tileH"]])
        r["bytes_must_move"] = r["bytes_survey_8d"]
    elif a.case.startswith("typedforward"):
        # typedforward<k>_<aggregate>: the dense forward mapping onto 1/k degree on PACKED SHORTS (SURVEY 8f n1 x a7)
        fw = workloads.ForwardLambert()
        head, aggr = a.case.split("_")
        k = int(head[len("typedforward"):] or 1)
        ox, oy = 360 * k, 180 * k
        tx = (np.arange(ox) + 0.5) / k - 180.0; ty = (np.arange(oy) + 0.5) / k - 90.0
        lon, lat = np.meshgrid(fw.src_lon, fw.src_lat)
        px = workloads.axis_positions_numpy(lon.ravel(), tx); py = workloads.axis_positions_numpy(lat.ravel(), ty)
        m = {"mean": fa.FORWARD_MEAN, "max": fa.FORWARD_MAX, "sum": fa.FORWARD_SUM}[aggr]
        nz = 100
        d_in = bench.make_slices(torch, fw.base_field(), nz)
        d_s = ((d_in - 280) * 100).nan_to_num(-32767).to(torch.int16)
        del d_in
        d_o = torch.empty((nz, oy, ox), dtype=torch.int16, device="cuda")
        plan = fa.RegridPlan(m, px, py, fw.inX, fw.inY, ox, oy)
        info = plan.info()
        ts = timed(lambda: fa.regrid_apply_typed_device(plan, d_s.data_ptr(), fa.CDM_SHORT, nz, -32767.0, d_o.data_ptr(), st))
        r.update(workload="0.1-degree global 3600x1800 packed shorts -> 1/%d-degree global %dx%d, forward, %d slices, %d source cells per bucket at most" % (k, ox, oy, nz, info["maxBucket"]),
                 kernel_pattern="forward_apply", bytes_survey_8d=nz * 2 * (fw.inX * fw.inY + ox * oy) + info["planBytes"], cells=nz * fw.inX * fw.inY)
        r["bytes_must_move"] = r["bytes_survey_8d"]
    elif a.case in ("rotate_values", "rotate_direction"):
        wl = workloads.BilinearRotatedPole()
        nz, n = 200, wl.outX * wl.outY
        vec = fa.VectorPlan(cases.rotation_matrix(wl.outX, wl.outY, seed=1), wl.outX, wl.outY)
        d_u = torch.randn((nz, wl.outY, wl.outX), device="cuda"); d_v = torch.randn((nz, wl.outY, wl.outX), device="cuda")
        if a.case == "rotate_values":
            ts = timed(lambda: vec.reproject_values_device(d_u.data_ptr(), d_v.data_ptr(), nz, st))
            r.update(kernel_pattern="rotate_values", bytes_survey_8d=nz * 16 * n + 16 * n)
        else:
            ts = timed(lambda: vec.reproject_direction_device(d_u.data_ptr(), nz, st))
            r.update(kernel_pattern="rotate_direction", bytes_survey_8d=nz * 8 * n + 8 * n)
        r.update(workload="rotation of %d slices of 2000x2000 in place (configs[4] step)" % nz, cells=nz * n)
        r["bytes_must_move"] = r["bytes_survey_8d"]
    elif a.case in ("fill2d_patchy_nz16", "creepfill_patchy_nz16"):
        # a regridded field as configs[4] produces it: two corners outside the source domain and one masked block, everything else defined
        nx = ny = 3000; nz = 16
        yy, xx = np.mgrid[0:ny, 0:nx]
        base = (280 + 5 * np.sin(xx * 2e-3) * np.cos(yy * 3e-3)).astype(np.float32)
        base[(yy * 0.4 + xx) < 210] = np.nan
        base[(yy * 0.4 + (nx - 1 - xx)) < 210] = np.nan
        base[460:900, 970:1500] = np.nan
        d_h = torch.from_numpy(np.stack([base * (1 + 0.001 * k) for k in range(nz)])).cuda()
        d = d_h.clone()
        def reset(): d.copy_(d_h)
        if a.case.startswith("fill2d"):
            ts = timed(lambda: fa.fill2d_device(d.data_ptr(), nx, ny, nz, 0.01, 1.6, 100, st), reset)
            r.update(workload="mifi_fill2d_f(0.01, 1.6, 100) on %d slices of 3000x3000: two undefined corners and one block (3.4 %% undefined)" % nz, kernel_pattern="fill")
        else:
            ts = timed(lambda: fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st), reset)
            r.update(workload="mifi_creepfill2d_f(20, 2) on %d slices of 3000x3000: two undefined corners and one block" % nz, kernel_pattern="fill")
        r.update(cells=nz * nx * ny, bytes_survey_8d=None, bytes_must_move=None, undefined_per_slice=int(np.isnan(base).sum()),
                 note="iteration dependent: time, no roofline claim (SURVEY 8d)")
    elif a.case in ("fill2d_nz16", "creepfill_nz16"):
        nx = ny = 3000; nz = 16
        holes = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
        d_h = torch.from_numpy(np.stack([holes] * nz)).cuda()
        d = d_h.clone()
        def reset(): d.copy_(d_h)
        if a.case == "fill2d_nz16":
            ts = timed(lambda: fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-9, 1.6, 100, st), reset)
            r.update(workload="mifi_fill2d_f(relaxCrit 1e-9, corrEff 1.6, maxLoop 100: no early exit) on %d slices of 3000x3000, 30 %% holes" % nz, kernel_pattern="fill")
        else:
            ts = timed(lambda: fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st), reset)
            r.update(workload="mifi_creepfill2d_f(repeat 20, setWeight 2) on %d slices of 3000x3000, 30 %% holes" % nz, kernel_pattern="fill")
        r.update(cells=nz * nx * ny, bytes_survey_8d=None, bytes_must_move=None, note="iteration dependent: time and sweeps, no roofline claim (SURVEY 8d)")
    elif a.case in ("bilinear_short", "bicubic_short", "bicubicfast_short", "typed_short_bilinear", "typed_short_nearest", "typed_uchar_bilinear", "typed_short_bilinear_short", "typed_short_bicubic"):
        wl = workloads.BilinearRotatedPole()
        typed = a.case.startswith("typed_")
        nz = 200 if (typed and not a.case.endswith("_short")) else 25
        method = fa.BICUBIC if (a.case.startswith("bicubic") or a.case.endswith("bicubic")) else (fa.NEAREST_NEIGHBOR if a.case.endswith("nearest") else fa.BILINEAR)
        plan, px, py = bench.build_plan(fa, torch, wl, method, st, bicubic=fa.BICUBIC_FAST if a.case == "bicubicfast_short" else None)
        info = plan.info()
        d_in = bench.make_slices(torch, wl.base_field(), nz)
        out = wl.outX * wl.outY
        if typed:
            if "uchar" in a.case:
                d_s = ((d_in - 200) * 1.2).nan_to_num(0.0).clamp(0, 255).to(torch.uint8); code, bad, eb = fa.CDM_UCHAR, 0.0, 1
            else:
                d_s = ((d_in - 280) * 100).nan_to_num(-32767).to(torch.int16); code, bad, eb = fa.CDM_SHORT, -32767.0, 2
            d_o = torch.empty((nz, wl.outY, wl.outX), dtype=d_s.dtype, device="cuda")
            del d_in
            ts = timed(lambda: fa.regrid_apply_typed_device(plan, d_s.data_ptr(), code, nz, bad, d_o.data_ptr(), st))
            # the plan bytes of the stored-type form: LDS offsets (4 B) and the two fractions (8 B) per cell, the chunk lists are small
            plan_bytes = (12 if method == fa.BILINEAR else (24 if method == fa.BICUBIC else 4)) * out
            r.update(workload="%s, %d slices, %s, fused conversion (SURVEY 8f n1)" % ("packed shorts" if eb == 2 else "unsigned bytes", nz, a.case.split("_")[2]),
                     kernel_pattern="_apply", bytes_survey_8d=nz * eb * (wl.inX * wl.inY + out) + plan_bytes)
        else:
            d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
            ts = timed(lambda: plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st))
            r.update(workload="BASELINE configs[2] per-GPU share: 25 slices, %s" % a.case.split("_")[0], kernel_pattern="_apply",
                     bytes_survey_8d=nz * 4 * (wl.inX * wl.inY + out) + info["planBytes"])
        r.update(cells=nz * out, bytes_must_move=r["bytes_survey_8d"])
    else:
        raise SystemExit("unknown case " + a.case)
    ms = float(np.mean(ts))
    med = float(np.median(ts))
    r.update(ms_avg=ms, ms_median=med, ms_min=float(np.min(ts)), ms_all=ts, Mcells_per_s=r["cells"] / ms / 1e3)
    for k in ("bytes_survey_8d", "bytes_must_move"):
        if r.get(k):
            r["frac_of_8TBps_" + k] = r[k] / ms / 1e6 / PEAK           # on the average of the timed calls
            r["frac_of_8TBps_" + k + "_median"] = r[k] / med / 1e6 / PEAK
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
