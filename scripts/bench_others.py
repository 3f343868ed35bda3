#!/usr/bin/env python3
"""Secondary measurements on one GPU (the headline metric is bench.py): every other kernel of the path at its
BASELINE.json configuration, device-resident, HIP-event timed, with the algorithmic bytes of SURVEY 8d.
Prints one JSON object per kernel.  usage: python scripts/bench_others.py [--quick]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK = 8000.0

def timed(torch, fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--quick", action="store_true"); args = ap.parse_args()
    import torch
    from fimex_amd import capi as fa
    import workloads, bench
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cases
    fa.load(); fa.set_device(0)
    st = torch.cuda.current_stream().cuda_stream
    out = []
    def emit(name, ms, mn, bytes_alg, cells, extra=None):
        r = {"kernel": name, "ms_median": ms, "ms_min": mn, "Mcells_per_s": cells / ms / 1e3 if cells else None,
             "algorithmic_GB": bytes_alg / 1e9 if bytes_alg else None,
             "achieved_GBps": bytes_alg / ms / 1e6 if bytes_alg else None, "frac_of_8TBps": bytes_alg / ms / 1e6 / PEAK if bytes_alg else None}
        if extra: r.update(extra)
        print(json.dumps(r), flush=True)
    # --- C2/C3 geometry: nearest, bilinear, bicubic at nz = 200
    wl = workloads.BilinearRotatedPole()
    nz = 50 if args.quick else 200
    d_in = bench.make_slices(torch, wl.base_field(), nz)
    d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    for mname, m in (("nearest", fa.NEAREST_NEIGHBOR), ("bilinear", fa.BILINEAR), ("bicubic", fa.BICUBIC)):
        plan, px, py = bench.build_plan(fa, torch, wl, m, st)
        info = plan.info()
        ms, mn = timed(torch, lambda: plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st))
        b = nz * 4 * (wl.inX * wl.inY + wl.outX * wl.outY) + info["planBytes"]
        emit(mname + "_apply nz=%d 4000x3000->2000x2000" % nz, ms, mn, b, nz * wl.outX * wl.outY, {"plan_bytes": info["planBytes"]})
        del plan
    # --- rotation on the 2000x2000 target, nz = 200 (u and v in place)
    m4 = cases.rotation_matrix(wl.outX, wl.outY, seed=1)
    vec = fa.VectorPlan(m4, wl.outX, wl.outY)
    d_u = torch.randn((nz, wl.outY, wl.outX), device="cuda"); d_v = torch.randn((nz, wl.outY, wl.outX), device="cuda")
    ms, mn = timed(torch, lambda: vec.reproject_values_device(d_u.data_ptr(), d_v.data_ptr(), nz, st))
    n = wl.outX * wl.outY
    emit("rotate_values nz=%d 2000x2000" % nz, ms, mn, nz * 16 * n + 16 * n, nz * n)
    ms, mn = timed(torch, lambda: vec.reproject_direction_device(d_u.data_ptr(), nz, st))
    emit("rotate_direction nz=%d 2000x2000" % nz, ms, mn, nz * 8 * n + 8 * n, nz * n)
    ms, mn = timed(torch, lambda: fa.bad2nan_device(d_u.data_ptr(), d_u.numel(), 1e30, st))
    emit("bad2nan %d floats (no fill values met: read only, groups are written back only where one was replaced)" % d_u.numel(), ms, mn, 4 * d_u.numel(), d_u.numel())
    # --- typed slice edges (n1) and 1-D blends (n4) on the same 800 M values
    d_s = (d_u * 100).to(torch.int16)
    ms, mn = timed(torch, lambda: fa.data2interpolation_device(d_s.data_ptr(), fa.CDM_SHORT, d_s.numel(), -32767.0, d_u.data_ptr(), st))
    emit("data2interpolation short->float %d values" % d_s.numel(), ms, mn, 6 * d_s.numel(), d_s.numel())
    ms, mn = timed(torch, lambda: fa.interpolation2data_device(d_u.data_ptr(), d_u.numel(), fa.CDM_SHORT, -32767.0, d_s.data_ptr(), st))
    emit("interpolation2data float->short %d values" % d_s.numel(), ms, mn, 6 * d_s.numel(), d_s.numel())
    ms, mn = timed(torch, lambda: fa.get_values_1d_device(fa.BLEND_LINEAR, d_u.data_ptr(), d_v.data_ptr(), d_in.data_ptr(), d_u.numel(), 0., 1., .3, st))
    emit("linear blend of two fields, %d values" % d_u.numel(), ms, mn, 12 * d_u.numel(), d_u.numel())
    del d_u, d_v, d_in, d_out, d_s
    # --- C4: forward methods, 3600x1800 -> 1500x1500 Lambert
    fw = workloads.ForwardLambert()
    x, y = fw.source_in_target_metres()
    dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    fa.points2position_device(dx.data_ptr(), dx.numel(), fw.x_axis, fa.PROJ_AXIS, st)
    fa.points2position_device(dy.data_ptr(), dy.numel(), fw.y_axis, fa.PROJ_AXIS, st)
    # bucket occupancy: source cells per target cell, RoundAndClamp positions (half away from zero)
    rx = torch.copysign(torch.floor(dx.abs() + 0.5), dx).long()
    ry = torch.copysign(torch.floor(dy.abs() + 0.5), dy).long()
    inside = (rx >= 0) & (rx < fw.outX) & (ry >= 0) & (ry < fw.outY)
    per_target = torch.bincount((ry * fw.outX + rx)[inside], minlength=fw.outX * fw.outY)
    hist = torch.bincount(per_target.clamp(max=8), minlength=9).tolist()
    occupancy = {("%d" % k if k < 8 else "8+"): hist[k] for k in range(9)}
    del rx, ry, inside, per_target
    nzf = 20 if args.quick else 100
    d_in = bench.make_slices(torch, fw.base_field(), nzf)
    d_out = torch.empty((nzf, fw.outY, fw.outX), dtype=torch.float32, device="cuda")
    for mname, m in (("forward_mean", fa.FORWARD_MEAN), ("forward_median", fa.FORWARD_MEDIAN), ("forward_max", fa.FORWARD_MAX)):
        t0 = time.perf_counter()
        plan = fa.RegridPlan.from_device(m, dx.data_ptr(), dy.data_ptr(), dx.numel(), fw.inX, fw.inY, fw.outX, fw.outY, st)
        tb = time.perf_counter() - t0
        info = plan.info()
        ms, mn = timed(torch, lambda: plan.apply_device(d_in.data_ptr(), nzf, d_out.data_ptr(), st))
        b = nzf * 4 * (fw.inX * fw.inY + fw.outX * fw.outY) + info["planBytes"]
        bt = nzf * 4 * (info["mappedSourceCells"] + fw.outX * fw.outY) + info["planBytes"]
        emit(mname + " nz=%d 3600x1800->1500x1500" % nzf, ms, mn, b, nzf * fw.inX * fw.inY,
             {"plan_build_s": tb, "frac_if_only_mapped_source_cells_counted": bt / ms / 1e6 / PEAK, "mapped_source_cells": info["mappedSourceCells"], "max_bucket": info["maxBucket"],
              "empty_targets": info["undefinedCells"], "bucket_occupancy": occupancy, "note": "Mcells/s counts SOURCE cells"})
    del d_in, d_out
    # --- C5: fills on 3000x3000 slices with land-mask like holes
    nx = ny = 1000 if args.quick else 3000
    nzh = 8 if args.quick else 16
    holes = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
    d_h = torch.from_numpy(np.stack([holes] * nzh)).cuda()
    for name, fn in (("creepfill2d(20,2)", lambda d: fa.creepfill2d_device(d.data_ptr(), nx, ny, nzh, 20, 2, st)),
                     ("fill2d(4,1.6,100)", lambda d: fa.fill2d_device(d.data_ptr(), nx, ny, nzh, 4.0, 1.6, 100, st))):
        d = d_h.clone(); torch.cuda.synchronize()
        t0 = time.perf_counter(); nch = fn(d); torch.cuda.synchronize(); t = time.perf_counter() - t0
        emit("%s nz=%d %dx%d" % (name, nzh, nx, ny), t * 1e3, t * 1e3, None, nzh * nx * ny,
             {"undefined_cells_per_slice": nch[0], "note": "wall time of one call incl. workspace allocation; no roofline claim (iteration dependent)"})

if __name__ == "__main__":
    main()
