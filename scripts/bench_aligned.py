#!/usr/bin/env python3
"""How much of the bilinear launch time is due to segment-end lines shared between neighbouring tiles?  Same kernel, same sizes
(4000x3000 -> 2000x2000, nz = 200) on a synthetic plan whose tiles own whole 128-byte lines: px = 2 i + 0.25, py = 1.5 j + 0.25."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True)  # FIMEX_AMD_<NAME> switches select the kernel shape
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
nz = 200
d_in = bench.make_slices(torch, wl.base_field(), nz)
d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
def timed(plan):
    for _ in range(3): plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.mean(ts))
i, j = np.meshgrid(np.arange(wl.outX), np.arange(wl.outY))
for name, px, py in (("aligned 2x1.5", 2.0 * i + 0.25, 1.5 * j + 0.25), ("misaligned 2x1.5 (+37 cells)", 2.0 * i + 37.25, 1.5 * j + 0.25),
                     ("1.73x1.73 unrotated", 1.73 * i + 0.3, 1.49 * j + 0.3)):
    plan = fa.RegridPlan(fa.BILINEAR, px.ravel(), py.ravel(), wl.inX, wl.inY, wl.outX, wl.outY)
    info = plan.info()
    ms = timed(plan)
    touchedBytes = nz * 4 * (info["stagedCells"] + wl.outX * wl.outY)
    print(json.dumps({"plan": name, "ms": ms, "staged_cells": info["stagedCells"], "tile": [info["tileW"], info["tileH"]],
                      "staged_plus_out_GB": touchedBytes / 1e9, "TBps_on_staged_bytes": touchedBytes / ms / 1e9}), flush=True)
    plan.close()
plan, _, _ = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
info = plan.info(); ms = timed(plan)
touchedBytes = nz * 4 * (info["stagedCells"] + wl.outX * wl.outY)
print(json.dumps({"plan": "C2 rotated pole", "ms": ms, "staged_cells": info["stagedCells"], "staged_plus_out_GB": touchedBytes / 1e9,
                  "TBps_on_staged_bytes": touchedBytes / ms / 1e9}), flush=True)
