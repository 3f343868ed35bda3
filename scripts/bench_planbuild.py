#!/usr/bin/env python3
"""Plan build of the benchmark geometry (4 M target cells): projection + axis positions + plan tables + rotation matrix,
device resident."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
n = wl.outX * wl.outY
ax, ay = wl.source_axes_rad()
tx, ty = np.radians(wl.target_axes_deg()[0]), np.radians(wl.target_axes_deg()[1])
d = torch.empty(2 * n, dtype=torch.float64, device="cuda")
m = torch.empty(4 * n, dtype=torch.float64, device="cuda")
def build(method):
    fa.project_axes_device(wl.target_proj, wl.source_proj, tx, ty, d.data_ptr(), d.data_ptr() + 8 * n, st)
    fa.points2position_device(d.data_ptr(), n, ax, fa.LONGITUDE, st)
    fa.points2position_device(d.data_ptr() + 8 * n, n, ay, fa.LATITUDE, st)
    return fa.RegridPlan.from_device(method, d.data_ptr(), d.data_ptr() + 8 * n, n, wl.inX, wl.inY, wl.outX, wl.outY, st)
for method, name in ((fa.BILINEAR, "bilinear"), (fa.BICUBIC, "bicubic"), (fa.NEAREST_NEIGHBOR, "nearest")):
    build(method); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); p = build(method); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); p.close()
    print(json.dumps({"step": "project_axes + 2 x points2position + plan tables", "method": name, "cells": n, "ms": min(ts) * 1e3}), flush=True)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    fa.get_vector_reproject_matrix_device(wl.source_proj, wl.target_proj, np.degrees(tx), np.degrees(ty), fa.LONGITUDE, fa.LATITUDE, m.data_ptr(), st)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(json.dumps({"step": "vector reproject matrix", "cells": n, "ms": min(ts[1:]) * 1e3}), flush=True)
# The CPU side of this comparison (numpy projection + C axis positions on one core: 346 + 188 ms, positions equal to 5e-12) was
# measured once through tests' oracle and is recorded in the round-1 notes (profiles/LAB_NOTES_r01_r02.md); scripts do not touch oracle/.
