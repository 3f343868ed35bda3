#!/usr/bin/env python3
"""Plan statistics of the benchmark geometry: staged cells per slice against touched / bbox cells."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads
fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
fa.load(); fa.set_device(0)
wl = workloads.BilinearRotatedPole()
lon, lat = wl.target_lonlat()
ax, ay = wl.source_axes_rad()
px = fa.points2position_host(lon, ax, fa.LONGITUDE)
py = fa.points2position_host(lat, ay, fa.LATITUDE)
for method, name in ((1, "bilinear"), (2, "bicubic")):
    for tw in ("32", "64", "128"):
        os.environ["FIMEX_AMD_STAGE_TW"] = tw
        plan = fa.RegridPlan(method, px, py, wl.inX, wl.inY, wl.outX, wl.outY)
        info = plan.info()
        print(json.dumps({"method": name, "STAGE_TW": int(tw), "tile": [info["tileW"], info["tileH"]], "stagedCells": info["stagedCells"],
                          "sourceCells": wl.inX * wl.inY, "touched": int(workloads.touched_source_cells(px, py, wl.inX, wl.inY, 2)) if method == 1 and tw == "64" else None,
                          "staged_over_source": info["stagedCells"] / (wl.inX * wl.inY)}), flush=True)
