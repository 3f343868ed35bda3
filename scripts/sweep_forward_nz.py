#!/usr/bin/env python3
"""Dense forward mapping (0.1-degree global -> 1/k-degree global) at short batches: the LDS-staged kernel against the lane kernels
(FIMEX_AMD_FWD_TILED=0 at launch, tuning build).  usage: python scripts/sweep_forward_nz.py [k ...]   (k = 0: configs[3])"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True); fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
fw = workloads.ForwardLambert()
lon, lat = np.meshgrid(fw.src_lon, fw.src_lat)
for k in [int(v) for v in sys.argv[1:]] or [1, 4]:
    os.environ["FIMEX_AMD_FWD_TILED"] = "1"  # read at plan creation too
    if k == 0:  # BASELINE configs[3]: the global source onto the 1500 x 1500 Lambert grid (sparse: most targets empty, buckets of one cell)
        x, y = fw.source_in_target_metres()
        px = workloads.axis_positions_numpy(x, fw.x_axis); py = workloads.axis_positions_numpy(y, fw.y_axis)
        ox, oy = fw.outX, fw.outY
    else:
        ox, oy = 360 * k, 180 * k
        tx = (np.arange(ox) + 0.5) / k - 180.0; ty = (np.arange(oy) + 0.5) / k - 90.0
        px = workloads.axis_positions_numpy(lon.ravel(), tx); py = workloads.axis_positions_numpy(lat.ravel(), ty)
    plan = fa.RegridPlan(fa.FORWARD_MEAN, px, py, fw.inX, fw.inY, ox, oy)
    d_in = bench.make_slices(torch, fw.base_field(), 32)
    d_out = torch.empty((32, oy, ox), dtype=torch.float32, device="cuda")
    for nz in (1, 2, 4, 8, 16, 32):
        rec = {"k": k, "nz": nz}
        for name, env in (("tiled", "1"), ("lanes", "0")):
            os.environ["FIMEX_AMD_FWD_TILED"] = env
            ts = []
            for r in range(8):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
                if r >= 3: ts.append(a.elapsed_time(b))
            rec[name + "_ms"] = round(float(np.median(ts)), 4)
        print(json.dumps(rec), flush=True)
