#!/usr/bin/env python3
"""configs[3] (0.1 deg global -> 1500x1500 Lambert, 100 slices), forward_mean / forward_median under FWD_* switches (tuning build).
usage: python scripts/bench_forward.py "FWD_ZPB=8" "FWD_ZPB=50" ..."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True)
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
fw = workloads.ForwardLambert()
x, y = fw.source_in_target_metres()
dx, dy = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
fa.points2position_device(dx.data_ptr(), dx.numel(), fw.x_axis, fa.PROJ_AXIS, st)
fa.points2position_device(dy.data_ptr(), dy.numel(), fw.y_axis, fa.PROJ_AXIS, st)
nz = 100
d_in = bench.make_slices(torch, fw.base_field(), nz)
d_out = torch.empty((nz, fw.outY, fw.outX), dtype=torch.float32, device="cuda")
ref = {}
for v in sys.argv[1:] or [""]:
    for kv in filter(None, v.split(",")):
        k, val = kv.split("="); os.environ["FIMEX_AMD_" + k] = val
    for mname, m in (("forward_mean", fa.FORWARD_MEAN), ("forward_median", fa.FORWARD_MEDIAN), ("forward_max", fa.FORWARD_MAX)):
        plan = fa.RegridPlan.from_device(m, dx.data_ptr(), dy.data_ptr(), dx.numel(), fw.inX, fw.inY, fw.outX, fw.outY, st)
        for _ in range(2): plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        h = float(torch.nan_to_num(d_out[::7], nan=-1.0).double().sum().item())
        same = ref.setdefault(mname, h) == h
        print(json.dumps({"variant": v, "method": mname, "ms": float(np.median(ts)), "same_result_as_first_variant": same}), flush=True)
    for kv in filter(None, v.split(",")):
        os.environ.pop("FIMEX_AMD_" + kv.split("=")[0], None)
