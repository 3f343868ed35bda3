#!/usr/bin/env python3
"""Device-resident regrid of packed shorts (benchmark geometry, nz = 200): one kernel on the stored type vs the three passes."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
nz = 200
d_f = bench.make_slices(torch, wl.base_field(), nz)
d_in = (torch.nan_to_num(d_f, nan=-327.67) * 100).to(torch.int16)
del d_f
d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.int16, device="cuda")
for mname, m in (("bilinear", fa.BILINEAR), ("nearest", fa.NEAREST_NEIGHBOR), ("bicubic", fa.BICUBIC)):
    plan, _, _ = bench.build_plan(fa, torch, wl, m, st)
    for fused, label in (("1", "one kernel on shorts (LDS-staged)"), ("1,nbuf3", "one kernel on shorts (LDS-staged, ring of 3)"),
                         ("1,gather", "one kernel on shorts (gather form)"), ("0", "to float + regrid + from float")):
        os.environ["FIMEX_AMD_TYPED_FUSED"] = "2" if "gather" in fused else fused[0]
        os.environ["FIMEX_AMD_TYPED_STAGED"] = "0" if "gather" in fused else "1"
        os.environ["FIMEX_AMD_TYPED_NBUF"] = "3" if "nbuf3" in fused else "2"
        for _ in range(2): fa.regrid_apply_typed_device(plan, d_in.data_ptr(), fa.CDM_SHORT, nz, -32767.0, d_out.data_ptr(), st)
        torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fa.regrid_apply_typed_device(plan, d_in.data_ptr(), fa.CDM_SHORT, nz, -32767.0, d_out.data_ptr(), st); b.record()
            torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
        ms = float(np.median(ts))
        print(json.dumps({"method": mname, "path": label,
                          "ms": ms, "Mcells_per_s": nz * wl.outX * wl.outY / ms / 1e3,
                          "algorithmic_GB_on_shorts": nz * 2 * (wl.inX * wl.inY + wl.outX * wl.outY) / 1e9}), flush=True)
    plan.close()
