import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from fimex_amd import capi as fa
import workloads, bench
fa.use_tuning_build(True); fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole(); nz = 200
d_f = bench.make_slices(torch, wl.base_field(), nz)
d_in = (torch.nan_to_num(d_f, nan=-327.67) * 100).to(torch.int16); del d_f
d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.int16, device="cuda")
variants = ["", "STAGE_ORDER=1", "STAGE_ORDER=1,STAGE_ZPB=50", "STAGE_ORDER=1,STAGE_ZPB=13", "STAGE_ORDER=1,XCD=0", "XCD=2"]
def setenv(v):
    for k in ("STAGE_ORDER", "STAGE_ZPB", "XCD"): os.environ.pop("FIMEX_AMD_" + k, None)
    for kv in filter(None, v.split(",")):
        k, val = kv.split("="); os.environ["FIMEX_AMD_" + k] = val
for mname, m in (("bilinear", fa.BILINEAR), ("nearest", fa.NEAREST_NEIGHBOR), ("bicubic", fa.BICUBIC)):
    plan, _, _ = bench.build_plan(fa, torch, wl, m, st)
    times = {v: [] for v in variants}
    for r in range(5):
        for v in variants:
            setenv(v)
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); fa.regrid_apply_typed_device(plan, d_in.data_ptr(), fa.CDM_SHORT, nz, -32767.0, d_out.data_ptr(), st); b.record()
                torch.cuda.synchronize()
                if r: times[v].append(a.elapsed_time(b))
    for v in variants: print(json.dumps({"method": mname, "variant": v, "ms_median": float(np.median(times[v]))}), flush=True)
    plan.close()
