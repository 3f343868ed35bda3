#!/usr/bin/env python3
"""The drop-in boundary as the reference calls it: interpolateValues on HOST arrays (fimex_amd_regrid_apply_host),
PCIe included.  Mcells/s for one time step of nz levels of the benchmark geometry."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from fimex_amd import capi as fa
import workloads, bench
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
wl = workloads.BilinearRotatedPole()
plan, px, py = bench.build_plan(fa, torch, wl, fa.BILINEAR, st)
base = wl.base_field()
for nz in (1, 10, 60):
    f = np.ascontiguousarray(np.stack([base + np.float32(0.01 * k) for k in range(nz)]))
    plan.apply_host(f[:1])
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = plan.apply_host(f); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(json.dumps({"entry": "regrid_apply_host", "nz": nz, "ms": t * 1e3, "Mcells_per_s": nz * wl.outX * wl.outY / t / 1e6,
                      "GB_moved": (f.nbytes + out.nbytes) / 1e9, "GBps": (f.nbytes + out.nbytes) / t / 1e9}), flush=True)
    s16 = (f * 50).astype(np.int16)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); o2 = fa.regrid_slice_typed_host(plan, s16, -32767.0); ts.append(time.perf_counter() - t0)
    t = min(ts)
    print(json.dumps({"entry": "regrid_slice_typed_host (short)", "nz": nz, "ms": t * 1e3, "Mcells_per_s": nz * wl.outX * wl.outY / t / 1e6,
                      "GB_moved": (s16.nbytes + o2.nbytes) / 1e9}), flush=True)
