#!/usr/bin/env python3
"""fill2d sweep time as a function of the number of 64-row bands (systolic kernel), no early exit."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
nx = 3000
for ny in (66, 130, 258, 1026, 2050, 3000):
    h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
    d0 = torch.from_numpy(h[None]).cuda()
    res = {}
    for loops in (20, 60):
        ts = []
        for _ in range(2):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, 1, 1e-12, 1.6, loops, st); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res[loops] = min(ts)
    per_sweep = (res[60] - res[20]) / 40
    bands = (ny - 2 + 63) // 64
    print(json.dumps({"ny": ny, "bands": bands, "ms_per_sweep": per_sweep * 1e3, "fixed_ms": (res[20] - 20 * per_sweep) * 1e3,
                      "us_per_step_if_serial_bands": per_sweep * 1e6 / (((bands + 15) // 16) * (nx + 61))}), flush=True)
