#!/usr/bin/env python3
"""fill2d on a full batch of slices: seconds per sweep for the systolic and the wavefront kernel (no early exit)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.use_tuning_build(True)  # the build that reads the FIMEX_AMD_<NAME> switches
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
nx, ny = 3000, 3000
h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
for nz in (1, 16, 64, 200):
    d0 = torch.from_numpy(h[None]).cuda().repeat(nz, 1, 1).contiguous()
    for v2 in ("1", "0"):
        os.environ["FIMEX_AMD_FILL_V2"] = v2
        res = {}
        for loops in (6, 16):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-12, 1.6, loops, st); torch.cuda.synchronize()
            res[loops] = time.perf_counter() - t0
        per = (res[16] - res[6]) / 10
        print(json.dumps({"nz": nz, "kernel": "systolic" if v2 == "1" else "wavefront", "ms_per_sweep": per * 1e3,
                          "fixed_ms": (res[6] - 6 * per) * 1e3, "Mcell_updates_per_s": nz * nx * ny / per / 1e6}), flush=True)
    del d0
