#!/usr/bin/env python3
"""creepfill2d(20, 2) on small fields with one undefined corner (the rectangles of a decomposed fill): several workgroups per slice
against one (FIMEX_AMD_FILL_MULTI, tuning build).  usage: python scripts/sweep_creep_small.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fimex_amd import capi as fa
fa.use_tuning_build(True); fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
os.environ["FIMEX_AMD_CREEP_RECTS"] = "0"
for (nx, ny, nz) in ((212, 550, 32), (536, 450, 16), (212, 550, 8), (1000, 1000, 16)):
    yy, xx = np.mgrid[0:ny, 0:nx]
    f = (280 + np.sin(xx * 0.01) + np.cos(yy * 0.02)).astype(np.float32)
    f[(yy * (nx / ny) + xx) < nx * 0.8] = np.nan
    h = torch.from_numpy(np.stack([f] * nz)).cuda()
    d = h.clone()
    for multi in ("1", "0"):
        os.environ["FIMEX_AMD_FILL_MULTI"] = multi
        ts = []
        for r in range(4):
            d.copy_(h); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st); b.record(); torch.cuda.synchronize()
            if r: ts.append(a.elapsed_time(b))
        print(json.dumps({"nx": nx, "ny": ny, "nz": nz, "multi": multi, "ms": round(float(np.median(ts)), 3)}), flush=True)
