#!/usr/bin/env python3
"""creepfill2d(20, 2) on a land-mask like field WITHOUT salt-and-pepper holes: idle chunks are passed over (CREEP_SKIP)."""
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
nx = ny = 3000
rng = np.random.default_rng(4)
f = cases.field(1, ny, nx, 4, nan_frac=0.0, extremes=False)[0]
y, x = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
mask = np.zeros((ny, nx), bool)
for _ in range(6):
    cy, cx, r = rng.uniform(0, ny), rng.uniform(0, nx), rng.uniform(0.05, 0.2) * nx
    mask |= (y - cy) ** 2 + (x - cx) ** 2 < r * r
f[mask] = np.nan
for nz in (16, 200):
    d0 = torch.from_numpy(f[None]).cuda().repeat(nz, 1, 1).contiguous()
    res = {}
    for skip in ("1", "0"):
        os.environ["FIMEX_AMD_CREEP_SKIP"] = skip
        ts = []
        for _ in range(3):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[skip] = (min(ts), d.cpu().numpy()[0])
    assert np.array_equal(res["1"][1].view(np.uint32), res["0"][1].view(np.uint32))
    print(json.dumps({"nz": nz, "hole_fraction": float(mask.mean()), "seconds_skip": res["1"][0], "seconds_noskip": res["0"][0]}), flush=True)
    del d0
