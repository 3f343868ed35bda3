import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from fimex_amd import capi as fa
import cases
fa.use_tuning_build(True)
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
nx = ny = 3000
h = cases.holes(1, ny, nx, seed=4, frac=0.3)[0]
for nz in (200, 64, 32):
    d0 = torch.from_numpy(np.stack([h] * nz)).cuda()
    for env in ({"FILL_GEOMETRY": "1"}, {"FILL_GEOMETRY": "2"}, {}):
        for k, v in env.items(): os.environ["FIMEX_AMD_" + k] = v
        ts = []
        for _ in range(2):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.fill2d_device(d.data_ptr(), nx, ny, nz, 1e-9, 1.6, 100, st); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t2 = []
        for _ in range(2):
            d = d0.clone(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fa.creepfill2d_device(d.data_ptr(), nx, ny, nz, 20, 2, st); torch.cuda.synchronize(); t2.append(time.perf_counter() - t0)
        for k in env: os.environ.pop("FIMEX_AMD_" + k, None)
        print(json.dumps({"nz": nz, "env": env, "fill2d_s": min(ts), "creepfill_s": min(t2)}), flush=True)
    del d0
