import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from fimex_amd import capi as fa
import cases, oracle
fa.load(); fa.set_device(0)
st = torch.cuda.current_stream().cuda_stream
inX, inY, outX, outY = 403, 301, 130, 77
nz = 6
px, py = cases.coherent_positions(inX, inY, outX, outY, seed=21, outliers=3)
rng = np.random.default_rng(22)
f = rng.standard_normal((nz, inY, inX)).astype(np.float32)
f *= np.float32(10.0) ** rng.integers(-3, 4, (nz, 1, 1)).astype(np.float32)
f[:, ::53, ::47] = np.nan
junk = []
bad_host = bad_dev = bad_plan = 0
ref = None
for it in range(150):
    # churn memory so that allocations land on dirty pages
    junk.append(torch.randn(int(np.random.randint(1, 64)) * 1024 * 1024 // 4, device="cuda"))
    if len(junk) > 6: junk.pop(0)
    plan = fa.RegridPlan(fa.BICUBIC, px, py, inX, inY, outX, outY, bicubic=fa.BICUBIC_FAST)
    got = plan.apply_host(f)
    d_in = torch.from_numpy(f).cuda(); d_out = torch.full((nz, outY, outX), 7.0, device="cuda")
    plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), st); torch.cuda.synchronize()
    gd = d_out.cpu().numpy()
    if ref is None: ref = got.copy()
    if not cases.same(got, ref): bad_host += 1
    if not cases.same(gd, ref): bad_dev += 1
    plan.close()
print("iterations 150: host-path mismatches %d, device-path mismatches %d" % (bad_host, bad_dev))
