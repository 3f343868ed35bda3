#!/usr/bin/env python3
"""Headline benchmark: Mcells/s regridded, bilinear, 4000x3000 -> 2000x2000 f32 (BASELINE.json).

One "step" = one pass of the hot path over one batch of slices: ONE launch of the bilinear apply
kernel over NZ time x level slices that already sit in HBM (default NZ = 200, the north-star batch;
the geometry is BASELINE.json configs[1], "C2").  N > 1: one process per GPU, every rank regrids its own
NZ slices with a replicated plan (weak scaling, no data-path collective -- the reference's MPI mode shards
time steps the same way, src/NetCDF_CDMWriter.cc:632-646); the RCCL gather of the finished slices to
rank 0 ("write-back") is timed separately and reported beside the metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_plan(fa, torch, wl, method, stream):
    """Plan build on the GPU: target axes -> geographic lon/lat (projection) -> fractional source indices -> compact plan."""
    ax, ay = wl.source_axes_rad()
    n = wl.outX * wl.outY
    d_px = torch.empty(n, dtype=torch.float64, device="cuda")
    d_py = torch.empty(n, dtype=torch.float64, device="cuda")
    rlon, rlat = wl.target_axes_deg()
    fa.project_axes_device(wl.target_proj, wl.source_proj, np.radians(rlon), np.radians(rlat), d_px.data_ptr(), d_py.data_ptr(), stream)
    fa.points2position_device(d_px.data_ptr(), d_px.numel(), ax, fa.LONGITUDE, stream)
    fa.points2position_device(d_py.data_ptr(), d_py.numel(), ay, fa.LATITUDE, stream)
    plan = fa.RegridPlan.from_device(method, d_px.data_ptr(), d_py.data_ptr(), d_px.numel(),
                                     wl.inX, wl.inY, wl.outX, wl.outY, stream)
    torch.cuda.synchronize()
    return plan, d_px.cpu().numpy(), d_py.cpu().numpy()


def make_slices(torch, base, nz):
    """slice k = base + 0.01 k, resident in HBM ([nz][inY][inX] f32)."""
    d_base = torch.from_numpy(base).cuda()
    d_in = torch.empty((nz,) + base.shape, dtype=torch.float32, device="cuda")
    for k0 in range(0, nz, 16):
        k1 = min(nz, k0 + 16)
        off = 0.01 * torch.arange(k0, k1, dtype=torch.float32, device="cuda")
        d_in[k0:k1] = d_base[None] + off[:, None, None]
    return d_in


def time_launches(torch, fn, steps, warmup, dist_on):
    """W untimed + K timed launches; returns (wall seconds of the K steps, per-launch kernel ms list)."""
    import torch.distributed as dist
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()  # on torch's current stream, the stream the kernel is launched on
        fn()
        b.record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return wall, [a.elapsed_time(b) for a, b in ev]


def cpu_baseline(wl, px, py, base, budget_s):
    """The CPU oracle (restatement of the reference's OpenMP loop, src/CachedInterpolation.cc:125-144) timed on
    this host: one time step of `levels` levels per call, repeated for about budget_s seconds."""
    import oracle
    # the GPU box exposes every host CPU but grants one GPU's share of them (16): more threads only oversubscribe
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, int(os.environ.get("FIMEX_BENCH_CPU_THREADS", "16")))
    levels = 4
    f = np.stack([base + np.float32(0.01 * k) for k in range(levels)])
    cells = levels * wl.outX * wl.outY
    times = []
    t_end = time.perf_counter() + budget_s
    while True:
        t0 = time.perf_counter()
        oracle.interpolate_values(oracle.BILINEAR, px, py, f, wl.inX, wl.inY, wl.outX, wl.outY, nthreads=cores)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() > t_end and len(times) >= 3:
            break
    t1 = time.perf_counter()
    oracle.interpolate_values(oracle.BILINEAR, px, py, f[:1], wl.inX, wl.inY, wl.outX, wl.outY, nthreads=1)
    single = time.perf_counter() - t1
    med = float(np.median(times))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {
        "value": cells / med / 1e6, "unit": "Mcells/s", "cores": cores, "kind": "port", "cpu_model": model, "host_cpus": avail,
        "sample": "%d calls of one time step x %d levels (%.0f Mcells each) of the same 4000x3000->2000x2000 bilinear "
                  "plan, OpenMP over output cells on %d threads, median; 1 thread: %.1f Mcells/s"
                  % (len(times), levels, cells / 1e6, cores, wl.outX * wl.outY / single / 1e6),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nz", type=int, default=200, help="time x level slices per GPU per step")
    ap.add_argument("--method", default="bilinear", choices=["bilinear", "bicubic", "nearest"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-slice and gather measurements")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --one-device rehearses the N > 1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fimex_amd import capi as fa
    import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            import datetime
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend="gloo")
    if args.gpus != world and rank == 0:
        log("note: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world))
    fa.load()
    fa.set_device(local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    method = {"bilinear": fa.BILINEAR, "bicubic": fa.BICUBIC, "nearest": fa.NEAREST_NEIGHBOR}[args.method]
    stencil = {"bilinear": 2, "bicubic": 4, "nearest": 1}[args.method]
    wl = workloads.BilinearRotatedPole()
    t0 = time.perf_counter()
    plan, px, py = build_plan(fa, torch, wl, method, stream)
    t_plan = time.perf_counter() - t0
    info = plan.info()
    base = wl.base_field()
    nz = args.nz
    d_in = make_slices(torch, base, nz)
    d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def step():
        plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), stream)

    wall, kernel_ms = time_launches(torch, step, args.steps, args.warmup, dist_on)
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
    if dist_on:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())
    cells_per_step = nz * wl.outX * wl.outY
    ms_per_step = wall_max / args.steps * 1e3
    value = world * cells_per_step / (wall_max / args.steps) / 1e6

    # roofline of the dominant (only) kernel: algorithmic bytes per launch / average launch duration
    # B_alg = nz*4*(N_src + ox*oy) + B_plan, N_src = source cells in the plan's reduced-domain bounding box
    # (what the reference itself reads, CachedInterpolation.cc:159-200); the count of distinct cells the stencils
    # actually touch is reported beside it (smaller where the target grid is coarser than 2 source cells)
    n_src_touched = workloads.touched_source_cells(px, py, wl.inX, wl.inY, stencil) if rank == 0 else 0
    n_src_bbox = workloads.reduced_domain_cells(px, py, wl.inX, wl.inY) if rank == 0 else 0
    alg_bytes = nz * 4 * (n_src_bbox + wl.outX * wl.outY) + info["planBytes"]
    alg_bytes_touched = nz * 4 * (n_src_touched + wl.outX * wl.outY) + info["planBytes"]
    avg_kernel_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
    traffic = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get("%s_nz%d" % (args.method, nz))
        except Exception:
            traffic = None

    result = {
        "metric": "Mcells/s regridded (%s, 4000x3000->2000x2000 f32)" % args.method,
        "value": value, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] geometry (4000x3000 0.01-deg lon/lat -> 2000x2000 rotated pole, %s), "
                        "%d time x level slices per GPU per step resident in HBM (north-star batch)" % (args.method, nz),
            "slices_per_gpu": nz, "sharding": "slices over GPUs, plan replicated, no data-path collective",
            "plan_build_s": t_plan, "undefined_target_cells": info["undefinedCells"], "border_cells": info["borderCells"],
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "kernel": ("staged_apply<%d, ...>" % (2 if args.method == "bilinear" else 4)) if info.get("stagedCells") else args.method + "_apply",
            "staged_cells_per_slice": info.get("stagedCells"), "tile": [info.get("tileW"), info.get("tileH")],
            "kernel_ms_avg": avg_kernel_ms, "kernel_ms_min": float(np.min(kernel_ms)),
            "algorithmic_bytes_per_launch": alg_bytes,
            "n_src_bbox": n_src_bbox, "n_src_touched": n_src_touched, "plan_bytes": info["planBytes"],
            "frac_if_only_touched_cells_counted": alg_bytes_touched / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        },
    }

    if not args.no_extras:
        # configs[1] proper: one time step, one slice (plan read not amortised over z)
        wall1, k1 = time_launches(torch, lambda: plan.apply_device(d_in.data_ptr(), 1, d_out.data_ptr(), stream),
                                  max(args.steps, 20), args.warmup, False)
        b1 = 4 * (n_src_bbox + wl.outX * wl.outY) + info["planBytes"]
        result["single_slice"] = {"workload": "configs[1], nz = 1 (112 MB working set: served from the 256 MB Infinity Cache "
                                              "when repeated, not from HBM)", "kernel_ms_avg": float(np.mean(k1)),
                                  "Mcells_per_s": wl.outX * wl.outY / (float(np.mean(k1)) * 1e-3) / 1e6,
                                  "achieved_GBps": b1 / (float(np.mean(k1)) * 1e-3) / 1e9}
        # the box's copy ceiling for context: a device-to-device copy of the output's size, read + write bytes
        flat_in, flat_out = d_in.view(-1)[:d_out.numel()], d_out.view(-1)
        _, kc = time_launches(torch, lambda: flat_out.copy_(flat_in), 10, 2, False)
        result["roofline"]["copy_kernel_GBps"] = 8 * d_out.numel() / (float(np.mean(kc)) * 1e-3) / 1e9
        if dist_on:
            # write-back: RCCL gather of every rank's finished slices to rank 0 over xGMI, outside the metric
            from fimex_amd import sharding
            try:
                dist.barrier()
                torch.cuda.synchronize()
                tg = time.perf_counter()
                # (gloo rehearsal: point-to-point on host copies; the measured path is RCCL on device buffers)
                full = sharding.gather_slices(d_out if args.backend == "nccl" else d_out.cpu(), world * nz, dst=0)
                torch.cuda.synchronize()
                dist.barrier()
                tg = time.perf_counter() - tg
                result["gather"] = {"seconds": tg, "bytes_per_peer": d_out.numel() * 4,
                                    "GBps_into_root": (world - 1) * d_out.numel() * 4 / tg / 1e9,
                                    "note": "point-to-point RCCL sends of every rank's output slices to rank 0 "
                                            "(fimex_amd/sharding.py); not part of value"}
                del full
            except Exception as e:  # the write-back is reported beside the metric; the metric stands without it
                result["gather"] = {"seconds": None, "error": repr(e)[:300]}

    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        try:
            result["cpu_baseline"] = cpu_baseline(wl, px, py, base, args.cpu_seconds)
        except Exception as e:  # the oracle is only the reported baseline; the GPU numbers stand without it
            result["cpu_baseline"] = {"value": None, "unit": "Mcells/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist_on:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:
            pass


if __name__ == "__main__":
    main()
