#!/usr/bin/env python3
"""Headline benchmark: Mcells/s regridded, bilinear, 4000x3000 -> 2000x2000 f32 (BASELINE.json).

One "step" = one pass of the hot path over one batch of slices: ONE launch of the bilinear apply
kernel over NZ time x level slices that already sit in HBM (default NZ = 200, the north-star batch;
the geometry is BASELINE.json configs[1], "C2").  N > 1: one process per GPU with a replicated plan and no
data-path collective -- the reference's MPI mode shards time steps the same way
(src/NetCDF_CDMWriter.cc:632-663).
  --scaling weak (default): every rank regrids its own NZ slices; the RCCL gather of the finished slices to
      rank 0 ("write-back") is timed separately and reported beside the metric.
  --scaling strong: the NZ slices are split over the ranks (fimex_amd.sharding.slice_range; BASELINE configs[2]
      is this with --method bicubic: 25 slices per GPU on 8 GPUs); besides the regrid-only metric the step
      "regrid in z chunks + write-back of every finished chunk while the next one is regridded" is timed and
      reported with its exposed gather time.

    python bench.py --gpus N --steps K --warmup W      (N > 1 without WORLD_SIZE: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

After the timed steps (outside the timed region) slices 0, NZ/2 and NZ-1 of the last launch's output are compared bit
for bit with the CPU oracle, and EVERY slice of it with the same batch regridded once more by the per-lane gather kernels
(fimex_amd_regrid_apply_gather_device) into a second buffer; a mismatch makes the run fail.

Both batches are allocated by the library: the source batch (fimex_amd_regrid_source_batch_alloc_device: --source-candidates whole
allocations timed with the plan's launch, the fastest kept) and the output batch (fimex_amd_regrid_batch_alloc_device: --placements
whole allocations, the source batch regridded into each, the fastest kept, the others freed); what plain allocations would have
got (the first candidate) and the cost of the probing are reported in config.source_placement and
config.output_placement.  --placements 1 = plain allocations for both.
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


def launch_ranks(n, argv):
    """--gpus N without WORLD_SIZE: start N ranks as children (torch.distributed.run) BEFORE this process touches the
    GPU, relay rank 0's JSON line and the exit status.  Nothing here imports torch.cuda or the product library."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in p.stdout:
        if out.startswith('{"metric"'):
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        log("bench.py: the ranks ended without a result line")
        rc = 1
    return rc


def verify_slices(torch, oracle_method, px, py, wl, d_in, d_out, first_slice, picks, tolerance=None):
    """Check of a few slices of the timed launch's output against the CPU oracle (test infrastructure used as the checker
    only) on the very input slices the launch read: bit for bit, or -- tolerance given (FIMEX_AMD_BICUBIC_FAST) -- within
    that fraction of the slice's largest magnitude with identical NaN positions.  picks: local slice indices."""
    import oracle
    bad = []
    for k in picks:
        f = d_in[k].cpu().numpy()[None]
        want = oracle.interpolate_values(oracle_method, px, py, f, wl.inX, wl.inY, wl.outX, wl.outY, nthreads=8)[0]
        got = d_out[k].cpu().numpy()
        if tolerance is None:
            same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        else:
            with np.errstate(invalid="ignore"):
                same = (np.abs(got - want) <= tolerance * float(np.nanmax(np.abs(f)))) | (np.isnan(got) & np.isnan(want))
        if not bool(same.all()):
            bad.append((int(first_slice + k), int((~same).sum())))
    return bad


def verify_all_slices_vs_gather(torch, plan, d_in, d_out, nz, stream, tolerance=None):
    """Every slice of the timed launch's output against the same batch regridded by the per-lane gather kernels (regrid.hip:
    one lane per output cell reads its stencil from memory -- no tiles, no LDS ring, no DMA) into a second buffer, on the
    device.  Bit for bit with NaNs at the same cells; tolerance given (FIMEX_AMD_BICUBIC_FAST: the gather kernel computes in
    the reference's arithmetic): within that fraction of the batch's largest magnitude.  Returns the slices that differ."""
    chk = torch.empty_like(d_out)
    plan.apply_gather_device(d_in.data_ptr(), nz, chk.data_ptr(), stream)
    torch.cuda.synchronize()
    bad = []
    lim = None
    if tolerance is not None:
        lim = tolerance * float(torch.nan_to_num(d_in[0], nan=0.0).abs().max().item() + 0.01 * nz)
    for k0 in range(0, nz, 25):
        a, b = d_out[k0:k0 + 25], chk[k0:k0 + 25]
        both_nan = a.isnan() & b.isnan()
        if lim is None:
            same = (a.view(torch.int32) == b.view(torch.int32)) | both_nan
        else:
            same = ((a - b).abs() <= lim) | both_nan
        per = same.flatten(1).all(dim=1)
        bad += [int(k0 + i) for i in torch.nonzero(~per).flatten().tolist()]
    del chk
    return bad


def build_plan(fa, torch, wl, method, stream, bicubic=None):
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
                                     wl.inX, wl.inY, wl.outX, wl.outY, stream, bicubic=bicubic)
    torch.cuda.synchronize()
    return plan, d_px.cpu().numpy(), d_py.cpu().numpy()


def make_slices(torch, base, nz, first=0, into=None):
    """slice k = base + 0.01 (first + k), resident in HBM ([nz][inY][inX] f32); into: a tensor of that shape to fill."""
    d_base = torch.from_numpy(base).cuda()
    d_in = into if into is not None else torch.empty((nz,) + base.shape, dtype=torch.float32, device="cuda")
    for k0 in range(0, nz, 16):
        k1 = min(nz, k0 + 16)
        off = 0.01 * torch.arange(first + k0, first + k1, dtype=torch.float32, device="cuda")
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
    ap.add_argument("--nz", type=int, default=200, help="time x level slices: per GPU per step (weak) or in all (strong)")
    ap.add_argument("--method", default="bilinear", choices=["bilinear", "bicubic", "nearest"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --nz slices on every GPU; strong: --nz slices split over the GPUs (configs[2] with --method bicubic)")
    ap.add_argument("--bicubic-fast", action="store_true",
                    help="--method bicubic with FIMEX_AMD_BICUBIC_FAST (float FMA, 1e-5 of the stencil's magnitude) instead of the bit-exact kernel")
    ap.add_argument("--chunk", type=int, default=5, help="strong scaling: slices per write-back chunk of the overlapped gather")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-slice, copy and gather measurements")
    ap.add_argument("--placements", type=int, default=8,
                    help="windows the library tries for the output batch (fimex_amd_regrid_batch_alloc_device; 1 = plain allocation)")
    ap.add_argument("--source-candidates", type=int, default=6,
                    help="allocations the library tries for the source batch (fimex_amd_regrid_source_batch_alloc_device; 1 = plain allocation)")
    ap.add_argument("--workload", default="default", choices=["default", "one_percent"],
                    help="target axes: round 1's (10.9 %% of the target cells undefined) or the ~1 %% variant of SURVEY 8d")
    ap.add_argument("--no-tune", action="store_true", help="keep the plan's default workgroup shape (skip fimex_amd_regrid_plan_tune_device)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of three output slices")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend; gloo + --one-device rehearses the N > 1 path on a single-GPU box")
    ap.add_argument("--one-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--tuning-build", action="store_true",
                    help="profiling sweeps only: run libfimex_amd_tuning.so, the build that reads the FIMEX_AMD_<NAME> experiment switches")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no rank environment: start the ranks ourselves, before anything in this process touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from fimex_amd import capi as fa
    from fimex_amd import sharding
    import workloads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    if args.one_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible); --one-device is the single-GPU rehearsal"
                         % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            import datetime
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend="gloo")
    if args.tuning_build:
        fa.use_tuning_build(True)
    fa.load()
    fa.set_device(local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    comm_dev = "cuda" if args.backend == "nccl" else "cpu"

    method = {"bilinear": fa.BILINEAR, "bicubic": fa.BICUBIC, "nearest": fa.NEAREST_NEIGHBOR}[args.method]
    stencil = {"bilinear": 2, "bicubic": 4, "nearest": 1}[args.method]
    wl = workloads.BilinearRotatedPole(variant=args.workload)
    t0 = time.perf_counter()
    fast = args.bicubic_fast and args.method == "bicubic"
    plan, px, py = build_plan(fa, torch, wl, method, stream, bicubic=fa.BICUBIC_FAST if fast else None)
    t_plan = time.perf_counter() - t0
    info = plan.info()
    base = wl.base_field()
    # slices of this rank: weak = its own nz, strong = its block of the nz slices of the job
    strong = args.scaling == "strong"
    nz_total = args.nz if strong else world * args.nz
    first, last = sharding.slice_range(nz_total, world, rank)
    nz = last - first
    if nz == 0:
        raise SystemExit("bench.py: rank %d holds no slice (%d slices over %d ranks)" % (rank, nz_total, world))
    out_layer, in_layer = wl.outX * wl.outY, wl.inX * wl.inY
    # The source batch comes from the library too (fimex_amd_regrid_source_batch_alloc_device): which allocation the source slices
    # lie in moves the launch by 4-5 %, more than the output's placement does (profiles/calib/r03_placement_matrix.jsonl).  A
    # resident pipeline asks the library for the buffer its reader fills; here the synthetic slices are written into it.
    source_placement, src_batch = None, None
    if args.placements > 1 and args.source_candidates > 1:
        try:
            src_batch = plan.alloc_source_batch(nz, candidates=args.source_candidates, stream=stream)
            si = src_batch.info
            source_placement = {"by": "fimex_amd_regrid_source_batch_alloc_device", "candidates": si["positions"], "chosen": si["chosen"],
                                "ms_at_each": si["msAtPosition"], "ms_at_first_candidate": si["msAtPosition"][0] if si["msAtPosition"] else None,
                                "bytes_allocated_while_probing": si["bytesProbed"], "bytes_held_afterwards": si["bytesHeld"],
                                "probe_seconds": si["probeSeconds"],
                                "note": "whole allocations, zero-filled, default workgroup shape, median of 3 launches each into a scratch output"}
        except fa.FimexAmdError as e:
            log("bench.py: no source placement (%s)" % str(e).splitlines()[0])
    d_in = make_slices(torch, base, nz, first, into=src_batch.as_tensor() if src_batch is not None else None)
    # rank 0 holds the job's whole output in strong mode (the write-back target); its own block is a view of it
    d_full = torch.empty((nz_total, wl.outY, wl.outX), dtype=torch.float32, device="cuda") if (strong and rank == 0 and dist_on) else None
    d_out = d_full[first:last] if d_full is not None else torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    # Which allocation the output batch lies in moves this launch too (same kernel, same traffic, per-channel request
    # counts identical; profiles/calib/r03_placement_pmc_*.json), so that allocation is a service of the library as well:
    # fimex_amd_regrid_batch_alloc_device makes --placements whole allocations, times the plan's apply on this source batch into
    # each, keeps the fastest and frees the others.  The time with the first candidate -- what a plain allocation would have got --
    # and the cost of the probing are reported beside the metric.
    placement = None
    batch = None
    if args.placements > 1 and d_full is None:
        del d_out
        torch.cuda.empty_cache()
        try:
            batch = plan.alloc_batch(d_in.data_ptr(), nz, positions=args.placements, stream=stream)
        except fa.FimexAmdError as e:  # the plain allocation instead
            log("bench.py: no placement search (%s)" % str(e).splitlines()[0])
            d_out = torch.empty((nz, wl.outY, wl.outX), dtype=torch.float32, device="cuda")
    if batch is not None:
        d_out = batch.as_tensor()
        bi = batch.info
        placement = {"by": "fimex_amd_regrid_batch_alloc_device", "positions": bi["positions"],
                     "chosen": bi["chosen"], "ms_at_each": bi["msAtPosition"],
                     "ms_at_first_position": bi["msAtPosition"][0] if bi["msAtPosition"] else None,
                     "bytes_allocated_while_probing": bi["bytesProbed"], "bytes_held_afterwards": bi["bytesHeld"], "batch_bytes": bi["bytes"],
                     "probe_seconds": bi["probeSeconds"],
                     "note": "whole allocations, default workgroup shape, median of 3 launches each, before the shape tuning"}

    # the plan's two workgroup shapes (identical results) timed on this device and this batch, the faster kept: part of the product
    # (fimex_amd_regrid_plan_tune_device), done once per plan like the plan build and outside the timed steps
    tuned_shape = None if args.no_tune else plan.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), stream)
    info = plan.info()

    def step():
        plan.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), stream)

    wall, kernel_ms = time_launches(torch, step, args.steps, args.warmup, dist_on)
    wall_t = torch.tensor([wall], dtype=torch.float64, device=comm_dev)
    if dist_on:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall_max = float(wall_t.item())
    cells_per_step = nz_total * out_layer  # all ranks together
    ms_per_step = wall_max / args.steps * 1e3
    value = cells_per_step / (wall_max / args.steps) / 1e6

    # parity of the timed launch itself: three slices of its output against the CPU oracle, bit for bit
    verified, failed = [], []
    if not args.no_verify:
        picks = sorted({0, (nz - 1) // 2, nz - 1})
        failed = verify_slices(torch, {"bilinear": 1, "bicubic": 2, "nearest": 0}[args.method], px, py, wl, d_in, d_out, first, picks,
                               tolerance=1e-5 if fast else None)
        verified = [first + k for k in picks]
        # ... and every slice of it against the gather kernels' result for the same batch (a second buffer, on the device)
        bad_slices = verify_all_slices_vs_gather(torch, plan, d_in, d_out, nz, stream, tolerance=1e-5 if fast else None)
        failed += [(first + k, -1) for k in bad_slices]
        flag = torch.tensor([len(failed)], dtype=torch.int64, device=comm_dev)
        if dist_on:
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) != 0:
            log("bench.py: PARITY FAILURE on rank %d: (slice, differing cells; -1 = differs from the gather kernels) %r" % (rank, failed))
            if dist_on:
                dist.destroy_process_group()
            sys.exit(1)

    # roofline of the dominant (only) kernel: algorithmic bytes per launch / average launch duration
    # B_alg = nz*4*(N_src + ox*oy) + B_plan, N_src = source cells in the plan's reduced-domain bounding box
    # (what the reference itself reads, CachedInterpolation.cc:159-200); the count of distinct cells the stencils
    # actually touch is reported beside it (smaller where the target grid is coarser than 2 source cells)
    n_src_touched = workloads.touched_source_cells(px, py, wl.inX, wl.inY, stencil) if rank == 0 else 0
    n_src_bbox = workloads.reduced_domain_cells(px, py, wl.inX, wl.inY) if rank == 0 else 0
    alg_bytes = nz * 4 * (n_src_bbox + out_layer) + info["planBytes"]
    alg_bytes_touched = nz * 4 * (n_src_touched + out_layer) + info["planBytes"]
    avg_kernel_ms = float(np.mean(kernel_ms))
    achieved = alg_bytes / (avg_kernel_ms * 1e-3) / 1e9
    # HBM-side bytes per launch from the PMC passes of scripts/collect_profiles.py (rocprofv3 cannot run inside this
    # process): used only if the recorded launch is this one (method, slices, tile shape), otherwise null
    traffic, traffic_source = None, None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tf):
        try:
            recs = json.load(open(tf))
            rec = recs.get("%s_nz%d_%dx%d" % (args.method + ("fast" if fast else ""), nz, info.get("tileW"), info.get("tileH"))) or \
                recs.get("%s_nz%d" % (args.method + ("fast" if fast else ""), nz))
            if isinstance(rec, dict) and rec.get("tile") == [info.get("tileW"), info.get("tileH")]:
                traffic, traffic_source = rec["bytes"], rec.get("source")
        except Exception:
            traffic = None

    kernel_name = ("staged_apply<%d, ...>" % stencil) if info.get("stagedCells") else args.method + "_apply"
    result = {
        "metric": "Mcells/s regridded (%s, 4000x3000->2000x2000 f32)" % (args.method + (" fast arithmetic" if fast else "")),
        "value": value, "unit": "Mcells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1] geometry (4000x3000 0.01-deg lon/lat -> 2000x2000 rotated pole, %s), "
                        "%d time x level slices %s per step resident in HBM (north-star batch)"
                        % (args.method, args.nz, "split over the GPUs" if strong else "per GPU"),
            "slices_per_gpu": nz, "slices_total": nz_total, "library": "libfimex_amd_tuning.so" if args.tuning_build else "libfimex_amd.so",
            "sharding": "slices over GPUs, plan replicated, no data-path collective",
            "plan_build_s": t_plan, "tuned_shape": tuned_shape, "source_placement": source_placement, "output_placement": placement, "undefined_target_cells": info["undefinedCells"], "border_cells": info["borderCells"],
        },
        "verified_slices": verified, "verified_all_slices_vs_gather": (not args.no_verify) or None, "verified_how": "1e-5 of the slice's largest magnitude (FIMEX_AMD_BICUBIC_FAST)" if fast else "bit for bit against the CPU oracle",
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kernel_name,
            "staged_cells_per_slice": info.get("stagedCells"), "tile": [info.get("tileW"), info.get("tileH")],
            "kernel_ms_avg": avg_kernel_ms, "kernel_ms_min": float(np.min(kernel_ms)),
            "algorithmic_bytes_per_launch": alg_bytes,
            "n_src_bbox": n_src_bbox, "n_src_touched": n_src_touched, "plan_bytes": info["planBytes"],
            "frac_if_only_touched_cells_counted": alg_bytes_touched / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "frac_on_staged_bytes": ((nz * 4 * (info.get("stagedCells") + out_layer) + info["planBytes"]) / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)
                                    if info.get("stagedCells") else None,
        },
    }

    if not args.no_extras:
        # configs[1] proper: one time step, one slice per call (src/CDMInterpolator.cc:251-259 hands over a new slice every
        # call).  Cold: the calls walk through the nz resident source / output slice pairs, so neither the L2s nor the 256 MB
        # Infinity Cache hold any of the field from the previous call (the plan, read by every call, may stay there, as in a
        # real run).  Warm: the same slice pair again and again -- cache resident, reported for comparison only.
        # bytes of plan a one-slice call reads: batches below 4 slices take the gather kernels (pos + fractions per output cell)
        plan_bytes_1 = {1: 4, 2: 12, 4: 20}[stencil] * out_layer
        b1 = 4 * (n_src_bbox + out_layer) + plan_bytes_1
        if nz >= 8:
            ring = [(k * 7) % nz for k in range(max(3 * args.steps, 60))]
            it = iter(ring * 2)

            def one_cold():
                k = next(it)
                plan.apply_device(d_in.data_ptr() + 4 * in_layer * k, 1, d_out.data_ptr() + 4 * out_layer * k, stream)
            _, kc1 = time_launches(torch, one_cold, len(ring), 5, False)
            t1 = float(np.mean(kc1))
            result["single_slice_cold"] = {
                "workload": "configs[1], nz = 1, a different slice pair every call (%d pairs, %.1f GB working set)" % (nz, nz * 4e-9 * (in_layer + out_layer)),
                "kernel_ms_avg": t1, "kernel_ms_min": float(np.min(kc1)), "Mcells_per_s": out_layer / (t1 * 1e-3) / 1e6,
                "algorithmic_bytes": b1, "achieved_GBps": b1 / (t1 * 1e-3) / 1e9, "frac": b1 / (t1 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "plan_bytes": plan_bytes_1}
        wall1, k1 = time_launches(torch, lambda: plan.apply_device(d_in.data_ptr(), 1, d_out.data_ptr(), stream),
                                  max(args.steps, 20), args.warmup, False)
        result["single_slice_warm"] = {"workload": "configs[1], nz = 1, the SAME slice pair every call: 112 MB working set, served from the "
                                                   "256 MB Infinity Cache, not from HBM -- not a roofline figure", "kernel_ms_avg": float(np.mean(k1)),
                                       "Mcells_per_s": out_layer / (float(np.mean(k1)) * 1e-3) / 1e6}
        # the box's copy ceiling for context: a device-to-device copy of the output's size, read + write bytes
        flat_in, flat_out = d_in.view(-1)[:d_out.numel()], d_out.reshape(-1)
        _, kc = time_launches(torch, lambda: flat_out.copy_(flat_in), 10, 2, False)
        result["roofline"]["copy_kernel_GBps"] = 8 * d_out.numel() / (float(np.mean(kc)) * 1e-3) / 1e9
        if args.workload == "default" and args.method == "bilinear" and world == 1:
            # the same source batch onto the target axes with ~1 % undefined cells (SURVEY 8d's proportions), beside the headline
            try:
                wl2 = workloads.BilinearRotatedPole(variant="one_percent")
                plan2, px2, py2 = build_plan(fa, torch, wl2, method, stream)
                shape2 = None if args.no_tune else plan2.tune_device(d_in.data_ptr(), nz, d_out.data_ptr(), stream)
                _, k2 = time_launches(torch, lambda: plan2.apply_device(d_in.data_ptr(), nz, d_out.data_ptr(), stream), args.steps, args.warmup, False)
                bad2 = [] if args.no_verify else verify_slices(torch, 1, px2, py2, wl2, d_in, d_out, first, [0, nz - 1])
                bad2 += [] if args.no_verify else [(k, -1) for k in verify_all_slices_vs_gather(torch, plan2, d_in, d_out, nz, stream)]
                i2 = plan2.info()
                n2 = workloads.reduced_domain_cells(px2, py2, wl2.inX, wl2.inY)
                b2 = nz * 4 * (n2 + out_layer) + i2["planBytes"]
                t2 = float(np.mean(k2))
                result["one_percent_undefined_variant"] = {
                    "workload": "same source batch, target rotated lon -8.8..8.8, lat 16.5..45.0: %.2f %% of the target cells undefined, reduced-domain "
                                "bounding box %.1f %% of the source" % (100.0 * i2["undefinedCells"] / out_layer, 100.0 * n2 / in_layer),
                    "kernel_ms_avg": t2, "Mcells_per_s": nz * out_layer / (t2 * 1e-3) / 1e6, "algorithmic_bytes_per_launch": b2, "n_src_bbox": n2,
                    "frac": b2 / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "tuned_shape": shape2, "tile": [i2.get("tileW"), i2.get("tileH")],
                    "staged_cells_per_slice": i2.get("stagedCells"), "n_src_touched": workloads.touched_source_cells(px2, py2, wl2.inX, wl2.inY, 2),
                    "frac_on_staged_bytes": (nz * 4 * (i2.get("stagedCells", 0) + out_layer) + i2["planBytes"]) / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                    "verified": (not bad2) if not args.no_verify else None,
                    "note": "output batch at the window chosen for the headline plan"}
                if bad2:
                    log("bench.py: PARITY FAILURE of the one_percent variant: %r" % (bad2,))
                    sys.exit(1)
                del plan2
            except fa.FimexAmdError as e:
                result["one_percent_undefined_variant"] = {"error": str(e)[:300]}
        step()  # d_out holds the regrid result again
        torch.cuda.synchronize()
        if dist_on:
            try:
                result["gather"] = (measure_strong_write_back if strong else measure_weak_write_back)(
                    torch, dist, sharding, args, plan, stream, d_in, d_out, d_full, nz, nz_total, first, in_layer, out_layer, world, rank)
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


def measure_weak_write_back(torch, dist, sharding, args, plan, stream, d_in, d_out, d_full, nz, nz_total, first, in_layer, out_layer, world, rank):
    """weak scaling: RCCL gather of every rank's finished slices to rank 0 over xGMI, after the regrid, outside the metric
    (gloo rehearsal: point-to-point on host copies; the measured path is RCCL on device buffers)."""
    src = d_out if args.backend == "nccl" else d_out.cpu()
    dist.barrier()
    torch.cuda.synchronize()
    tg = time.perf_counter()
    full = sharding.gather_slices(src, nz_total, dst=0)
    torch.cuda.synchronize()
    dist.barrier()
    tg = time.perf_counter() - tg
    del full
    return {"seconds": tg, "bytes_per_peer": d_out.numel() * 4, "GBps_into_root": (world - 1) * d_out.numel() * 4 / tg / 1e9,
            "note": "point-to-point RCCL sends of every rank's output slices to rank 0 (fimex_amd/sharding.py); not part of value"}


def measure_strong_write_back(torch, dist, sharding, args, plan, stream, d_in, d_out, d_full, nz, nz_total, first, in_layer, out_layer, world, rank):
    """strong scaling: the job as the reference's writer sees it (src/NetCDF_CDMWriter.cc:632-663) -- every rank regrids
    its block in chunks of --chunk slices and each finished chunk travels to rank 0 while the next one is regridded.
    Timed: the chunked regrid alone, and chunked regrid + overlapped write-back; the difference is the exposed gather."""
    nmax = -(-nz_total // world)
    chunk = max(1, args.chunk)
    on_device = args.backend == "nccl"

    def regrid_chunk(c0, c1):
        plan.apply_device(d_in.data_ptr() + 4 * in_layer * c0, c1 - c0, d_out.data_ptr() + 4 * out_layer * c0, stream)

    def chunked(with_write_back):
        reqs = []
        for c0 in range(0, nmax, chunk):
            c1 = min(c0 + chunk, nmax)
            l1 = min(c1, nz)
            if l1 > c0:
                regrid_chunk(c0, l1)
            if with_write_back:
                mine = d_out[c0:l1] if l1 > c0 else d_out[0:0]
                if not on_device:
                    mine = mine.cpu()
                reqs += sharding.post_chunk_write_back(mine, d_full if on_device else host_full, nz_total, c0, c1, dst=0)
        for r in reqs:
            r.wait()
        torch.cuda.synchronize()

    host_full = torch.empty((nz_total,) + tuple(d_out.shape[1:]), dtype=torch.float32) if (rank == 0 and not on_device) else None

    def timed(fn, reps):
        fn()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([(time.perf_counter() - t0) / reps], dtype=torch.float64, device="cuda" if on_device else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    reps = max(3, min(args.steps, 10))
    t_regrid = timed(lambda: chunked(False), reps)
    t_both = timed(lambda: chunked(True), reps)
    ok = None
    if rank == 0:  # the gathered job equals what the ranks computed: spot check of the last rank's last slice against rank 0's kernel
        got = (d_full if on_device else host_full)[nz_total - 1]
        last_in = make_slices(torch, workloads_base(), 1, nz_total - 1)
        chk = torch.empty_like(d_out[0:1])
        plan.apply_device(last_in.data_ptr(), 1, chk.data_ptr(), stream)
        torch.cuda.synchronize()
        a, b = got.cpu().numpy(), chk[0].cpu().numpy()
        ok = bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return {"seconds_regrid_chunked": t_regrid, "seconds_regrid_and_write_back": t_both, "seconds_exposed": max(0.0, t_both - t_regrid),
            "chunk_slices": chunk, "bytes_per_peer": nz * out_layer * 4, "write_back_verified": ok,
            "Mcells_per_s_with_write_back": nz_total * out_layer / t_both / 1e6,
            "note": "every finished chunk of --chunk slices is sent to rank 0 (point-to-point, batched per chunk) while the next chunk is "
                    "regridded; value above is the regrid alone"}


def workloads_base():
    import workloads
    return workloads.BilinearRotatedPole().base_field()


if __name__ == "__main__":
    main()
