"""Multi-process path on CPU: slice sharding + write-back gather with world_size 2 (and 3) over gloo.
Each rank regrids its block of slices (with the CPU oracle standing in for the GPU kernel -- this test is about
the partitioning and the exchange, not the arithmetic) and rank 0 must end up with exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
import oracle
from fimex_amd import sharding


def test_slice_range_partitions_exactly():
    for n in (0, 1, 2, 7, 200, 201):
        for world in (1, 2, 3, 8):
            blocks = [sharding.slice_range(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.slice_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nz, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inX, inY, outX, outY = 60, 45, 50, 40
        px, py = cases.backward_positions(inX, inY, outX, outY, seed=1)  # plan inputs replicated on every rank
        f = cases.field(nz, inY, inX, seed=2)
        b, e = sharding.slice_range(nz, world, rank)
        local = oracle.interpolate_values(oracle.BILINEAR, px, py, f[b:e], inX, inY, outX, outY) if e > b \
            else np.empty((0, outY, outX), np.float32)
        full = sharding.gather_slices(torch.from_numpy(local), nz, dst=0)
        if rank == 0:
            want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
            assert cases.same(full.numpy(), want)
            open(os.path.join(tmp, "ok"), "w").write("ok")
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nz", [(2, 7), (2, 8), (3, 2), (2, 1)])
def test_sharded_regrid_and_gather_gloo(tmp_path, world, nz):
    mp.spawn(_worker, args=(world, _free_port(), nz, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def _chunk_worker(rank, world, port, nz, chunk, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        inX, inY, outX, outY = 60, 45, 50, 40
        px, py = cases.backward_positions(inX, inY, outX, outY, seed=1)
        f = cases.field(nz, inY, inX, seed=2)
        b, e = sharding.slice_range(nz, world, rank)
        nmax = -(-nz // world)
        full = torch.full((nz, outY, outX), -1.0) if rank == 0 else None
        local = full[b:e] if rank == 0 else torch.empty((e - b, outY, outX))
        reqs = []
        for c0 in range(0, nmax, chunk):  # bench.py's strong-scaling loop: regrid a chunk, post its write-back, go on
            c1 = min(c0 + chunk, nmax)
            l1 = min(c1, e - b)
            if l1 > c0:
                local[c0:l1] = torch.from_numpy(oracle.interpolate_values(oracle.BILINEAR, px, py, f[b + c0:b + l1], inX, inY, outX, outY))
            reqs += sharding.post_chunk_write_back(local[c0:l1] if l1 > c0 else local[0:0], full, nz, c0, c1, dst=0)
        for r in reqs:
            r.wait()
        if rank == 0:
            want = oracle.interpolate_values(oracle.BILINEAR, px, py, f, inX, inY, outX, outY)
            assert cases.same(full.numpy(), want)
            open(os.path.join(tmp, "ok"), "w").write("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,nz,chunk", [(2, 8, 2), (2, 7, 3), (3, 5, 1), (2, 1, 4)])
def test_chunked_overlapped_write_back_gloo(tmp_path, world, nz, chunk):
    """strong scaling: every rank's block goes to rank 0 chunk by chunk while the next chunk is computed; uneven blocks and
    a rank without any slice included; rank 0 ends up with the single-process result."""
    mp.spawn(_chunk_worker, args=(world, _free_port(), nz, chunk, str(tmp_path)), nprocs=world, join=True)
    assert (tmp_path / "ok").read_text() == "ok"


def test_bench_starts_its_own_ranks_before_touching_the_gpu():
    """bench.py --gpus N without WORLD_SIZE must spawn N ranks itself (torch.distributed.run) and may not import torch or
    the product library before it does; without a GPU the ranks fail loudly and the launcher returns non-zero."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "bench.py")).read()
    launcher = src[src.index("def launch_ranks"):src.index("def verify_slices")]
    assert "torch.distributed.run" in launcher and "import torch" not in launcher and "capi" not in launcher
    main = src[src.index("def main():"):]
    assert main.index("launch_ranks(") < main.index("import torch")
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and '{"metric"' not in r.stdout
    assert "needs an MI355X" in r.stderr or "no GPU" in r.stderr
