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
