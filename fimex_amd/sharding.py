"""Slice sharding across the GPUs of one node and the write-back gather.

Time x level slices are independent units: each rank regrids a contiguous block of them with its own
replica of the plan, exactly as the reference's MPI mode gives each rank its own time steps
(src/NetCDF_CDMWriter.cc:632-646) -- there is no data-path collective.  The only exchange is the final
write-back of finished output slices to the writer rank, done here with point-to-point sends over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests): on MI355X
every peer has its own direct xGMI link to the root, so all peers send concurrently.
"""
import torch
import torch.distributed as dist


def slice_range(n_slices, world_size, rank):
    """[begin, end) of the contiguous block of slices rank owns; the first n_slices % world_size ranks hold one more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d of %d" % (rank, world_size))
    base, extra = divmod(n_slices, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def gather_slices(local_out, n_slices, dst=0, group=None):
    """Write-back: every rank contributes its [n_local][oy][ox] block; rank dst returns [n_slices][oy][ox]
    (None elsewhere).  Blocks may differ in length (slice_range); empty blocks are skipped."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = slice_range(n_slices, world, rank)
    if local_out.shape[0] != e - b:
        raise ValueError("rank %d holds %d slices, expected %d" % (rank, local_out.shape[0], e - b))
    if rank == dst:
        full = torch.empty((n_slices,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
        full[b:e] = local_out
        ops = []
        for r in range(world):
            rb, re = slice_range(n_slices, world, r)
            if r != dst and re > rb:
                ops.append(dist.P2POp(dist.irecv, full[rb:re], r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return full
    if e > b:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_out.contiguous(), dst, group)]):
            req.wait()
    return None


def post_chunk_write_back(local_chunk, full, n_slices, c0, c1, dst=0, group=None):
    """Overlapped write-back, one chunk: posts (does not wait for) the transfer of block-local slices [c0, c1) of every
    rank's block to rank dst and returns the pending requests.  Called by every rank once per chunk, in the same order,
    right after the chunk's regrid has been enqueued: the next chunk is regridded while this one travels (the
    reference's writer receives finished slices the same way, one time step at a time, src/NetCDF_CDMWriter.cc:632-663).

    local_chunk: this rank's slices [c0, min(c1, n_local)) -- may be empty when its block is shorter than the longest.
    full: on dst the job's [n_slices][oy][ox] tensor (None elsewhere); dst's own slices are copied into it unless
    local_chunk already is that part of it."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = slice_range(n_slices, world, rank)
    mine = max(0, min(c1, e - b) - c0)
    if local_chunk.shape[0] != mine:
        raise ValueError("rank %d passes %d slices for chunk [%d, %d), expected %d" % (rank, local_chunk.shape[0], c0, c1, mine))
    ops = []
    if rank == dst:
        if mine and full[b + c0:b + c0 + mine].data_ptr() != local_chunk.data_ptr():
            full[b + c0:b + c0 + mine].copy_(local_chunk)
        for r in range(world):
            rb, re = slice_range(n_slices, world, r)
            lo, hi = rb + c0, min(rb + c1, re)
            if r != dst and hi > lo:
                ops.append(dist.P2POp(dist.irecv, full[lo:hi], r, group))
    elif mine:
        ops.append(dist.P2POp(dist.isend, local_chunk.contiguous(), dst, group))
    return list(dist.batch_isend_irecv(ops)) if ops else []
