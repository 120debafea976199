"""One process per GPU: sharding helpers and the torch.distributed plumbing around the
library's collectives.

The hypercube is sharded by the TOP log2(world) index bits: rank d owns the contiguous
range [d*2^(n-g), (d+1)*2^(n-g)) of every table.  Folding is LE (variable j = index bit j),
so fold pairs (2b, 2b+1) stay shard-local until only g variables are left.  Per pass each
rank produces partial sums; they are summed across ranks as 32-bit limbs in uint64 slots
(a plain u64 sum would wrap mod 2^64, which is wrong mod p) and recombined mod p.

Transports for that sum:
  * peer (`attach_peer`): the last workgroup of each sharded pass exchanges the limbs with the peers itself
    through HIP-IPC-mapped inboxes - no collective launch.
  * RCCL inside the library (`Context.comm_init_rccl`): ncclAllReduce on the context's
    stream.  torch.distributed is only the control plane that broadcasts the unique id.
  * host callbacks (`Context.comm_init_host`) - e.g. torch.distributed all_reduce on CPU
    tensors (gloo); used by the CPU tests and available as a fallback transport.
"""
import os

import numpy as np

MASK32 = 0xFFFFFFFF


def shard_range(num_vars, rank, world):
    """(start, length) of rank's shard of a 2^num_vars table"""
    g = world.bit_length() - 1
    if world != 1 << g:
        raise ValueError("world size must be a power of two")
    if num_vars < g:
        raise ValueError("table of 2^%d entries cannot be split over %d ranks" % (num_vars, world))
    length = 1 << (num_vars - g)
    return rank * length, length


def split_limbs(values):
    """[v0, v1, ...] -> uint64 array [lo0, hi0, lo1, hi1, ...] of 32-bit limbs"""
    out = np.empty(2 * len(values), dtype=np.uint64)
    for i, v in enumerate(values):
        out[2 * i] = int(v) & MASK32
        out[2 * i + 1] = int(v) >> 32
    return out


def recombine_limbs(limbs, p):
    """inverse of split_limbs after the limb-wise sum over ranks, reduced mod p"""
    return [(int(limbs[2 * i]) + (int(limbs[2 * i + 1]) << 32)) % p for i in range(len(limbs) // 2)]


def torch_collectives(group=None):
    """(allreduce, allgather) callables over torch.distributed for Context.comm_init_host.
    Limb sums stay far below 2^63, so the int64 view torch needs is exact."""
    import torch
    import torch.distributed as dist

    def allreduce(arr):
        t = torch.from_numpy(arr.view(np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    def allgather(send):
        world = dist.get_world_size(group)
        t = torch.from_numpy(send.view(np.int64).copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        return np.concatenate([o.numpy().view(np.uint64) for o in outs])

    return allreduce, allgather


def init_process_group_from_env(backend="gloo"):
    """control-plane process group from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (torch.distributed.run sets them).  Returns (rank, world, local_rank)."""
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def broadcast_bytes(payload, src=0, group=None):
    """broadcast a bytes object from `src` (any backend that moves CPU tensors)"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return payload
    obj = [payload if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(obj, src=src, group=group)
    return obj[0]


def attach_rccl(ctx, rank, world, group=None):
    """create the in-library RCCL communicator: rank 0 makes the unique id, everyone joins"""
    from .dense_mle import Context
    uid = Context.rccl_unique_id() if rank == 0 else None
    uid = broadcast_bytes(uid, src=0, group=group)
    ctx.comm_init_rccl(uid, rank, world)


def attach_peer(ctx, rank, world, group=None):
    """peer transport: export this rank's region, all-gather the IPC handles over the control plane
    (torch.distributed, any backend that moves Python objects), map the peers' regions.  The connect ends with the
    library's hello handshake (every rank has mapped, loaded its code and run a kernel) and self-test.  Every rank
    takes part in the all-gather whatever happened locally, so a one-sided failure raises on every rank."""
    import torch.distributed as dist
    handle, err = None, None
    try:
        handle = ctx.comm_peer_export(rank, world)
    except Exception as e:
        err = e
    if world == 1:
        handles = [handle]
    else:
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
    if err is not None:
        raise err
    if any(h is None for h in handles):
        raise RuntimeError("a peer could not export its region")
    ctx.comm_peer_connect(handles)


def attach_default(ctx_factory, rank, world, group=None):
    """The data-plane transport a multi-GPU run uses unless told otherwise.  `ctx_factory()` makes a fresh Context for
    this rank.  First choice: the peer transport (no collective launch); its connect runs a hello handshake and an
    exchange / gather self-test inside the library, and the ranks then agree over the control plane - if it failed on
    ANY rank (e.g. fine-grained peer memory that does not behave as the kernels assume on this node), every rank
    falls back to RCCL together.  Returns (ctx, "peer" | "rccl")."""
    import torch
    import torch.distributed as dist
    ctx = ctx_factory()
    ok = True
    try:
        attach_peer(ctx, rank, world, group)
    except Exception:
        ok = False
    if world > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ok = int(flag.item()) == 1
    if ok:
        return ctx, "peer"
    ctx.close()
    ctx = ctx_factory()
    attach_rccl(ctx, rank, world, group)
    return ctx, "rccl"
