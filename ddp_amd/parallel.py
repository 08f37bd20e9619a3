"""Multi-GPU layout of the sampling loop: one process per GPU, images are independent units
(the reference loop is b=1), so a batch is split into contiguous per-rank shards and
NOTHING is exchanged inside the loop.  Collectives (RCCL over xGMI when the backend is 'nccl'; gloo
in the CPU tests): one broadcast of the packed frozen weights at start-up, an optional all_gather of
the per-rank outputs at the end.  The broadcast is ONE ``dist.broadcast`` of the 34 MB blob: which
algorithm RCCL runs for it over the point-to-point xGMI links (ring, tree, direct) is RCCL's choice and has
never been observed here - no box with more than one GPU was available in any round; at 34 MB, once per
process, it does not matter for the throughput.  No bucketing or ring tuning exists because nothing is
exchanged in the loop."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous [start, stop) of ``total`` items owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(total, world):
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def is_distributed(force=False):
    """a process group with more than one rank - or, with ``force``, any initialised process group: the collectives then run
    even at world size 1 (RCCL executes them as device-side copies), which is how a 1-GPU box exercises the RCCL code path"""
    return dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1)


def broadcast_weights(flat, src=0, force=False):
    """Replicate the packed weight blob (PackedWeights.flat) from ``src`` to all ranks, in place."""
    if is_distributed(force):
        dist.broadcast(flat, src=src)
    return flat


def gather_outputs(local_out, total, force=False):
    """all_gather per-rank output shards (B_local, ...) back into (total, ...) on every rank.  Shards may
    be ragged (total not divisible by world): they are padded to the largest shard for the collective."""
    if not is_distributed(force):
        return local_out
    world = dist.get_world_size()
    sizes = shard_sizes(total, world)
    mx = max(sizes)
    pad = local_out
    if local_out.shape[0] < mx:
        pad = torch.cat([local_out, local_out.new_zeros((mx - local_out.shape[0],) + tuple(local_out.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous())
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def sample_sharded(make_engine, x, noise, total=None):
    """Run the loop on this rank's shard of a global batch.  ``x`` / ``noise`` hold the GLOBAL batch
    (e.g. produced identically on every rank); ``make_engine(b_local)`` builds the rank-local engine.
    Returns (local_out, (start, stop))."""
    rank = dist.get_rank() if is_distributed() else 0
    world = dist.get_world_size() if is_distributed() else 1
    total = x.shape[0] if total is None else total
    a, b = shard_range(total, rank, world)
    if b == a:
        return None, (a, b)
    eng = make_engine(b - a)
    return eng.sample(x[a:b].contiguous(), noise[a:b].contiguous()), (a, b)
