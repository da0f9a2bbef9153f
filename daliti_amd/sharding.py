"""Multi-GPU plumbing of the scan-to-map path: contiguous point shards + one all-reduce per pass.

Every scan point's kNN / plane / residual / Jacobian row is independent given the pose and the
(replicated) map; the only coupling is the sum of the normal block (HtH 144, Htz 12, effct,
total_residual = 158 doubles), so each rank reduces its shard on its GPU and the blocks are summed
with one RCCL all-reduce over xGMI per ESKF iteration.  Every rank then runs the identical fp64
update, so states stay bit-identical without a broadcast.  The reference is single-threaded
(eskf_lio/src/laserMapping.cpp:827-828); this split has no counterpart there.
"""
BLOCK_DOUBLES = 160


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of n scan points for this rank (keeps laserCloudOri order
    reconstructible by concatenating shards in rank order)."""
    if world < 1 or not (0 <= rank < world) or n < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_block(block, group=None):
    """Sum a 160-double block tensor in place across ranks (RCCL on GPU tensors, gloo on CPU)."""
    import torch.distributed as dist
    if block.numel() != BLOCK_DOUBLES:
        raise ValueError("normal block must have %d doubles" % BLOCK_DOUBLES)
    dist.all_reduce(block, op=dist.ReduceOp.SUM, group=group)
    return block
