"""Throughput-mode sharding across GPUs: stereo keyframes / BA windows are independent, so each rank owns a
contiguous chunk and no data-path collective is needed.  The only exchange is a gather of the per-keyframe poses
(7 f64 = 56 B each) once per step -- latency-bound, RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.
Live SLAM mode (one trajectory, one map) is inherently serial: replicas only.
"""
import torch


def shard_range(total, rank, world):
    """contiguous chunk [lo, hi) of `total` items owned by `rank`: the first (total % world) ranks get one extra item
    (50 pairs over 8 ranks -> 7,7,6,6,6,6,6,6)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_poses(local_poses, dist=None):
    """all-gather equally sized per-rank pose blocks [B,7] -> [world*B,7] in rank order (rank r owns rows r*B..)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_poses
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(local_poses.shape), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out.view(-1), local_poses.contiguous().view(-1))
    return out.view(world * local_poses.shape[0], *local_poses.shape[1:])


def gather_ragged_poses(local_poses, total, dist):
    """chunks from shard_range may differ by one row: pad to the largest chunk, gather, then strip the padding"""
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total, r, world) for r in range(world)]
    cap = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((cap, local_poses.shape[1]), dtype=local_poses.dtype, device=local_poses.device)
    pad[: local_poses.shape[0]] = local_poses
    allp = gather_poses(pad, dist)
    return torch.cat([allp[r * cap: r * cap + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)
