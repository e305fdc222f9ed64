"""Sharding across GPUs.  Two modes, both with ONE collective: a gather of per-frame poses (7 f64 = 56 B each),
latency-bound, RCCL over xGMI on GPUs (backend "nccl"), gloo in the CPU tests.

* throughput mode: stereo keyframes / BA windows are independent, each rank owns a contiguous chunk, no data-path collective;
  `gather_poses` collects the equally sized per-rank pose blocks once per step.
* sequence mode (BASELINE config 5, the loop of run_vslam.cpp:40 split across GPUs): an F-frame sequence is cut into contiguous
  chunks; rank r owns frames [lo, hi) and also processes frame lo-1 as a HALO, because the frame-to-frame stage of frame lo
  needs its predecessor's keypoints and landmarks.  Every rank estimates the relative poses T_{j,j-1} of the frames it owns;
  the ragged gather puts them in frame order and rank 0 chains them into one trajectory T_j = T_{j,j-1} ... T_{1,0}.
Live SLAM (one map, keyframe decisions that depend on the previous BA) is inherently serial: replicas only.
"""
import numpy as np
import torch


def shard_range(total, rank, world):
    """contiguous chunk [lo, hi) of `total` items owned by `rank`: the first (total % world) ranks get one extra item
    (50 pairs over 8 ranks -> 7,7,6,6,6,6,6,6)"""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def halo_start(lo):
    """first frame a rank has to PROCESS to own frames from `lo` on: its predecessor, except at the start of the sequence"""
    return max(lo - 1, 0)


def owned_pose_range(total, rank, world):
    """global indices j of the relative poses T_{j,j-1} rank `rank` produces: its frames, minus frame 0 (no predecessor)"""
    lo, hi = shard_range(total, rank, world)
    return max(lo, 1), max(hi, 1)


def gather_poses(local_poses, dist=None):
    """all-gather equally sized per-rank pose blocks [B,7] -> [world*B,7] in rank order (rank r owns rows r*B..)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_poses
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(local_poses.shape), dtype=local_poses.dtype, device=local_poses.device)
    dist.all_gather_into_tensor(out.view(-1), local_poses.contiguous().view(-1))
    return out.view(world * local_poses.shape[0], *local_poses.shape[1:])


def _gather_ragged(local, sizes, dist):
    """rows of every rank (sizes[r] rows each) concatenated in rank order: pad to the largest block, one all-gather, strip"""
    cap = max(max(sizes), 1)
    pad = torch.zeros((cap, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    allp = gather_poses(pad, dist)
    return torch.cat([allp[r * cap: r * cap + n] for r, n in enumerate(sizes)], 0)


def gather_ragged_poses(local_poses, total, dist):
    """chunks from shard_range may differ by one row: pad to the largest chunk, gather, then strip the padding"""
    world = dist.get_world_size()
    return _gather_ragged(local_poses, [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))], dist)


# ---- SE3 on (n, 7) tensors: unit quaternion (x, y, z, w) + translation, the layout of every pose in this package
def se3_compose(A, B):
    """A * B, row-wise (first apply B, then A), torch tensors (n, 7) f64"""
    ax, ay, az, aw = A[:, 0], A[:, 1], A[:, 2], A[:, 3]
    bx, by, bz, bw = B[:, 0], B[:, 1], B[:, 2], B[:, 3]
    q = torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], 1)
    q = q / q.norm(dim=1, keepdim=True)
    qv, t = A[:, :3], B[:, 4:]
    c1 = torch.cross(qv, t, dim=1)
    rot = t + 2.0 * (aw[:, None] * c1 + torch.cross(qv, c1, dim=1))  # R(qa) t
    return torch.cat([q, rot + A[:, 4:]], 1)


def chain_poses(rel):
    """rel[j-1] = T_{j,j-1} for j = 1..F-1  ->  (F, 7) absolute poses T_j = T_{j,j-1} T_{j-1}, T_0 = identity (world = first camera).
    Inclusive scan with log2(F) vectorised rounds (SE3 composition is associative), on whatever device `rel` lives."""
    F = rel.shape[0] + 1
    ident = torch.zeros((1, 7), dtype=rel.dtype, device=rel.device); ident[0, 3] = 1.0
    X = torch.cat([ident, rel], 0)
    d = 1
    while d < F:
        Y = X.clone()
        Y[d:] = se3_compose(X[d:], X[:-d])
        X = Y
        d *= 2
    return X


def gather_and_chain(local_rel, total, dist, world, rank):
    """sequence mode: local_rel = this rank's relative poses (owned_pose_range rows, frame order).  Returns the chained (total, 7)
    trajectory on rank 0 (the gather is an all-gather, so other ranks could chain too; they return None)."""
    if world > 1:
        sizes = [hi - lo for lo, hi in (owned_pose_range(total, r, world) for r in range(world))]
        rel = _gather_ragged(local_rel, sizes, dist)
    else:
        rel = local_rel
    return chain_poses(rel) if rank == 0 else None


def camera_centre(T):
    """camera position in the world frame, -R^T t, of one pose (7,) tensor / array"""
    T = np.asarray(T.detach().cpu() if hasattr(T, "detach") else T, np.float64)
    x, y, z, w = T[:4]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return -R.T @ T[4:]
