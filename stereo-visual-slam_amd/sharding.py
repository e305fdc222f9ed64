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


WINDOW_HALO = 9   # Map::num_keyframes_ - 1 (map.hpp:22): window b holds keyframes [b - 9, b]


def halo_start(lo, halo=1):
    """first frame a rank has to PROCESS to own frames from `lo` on: `halo` frames before them (1: the pose stage needs the predecessor of its first
    frame; WINDOW_HALO: the local-BA window of its first frame reaches nine keyframes back), clipped at the start of the sequence"""
    return max(lo - halo, 0)


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


def throughput_step(ring, dist, world, serial=False):
    """One step of throughput mode on a ring of pipelines in flight (pipeline.PipelineRing): step k runs on pipeline k mod P without waiting for step
    k - 1, then -- N > 1 -- its poses are all-gathered, ordered behind the step on THAT pipeline's stream.  Every rank calls this the same number of
    times with the same P, so the all-gathers of the communicator are issued in the same order on every rank whatever stream each one rides on (the
    one thing RCCL requires; tests/test_sharding_gloo.py::test_two_pipelines_in_flight_gather_in_step_order).  Returns (pipeline, gathered or None)."""
    import contextlib
    pipe = ring.pipes[0] if serial else ring.next_pipe()
    pipe.step()
    if not serial:
        ring.k += 1
    if world <= 1:
        return pipe, None
    stream = getattr(pipe, "stream", None)
    with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
        return pipe, gather_poses(pipe.d_Tpnp, dist)


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


def gather_relative_poses(local_rel, total, dist, world):
    """every rank's owned relative poses (owned_pose_range rows each) -> the (total - 1, 7) list in frame order, on EVERY rank (an all-gather)"""
    if world <= 1:
        return local_rel
    sizes = [hi - lo for lo, hi in (owned_pose_range(total, r, world) for r in range(world))]
    return _gather_ragged(local_rel, sizes, dist)


def carry_out_frame(total, rank, world, halo=WINDOW_HALO):
    """local index (inside rank `rank`'s chunk, which starts at halo_start(lo, halo)) of the frame where the NEXT rank's chunk starts, or 0: no carry needed
    (last rank, or the next chunk starts at frame 0 of the sequence)"""
    if rank + 1 >= world:
        return 0
    lo, _ = shard_range(total, rank, world)
    lo_n, _ = shard_range(total, rank + 1, world)
    return max(halo_start(lo_n, halo) - halo_start(lo, halo), 0)


def chain_carry(dist, rank, world, build, carry_buf, needs_in, sends_out):
    """The one SERIAL step of sequence mode: the track state at a chunk boundary depends on every frame before it, so rank r can build its windows only
    after rank r - 1 has (a 64 KB record, two kernels: microseconds against the chunk's front end and BA).  rank r receives the carry of rank r - 1
    (if needs_in), calls build(carry or None) -> its own carry-out tensor (or None), sends it on (if sends_out)."""
    if rank > 0 and needs_in:
        dist.recv(carry_buf, src=rank - 1)
    out = build(carry_buf if (rank > 0 and needs_in) else None)
    if rank + 1 < world and sends_out:
        dist.send(out.contiguous(), dst=rank + 1)
    return out


def sequence_windows_and_ba(pipe, total, rank, world, dist, rel_all):
    """sequence mode, after the front end and the pose stage ran on the chunk [halo_start(lo, WINDOW_HALO), hi): the chunk's frames get their poses in the
    SEQUENCE's world (rel_all = every frame's relative pose, gathered; chained identically on every rank), the window builder continues the tracks
    that cross the chunk's start (carry from the previous rank), and the BA schedule runs on the windows of the frames this rank owns -- the same
    windows, bit for bit, as one unsharded pass over the sequence builds (tests/test_gpu_sequence.py).  Returns the local index of the first owned window."""
    if total < world:
        raise ValueError("sequence mode needs at least one frame per rank (%d frames, %d ranks): an empty shard would neither send nor need a carry" % (total, world))
    lo, hi = shard_range(total, rank, world)
    h_lo = halo_start(lo, WINDOW_HALO)
    c_out = carry_out_frame(total, rank, world)
    # sender and receiver decide from the SAME number: rank r expects a carry exactly when rank r - 1 sends one
    needs_in = rank > 0 and carry_out_frame(total, rank - 1, world) > 0
    # Everything below runs on the pipeline's stream: rel_all is usually a view of pipe.d_Tpnp (written by the pose stage on pipe.stream, a
    # non-blocking stream) or a gathered tensor produced on the caller's current stream; G / T_abs / buf are read by the builder on pipe.stream.
    # Allocating or chaining them on another stream would read half-written poses and let the allocator recycle G under a pending copy.
    cur = torch.cuda.current_stream(rel_all.device) if rel_all.is_cuda else None
    if cur is not None and cur != pipe.stream:
        pipe.stream.wait_stream(cur)               # (a gathered rel_all was produced on the caller's stream)
        rel_all.record_stream(pipe.stream)
    with torch.cuda.stream(pipe.stream):
        G = chain_poses(rel_all)                   # (total, 7): the same bits on every rank
        T_abs = G[h_lo:hi]
        buf = torch.zeros((pipe.cap, 4), dtype=torch.float32, device=rel_all.device)

        def build(carry):
            return pipe.stage_build_windows_chunk(T_abs, carry, c_out)

        if world > 1:
            chain_carry(dist, rank, world, build, buf, needs_in, c_out > 0)
        else:
            build(None)
    first = lo - h_lo
    pipe.ba_schedule_from(first)
    return first


def camera_centre(T):
    """camera position in the world frame, -R^T t, of one pose (7,) tensor / array"""
    T = np.asarray(T.detach().cpu() if hasattr(T, "detach") else T, np.float64)
    x, y, z, w = T[:4]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return -R.T @ T[4:]
