"""CPU, world_size 2 over gloo: the N>1 host path of throughput mode (contiguous keyframe shards, pose gather)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_range():
    from stereo_visual_slam_amd.sharding import shard_range
    assert [shard_range(50, r, 8) for r in range(8)] == [(0, 7), (7, 14), (14, 20), (20, 26), (26, 32), (32, 38), (38, 44), (44, 50)]
    for total in (0, 1, 7, 64, 4541):
        for world in (1, 2, 3, 8):
            ch = [shard_range(total, r, world) for r in range(world)]
            assert ch[0][0] == 0 and ch[-1][1] == total and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
            assert max(h - l for l, h in ch) - min(h - l for l, h in ch) <= 1


def _worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stereo_visual_slam_amd.sharding import gather_poses, gather_ragged_poses, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, rank, world)
        # "pose" of global keyframe g = [g, g+0.5, ..., rank marker]: each rank only knows its own chunk
        local = torch.tensor([[g + 0.125 * c for c in range(7)] for g in range(lo, hi)], dtype=torch.float64).reshape(-1, 7)
        allp = gather_ragged_poses(local, total, dist)
        eq = gather_poses(torch.full((3, 7), float(rank), dtype=torch.float64), dist)
        # barrier + max-over-ranks timing reduction, as bench.py does
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, allp.numpy(), eq.numpy(), float(t.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_pose_gather():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, world = 11, 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.array([[g + 0.125 * c for c in range(7)] for g in range(total)])
    for rank, allp, eq, tmax in res:
        assert allp.shape == (total, 7) and np.array_equal(allp, want)
        assert np.array_equal(eq, np.repeat(np.arange(world, dtype=np.float64), 3)[:, None] * np.ones((1, 7)))
        assert tmax == float(world)


# ---------------------------------------------------------------------------------------------- sequence mode (BASELINE config 5)
def _rel_pose(j):
    """deterministic relative pose T_{j,j-1} of global frame j (forward ~1 m, small yaw/pitch), numpy (7,)"""
    rng = np.random.default_rng(1000 + j)
    w = rng.normal(0, 0.03, 3); th = np.linalg.norm(w)
    q = np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
    return np.concatenate([q, rng.normal(0, 0.05, 3) + np.array([0, 0, -1.0])])


def _chain_reference(F):
    """plain sequential chaining T_j = T_{j,j-1} T_{j-1} in numpy, the definition the scan must reproduce"""
    def mul(A, B):
        ax, ay, az, aw = A[:4]; bx, by, bz, bw = B[:4]
        q = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
        x, y, z, w = A[:4]
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        return np.concatenate([q / np.linalg.norm(q), R @ B[4:] + A[4:]])
    T = [np.array([0, 0, 0, 1, 0, 0, 0.0])]
    for j in range(1, F):
        T.append(mul(_rel_pose(j), T[-1]))
    return np.array(T)


def test_halo_and_owned_ranges():
    from stereo_visual_slam_amd.sharding import halo_start, owned_pose_range, shard_range
    for F in (2, 7, 50, 4541):
        for world in (1, 2, 3, 8):
            if F < 2 * world:
                continue
            covered = []
            for r in range(world):
                lo, hi = shard_range(F, r, world)
                h = halo_start(lo)
                assert h == (lo - 1 if lo > 0 else 0)
                produced = list(range(h + 1, hi))          # a rank that processes frames [h, hi) estimates T_{j,j-1} for j = h+1 .. hi-1
                assert produced == list(range(*owned_pose_range(F, r, world)))
                covered += produced
            assert covered == list(range(1, F))             # every relative pose exactly once, in frame order


def test_chain_poses_scan_equals_sequential():
    from stereo_visual_slam_amd.sharding import chain_poses
    for F in (2, 3, 17, 50, 129):
        rel = torch.tensor(np.array([_rel_pose(j) for j in range(1, F)]), dtype=torch.float64)
        got = chain_poses(rel).numpy(); want = _chain_reference(F)
        assert got.shape == (F, 7) and np.allclose(got, want, rtol=0, atol=1e-12)


def _seq_worker(rank, world, port, F, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stereo_visual_slam_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(F, rank, world)
        h = sharding.halo_start(lo)
        # what the per-rank pipeline produces: one relative pose per processed frame except the first (halo or frame 0)
        local = torch.tensor(np.array([_rel_pose(j) for j in range(h + 1, hi)]).reshape(-1, 7), dtype=torch.float64)
        traj = sharding.gather_and_chain(local, F, dist, world, rank)
        q.put((rank, None if traj is None else traj.numpy()))
    finally:
        dist.destroy_process_group()


def _run_seq(world, F):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_seq_worker, args=(r, world, port, F, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_sequence_sharding_two_and_three_ranks_match_single_rank():
    """config 5 host logic with injected poses: contiguous chunks + 1-frame halo, ragged gather, chaining on rank 0 == the
    single-rank trajectory of the same 50-frame sequence"""
    F = 50
    want = _chain_reference(F)
    for world in (2, 3, 8):   # (8: BASELINE config 5's rank count -- chunks of 6-7 frames)
        res = _run_seq(world, F)
        assert all(res[r] is None for r in range(1, world))
        assert res[0].shape == (F, 7) and np.allclose(res[0], want, rtol=0, atol=1e-12)


# ---------------------------------------------------------------- round 5: the carry chain of sequence mode with local BA
def _carry_worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stereo_visual_slam_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = sharding.shard_range(total, rank, world)
        h_lo = sharding.halo_start(lo, sharding.WINDOW_HALO)
        c_out = sharding.carry_out_frame(total, rank, world)
        buf = torch.zeros((8, 4), dtype=torch.float32)
        seen = {}

        def build(carry):
            # injected "window builder": what it was handed, and a carry-out that names the frame it describes and everything upstream of it
            seen["in"] = None if carry is None else carry.clone()
            up = 0.0 if carry is None else float(carry[0, 0])
            out = torch.zeros((8, 4), dtype=torch.float32)
            out[:, 0] = up + 1.0                    # chain depth: rank r's carry-out has passed through r + 1 builders
            out[:, 1] = float(h_lo + c_out)         # the global frame the record is for
            return out

        sharding.chain_carry(dist, rank, world, build, buf, h_lo > 0, c_out > 0)
        q.put((rank, lo, hi, h_lo, c_out, None if seen["in"] is None else seen["in"][0].tolist()))
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,total", [(2, 50), (3, 50), (3, 20), (8, 50), (8, 4541), (8, 11)])
def test_carry_chain_reaches_every_rank_in_order(world, total):
    """sequence mode's one serial step over gloo: rank r gets exactly the record rank r - 1 built for the first frame of r's chunk (halo included), built
    after r - 1 received its own; ranks whose chunk starts at frame 0 get none"""
    import pytest  # noqa: F401
    from stereo_visual_slam_amd import sharding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = [ctx.Process(target=_carry_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p_ in procs:
        p_.join(60)
    depth = 0
    for rank, lo, hi, h_lo, c_out, got in res:
        assert h_lo == max(lo - sharding.WINDOW_HALO, 0)
        if h_lo == 0:
            assert got is None                      # nothing upstream of frame 0
        else:
            assert got is not None and got[1] == float(h_lo)          # the record describes THIS chunk's first frame ...
            assert got[0] == float(depth)                              # ... and was built by the rank before, after its own carry arrived
        if c_out > 0:
            depth += 1
        else:
            assert rank == world - 1 or sharding.halo_start(sharding.shard_range(total, rank + 1, world)[0], sharding.WINDOW_HALO) == 0


def test_owned_windows_cover_every_frame_once():
    from stereo_visual_slam_amd import sharding
    for total in (20, 50, 4541):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_range(total, r, world)
                h = sharding.halo_start(lo, sharding.WINDOW_HALO)
                assert lo - h == min(lo, sharding.WINDOW_HALO)
                seen += list(range(lo, hi))
                c = sharding.carry_out_frame(total, r, world)
                if r + 1 < world:
                    lo_n = sharding.shard_range(total, r + 1, world)[0]
                    assert h + c == sharding.halo_start(lo_n, sharding.WINDOW_HALO) or (c == 0 and sharding.halo_start(lo_n, sharding.WINDOW_HALO) <= h)
                    assert c < hi - h
            assert seen == list(range(total))


def test_eight_rank_pose_gather():
    """the ragged all-gather of config 5 at its real rank count (50 frames -> 7, 7, 6, 6, 6, 6, 6, 6)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    total, world = 50, 8
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.array([[g + 0.125 * c for c in range(7)] for g in range(total)])
    for rank, allp, eq, tmax in res:
        assert allp.shape == (total, 7) and np.array_equal(allp, want) and tmax == float(world)


# ---------------------------------------------------------------- round 6: two pipelines in flight and their collectives
class _FakePipe:
    """stands in for a KeyframePipeline: step() "computes" poses that name (rank, pipeline, how often it has stepped) after a rank-dependent delay"""
    def __init__(self, rank, idx, B):
        self.rank, self.idx, self.B, self.n = rank, idx, B, 0
        self.d_Tpnp = torch.zeros((B, 7), dtype=torch.float64)
        self.stream = None

    def step(self):
        import time
        time.sleep(0.002 * ((self.rank * 7 + self.n * 3 + self.idx) % 5))     # ranks drift apart: a reordered collective would pair the wrong steps
        self.n += 1
        self.d_Tpnp[:] = 1000.0 * self.rank + 100.0 * self.idx + self.n


class _FakeRing:
    def __init__(self, rank, P, B):
        self.pipes = [_FakePipe(rank, i, B) for i in range(P)]
        self.k = 0

    def next_pipe(self):
        return self.pipes[self.k % len(self.pipes)]


def _inflight_worker(rank, world, port, steps, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from stereo_visual_slam_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ring = _FakeRing(rank, 2, 3)
        got = []
        for k in range(steps):
            pipe, allp = sharding.throughput_step(ring, dist, world)
            got.append((pipe.idx, allp.clone().numpy()))
        _, one = sharding.throughput_step(ring, dist, world, serial=True)     # the one-batch-in-flight repeat: always pipeline 0, the ring does not advance
        q.put((rank, got, one.numpy(), ring.k))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_two_pipelines_in_flight_gather_in_step_order(world):
    """throughput mode, N > 1, two batches in flight: step k runs on pipeline k mod 2 and then all-gathers THAT pipeline's poses.  Every rank must issue
    these collectives in the same order (RCCL pairs collectives of a communicator by issue order, whichever stream carries them): block r of step k's
    gather is rank r's pipeline k mod 2 after its (k // 2 + 1)-th step -- for every k, on every rank, with the ranks drifting apart in time."""
    steps = 9
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_inflight_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, got, one, k_end in res:
        assert k_end == steps and len(got) == steps
        for k, (idx, allp) in enumerate(got):
            assert idx == k % 2 and allp.shape == (world * 3, 7)
            for r in range(world):
                assert np.all(allp[3 * r: 3 * r + 3] == 1000.0 * r + 100.0 * (k % 2) + (k // 2 + 1)), (rank, k, r)
        n0 = (steps + 1) // 2 + 1                                              # pipeline 0 has stepped ceil(steps / 2) times, then once more
        for r in range(world):
            assert np.all(one[3 * r: 3 * r + 3] == 1000.0 * r + n0)
