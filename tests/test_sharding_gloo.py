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
