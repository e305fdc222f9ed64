"""BASELINE config 5 on ONE GPU: the loop of run_vslam.cpp:40 cut into contiguous chunks with a 1-frame halo (sharding.py), each chunk
run through the device pipeline exactly as a rank would run it, the per-chunk relative poses put through the ragged gather and the
chaining scan -- and the result required to equal the unsharded run of the same 50 frames bit for bit, and the CPU oracle to 1e-4.
(The collective itself is covered with gloo on CPU in tests/test_sharding_gloo.py; here a stand-in `dist` object plays the wire so
that the product's own gather / strip / chain code runs on device tensors.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = 50          # BASELINE.json configs[0] / configs[4]: the first 50 stereo pairs
ANMS = 500      # the reference's own ANMS size (visual_odometry.cpp:82); keeps the oracle leg of this test short


class WireStandIn:
    """all_gather_into_tensor over ranks that run one after the other in this process: every rank's call deposits its block; the call that
    completes the set (rank 0 is run last) receives the concatenation in rank order -- what RCCL delivers to every rank at once"""

    def __init__(self, world):
        self.world, self.rank, self.blocks = world, 0, {}

    def is_initialized(self):
        return True

    def get_world_size(self):
        return self.world

    def all_gather_into_tensor(self, out, inp):
        import torch
        self.blocks[self.rank] = inp.clone()
        if len(self.blocks) == self.world:
            out.copy_(torch.cat([self.blocks[r] for r in range(self.world)]))
        else:
            out.zero_()


@pytest.fixture(scope="module")
def rendered(synth):
    return synth.stereo_sequence(F, seed=0, workers=8)


def _run_chunk(rendered, lo, hi):
    import torch
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    pipe = KeyframePipeline(hi - lo, device=0, anms_num=ANMS, with_ba=False, unique_frames=F, frame_range=(lo, hi, F), sequence=rendered)
    try:
        pipe.step()
        out = pipe.download()
        assert (pipe.vo.orb_status(2 * (hi - lo)) == 0).all()
        rel = pipe.d_Tpnp[:hi - lo - 1].clone()      # item i = T_{lo+i+1, lo+i}
        torch.cuda.synchronize()
        return rel, out, pipe.frame_of
    finally:
        pipe.close()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_sequence_equals_unsharded(rendered, world):
    import torch
    from stereo_visual_slam_amd import sharding
    rel_full, out_full, frame_of = _run_chunk(rendered, 0, F)
    assert frame_of == list(range(F))
    traj_full = sharding.chain_poses(rel_full).cpu().numpy()
    wire = WireStandIn(world)
    traj = None
    for rank in reversed(range(world)):              # rank 0 last: it is the one that chains
        lo, hi = sharding.shard_range(F, rank, world)
        h_lo = sharding.halo_start(lo)
        rel, out, _ = _run_chunk(rendered, h_lo, hi)
        p_lo, p_hi = sharding.owned_pose_range(F, rank, world)
        assert rel.shape[0] == p_hi - p_lo           # the halo frame adds the pose of the chunk's first frame, nothing else
        # the chunk's relative poses are the same bits as the unsharded run's rows (same kernels, same inputs, batch-size independent)
        assert np.array_equal(rel.cpu().numpy(), rel_full[p_lo - 1:p_hi - 1].cpu().numpy()), rank
        wire.rank = rank
        traj = sharding.gather_and_chain(rel, F, wire, world, rank)
    assert traj is not None and traj.shape == (F, 7)
    assert np.array_equal(traj.cpu().numpy(), traj_full)          # bit for bit
    c = sharding.camera_centre(traj[-1])
    assert 30.0 < c[2] < 70.0                                      # ~1 m per frame forward


def test_sequence_relative_poses_match_oracle(rendered, oracle):
    """the frame-to-frame poses of the device pipeline vs the CPU oracle's composite on the same frames (1e-4), counts exact"""
    rel, out, _ = _run_chunk(rendered, 0, F)
    rel = rel.cpu().numpy()
    O = oracle
    ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
    n_f = 16                                                        # oracle leg: the first 16 frames (about 10 s of CPU)
    prev = None
    for f in range(n_f):
        L, R = rendered[f][0], rendered[f][1]
        kL, dL = O.feature_detection(L, 3000, ANMS); kR, dR = O.feature_detection(R, 3000, ANMS)
        assert out["cnt"][f] == len(kL) and out["cnt"][F + f] == len(kR)
        m = O.feature_matching(dL, dR, 1.0)
        assert out["nlr"][f] == len(m)
        uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1); uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
        xyz, valid, _ = O.triangulate_dlt(uvL, uvR, ident)
        if prev is not None:
            pk, pd, pm, pxyz, pvalid = prev
            fm = O.feature_matching(pd, dL, 1.0)
            kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
            li = kp2lr[fm["queryIdx"]]
            ok = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
            assert out["nf2f"][f - 1] == len(fm) and out["pn"][f - 1] == ok.sum()
            T, _, ninl, _ = O.pnp_motion_only(pxyz[li[ok]], np.stack([kL["x"][fm["trainIdx"][ok]], kL["y"][fm["trainIdx"][ok]]], 1), ident, iters=10)
            assert out["ninl"][f - 1] == ninl
            q = T[:4] if np.dot(T[:4], rel[f - 1][:4]) >= 0 else -T[:4]
            assert np.allclose(rel[f - 1][:4], q, rtol=1e-4, atol=1e-7) and np.allclose(rel[f - 1][4:], T[4:], rtol=1e-4, atol=1e-6), f
        prev = (kL, dL, m, xyz, valid)
