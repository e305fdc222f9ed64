"""BASELINE config 5 on ONE GPU: the loop of run_vslam.cpp:40 cut into contiguous chunks with a halo (sharding.py), each chunk
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


# ---------------------------------------------------------------- round 5: the BA windows of a sharded sequence ARE the unsharded run's windows
class WireP2P(WireStandIn):
    """... plus send / recv between ranks that run one after the other in ascending order (the carry chain is serial by nature)"""

    def __init__(self, world):
        super().__init__(world)
        self.mail = {}

    def send(self, t, dst):
        self.mail[dst] = t.clone()

    def recv(self, t, src):
        t.copy_(self.mail.pop(self.rank))


def _front_and_pose(rendered, h_lo, hi, sync=True):
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    pipe = KeyframePipeline(hi - h_lo, device=0, anms_num=ANMS, with_ba=True, ba_windows="tracks", unique_frames=F, frame_range=(h_lo, hi, F), sequence=rendered)
    pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track()
    if sync:
        pipe.vo.sync()
        import torch
        torch.cuda.synchronize()   # (the test reads the poses from torch's default stream; the product orders its gather on the pipeline's stream)
    return pipe


def _owned_results(pipe, first):
    import torch
    pipe.vo.sync(); torch.cuda.synchronize()
    n = pipe.B - first
    assert (pipe.vo.ba_status(n) == 0).all() and int(pipe.ba_build_status.item()) == 0
    lm_off = pipe.ba_lm_off.cpu().numpy(); e_off = pipe.ba_e_off.cpu().numpy()
    T = pipe.ba_T.cpu().numpy(); inl = pipe.ba_inl.cpu().numpy(); xyz = pipe.ba_xyz.cpu().numpy(); uv = pipe.ba_uv.cpu().numpy()
    kf = pipe.ba_kf.cpu().numpy(); lm = pipe.ba_lm.cpu().numpy(); nkf = pipe.ba_nkf.cpu().numpy()
    out = []
    for b in range(first, pipe.B):
        out.append(dict(T=T[b, :nkf[b]].copy(), inl=inl[lm_off[b]:lm_off[b + 1]].copy(), xyz=xyz[lm_off[b]:lm_off[b + 1]].copy(), n_kf=int(nkf[b]),
                        uv=uv[e_off[b]:e_off[b + 1]].copy(), kf=kf[e_off[b]:e_off[b + 1]].copy(), lm=lm[e_off[b]:e_off[b + 1]].copy()))
    return out


def test_sequence_windows_without_global_synchronize(rendered):
    """sequence_windows_and_ba used to chain the poses and allocate the carry buffer on the CALLER's stream while the pose stage was still
    writing d_Tpnp on the pipeline's (non-blocking) stream; the other tests hid that behind a device-wide synchronize.  Here nothing synchronises between
    the pose stage and the call, and the call is made from the default stream with a VIEW of d_Tpnp: same windows, same BA results."""
    from stereo_visual_slam_amd import sharding
    pipe = _front_and_pose(rendered, 0, F)
    try:
        sharding.sequence_windows_and_ba(pipe, F, 0, 1, None, pipe.d_Tpnp[:F - 1].clone())
        ref = _owned_results(pipe, 0)
    finally:
        pipe.close()
    for _ in range(3):   # (a race would not show every time)
        pipe = _front_and_pose(rendered, 0, F, sync=False)
        try:
            sharding.sequence_windows_and_ba(pipe, F, 0, 1, None, pipe.d_Tpnp[:F - 1])
            got = _owned_results(pipe, 0)
        finally:
            pipe.close()
        assert len(got) == len(ref)
        for g, r in zip(got, ref):
            for k in ("n_kf", "kf", "lm", "uv", "xyz", "inl", "T"):
                assert np.array_equal(g[k], r[k]), k


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_sequence_ba_windows_equal_unsharded(rendered, world):
    """Sequence mode with local BA: every rank processes its chunk plus a 9-frame halo, the relative poses are gathered and chained, the track state at a chunk's
    first frame arrives from the rank before it (a 64 KB carry), and the BA schedule runs on the windows of the OWNED frames.  Every such window -- its
    keyframe poses, landmark positions, reliable flags, edges -- and its result (poses after the 5+5+10+10 schedule, landmark flags) must equal the unsharded
    pass over the 50 frames bit for bit, for 2 and for 8 ranks (chunks of 6-7 frames: shorter than the halo)."""
    import torch
    from stereo_visual_slam_amd import sharding
    # unsharded: one chunk, world 1, the same product code
    pipe = _front_and_pose(rendered, 0, F)
    try:
        rel_full = pipe.d_Tpnp[:F - 1].clone()
        first = sharding.sequence_windows_and_ba(pipe, F, 0, 1, None, rel_full)
        assert first == 0
        ref = _owned_results(pipe, 0)
    finally:
        pipe.close()
    assert len(ref) == F and max(r["n_kf"] for r in ref) == 10 and sum(len(r["inl"]) for r in ref) > 10000
    # sharded: phase 1 on every rank (front end, pose stage, deposit of the owned relative poses), then the gathered list, then the serial carry chain
    wire = WireP2P(world)
    pipes, rel_all = {}, None
    try:
        for rank in range(world):
            lo, hi = sharding.shard_range(F, rank, world)
            h_lo = sharding.halo_start(lo, sharding.WINDOW_HALO)
            pipes[rank] = _front_and_pose(rendered, h_lo, hi)
            p_lo, p_hi = sharding.owned_pose_range(F, rank, world)
            rel = pipes[rank].d_Tpnp[p_lo - h_lo - 1: p_hi - h_lo - 1]
            assert np.array_equal(rel.cpu().numpy(), rel_full[p_lo - 1:p_hi - 1].cpu().numpy()), rank
            wire.rank = rank
            rel_all = sharding.gather_relative_poses(rel, F, wire, world)     # (complete on the last caller: what the all-gather hands every rank)
        assert np.array_equal(rel_all.cpu().numpy(), rel_full.cpu().numpy())
        n_checked = 0
        for rank in range(world):                                              # ascending: rank r needs the carry of rank r - 1
            lo, hi = sharding.shard_range(F, rank, world)
            wire.rank = rank
            first = sharding.sequence_windows_and_ba(pipes[rank], F, rank, world, wire, rel_all)
            assert first == lo - sharding.halo_start(lo, sharding.WINDOW_HALO)
            got = _owned_results(pipes[rank], first)
            assert len(got) == hi - lo
            for j, g in enumerate(got):
                r = ref[lo + j]
                for k in ("n_kf", "kf", "lm", "uv", "xyz", "inl", "T"):
                    assert np.array_equal(g[k], r[k]), (rank, lo + j, k)
                n_checked += 1
        assert n_checked == F
    finally:
        for p_ in pipes.values():
            p_.close()
