"""CPU baseline worker: one stereo keyframe of the hot path through the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Used by bench.py's `cpu_baseline` leg (single thread in-process, and all cores through a process pool whose workers import
THIS module -- numpy + the oracle, no torch, no GPU).  The work per keyframe mirrors one item of the GPU step
(stereo-visual-slam_amd/pipeline.py): ORB(3000) -> ANMS -> rBRIEF on the left and right image, L/R cross-check match + gate,
DLT triangulation, frame-to-frame match against the previous keyframe, motion-only LM pose (10 its), and the local-BA
schedule 5+5+10 LM + 10 pose-only (run_vslam.cpp:58-71) on one 10-keyframe window.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

IDENT = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
_state = {}


def init(images_npy, w, anms, n_kf, n_lm, window_seed0):
    """worker initialiser: map the image stack ((2U, h, pitch) u8: U left images then U right images), load the oracle"""
    import oracle as O
    O.lib()
    _state.update(imgs=np.load(images_npy, mmap_mode="r"), w=int(w), anms=int(anms), n_kf=int(n_kf), n_lm=int(n_lm), seed0=int(window_seed0), O=O)
    return True


def front_end(u):
    """ORB + L/R match + DLT of unique frame u -> (kL, dL, lr matches, xyz, valid)"""
    O, imgs, w, anms = _state["O"], _state["imgs"], _state["w"], _state["anms"]
    U = imgs.shape[0] // 2
    kL, dL = O.feature_detection(np.ascontiguousarray(imgs[u][:, :w]), 3000, anms)
    kR, dR = O.feature_detection(np.ascontiguousarray(imgs[U + u][:, :w]), 3000, anms)
    m = O.feature_matching(dL, dR, 1.0)
    uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1)
    uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
    xyz, valid, rel = O.triangulate_dlt(uvL, uvR, IDENT)
    return kL, dL, m, xyz, valid, len(kR)


def track(prev, cur):
    """frame-to-frame match + 3D-2D gather + motion-only LM (prev -> cur); returns (T, n_points, n_inliers, n_f2f)"""
    O = _state["O"]
    pk, pd, pm, pxyz, pvalid = prev[:5]
    kL, dL = cur[0], cur[1]
    f = O.feature_matching(pd, dL, 1.0)
    kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
    li = kp2lr[f["queryIdx"]]
    ok = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
    T, ninl = IDENT.copy(), 0
    if ok.sum() >= 1:
        T, _, ninl, _ = O.pnp_motion_only(pxyz[li[ok]], np.stack([kL["x"][f["trainIdx"][ok]], kL["y"][f["trainIdx"][ok]]], 1), IDENT, iters=10)
    return T, int(ok.sum()), int(ninl), len(f)


def ba_schedule(win):
    """run_vslam.cpp:58-71 on one window dict (synth.ba_window*): returns (poses after the schedule, landmark inlier flags)"""
    O = _state["O"]
    T = win["T0"].copy(); inl = np.ones(len(win["xyz"]), np.uint8)
    for iters, upd in ((5, False), (5, False), (10, True)):
        act = inl.astype(bool)[win["lm_idx"]]
        T2, _, chi2, _ = O.local_ba(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=iters)
        _, inl, _, _ = O.chi2_classify(chi2, win["lm_idx"][act], inl)
        if upd:
            T = T2
    act = inl.astype(bool)[win["lm_idx"]]
    T2, chi2, _ = O.pose_only_window(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=10)
    _, inl, _, _ = O.chi2_classify(chi2, win["lm_idx"][act], inl)
    return T2, inl


def ba_schedule_built(win):
    """the same schedule on a window built from tracks (oracle.build_windows slices): optimize_map's landmark filter is
    is_inlier && reliable_depth_ (optimization.cpp:160), optimize_pose_only's is_inlier (:334)"""
    O = _state["O"]
    T = win["T0"].copy(); inl = np.ones(len(win["xyz"]), np.uint8); rel = win["reliable"].astype(bool)
    for iters, upd in ((5, False), (5, False), (10, True)):
        act = (inl.astype(bool) & rel)[win["lm_idx"]]
        T2, _, chi2, _ = O.local_ba(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=iters)
        _, inl, _, _ = O.chi2_classify(chi2, win["lm_idx"][act], inl)
        if upd:
            T = T2
    act = inl.astype(bool)[win["lm_idx"]]
    T2, chi2, _ = O.pose_only_window(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=10)
    _, inl, _, _ = O.chi2_classify(chi2, win["lm_idx"][act], inl)
    return T2, inl


def track_full(prev, cur):
    """like track(), returning what the window builder needs as well: (T, f2f matches, inlier flags of the pose inputs)"""
    O = _state["O"]
    pk, pd, pm, pxyz, pvalid = prev[:5]
    kL, dL = cur[0], cur[1]
    f = O.feature_matching(pd, dL, 1.0)
    kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
    li = kp2lr[f["queryIdx"]]
    ok = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
    T, inl = IDENT.copy(), np.zeros(0, np.uint8)
    if ok.sum() >= 1:
        T, inl, _, _ = O.pnp_motion_only(pxyz[li[ok]], np.stack([kL["x"][f["trainIdx"][ok]], kL["y"][f["trainIdx"][ok]]], 1), IDENT, iters=10)
    return T, f, np.asarray(inl, np.uint8)


def front_end_full(u):
    """front_end() plus the reliable flags (the window builder's input)"""
    O, imgs, w, anms = _state["O"], _state["imgs"], _state["w"], _state["anms"]
    U = imgs.shape[0] // 2
    kL, dL = O.feature_detection(np.ascontiguousarray(imgs[u][:, :w]), 3000, anms)
    kR, dR = O.feature_detection(np.ascontiguousarray(imgs[U + u][:, :w]), 3000, anms)
    m = O.feature_matching(dL, dR, 1.0)
    uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1)
    uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
    xyz, valid, rel = O.triangulate_dlt(uvL, uvR, IDENT)
    return kL, dL, m, xyz, valid, len(kR), rel


def pack_tracks(fronts, tracks, cap):
    """per-frame results -> the padded arrays of vslam_tracks_in (host side), for oracle.build_windows"""
    O = _state["O"]
    F = len(fronts)
    kps = np.zeros((F, cap), O.KEYPOINT_DTYPE); lr = np.zeros((F, cap), O.DMATCH_DTYPE); nlr = np.zeros(F, np.int32)
    xyz = np.zeros((F, cap, 3), np.float32); valid = np.zeros((F, cap), np.uint8); rel = np.zeros((F, cap), np.uint8)
    f2f = np.zeros((max(F - 1, 1), cap), O.DMATCH_DTYPE); nf2f = np.zeros(max(F - 1, 1), np.int32)
    inl = np.zeros((max(F - 1, 1), cap), np.uint8); T_rel = np.tile(IDENT, (max(F - 1, 1), 1))
    for f, fr in enumerate(fronts):
        kL, _, m, x, v, _, r = fr
        kps[f, :len(kL)] = kL; lr[f, :len(m)] = m; nlr[f] = len(m); xyz[f, :len(m)] = x; valid[f, :len(m)] = v; rel[f, :len(m)] = r
    for i, (T, f, il) in enumerate(tracks):
        f2f[i, :len(f)] = f; nf2f[i] = len(f); inl[i, :len(il)] = il; T_rel[i] = T
    return kps, lr, nlr, xyz, valid, rel, f2f[:max(F - 1, 0)] if F > 1 else f2f[:0], nf2f[:max(F - 1, 0)], inl[:max(F - 1, 0)] if F > 1 else inl[:0], T_rel[:max(F - 1, 0)]


def track_pair(pair):
    """pool task: (front-end result of frame i, of frame i + 1) -> track_full"""
    return track_full(pair[0], pair[1])


def window_slice(w, b):
    """window b of an oracle.build_windows result as the dict ba_schedule_built takes"""
    lo, hi, e0, e1 = int(w["lm_off"][b]), int(w["lm_off"][b + 1]), int(w["edge_off"][b]), int(w["edge_off"][b + 1])
    nk = int(w["n_kf"][b])
    return dict(T0=w["T"][b][:nk].copy(), xyz=w["xyz"][lo:hi].copy(), reliable=w["reliable"][lo:hi].copy(), kf_idx=w["kf_idx"][e0:e1].copy(),
                lm_idx=w["lm_idx"][e0:e1].copy(), uv=w["uv"][e0:e1].copy())


def window_of(b):
    from stereo_visual_slam_amd import synth
    return synth.ba_window_fast(n_kf=_state["n_kf"], n_lm=_state["n_lm"], seed=_state["seed0"] + b)


def chunk(task):
    """all-cores task: keyframes [b0, b1) of the batch; frames[b] = unique frame shown at batch position b.  The predecessor of
    b0 is recomputed as a halo (front end only) so that every keyframe of the chunk has its frame-to-frame stage."""
    b0, b1, frames, n_windows = task
    prev = front_end(frames[b0 - 1]) if b0 > 0 else None
    done = 0
    for b in range(b0, b1):
        cur = front_end(frames[b])
        if prev is not None:
            track(prev, cur)
        ba_schedule(window_of(b % n_windows))
        prev = cur
        done += 1
    return done


def warm(_):
    import time
    time.sleep(0.05)
    return os.getpid()
