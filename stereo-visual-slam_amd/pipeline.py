"""Throughput-mode keyframe pipeline: B independent stereo keyframes per step, everything device-resident.

One step = the hot path BASELINE.json's metric names, over a batch of B stereo keyframes:
    ORB detect+ANMS+describe on the B left and B right images (one 2B-image launch set)
 -> L/R cross-checked Hamming match + gate          (stereo association, north_star)
 -> gather matched pixels -> rectified-stereo DLT triangulation (+ depth gates of set_ref_3d_position)
 -> frame-to-frame match: keyframe b-1 (query) vs keyframe b (train)      (VO::feature_matching, :575)
 -> 3D(prev, triangulated) - 2D(cur) gather -> motion-only LM pose (10 its) (VO::motion_estimation substitute)
 -> BA windows built ON THE DEVICE from the step's own tracks (ba_windows="tracks": window b = keyframes [b-9, b] of the batch, the
    landmarks / observations VO::insert_key_frame would have recorded, optimization.cpp:127-214) -- or B canned synthetic windows of the
    BASELINE config-4 shape (ba_windows="synthetic": 10 KF x ~3000 landmarks)
 -> local BA on the B windows: schedule 5+5+10 LM + 10 pose-only (run_vslam.cpp:58-71)
torch is used only for device memory and the stream; every stage is a C-ABI call into libvslam_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import BaBatch, DMATCH_DTYPE, KEYPOINT_DTYPE, TracksIn, VO, default_params
from . import synth


class KeyframePipeline:
    def __init__(self, B, device=0, anms_num=1500, n_lm=3000, n_kf=10, unique_frames=64, unique_windows=None, seed=0, verbose=False,
                 with_ba=True, depth="match", frame_range=None, render_workers=0, sequence=None, ba_windows="synthetic",
                 lm_per_window=None, edges_per_window=None, pose="lm"):
        """depth = "match": north_star stage (right-image ORB, L/R match, DLT); "sgbm": the reference's own depth path
        (VO::disparity_map + Frame::find_3d on the left keypoints; the right image is only consumed by SGBM).
        Inputs: ONE rendered sequence of `unique_frames` consecutive stereo keyframes, laid over the batch as a ping-pong
        (0, 1, ..., n-1, n-2, ..., 1, 0, 1, ...), so that every item b >= 1 and its predecessor are adjacent frames of the same
        scene (driving the sequence backwards is as valid a frame-to-frame pair as driving it forwards); `unique_windows` BA
        windows (default: one per batch item).  frame_range = (first, last + 1, F): sequence mode -- the batch is the contiguous
        chunk [first, last] of an F-frame sequence (frame f shows ping-pong frame f of the SAME rendered scene on every rank)."""
        assert depth in ("match", "sgbm") and ba_windows in ("synthetic", "tracks") and pose in ("lm", "ransac")
        self.depth = depth
        self.pose = pose   # "lm": north_star motion-only LM; "ransac": the reference's cv::solvePnPRansac(..., 100, 4.0, 0.99) (visual_odometry.cpp:277)
        self.ba_windows = ba_windows
        self.B = B
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        # One explicit stream for the library AND for the few torch ops between its calls (copies of initial guesses, the pose
        # gather): torch's default stream has handle 0, which the C-ABI reads as "create your own stream" -- the two would race.
        self.stream = torch.cuda.Stream(self.dev)
        self.with_ba = with_ba
        p = default_params(max_batch=2 * B, anms_num=anms_num)
        self.vo = VO(params=p, device=device, stream=self.stream.cuda_stream)
        self.cap = p.kp_capacity
        self.w, self.h = p.img_w, p.img_h
        self.pitch = (self.w + 63) // 64 * 64
        self.img_bytes = self.pitch * self.h
        d = self.dev
        # ---- inputs: 2B images [left 0..B-1 | right 0..B-1]; consecutive keyframes of `unique_scenes` short sequences
        imgs = np.zeros((2 * B, self.h, self.pitch), np.uint8)
        n_u = max(2, min(unique_frames, B if frame_range is None else int(frame_range[2]))) if (B > 1 or frame_range is not None) else 1
        # `sequence`: an already rendered synth.stereo_sequence(n_u, seed) (tests that build several pipelines over the same frames)
        seq = sequence if sequence is not None else synth.stereo_sequence(n_u, seed=seed, w=self.w, h=self.h, workers=render_workers)
        assert len(seq) == n_u, (len(seq), n_u)
        if verbose:
            print("rendered %d stereo keyframes" % n_u, flush=True)
        period = max(2 * (n_u - 1), 1)
        f0 = 0
        if frame_range is not None:
            f0 = int(frame_range[0])
            assert int(frame_range[1]) - f0 == B
        self.frame_of = [(t if t < n_u else period - t) for t in ((f0 + b) % period for b in range(B))]
        for b in range(B):
            L, R, _, _ = seq[self.frame_of[b]]
            imgs[b, :, :self.w] = L
            imgs[B + b, :, :self.w] = R
        self.unique_frames = n_u
        self.h_seq = seq   # (kept: a second pipeline over the same frames needs no second rendering)
        self.h_imgs = imgs
        self.h_imgs_unique_left = np.stack([np.pad(f[0], ((0, 0), (0, self.pitch - self.w))) for f in seq])
        self.h_imgs_unique_right = np.stack([np.pad(f[1], ((0, 0), (0, self.pitch - self.w))) for f in seq])
        self.d_imgs = torch.from_numpy(imgs).to(d)
        # ---- ORB outputs
        self.d_kps = torch.zeros((2 * B, self.cap, 28), dtype=torch.uint8, device=d)
        self.d_desc = torch.zeros((2 * B, self.cap, 32), dtype=torch.uint8, device=d)
        self.d_cnt = torch.zeros(2 * B, dtype=torch.int32, device=d)
        # ---- matches (L/R and frame-to-frame)
        self.d_gap = torch.ones(B, dtype=torch.float64, device=d)
        self.d_lr = torch.zeros((B, self.cap, 16), dtype=torch.uint8, device=d)
        self.d_nlr = torch.zeros(B, dtype=torch.int32, device=d)
        self.d_f2f = torch.zeros((B, self.cap, 16), dtype=torch.uint8, device=d)
        self.d_nf2f = torch.zeros(B, dtype=torch.int32, device=d)
        # ---- triangulation
        self.d_uvL = torch.zeros((B, self.cap, 2), dtype=torch.float32, device=d)
        self.d_uvR = torch.zeros((B, self.cap, 2), dtype=torch.float32, device=d)
        ident = np.tile(np.array([0, 0, 0, 1, 0, 0, 0], np.float64), (B, 1))
        self.d_Tident = torch.from_numpy(ident).to(d)
        self.d_xyz = torch.zeros((B, self.cap, 3), dtype=torch.float32, device=d)
        self.d_valid = torch.zeros((B, self.cap), dtype=torch.uint8, device=d)
        self.d_rel = torch.zeros((B, self.cap), dtype=torch.uint8, device=d)
        if depth == "sgbm":
            self.d_disp = torch.zeros((B, self.h, self.w), dtype=torch.float32, device=d)
            ident_lr = np.zeros((B, self.cap), DMATCH_DTYPE)
            ident_lr["queryIdx"] = np.arange(self.cap)[None, :]; ident_lr["trainIdx"] = np.arange(self.cap)[None, :]
            self.d_lr = torch.from_numpy(ident_lr.view(np.uint8).reshape(B, self.cap, 16)).to(d)  # keypoint i <-> landmark slot i
        # ---- PnP
        self.d_kp2lr = torch.zeros((B, self.cap), dtype=torch.int32, device=d)
        self.d_pxyz = torch.zeros((B, self.cap, 3), dtype=torch.float32, device=d)
        self.d_puv = torch.zeros((B, self.cap, 2), dtype=torch.float32, device=d)
        self.d_pn = torch.zeros(B, dtype=torch.int32, device=d)
        self.d_Tpnp = torch.from_numpy(ident.copy()).to(d)
        self.d_inl = torch.zeros((B, self.cap), dtype=torch.uint8, device=d)
        self.d_ninl = torch.zeros(B, dtype=torch.int32, device=d)
        # ---- local-BA windows built on the device from this step's tracks (vslam_build_windows_dev)
        if with_ba and ba_windows == "tracks":
            self.n_kf = n_kf
            # capacities of the concatenated window arrays: L/R match + DLT gives ~1/3 of the keypoints a depth, the disparity map nearly all
            if lm_per_window is None:
                lm_per_window = n_kf * anms_num if depth == "sgbm" else max(4 * n_kf * anms_num // 10, 1024)
            if edges_per_window is None:
                edges_per_window = lm_per_window + lm_per_window // 3
            self.lm_capacity, self.edge_capacity = B * lm_per_window, B * edges_per_window
            self.ba_T = torch.zeros((B, n_kf, 7), dtype=torch.float64, device=d)
            self.ba_xyz = torch.zeros((self.lm_capacity, 3), dtype=torch.float32, device=d)
            self.ba_rel = torch.zeros(self.lm_capacity, dtype=torch.uint8, device=d)
            self.ba_inl = torch.zeros(self.lm_capacity, dtype=torch.uint8, device=d)
            self.ba_kf = torch.zeros(self.edge_capacity, dtype=torch.int32, device=d)
            self.ba_lm = torch.zeros(self.edge_capacity, dtype=torch.int32, device=d)
            self.ba_uv = torch.zeros((self.edge_capacity, 2), dtype=torch.float32, device=d)
            self.ba_lm_off = torch.zeros(B + 1, dtype=torch.int32, device=d)
            self.ba_e_off = torch.zeros(B + 1, dtype=torch.int32, device=d)
            self.ba_nkf = torch.zeros(B, dtype=torch.int32, device=d)
            self.ba_build_status = torch.zeros(1, dtype=torch.int32, device=d)
            self.ba_chi2 = torch.zeros(1, dtype=torch.float64, device=d)
            tr = TracksIn()
            tr.n_frames = B; tr.kp_capacity = self.cap; tr.lr_capacity = self.cap; tr.match_capacity = self.cap; tr.pnp_capacity = self.cap
            tr.d_kps = self.d_kps.data_ptr(); tr.d_lr = self.d_lr.data_ptr(); tr.d_nlr = self.d_nlr.data_ptr(); tr.d_xyz = self.d_xyz.data_ptr()
            tr.d_valid = self.d_valid.data_ptr(); tr.d_reliable = self.d_rel.data_ptr(); tr.d_f2f = self.d_f2f.data_ptr()
            tr.d_nf2f = self.d_nf2f.data_ptr(); tr.d_pose_inlier = self.d_inl.data_ptr(); tr.d_T_rel = self.d_Tpnp.data_ptr()
            tr.d_nkps = self.d_cnt.data_ptr()   # (the first B counts: the left images)
            self.tracks = tr
            bb = BaBatch()
            bb.n_windows = B; bb.n_kf = n_kf
            bb.d_lm_off = self.ba_lm_off.data_ptr(); bb.d_edge_off = self.ba_e_off.data_ptr(); bb.d_T_c_w = self.ba_T.data_ptr()
            bb.d_xyz = self.ba_xyz.data_ptr(); bb.d_reliable = self.ba_rel.data_ptr(); bb.d_lm_inlier = self.ba_inl.data_ptr()
            bb.d_kf_idx = self.ba_kf.data_ptr(); bb.d_lm_idx = self.ba_lm.data_ptr(); bb.d_uv = self.ba_uv.data_ptr()
            bb.d_chi2 = None; bb.d_stats = None; bb.K4 = None; bb.d_n_kf = self.ba_nkf.data_ptr()
            bb.total_lm = self.lm_capacity; bb.total_edge = self.edge_capacity
            self.ba_batch = bb
            self.unique_windows = B
        # ---- canned local-BA windows
        if with_ba and ba_windows == "synthetic":
            unique_windows = B if unique_windows is None else max(1, min(unique_windows, B))
            self.unique_windows = unique_windows
            self.window_seed0 = seed + 100
            wins = [synth.ba_window_fast(n_kf=n_kf, n_lm=n_lm, seed=self.window_seed0 + i) for i in range(unique_windows)]
            self.h_windows = wins
            lm_off, e_off = [0], [0]
            T0, xyz, kf, lm, uv = [], [], [], [], []
            for b in range(B):
                wn = wins[b % unique_windows]
                T0.append(wn["T0"]); xyz.append(wn["xyz"]); kf.append(wn["kf_idx"]); lm.append(wn["lm_idx"]); uv.append(wn["uv"])
                lm_off.append(lm_off[-1] + len(wn["xyz"])); e_off.append(e_off[-1] + len(wn["kf_idx"]))
            self.n_kf = n_kf
            self.ba_T0 = torch.from_numpy(np.stack(T0)).to(d)
            self.ba_T = self.ba_T0.clone()
            self.ba_xyz = torch.from_numpy(np.concatenate(xyz)).to(d)
            self.ba_kf = torch.from_numpy(np.concatenate(kf)).to(d)
            self.ba_lm = torch.from_numpy(np.concatenate(lm)).to(d)
            self.ba_uv = torch.from_numpy(np.concatenate(uv)).to(d)
            self.ba_lm_off = torch.tensor(lm_off, dtype=torch.int32, device=d)
            self.ba_e_off = torch.tensor(e_off, dtype=torch.int32, device=d)
            self.ba_inl = torch.ones(lm_off[-1], dtype=torch.uint8, device=d)
            self.ba_chi2 = torch.zeros(e_off[-1], dtype=torch.float64, device=d)
            self.total_lm, self.total_edge = lm_off[-1], e_off[-1]
            self.h_lm_off = np.array(lm_off, np.int64)
            self.edges_per_window = e_off[-1] / B
            self.lms_per_window = lm_off[-1] / B
            bb = BaBatch()
            bb.n_windows = B; bb.n_kf = n_kf
            bb.d_lm_off = self.ba_lm_off.data_ptr(); bb.d_edge_off = self.ba_e_off.data_ptr(); bb.d_T_c_w = self.ba_T.data_ptr()
            bb.d_xyz = self.ba_xyz.data_ptr(); bb.d_reliable = None; bb.d_lm_inlier = self.ba_inl.data_ptr()
            bb.d_kf_idx = self.ba_kf.data_ptr(); bb.d_lm_idx = self.ba_lm.data_ptr(); bb.d_uv = self.ba_uv.data_ptr()
            bb.d_chi2 = None; bb.d_stats = None  # per-edge chi2 is internal to optimize_map, not one of its outputs
            bb.total_lm = self.total_lm; bb.total_edge = self.total_edge
            self.ba_batch = bb
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ stages
    def stage_orb(self):
        n_img = self.B if self.depth == "sgbm" else 2 * self.B  # the reference never detects on the right image
        self.vo.feature_detection_dev(self.d_imgs.data_ptr(), self.img_bytes, self.pitch, n_img, self.d_kps.data_ptr(),
                                      self.d_desc.data_ptr(), self.d_cnt.data_ptr())

    def stage_stereo_match(self):
        B, cap = self.B, self.cap
        vo = self.vo
        if self.depth == "sgbm":
            # VO::disparity_map + Frame::find_3d / gates of set_ref_3d_position on every left keypoint (visual_odometry.cpp:159-217)
            vo.disparity_map_dev(self.d_imgs.data_ptr(), self.d_imgs.data_ptr() + B * self.img_bytes, self.img_bytes, self.pitch, self.w, self.h, B,
                                 self.d_disp.data_ptr())
            vo.find_3d_disparity_dev(self.d_kps.data_ptr(), self.d_cnt.data_ptr(), cap, B, self.d_disp.data_ptr(), self.w, self.h,
                                     self.d_Tident.data_ptr(), self.d_xyz.data_ptr(), self.d_valid.data_ptr(), self.d_rel.data_ptr())
            with torch.cuda.stream(self.stream):
                self.d_nlr.copy_(self.d_cnt[:B])
            return
        # L/R: query = left descriptors of keyframe b, train = right descriptors of keyframe b
        vo.feature_matching_dev(self.d_desc.data_ptr(), cap * 32, self.d_cnt.data_ptr(), self.d_desc.data_ptr() + B * cap * 32, cap * 32,
                                self.d_cnt.data_ptr() + 4 * B, self.d_gap.data_ptr(), 1, B, cap, self.d_lr.data_ptr(), cap, self.d_nlr.data_ptr())
        vo.gather_matched_uv_dev(self.d_kps.data_ptr(), self.d_kps.data_ptr() + B * cap * 28, cap, self.d_lr.data_ptr(), self.d_nlr.data_ptr(),
                                 cap, B, self.d_uvL.data_ptr(), self.d_uvR.data_ptr())
        vo.triangulate_dev(self.d_uvL.data_ptr(), self.d_uvR.data_ptr(), self.d_nlr.data_ptr(), cap, B, self.d_Tident.data_ptr(),
                           self.d_xyz.data_ptr(), self.d_valid.data_ptr(), self.d_rel.data_ptr())

    def stage_track(self):
        """keyframe b-1 -> keyframe b for b = 1..B-1 (the first keyframe of the batch has no predecessor in the batch)"""
        B, cap = self.B, self.cap
        if B < 2:
            return
        vo, n = self.vo, B - 1
        # query = left descriptors of keyframe b-1 (item i = b-1), train = left descriptors of keyframe b
        vo.feature_matching_dev(self.d_desc.data_ptr(), cap * 32, self.d_cnt.data_ptr(), self.d_desc.data_ptr() + cap * 32, cap * 32,
                                self.d_cnt.data_ptr() + 4, self.d_gap.data_ptr(), 1, n, cap, self.d_f2f.data_ptr(), cap, self.d_nf2f.data_ptr())
        vo.build_pnp_inputs_dev(self.d_f2f.data_ptr(), self.d_nf2f.data_ptr(), cap, self.d_lr.data_ptr(), self.d_nlr.data_ptr(), cap,
                                self.d_xyz.data_ptr(), self.d_valid.data_ptr(), self.d_kps.data_ptr() + cap * 28, cap, n, self.d_kp2lr.data_ptr(),
                                self.d_pxyz.data_ptr(), self.d_puv.data_ptr(), self.d_pn.data_ptr(), cap)
        if self.pose == "ransac":   # no pose guess is consumed (useExtrinsicGuess = false)
            vo.pnp_ransac_dev(self.d_pxyz.data_ptr(), self.d_puv.data_ptr(), self.d_pn.data_ptr(), cap, n, self.d_Tpnp.data_ptr(), 100, 4.0, 0.99,
                              self.d_inl.data_ptr(), self.d_ninl.data_ptr(), None)
            return
        with torch.cuda.stream(self.stream):
            self.d_Tpnp.copy_(self.d_Tident)
        vo.motion_estimation_dev(self.d_pxyz.data_ptr(), self.d_puv.data_ptr(), self.d_pn.data_ptr(), cap, n, self.d_Tpnp.data_ptr(), 10,
                                 self.d_inl.data_ptr(), self.d_ninl.data_ptr())

    def stage_build_windows(self):
        """optimize_map's graph build (optimization.cpp:127-214) + insert_key_frame's bookkeeping (visual_odometry.cpp:363-424) on the device"""
        self.vo.build_windows_dev(self.tracks, self.n_kf, self.lm_capacity, self.edge_capacity, self.ba_batch, self.ba_build_status.data_ptr())

    def stage_build_windows_chunk(self, T_abs, carry_in=None, carry_out_frame=0):
        """sequence mode: this batch is the chunk [first, last] of a longer sequence.  T_abs (B, 7): the poses of its frames in the SEQUENCE's world (gathered
        relative poses, chained); carry_in (kp_capacity, 4) f32 or None: the tracks that reach the chunk's first frame from before it (the previous rank's
        carry-out); carry_out_frame > 0: also export that record for this local frame (returned tensor) -- the first frame of the next rank's chunk."""
        d = self.dev
        if not hasattr(self, "d_T_abs"):
            self.d_T_abs = torch.zeros((self.B, 7), dtype=torch.float64, device=d)
            self.d_carry_in = torch.zeros((self.cap, 4), dtype=torch.float32, device=d)
            self.d_carry_out = torch.zeros((self.cap, 4), dtype=torch.float32, device=d)
        with torch.cuda.stream(self.stream):
            self.d_T_abs.copy_(T_abs.to(d))
            if carry_in is not None:
                self.d_carry_in.copy_(carry_in.to(d))
        tr = self.tracks
        tr.d_T_abs = self.d_T_abs.data_ptr()
        tr.d_carry_in = self.d_carry_in.data_ptr() if carry_in is not None else None
        tr.d_carry_out = self.d_carry_out.data_ptr() if carry_out_frame > 0 else None
        tr.carry_out_frame = int(carry_out_frame)
        self.stage_build_windows()
        tr.d_T_abs = None; tr.d_carry_in = None; tr.d_carry_out = None; tr.carry_out_frame = 0
        return self.d_carry_out if carry_out_frame > 0 else None

    def ba_schedule_from(self, first):
        """the BA schedule on the built windows [first, B) only (sequence mode: the windows of the frames this rank OWNS; the ones before belong to its halo)"""
        from . import BaBatch
        b = BaBatch()
        for f_, _ in BaBatch._fields_:
            setattr(b, f_, getattr(self.ba_batch, f_))
        b.n_windows = self.B - first
        b.d_lm_off = self.ba_lm_off[first:].data_ptr(); b.d_edge_off = self.ba_e_off[first:].data_ptr()
        b.d_T_c_w = self.ba_T[first:].data_ptr(); b.d_n_kf = self.ba_nkf[first:].data_ptr()
        self._ba_view = b   # (kept alive until the next call)
        self.vo.ba_batch_dev(b, schedule=1)

    def stage_ba(self):
        if not self.with_ba:
            return
        if self.ba_windows == "tracks":
            self.stage_build_windows()
            self.vo.ba_batch_dev(self.ba_batch, schedule=1)
            return
        with torch.cuda.stream(self.stream):
            self.ba_T.copy_(self.ba_T0)
            self.ba_inl.fill_(1)
        self.vo.ba_batch_dev(self.ba_batch, schedule=1)

    def step(self):
        self.stage_orb()
        self.stage_stereo_match()
        self.stage_track()
        self.stage_ba()

    # ------------------------------------------------------------------ host views (tests / reports)
    def download(self):
        self.vo.sync()
        torch.cuda.synchronize(self.dev)
        B, cap = self.B, self.cap
        out = dict(cnt=self.d_cnt.cpu().numpy(), nlr=self.d_nlr.cpu().numpy(), nf2f=self.d_nf2f.cpu().numpy(), pn=self.d_pn.cpu().numpy(),
                   ninl=self.d_ninl.cpu().numpy(), Tpnp=self.d_Tpnp.cpu().numpy())
        out["kps"] = self.d_kps.cpu().numpy().reshape(2 * B, -1).view(KEYPOINT_DTYPE).reshape(2 * B, cap)
        out["desc"] = self.d_desc.cpu().numpy()
        out["lr"] = self.d_lr.cpu().numpy().reshape(B, -1).view(DMATCH_DTYPE).reshape(B, cap)
        out["f2f"] = self.d_f2f.cpu().numpy().reshape(B, -1).view(DMATCH_DTYPE).reshape(B, cap)
        out["xyz"] = self.d_xyz.cpu().numpy(); out["valid"] = self.d_valid.cpu().numpy(); out["rel"] = self.d_rel.cpu().numpy()
        out["pxyz"] = self.d_pxyz.cpu().numpy(); out["puv"] = self.d_puv.cpu().numpy(); out["inl"] = self.d_inl.cpu().numpy()
        if self.with_ba:
            out["ba_T"] = self.ba_T.cpu().numpy(); out["ba_inl"] = self.ba_inl.cpu().numpy(); out["ba_chi2"] = self.ba_chi2.cpu().numpy()
        if self.with_ba and self.ba_windows == "tracks":
            for k, t in (("ba_lm_off", self.ba_lm_off), ("ba_e_off", self.ba_e_off), ("ba_nkf", self.ba_nkf), ("ba_xyz", self.ba_xyz), ("ba_rel", self.ba_rel),
                         ("ba_kf", self.ba_kf), ("ba_lm", self.ba_lm), ("ba_uv", self.ba_uv), ("ba_build_status", self.ba_build_status)):
                out[k] = t.cpu().numpy()
        return out

    def close(self):
        self.vo.close()


class PipelineRing:
    """P keyframe pipelines -- a context, a HIP stream and the device buffers of one batch each -- whose steps are in flight TOGETHER.

    Batches are independent (throughput mode), and a single step cannot keep the chip busy on its own: the BA kernel runs one window per
    CU for milliseconds (its last windows leave CUs empty, its SIMDs issue ~half the time), every ORB launch ends in a tail, the small
    bookkeeping kernels between them are latency.  With step k + 1 queued on a second stream the hardware scheduler fills those holes
    with the other batch's workgroups: 38.6 k -> 41.6 k keyframes/s at 2 x 512 keyframes in flight, results bit-identical to one pipeline
    stepping alone (tests/test_gpu_pipeline.py).  A third pipeline adds nothing (41.8 k at 3 x 256).
    step() hands the next batch to pipeline k mod P and returns without synchronising; sync() waits for all of them."""

    def __init__(self, P, B, distinct=False, sequences=None, **kw):
        """distinct: every pipeline renders (or is handed, `sequences[i]`) ITS OWN sequence of frames -- seed + 7919 i -- so that the batches in flight
        together are different data; otherwise all pipelines show the first one's frames (tests compare their results bit for bit)"""
        assert P >= 1
        kw0 = dict(kw)
        if sequences is not None:
            kw0["sequence"] = sequences[0]
        first = KeyframePipeline(B, **kw0)
        self.pipes = [first]
        for i in range(1, P):
            kw2 = dict(kw)
            kw2["unique_frames"] = first.unique_frames
            kw2["verbose"] = False
            if sequences is not None and i < len(sequences):
                kw2["sequence"] = sequences[i]; kw2["render_workers"] = 0
            elif distinct:
                kw2["seed"] = kw.get("seed", 0) + 7919 * i
            else:
                kw2["sequence"] = first.h_seq          # the other pipelines show the same rendered frames: no second rendering
                kw2["render_workers"] = 0
            self.pipes.append(KeyframePipeline(B, **kw2))
        self.distinct = distinct or (sequences is not None and len(sequences) > 1)
        self.k = 0

    def __len__(self):
        return len(self.pipes)

    def next_pipe(self):
        return self.pipes[self.k % len(self.pipes)]

    def step(self):
        p = self.next_pipe()
        p.step()
        self.k += 1
        return p

    def sync(self):
        for p in self.pipes:
            p.vo.sync()

    def close(self):
        for p in self.pipes:
            if p.vo.h:
                p.close()
