"""stereo-visual-slam_amd -- MI355X (gfx950) build of the stereo-VO hot path of shangzhouye/stereo-visual-slam.

This Python layer is plumbing: a ctypes binding of the C-ABI in include/vslam_hip.h (libvslam_hip.so, hand-written
HIP kernels) whose method names mirror the reference's C++ surface (visual_odometry.hpp, optimization.hpp) so that
the parity tests read like calls into the reference.  There is NO CPU fallback: if libvslam_hip.so is missing or no
GPU is visible, every compute call raises.

Import name: `stereo_visual_slam_amd` (the repo-root shim maps it to this hyphenated directory).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSLAM_LIB: path of an alternative build of the SAME library (kernel tuning variants, tools/build_variant.sh); never a fallback
_SO = os.environ.get("VSLAM_LIB") or os.path.join(_HERE, "libvslam_hip.so")

VSLAM_OK, VSLAM_ERR_ARG, VSLAM_ERR_HIP, VSLAM_ERR_CAPACITY, VSLAM_ERR_NO_DEVICE = 0, -1, -2, -3, -4
MAX_KF = 12
MAX_ROWS = 4096

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class Params(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("max_batch", C.c_int32), ("orb_nfeatures", C.c_int32),
                ("anms_num", C.c_int32), ("fast_threshold", C.c_int32), ("kp_capacity", C.c_int32),
                ("cam", C.c_double * 5), ("depth_min", C.c_double), ("depth_max", C.c_double),
                ("depth_reliable", C.c_double), ("match_ratio", C.c_double), ("match_gap_thr", C.c_double),
                ("huber_delta", C.c_double), ("pnp_reproj_thr", C.c_double), ("stereo_row_tol", C.c_double),
                ("struct_size", C.c_int32), ("abi_version", C.c_int32)]


ABI_VERSION = 5  # VSLAM_ABI_VERSION of include/vslam_hip.h this binding was written against


class LmStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("total_trials", C.c_int32), ("chi2_init", C.c_double),
                ("chi2_final", C.c_double), ("lambda_final", C.c_double), ("chi2_iter", C.c_double * 32),
                ("lambda_iter", C.c_double * 32), ("trials_iter", C.c_int32 * 32)]

    def as_dict(self):
        n = min(self.iterations, 32)
        return dict(iterations=self.iterations, total_trials=self.total_trials, chi2_init=self.chi2_init,
                    chi2_final=self.chi2_final, lambda_final=self.lambda_final, chi2_iter=list(self.chi2_iter)[:n],
                    lambda_iter=list(self.lambda_iter)[:n], trials_iter=list(self.trials_iter)[:n])


class BaBatch(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("n_kf", C.c_int32), ("d_lm_off", C.c_void_p), ("d_edge_off", C.c_void_p),
                ("d_T_c_w", C.c_void_p), ("d_xyz", C.c_void_p), ("d_reliable", C.c_void_p), ("d_lm_inlier", C.c_void_p),
                ("d_kf_idx", C.c_void_p), ("d_lm_idx", C.c_void_p), ("d_uv", C.c_void_p), ("d_chi2", C.c_void_p),
                ("d_stats", C.c_void_p), ("total_lm", C.c_int32), ("total_edge", C.c_int32), ("K4", C.c_void_p), ("d_n_kf", C.c_void_p)]


class TracksIn(C.Structure):
    """vslam_tracks_in: the front end's per-frame results (device pointers) that vslam_build_windows_dev turns into BA windows"""
    _fields_ = [("n_frames", C.c_int32), ("kp_capacity", C.c_int32), ("lr_capacity", C.c_int32), ("match_capacity", C.c_int32),
                ("pnp_capacity", C.c_int32), ("d_kps", C.c_void_p), ("d_lr", C.c_void_p), ("d_nlr", C.c_void_p), ("d_xyz", C.c_void_p),
                ("d_valid", C.c_void_p), ("d_reliable", C.c_void_p), ("d_f2f", C.c_void_p), ("d_nf2f", C.c_void_p),
                ("d_pose_inlier", C.c_void_p), ("d_T_rel", C.c_void_p), ("d_nkps", C.c_void_p),
                # ABI rev 5: a chunk of a longer sequence -- absolute poses, carry across the chunk boundaries (all optional)
                ("d_T_abs", C.c_void_p), ("d_carry_in", C.c_void_p), ("d_carry_out", C.c_void_p), ("carry_out_frame", C.c_int32)]


# every symbol include/vslam_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "vslam_default_params", "vslam_create", "vslam_destroy", "vslam_last_error", "vslam_version", "vslam_sync",
    "vslam_device_bytes", "vslam_kernel_names", "vslam_feature_detection", "vslam_orb_detect", "vslam_anms",
    "vslam_orb_compute", "vslam_feature_detection_dev", "vslam_feature_matching", "vslam_feature_matching_dev",
    "vslam_find_3d_disparity", "vslam_triangulate", "vslam_triangulate_dev", "vslam_gather_matched_uv_dev",
    "vslam_pnp_motion_only", "vslam_pnp_motion_only_dev", "vslam_check_motion", "vslam_local_ba",
    "vslam_pose_only_window", "vslam_ba_batch_dev", "vslam_ba_status_dev", "vslam_ba_schedule_passes_dev", "vslam_ba_deferred_dev", "vslam_edge_jacobians", "vslam_orb_status_dev", "vslam_orb_level", "vslam_dev_alloc",
    "vslam_dev_free", "vslam_dev_upload", "vslam_dev_download", "vslam_dev_memset", "vslam_build_pnp_inputs_dev",
    "vslam_profile_enable", "vslam_profile_read", "vslam_profile_intervals", "vslam_hbm_copy_probe", "vslam_disparity_map", "vslam_disparity_map_dev", "vslam_pnp_ransac", "vslam_pnp_ransac_models", "vslam_find_3d_disparity_dev",
    "vslam_abi_version", "vslam_hbm_copy_probe_variants", "vslam_hbm_copy_probe_variant", "vslam_sgbm_status_dev", "vslam_set_tuning", "vslam_build_windows_dev", "vslam_pnp_ransac_dev",
]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_int32), ("calls", C.c_int32)]


class StageInterval(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("t0_ms", C.c_double), ("t1_ms", C.c_double)]


class VslamError(RuntimeError):
    pass


def build(force=False):
    """compile libvslam_hip.so for gfx950 (hipcc cross-compiles without a GPU)"""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-j8", "-s"]
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean", "-s"])
    subprocess.check_call(cmd)
    return _SO


_lib = None


def load_library():
    """load libvslam_hip.so; raises (loudly) when it has not been built"""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # torch bundles its own libamdhip64 (same SONAME).  Import it first so that this library binds to the HIP runtime
        # torch already loaded: two HIP runtimes in one process cannot both own the GPU.
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(_SO):
        raise VslamError("libvslam_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(_SO)
    lib.vslam_last_error.restype = C.c_char_p
    lib.vslam_version.restype = C.c_char_p
    lib.vslam_kernel_names.restype = C.c_char_p
    lib.vslam_device_bytes.restype = C.c_size_t
    lib.vslam_device_bytes.argtypes = [C.c_void_p]
    lib.vslam_create.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.vslam_destroy.argtypes = [C.c_void_p]
    lib.vslam_sync.argtypes = [C.c_void_p]
    if not hasattr(lib, "vslam_abi_version") or lib.vslam_abi_version() != ABI_VERSION:
        raise VslamError("libvslam_hip.so was built from a different ABI revision than this binding (%s vs %d): rebuild it"
                         % (lib.vslam_abi_version() if hasattr(lib, "vslam_abi_version") else "pre-3", ABI_VERSION))
    # positional signatures that changed between ABI revisions get explicit argtypes: a stale call site fails in ctypes, not in the kernel
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    lib.vslam_local_ba.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.vslam_pose_only_window.argtypes = [vp, i32, vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp]
    lib.vslam_ba_batch_dev.argtypes = [vp, vp, i32, i32, i32, i32, i32]
    lib.vslam_pnp_ransac.argtypes = [vp, vp, vp, i32, vp, i32, dbl, dbl, i32, vp, vp, vp]
    lib.vslam_pnp_ransac_dev.argtypes = [vp, vp, vp, vp, i32, i32, vp, i32, dbl, dbl, vp, vp, vp]
    lib.vslam_pnp_ransac_models.argtypes = [vp, vp, vp, i32, vp, i32, dbl, dbl, i32, vp, vp, vp, vp, vp]
    lib.vslam_feature_matching_dev.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t, vp, vp, i32, i32, i32, vp, i32, vp]
    lib.vslam_hbm_copy_probe_variant.argtypes = [vp, C.c_size_t, i32, i32, vp, vp]
    _lib = lib
    return lib


def default_params(**kw):
    p = Params()
    load_library().vslam_default_params(C.byref(p))
    for k, v in kw.items():
        if k == "cam":
            for i in range(5):
                p.cam[i] = float(v[i])
        else:
            setattr(p, k, v)
    return p


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _desc(d):
    d = np.ascontiguousarray(d, np.uint8)
    if d.size == 0:
        d = d.reshape(0, 32)
    assert d.ndim == 2 and d.shape[1] == 32, d.shape
    return d


class VO:
    """Device context + the reference's VO / optimisation method names over the C-ABI.

    Host-buffer methods take and return numpy arrays (one call per reference method); `*_dev` methods take raw
    device pointers (ints, e.g. torch.Tensor.data_ptr()) and run batched + asynchronously on the context stream.
    """

    def __init__(self, params=None, device=0, stream=None, **kw):
        self.lib = load_library()
        self.params = params if params is not None else default_params(**kw)
        h = C.c_void_p()
        rc = self.lib.vslam_create(C.byref(self.params), int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != VSLAM_OK:
            raise VslamError("vslam_create failed (%d): %s" % (rc, self.lib.vslam_last_error().decode()))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vslam_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != VSLAM_OK:
            raise VslamError("%s failed (%d): %s" % (what, rc, self.lib.vslam_last_error().decode()))

    def sync(self):
        self._chk(self.lib.vslam_sync(self.h), "vslam_sync")

    @property
    def device_bytes(self):
        return int(self.lib.vslam_device_bytes(self.h))

    # ------------------------------------------------------------ VO::feature_detection (visual_odometry.cpp:70-94)
    def _img(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.ndim == 2
        return img

    def feature_detection(self, img):
        img = self._img(img)
        cap = self.params.kp_capacity
        kps = np.zeros(cap, KEYPOINT_DTYPE); desc = np.zeros((cap, 32), np.uint8); n = C.c_int()
        self._chk(self.lib.vslam_feature_detection(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps),
                                                   _p(desc), cap, C.byref(n)), "vslam_feature_detection")
        return kps[:n.value].copy(), desc[:n.value].copy()

    def orb_detect(self, img):
        img = self._img(img)
        cap = self.params.kp_capacity
        kps = np.zeros(cap, KEYPOINT_DTYPE); n = C.c_int()
        self._chk(self.lib.vslam_orb_detect(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), cap,
                                            C.byref(n)), "vslam_orb_detect")
        return kps[:n.value].copy()

    def adaptive_non_maximal_suppresion(self, kps, num=500):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        buf = np.zeros(max(len(kps), 1), KEYPOINT_DTYPE); buf[:len(kps)] = kps
        n = C.c_int()
        self._chk(self.lib.vslam_anms(self.h, _p(buf), len(kps), int(num), C.byref(n)), "vslam_anms")
        return buf[:n.value].copy()

    def orb_compute(self, img, kps):
        img = self._img(img)
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
        buf = np.zeros(max(len(kps), 1), KEYPOINT_DTYPE); buf[:len(kps)] = kps
        desc = np.zeros((max(len(kps), 1), 32), np.uint8); n = C.c_int()
        self._chk(self.lib.vslam_orb_compute(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(buf), len(kps),
                                             _p(desc), C.byref(n)), "vslam_orb_compute")
        return buf[:n.value].copy(), desc[:n.value].copy()

    def feature_detection_dev(self, d_imgs, img_bytes, pitch, B, d_kps, d_desc, d_count):
        self._chk(self.lib.vslam_feature_detection_dev(self.h, _p(d_imgs), C.c_size_t(img_bytes), int(pitch), int(B), _p(d_kps),
                                                       _p(d_desc), _p(d_count)), "vslam_feature_detection_dev")

    def orb_level(self, item, level, blurred):
        """one level of the (blurred) pyramid of image `item` of the most recent ORB launch (diagnostic)"""
        w, h = C.c_int(0), C.c_int(0)
        buf = np.zeros((self.params.img_h, (self.params.img_w + 63) & ~63), np.uint8)
        self._chk(self.lib.vslam_orb_level(self.h, int(item), int(level), int(bool(blurred)), _p(buf), int(buf.strides[0]), int(buf.shape[0]),
                                           C.byref(w), C.byref(h)), "vslam_orb_level")
        return buf[:h.value, :w.value].copy()

    def orb_status(self, B):
        st = np.zeros(B, np.int32)
        self._chk(self.lib.vslam_orb_status_dev(self.h, int(B), _p(st)), "vslam_orb_status_dev")
        return st

    # ------------------------------------------------------------ VO::feature_matching (visual_odometry.cpp:219-251)
    def feature_matching(self, descriptors_1, descriptors_2, frame_gap=1.0, gate=True):
        q, t = _desc(descriptors_1), _desc(descriptors_2)
        out = np.zeros(max(len(q), 1), DMATCH_DTYPE); n = C.c_int()
        self._chk(self.lib.vslam_feature_matching(self.h, _p(q), len(q), _p(t), len(t), C.c_double(frame_gap), int(gate),
                                                  _p(out), C.byref(n)), "vslam_feature_matching")
        return out[:n.value].copy()

    def feature_matching_dev(self, d_q, q_stride, d_nq, d_t, t_stride, d_nt, d_gap, gate, B, max_rows, d_out, out_cap, d_nout):
        self._chk(self.lib.vslam_feature_matching_dev(self.h, _p(d_q), C.c_size_t(q_stride), _p(d_nq), _p(d_t), C.c_size_t(t_stride),
                                                      _p(d_nt), _p(d_gap), int(gate), int(B), int(max_rows), _p(d_out), int(out_cap),
                                                      _p(d_nout)), "vslam_feature_matching_dev")

    # ------------------------------------------------------------ VO::disparity_map (StereoSGBM + convertTo 1/16)
    def disparity_map(self, left, right, return_i16=False):
        """visual_odometry.cpp:159-174.  Returns the f32 disparity map (invalid = -1); with return_i16 also the CV_16S
        map after median/speckle filtering and the raw SGBM output before them."""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        assert left.ndim == 2 and left.shape == right.shape
        h, w = left.shape
        out = np.zeros((h, w), np.float32)
        i16 = np.zeros((h, w), np.int16) if return_i16 else None
        raw = np.zeros((h, w), np.int16) if return_i16 else None
        self._chk(self.lib.vslam_disparity_map(self.h, _p(left), _p(right), w, h, w, _p(out), _p(i16) if return_i16 else None,
                                               _p(raw) if return_i16 else None), "vslam_disparity_map")
        return (out, i16, raw) if return_i16 else out

    def disparity_map_dev(self, d_left, d_right, img_stride_bytes, pitch, w, h, B, d_disp, d_i16=None, d_raw=None):
        self._chk(self.lib.vslam_disparity_map_dev(self.h, _p(d_left), _p(d_right), C.c_size_t(int(img_stride_bytes)), int(pitch), int(w),
                                                   int(h), int(B), _p(d_disp) if d_disp is not None else None,
                                                   _p(d_i16) if d_i16 is not None else None,
                                                   _p(d_raw) if d_raw is not None else None), "vslam_disparity_map_dev")

    def set_tuning(self, **kw):
        """kernel-choice overrides of this context, e.g. set_tuning(sgbm_fwd_min=1, sgbm_fw_rows=32); -1 = library default"""
        for k, v in kw.items():
            self._chk(self.lib.vslam_set_tuning(self.h, k.encode(), int(v)), "vslam_set_tuning")

    def sgbm_status(self):
        """status word of the most recent disparity_map_dev launch (synchronises): 0, or raises VslamError"""
        st = C.c_int32(0)
        self._chk(self.lib.vslam_sgbm_status_dev(self.h, C.byref(st)), "vslam_sgbm_status_dev")
        return st.value

    # ------------------------------------------------------------ Frame::find_3d / VO::set_ref_3d_position
    def find_3d_disparity(self, kps, disparity, T_c_w):
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); disparity = np.ascontiguousarray(disparity, np.float32)
        T = np.ascontiguousarray(T_c_w, np.float64); n = len(kps)
        xyz = np.zeros((max(n, 1), 3), np.float32); valid = np.zeros(max(n, 1), np.uint8); rel = np.zeros(max(n, 1), np.uint8)
        self._chk(self.lib.vslam_find_3d_disparity(self.h, _p(kps), n, _p(disparity), disparity.shape[1], disparity.shape[0],
                                                   disparity.shape[1], _p(T), _p(xyz), _p(valid), _p(rel), None), "vslam_find_3d_disparity")
        return xyz[:n], valid[:n], rel[:n]

    def find_3d_disparity_dev(self, d_kps, d_n, kp_capacity, B, d_disp, w, h, d_T, d_xyz, d_valid, d_rel):
        self._chk(self.lib.vslam_find_3d_disparity_dev(self.h, _p(d_kps), _p(d_n), int(kp_capacity), int(B), _p(d_disp), int(w), int(h), _p(d_T),
                                                       _p(d_xyz), _p(d_valid), _p(d_rel)), "vslam_find_3d_disparity_dev")

    def triangulate(self, uvL, uvR, T_c_w):
        uvL = np.ascontiguousarray(uvL, np.float32).reshape(-1, 2); uvR = np.ascontiguousarray(uvR, np.float32).reshape(-1, 2)
        T = np.ascontiguousarray(T_c_w, np.float64); n = len(uvL)
        xyz = np.zeros((max(n, 1), 3), np.float32); valid = np.zeros(max(n, 1), np.uint8); rel = np.zeros(max(n, 1), np.uint8)
        self._chk(self.lib.vslam_triangulate(self.h, _p(uvL), _p(uvR), n, _p(T), _p(xyz), _p(valid), _p(rel), None), "vslam_triangulate")
        return xyz[:n], valid[:n], rel[:n]

    def triangulate_dev(self, d_uvL, d_uvR, d_n, capacity, B, d_T, d_xyz, d_valid, d_rel):
        self._chk(self.lib.vslam_triangulate_dev(self.h, _p(d_uvL), _p(d_uvR), _p(d_n), int(capacity), int(B), _p(d_T), _p(d_xyz),
                                                 _p(d_valid), _p(d_rel)), "vslam_triangulate_dev")

    def gather_matched_uv_dev(self, d_kpsQ, d_kpsT, kp_cap, d_matches, d_nmatch, match_cap, B, d_uvQ, d_uvT):
        self._chk(self.lib.vslam_gather_matched_uv_dev(self.h, _p(d_kpsQ), _p(d_kpsT), int(kp_cap), _p(d_matches), _p(d_nmatch),
                                                       int(match_cap), int(B), _p(d_uvQ), _p(d_uvT)), "vslam_gather_matched_uv_dev")

    # ------------------------------------------------------------ VO::motion_estimation (north_star motion-only stage)
    def motion_estimation_ransac(self, xyz_w, uv, T_init=None, max_iters=100, reproj_err=4.0, confidence=0.99, lm_iters=0):
        """VO::motion_estimation's pose stage (cv::solvePnPRansac(..., false, 100, 4.0, 0.99), visual_odometry.cpp:277): EPnP per
        5-point hypothesis, no pose guess is consumed (T_init only pre-fills the output, which stays untouched on failure).
        lm_iters = 0: the best RANSAC model itself (OpenCV 3.2.0, the reference's pinned version); > 0: refined on the inliers (3.4.2+).
        Returns (T, inlier mask, n_inliers, iterations evaluated); n_inliers == 0 means no model was found."""
        xyz = np.ascontiguousarray(xyz_w, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        T = np.array([0, 0, 0, 1, 0, 0, 0], np.float64) if T_init is None else np.ascontiguousarray(T_init, np.float64).copy(); n = len(xyz)
        inl = np.zeros(max(n, 1), np.uint8); ni = C.c_int(); it = C.c_int()
        self._chk(self.lib.vslam_pnp_ransac(self.h, _p(xyz), _p(uv), n, _p(T), int(max_iters), C.c_double(reproj_err), C.c_double(confidence),
                                            int(lm_iters), _p(inl), C.byref(ni), C.byref(it)), "vslam_pnp_ransac")
        return T, inl[:n], ni.value, it.value

    def motion_estimation_ransac_models(self, xyz_w, uv, max_iters=100, reproj_err=4.0, confidence=0.99, lm_iters=0):
        """diagnostic form: (T, mask, n_inliers, iterations, models (max_iters, 12) [R | t], counts (max_iters,))"""
        xyz = np.ascontiguousarray(xyz_w, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        T = np.array([0, 0, 0, 1, 0, 0, 0], np.float64); n = len(xyz)
        inl = np.zeros(max(n, 1), np.uint8); ni = C.c_int(); it = C.c_int()
        models = np.zeros((max_iters, 12)); counts = np.zeros(max_iters, np.int32)
        self._chk(self.lib.vslam_pnp_ransac_models(self.h, _p(xyz), _p(uv), n, _p(T), int(max_iters), C.c_double(reproj_err), C.c_double(confidence),
                                                   int(lm_iters), _p(inl), C.byref(ni), C.byref(it), _p(models), _p(counts)), "vslam_pnp_ransac_models")
        return T, inl[:n], ni.value, it.value, models, counts

    def motion_estimation(self, xyz_w, uv, T_guess, iters=10):
        xyz = np.ascontiguousarray(xyz_w, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        T = np.ascontiguousarray(T_guess, np.float64).copy(); n = len(xyz)
        inl = np.zeros(max(n, 1), np.uint8); ni = C.c_int(); st = LmStats()
        self._chk(self.lib.vslam_pnp_motion_only(self.h, _p(xyz), _p(uv), n, _p(T), int(iters), _p(inl), C.byref(ni), C.byref(st)),
                  "vslam_pnp_motion_only")
        return T, inl[:n], ni.value, st.as_dict()

    def motion_estimation_dev(self, d_xyz, d_uv, d_n, capacity, B, d_T, iters, d_inlier, d_ninl):
        self._chk(self.lib.vslam_pnp_motion_only_dev(self.h, _p(d_xyz), _p(d_uv), _p(d_n), int(capacity), int(B), _p(d_T), int(iters),
                                                     _p(d_inlier), _p(d_ninl)), "vslam_pnp_motion_only_dev")

    def pnp_ransac_dev(self, d_xyz, d_uv, d_n, capacity, B, d_T, max_iters=100, reproj_err=4.0, confidence=0.99, d_inlier=None, d_ninl=None, d_iters=None):
        """cv::solvePnPRansac(..., 100, 4.0, 0.99) of VO::motion_estimation (visual_odometry.cpp:277) for B device-resident problems"""
        self._chk(self.lib.vslam_pnp_ransac_dev(self.h, _p(d_xyz), _p(d_uv), _p(d_n), int(capacity), int(B), _p(d_T), int(max_iters), C.c_double(reproj_err),
                                                C.c_double(confidence), _p(d_inlier) if d_inlier is not None else None,
                                                _p(d_ninl) if d_ninl is not None else None, _p(d_iters) if d_iters is not None else None),
                  "vslam_pnp_ransac_dev")

    def check_motion_estimation(self, num_inliers, T_c_l, frame_gap):
        T = np.ascontiguousarray(T_c_l, np.float64)
        return bool(self.lib.vslam_check_motion(int(num_inliers), _p(T), C.c_double(frame_gap)))

    # ------------------------------------------------------------ optimize_map / optimize_pose_only (optimization.cpp)
    def _window(self, fn, name, T, xyz, kf_idx, lm_idx, uv, flag_lm, iters, update_poses, update_lms, lm_inlier, with_lms, K=None):
        T = np.ascontiguousarray(T, np.float64).reshape(-1, 7).copy()
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3).copy()
        kf_idx = np.ascontiguousarray(kf_idx, np.int32); lm_idx = np.ascontiguousarray(lm_idx, np.int32)
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        fl = None if flag_lm is None else np.ascontiguousarray(flag_lm, np.int32)
        K4 = None if K is None else np.ascontiguousarray(K, np.float64).reshape(4)  # {fx, fy, cx, cy}; None = context intrinsics
        inl = np.ones(len(xyz), np.uint8) if lm_inlier is None else np.ascontiguousarray(lm_inlier, np.uint8).copy()
        chi2 = np.zeros(len(kf_idx)); thr = C.c_double(); st = LmStats()
        if with_lms:
            rc = fn(self.h, len(T), _p(T), len(xyz), _p(xyz), len(kf_idx), _p(kf_idx), _p(lm_idx), _p(uv), _p(K4), _p(fl), int(iters),
                    int(update_poses), int(update_lms), _p(inl), _p(chi2), C.byref(thr), C.byref(st))
        else:
            rc = fn(self.h, len(T), _p(T), len(xyz), _p(xyz), len(kf_idx), _p(kf_idx), _p(lm_idx), _p(uv), _p(K4), _p(fl), int(iters),
                    int(update_poses), _p(inl), _p(chi2), C.byref(thr), C.byref(st))
        self._chk(rc, name)
        return dict(T=T, xyz=xyz, chi2=chi2, threshold=thr.value, lm_inlier=inl, stats=st.as_dict())

    def optimize_map(self, T, xyz, kf_idx, lm_idx, uv, if_update_map=True, if_update_landmark=False, num_ite=10, flag_lm=None,
                     lm_inlier=None, K=None):
        return self._window(self.lib.vslam_local_ba, "vslam_local_ba", T, xyz, kf_idx, lm_idx, uv, flag_lm, num_ite, if_update_map,
                            if_update_landmark, lm_inlier, True, K)

    def optimize_pose_only(self, T, xyz, kf_idx, lm_idx, uv, if_update_map=True, num_ite=10, flag_lm=None, lm_inlier=None, K=None):
        return self._window(self.lib.vslam_pose_only_window, "vslam_pose_only_window", T, xyz, kf_idx, lm_idx, uv, flag_lm, num_ite,
                            if_update_map, False, lm_inlier, False, K)

    def ba_batch_dev(self, batch, schedule=1, mode=0, iters=10, update_poses=1, update_lms=0):
        self._chk(self.lib.vslam_ba_batch_dev(self.h, C.byref(batch), int(schedule), int(mode), int(iters), int(update_poses),
                                              int(update_lms)), "vslam_ba_batch_dev")

    def build_windows_dev(self, tracks, n_kf, lm_capacity, edge_capacity, batch, d_status):
        """optimize_map's graph build on the device (optimization.cpp:127-214 + visual_odometry.cpp:363-424) for a batch of consecutive
        keyframes; fills the device arrays of `batch` (a BaBatch) and its scalar members"""
        self._chk(self.lib.vslam_build_windows_dev(self.h, C.byref(tracks), int(n_kf), int(lm_capacity), int(edge_capacity), C.byref(batch),
                                                   _p(d_status)), "vslam_build_windows_dev")

    def build_pnp_inputs_dev(self, d_f2f, d_nf2f, match_cap, d_lr, d_nlr, lr_cap, d_xyz_lr, d_valid_lr, d_kps_cur, kp_cap, B, d_kp2lr,
                             d_xyz_out, d_uv_out, d_nout, out_cap):
        self._chk(self.lib.vslam_build_pnp_inputs_dev(self.h, _p(d_f2f), _p(d_nf2f), int(match_cap), _p(d_lr), _p(d_nlr), int(lr_cap),
                                                      _p(d_xyz_lr), _p(d_valid_lr), _p(d_kps_cur), int(kp_cap), int(B), _p(d_kp2lr),
                                                      _p(d_xyz_out), _p(d_uv_out), _p(d_nout), int(out_cap)), "vslam_build_pnp_inputs_dev")

    def profile_enable(self, on=True):
        self._chk(self.lib.vslam_profile_enable(self.h, int(on)), "vslam_profile_enable")

    def profile_read(self):
        """{kernel family: (total_ms, launches, calls)} since the last read; synchronises the stream"""
        buf = (KernelTime * 32)(); n = C.c_int()
        self._chk(self.lib.vslam_profile_read(self.h, buf, 32, C.byref(n)), "vslam_profile_read")
        return {buf[i].name.decode(): (buf[i].total_ms, buf[i].launches, buf[i].calls) for i in range(n.value)}

    def profile_intervals(self, cap=4096):
        """[(kernel family, t0_ms, t1_ms)] of the brackets recorded since the last read, on the device's time axis (comparable across the contexts of one device)"""
        buf = (StageInterval * cap)(); n = C.c_int()
        self._chk(self.lib.vslam_profile_intervals(self.h, buf, cap, C.byref(n)), "vslam_profile_intervals")
        return [(buf[i].name.decode(), buf[i].t0_ms, buf[i].t1_ms) for i in range(n.value)]

    def hbm_copy_probe(self, nbytes=1 << 30, reps=5):
        """GB/s (read + write) of a float4 streaming copy on this GPU, timed on the context stream"""
        g = C.c_double()
        self._chk(self.lib.vslam_hbm_copy_probe(self.h, C.c_size_t(nbytes), int(reps), C.byref(g)), "vslam_hbm_copy_probe")
        return g.value

    def hbm_copy_probe_best(self, nbytes=1 << 30, reps=5):
        """every shape of the streaming copy (csrc/geom_kernels.hip): {"gbs": best GB/s, "variant": its description, "all": {name: GB/s}}"""
        res = {}
        for v in range(self.lib.vslam_hbm_copy_probe_variants()):
            g = C.c_double(); name = C.create_string_buffer(64)
            self._chk(self.lib.vslam_hbm_copy_probe_variant(self.h, C.c_size_t(nbytes), int(reps), v, C.byref(g), name), "vslam_hbm_copy_probe_variant")
            res[name.value.decode()] = round(g.value, 1)
        best = max(res, key=res.get)
        return {"gbs": res[best], "variant": best, "bytes": int(nbytes), "all": res}

    def ba_status(self, n_windows):
        st = np.zeros(n_windows, np.int32)
        self._chk(self.lib.vslam_ba_status_dev(self.h, int(n_windows), _p(st)), "vslam_ba_status_dev")
        return st

    def edge_jacobians(self, xyz_w, uv, T_c_w, K=None):
        """residual / Jacobians of EdgeProjection and PoseOnlyEdgeProjection (optimization.cpp:41-101) as the LM kernels' device functions evaluate
        them: {err (n, 2), J_pose (n, 2, 6), J_point (n, 2, 3), chi2 (n), huber_w (n)} for n world points seen through one pose"""
        xyz = np.ascontiguousarray(xyz_w, np.float32); z = np.ascontiguousarray(uv, np.float32); T = np.ascontiguousarray(T_c_w, np.float64)
        n = len(xyz)
        out = dict(err=np.zeros((n, 2)), J_pose=np.zeros((n, 2, 6)), J_point=np.zeros((n, 2, 3)), chi2=np.zeros(n), huber_w=np.zeros(n))
        K4 = None if K is None else np.ascontiguousarray(K, np.float64)
        self._chk(self.lib.vslam_edge_jacobians(self.h, n, _p(xyz), _p(z), _p(T), _p(K4) if K4 is not None else None, _p(out["err"]), _p(out["J_pose"]),
                                                _p(out["J_point"]), _p(out["chi2"]), _p(out["huber_w"])), "vslam_edge_jacobians")
        return out

    def ba_deferred(self, n_windows):
        """per window of the last BA launch: 0 = ba_resident_kernel ran it, 1 = lm_window_kernel did"""
        st = np.zeros(n_windows, np.int32)
        self._chk(self.lib.vslam_ba_deferred_dev(self.h, int(n_windows), _p(st)), "vslam_ba_deferred_dev")
        return st

    def ba_schedule_passes(self, n_windows):
        """optimize_map passes the last schedule executed per window (3, or 1 / 2 when a pass flagged nothing new and was continued instead of repeated)"""
        st = np.zeros(n_windows, np.int32)
        self._chk(self.lib.vslam_ba_schedule_passes_dev(self.h, int(n_windows), _p(st)), "vslam_ba_schedule_passes_dev")
        return st
