"""ctypes binding of the CPU oracle (oracle/libvo_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product
package (stereo-visual-slam_amd/) never does.  PARITY UNPINNED (see oracle/vo_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libvo_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                           ("octave", "<i4"), ("class_id", "<i4")])
DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])
assert KEYPOINT_DTYPE.itemsize == 28 and DMATCH_DTYPE.itemsize == 16


class LmStats(C.Structure):
    _fields_ = [("iterations", C.c_int), ("total_trials", C.c_int), ("chi2_init", C.c_double),
                ("chi2_final", C.c_double), ("lambda_final", C.c_double), ("chi2_iter", C.c_double * 32),
                ("lambda_iter", C.c_double * 32), ("trials_iter", C.c_int * 32)]


class OrbLayout(C.Structure):
    _fields_ = [("w", C.c_int * 8), ("h", C.c_int * 8), ("scale", C.c_float * 8), ("nfeat", C.c_int * 8)]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("orb.c", "match.c", "geom.c", "lm.c", "sgbm.c", "ransac.c", "epnp.c", "windows.c", "cpu_shim.c", "vo_oracle.h", "orb_pattern.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.vo_se3_angle_y.restype = C.c_double
        _lib.vo_chi2_classify.restype = C.c_double
        _lib.vo_harris_response.restype = C.c_float
        _lib.vo_fast_atan2.restype = C.c_float
        _lib.vo_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.vo_ic_angle.restype = C.c_float
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


# ---------------------------------------------------------------- ORB
def orb_layout(w, h, nfeatures):
    L = OrbLayout()
    lib().vo_orb_layout_init(int(w), int(h), int(nfeatures), C.byref(L))
    return dict(w=list(L.w), h=list(L.h), scale=list(L.scale), nfeat=list(L.nfeat))


def resize_linear(img, dw, dh):
    img = _u8img(img)
    out = np.empty((dh, dw), np.uint8)
    lib().vo_resize_linear_u8(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), dw, dh, dw)
    return out


def gaussian_blur7(img):
    img = _u8img(img)
    out = np.empty_like(img)
    lib().vo_gaussian_blur7_u8(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), img.shape[1])
    return out


def gaussian_kernel7_fixed():
    k = (C.c_int * 7)()
    lib().vo_gaussian_kernel7_fixed(k)
    return list(k)


def build_pyramid(img, nlevels=8, nfeatures=3000):
    img = _u8img(img)
    h, w = img.shape
    L = OrbLayout()
    lib().vo_orb_layout_init(w, h, nfeatures, C.byref(L))
    lv = [np.empty((L.h[l], L.w[l]), np.uint8) for l in range(nlevels)]
    ptrs = (C.c_void_p * nlevels)(*[a.ctypes.data for a in lv])
    lib().vo_orb_build_pyramid(_p(img), img.strides[0], C.byref(L), nlevels, ptrs)
    return lv


def fast_corner_score(img, x, y, threshold=20):
    img = _u8img(img)
    return lib().vo_fast_corner_score(C.c_void_p(int(img.ctypes.data) + int(y) * int(img.strides[0]) + int(x)), int(img.strides[0]), int(threshold))


def fast9_16(img, threshold=20, nonmax=True, cap=None):
    img = _u8img(img)
    cap = cap or img.size // 2 + 16
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().vo_fast9_16(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), _p(out), cap)
    assert n >= 0
    return out[:n].copy()


def harris_response(img, x, y):
    img = _u8img(img)
    return float(lib().vo_harris_response(_p(img), img.strides[0], int(x), int(y)))


def fast_atan2(y, x):
    return float(lib().vo_fast_atan2(float(y), float(x)))


def ic_angle(img, x, y):
    img = _u8img(img)
    return float(lib().vo_ic_angle(_p(img), img.strides[0], int(x), int(y)))


def retain_best(kps, npoints):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
    n = lib().vo_retain_best(_p(kps), len(kps), int(npoints))
    return kps[:n].copy()


def orb_detect(img, nfeatures=3000, cap=8192):
    img = _u8img(img)
    out = np.zeros(cap, KEYPOINT_DTYPE)
    n = lib().vo_orb_detect(_p(img), img.shape[1], img.shape[0], img.strides[0], nfeatures, _p(out), cap)
    assert n >= 0, n
    return out[:n].copy()


def anms(kps, num=500):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
    n = lib().vo_anms(_p(kps), len(kps), int(num))
    return kps[:n].copy()


def orb_compute(img, kps):
    img = _u8img(img)
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE).copy()
    desc = np.zeros((max(len(kps), 1), 32), np.uint8)
    n = lib().vo_orb_compute(_p(img), img.shape[1], img.shape[0], img.strides[0], _p(kps), len(kps), _p(desc))
    assert n >= 0
    return kps[:n].copy(), desc[:n].copy()


def feature_detection(img, nfeatures=3000, anms_num=500, cap=8192):
    img = _u8img(img)
    kps = np.zeros(cap, KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = lib().vo_feature_detection(_p(img), img.shape[1], img.shape[0], img.strides[0], nfeatures, anms_num,
                                   _p(kps), cap, _p(desc))
    assert n >= 0, n
    return kps[:n].copy(), desc[:n].copy()


# ---------------------------------------------------------------- matcher
def _desc(d):
    d = np.ascontiguousarray(d, np.uint8)
    if d.size == 0:
        d = d.reshape(0, 32)
    assert d.ndim == 2 and d.shape[1] == 32
    return d


def bf_match_xcheck(q, t):
    q, t = _desc(q), _desc(t)
    out = np.zeros(max(len(q), 1), DMATCH_DTYPE)
    n = lib().vo_bf_match_hamming_xcheck(_p(q), len(q), _p(t), len(t), _p(out))
    return out[:n].copy()


def feature_matching(q, t, frame_gap=1.0):
    q, t = _desc(q), _desc(t)
    out = np.zeros(max(len(q), 1), DMATCH_DTYPE)
    n = lib().vo_feature_matching(_p(q), len(q), _p(t), len(t), C.c_double(frame_gap), _p(out))
    return out[:n].copy()


# ---------------------------------------------------------------- SGBM
def sgbm_compute(left, right, num_disp=96, block=9, P1=648, P2=2592, disp12_max_diff=1, pre_filter_cap=63, uniqueness=10,
                 speckle_window=100, speckle_range=32, return_raw=False):
    left, right = _u8img(left), _u8img(right)
    assert left.shape == right.shape
    h, w = left.shape
    disp = np.zeros((h, w), np.int16); raw = np.zeros((h, w), np.int16)
    rc = lib().vo_sgbm_compute(_p(left), _p(right), w, h, left.strides[0], num_disp, block, P1, P2, disp12_max_diff, pre_filter_cap,
                               uniqueness, speckle_window, speckle_range, _p(disp), _p(raw))
    assert rc == 0, rc
    return (disp, raw) if return_raw else disp


def disparity_map(left, right):
    left, right = _u8img(left), _u8img(right)
    h, w = left.shape
    out = np.zeros((h, w), np.float32)
    rc = lib().vo_disparity_map(_p(left), _p(right), w, h, left.strides[0], _p(out))
    assert rc == 0
    return out


# ---------------------------------------------------------------- geometry
def _d(a, n=None):
    a = np.ascontiguousarray(a, np.float64)
    if n is not None:
        assert a.size == n
    return a


def se3_exp(xi):
    T = np.zeros(7)
    lib().vo_se3_exp(_p(_d(xi, 6)), _p(T))
    return T


def se3_log(T):
    xi = np.zeros(6)
    lib().vo_se3_log(_p(_d(T, 7)), _p(xi))
    return xi


def se3_mul(A, B):
    out = np.zeros(7)
    lib().vo_se3_mul(_p(_d(A, 7)), _p(_d(B, 7)), _p(out))
    return out


def se3_inv(A):
    out = np.zeros(7)
    lib().vo_se3_inv(_p(_d(A, 7)), _p(out))
    return out


def se3_act(T, p):
    out = np.zeros(3)
    lib().vo_se3_act(_p(_d(T, 7)), _p(_d(p, 3)), _p(out))
    return out


def se3_rotmat(T):
    R = np.zeros(9)
    lib().vo_se3_rotmat(_p(_d(T, 7)), _p(R))
    return R.reshape(3, 3)


def se3_angle_y(T):
    return float(lib().vo_se3_angle_y(_p(_d(T, 7))))


CAM_KITTI = np.array([718.856, 718.856, 607.1928, 185.2157, 0.573])  # types_def.hpp:53-54
K_KITTI = CAM_KITTI[:4].copy()


def find_3d_disparity(kps, disparity, T_c_w, cam=CAM_KITTI):
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    disparity = np.ascontiguousarray(disparity, np.float32)
    n = len(kps)
    xyz = np.zeros((n, 3), np.float32); valid = np.zeros(n, np.uint8); rel = np.zeros(n, np.uint8)
    lib().vo_find_3d_disparity(_p(kps), n, _p(disparity), disparity.shape[1], disparity.shape[0], disparity.shape[1],
                               _p(_d(T_c_w, 7)), _p(_d(cam, 5)), _p(xyz), _p(valid), _p(rel))
    return xyz, valid, rel


def triangulate_dlt(uvL, uvR, T_c_w, cam=CAM_KITTI, row_tol=2.0):
    uvL = np.ascontiguousarray(uvL, np.float32).reshape(-1, 2); uvR = np.ascontiguousarray(uvR, np.float32).reshape(-1, 2)
    n = len(uvL)
    xyz = np.zeros((n, 3), np.float32); valid = np.zeros(n, np.uint8); rel = np.zeros(n, np.uint8)
    lib().vo_triangulate_dlt(_p(uvL), _p(uvR), n, _p(_d(T_c_w, 7)), _p(_d(cam, 5)), C.c_double(row_tol), _p(xyz), _p(valid), _p(rel))
    return xyz, valid, rel


def check_motion(num_inliers, T_c_l, frame_gap):
    return bool(lib().vo_check_motion(int(num_inliers), _p(_d(T_c_l, 7)), C.c_double(frame_gap)))


# ---------------------------------------------------------------- LM
def pose_only_residual(T, pw, z, K=K_KITTI):
    e = np.zeros(2); J = np.zeros(12)
    lib().vo_pose_only_residual(_p(_d(T, 7)), _p(_d(pw, 3)), _p(_d(z, 2)), _p(_d(K, 4)), _p(e), _p(J))
    return e, J.reshape(2, 6)


def projection_residual(T, pw, z, K=K_KITTI):
    e = np.zeros(2); Jp = np.zeros(12); Jl = np.zeros(6)
    lib().vo_projection_residual(_p(_d(T, 7)), _p(_d(pw, 3)), _p(_d(z, 2)), _p(_d(K, 4)), _p(e), _p(Jp), _p(Jl))
    return e, Jp.reshape(2, 6), Jl.reshape(2, 3)


def _stats(st):
    n = min(st.iterations, 32)
    return dict(iterations=st.iterations, total_trials=st.total_trials, chi2_init=st.chi2_init,
                chi2_final=st.chi2_final, lambda_final=st.lambda_final, chi2_iter=list(st.chi2_iter)[:n],
                lambda_iter=list(st.lambda_iter)[:n], trials_iter=list(st.trials_iter)[:n])


def _edges(kf_idx, lm_idx, uv):
    kf_idx = np.ascontiguousarray(kf_idx, np.int32); lm_idx = np.ascontiguousarray(lm_idx, np.int32)
    uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    assert len(kf_idx) == len(lm_idx) == len(uv)
    return kf_idx, lm_idx, uv


def local_ba(T, xyz, kf_idx, lm_idx, uv, K=K_KITTI, iters=10, huber_delta=5.991, update_poses=True, update_lms=False):
    T = np.ascontiguousarray(T, np.float64).reshape(-1, 7).copy()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3).copy()
    kf_idx, lm_idx, uv = _edges(kf_idx, lm_idx, uv)
    chi2 = np.zeros(max(len(kf_idx), 1)); st = LmStats()
    rc = lib().vo_local_ba(len(T), _p(T), len(xyz), _p(xyz), len(kf_idx), _p(kf_idx), _p(lm_idx), _p(uv), _p(_d(K, 4)),
                           int(iters), C.c_double(huber_delta), int(update_poses), int(update_lms), _p(chi2), C.byref(st))
    assert rc == 0, rc
    return T, xyz, chi2[:len(kf_idx)], _stats(st)


def pose_only_window(T, xyz, kf_idx, lm_idx, uv, K=K_KITTI, iters=10, huber_delta=5.991, update_poses=True):
    T = np.ascontiguousarray(T, np.float64).reshape(-1, 7).copy()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    kf_idx, lm_idx, uv = _edges(kf_idx, lm_idx, uv)
    chi2 = np.zeros(max(len(kf_idx), 1)); st = LmStats()
    rc = lib().vo_pose_only_window(len(T), _p(T), len(xyz), _p(xyz), len(kf_idx), _p(kf_idx), _p(lm_idx), _p(uv),
                                   _p(_d(K, 4)), int(iters), C.c_double(huber_delta), int(update_poses), _p(chi2), C.byref(st))
    assert rc == 0, rc
    return T, chi2[:len(kf_idx)], _stats(st)


def chi2_classify(chi2, flag_lm, lm_inlier):
    chi2 = _d(chi2); flag_lm = np.ascontiguousarray(flag_lm, np.int32)
    lm_inlier = np.ascontiguousarray(lm_inlier, np.uint8).copy()
    ni = C.c_int(); no = C.c_int()
    th = lib().vo_chi2_classify(_p(chi2), len(chi2), _p(flag_lm), _p(lm_inlier), len(lm_inlier), C.byref(ni), C.byref(no))
    return float(th), lm_inlier, ni.value, no.value


def ransac_subsets(count, max_iters=100, model_points=5):
    out = np.zeros((max_iters, model_points), np.int32)
    lib().vo_ransac_subsets(int(count), int(model_points), int(max_iters), _p(out))
    return out


def ransac_update_num_iters(p, ep, model_points, max_iters):
    f = lib().vo_ransac_update_num_iters
    f.argtypes = [C.c_double, C.c_double, C.c_int, C.c_int]
    return f(p, ep, model_points, max_iters)


def epnp(xyz, uv, K=K_KITTI):
    """EPnP pose (R 3x3, t 3, mean reprojection error) of n >= 4 correspondences; err < 0: degenerate"""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    R = np.zeros(9); t = np.zeros(3)
    f = lib().vo_epnp; f.restype = C.c_double
    err = f(_p(xyz), _p(uv), len(xyz), _p(_d(K, 4)), _p(R), _p(t))
    return R.reshape(3, 3), t, err


def jacobi_eig12(A):
    A = np.ascontiguousarray(A, np.float64).reshape(12, 12).copy(); V = np.zeros((12, 12))
    lib().vo_jacobi_eig12(_p(A), _p(V))
    return np.diag(A).copy(), V


def epnp_subset(xyz, uv, subset, K=K_KITTI):
    """EPnP model [R | t] (12,) of the given 5 point indices, or None if degenerate"""
    R, t, err = epnp(np.asarray(xyz, np.float32)[list(subset)], np.asarray(uv, np.float32)[list(subset)], K)
    return None if err < 0 else np.concatenate([R.reshape(9), t])


def pnp_ransac_hypothesis(xyz, uv, it, K=K_KITTI, reproj_err=4.0):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    T = np.zeros(7); sub = np.zeros(5, np.int32)
    n = lib().vo_pnp_ransac_hypothesis(_p(xyz), _p(uv), len(xyz), _p(_d(K, 4)), int(it), C.c_double(reproj_err), _p(T), _p(sub))
    return T, n, sub


def pnp_ransac(xyz, uv, T0=None, K=K_KITTI, max_iters=100, reproj_err=4.0, confidence=0.99, lm_iters=0):
    """T0 is ignored (kept for call compatibility): like the reference's call, no pose guess is consumed"""
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    T = np.array([0, 0, 0, 1, 0, 0, 0], np.float64) if T0 is None else _d(T0, 7).copy(); inl = np.zeros(max(len(xyz), 1), np.uint8); it = C.c_int()
    n = lib().vo_pnp_ransac(_p(xyz), _p(uv), len(xyz), _p(_d(K, 4)), _p(T), int(max_iters), C.c_double(reproj_err), C.c_double(confidence),
                            int(lm_iters), _p(inl), C.byref(it))
    return T, inl[:len(xyz)], n, it.value


def build_windows(kps, lr, nlr, xyz, valid, reliable, f2f, nf2f, pose_inlier, T_rel, n_kf=10, lm_capacity=None, edge_capacity=None, K=None, reproj_thr=4.0,
                  track_rule=1):
    """windows.c: the BA windows of a batch of consecutive keyframes from the front end's per-frame results.
    kps (F, kp_cap) KEYPOINT_DTYPE; lr (F, lr_cap) DMATCH_DTYPE; xyz (F, lr_cap, 3); valid / reliable (F, lr_cap);
    f2f (F-1, match_cap) DMATCH_DTYPE; pose_inlier (F-1, pnp_cap); T_rel (F-1, 7).  Returns a dict of the vslam_ba_batch arrays."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE); lr = np.ascontiguousarray(lr, DMATCH_DTYPE); f2f = np.ascontiguousarray(f2f, DMATCH_DTYPE)
    F, kp_cap = kps.shape; lr_cap = lr.shape[1]
    match_cap = f2f.shape[1] if f2f.ndim == 2 and f2f.shape[0] else 1
    pose_inlier = np.ascontiguousarray(pose_inlier, np.uint8)
    pnp_cap = pose_inlier.shape[1] if pose_inlier.ndim == 2 and pose_inlier.shape[0] else 1
    nlr = np.ascontiguousarray(nlr, np.int32); nf2f = np.ascontiguousarray(nf2f, np.int32)
    xyz = np.ascontiguousarray(xyz, np.float32); valid = np.ascontiguousarray(valid, np.uint8); reliable = np.ascontiguousarray(reliable, np.uint8)
    T_rel = np.ascontiguousarray(T_rel, np.float64)
    lm_capacity = lm_capacity or F * kp_cap; edge_capacity = edge_capacity or F * kp_cap * 2
    out = dict(lm_off=np.zeros(F + 1, np.int32), edge_off=np.zeros(F + 1, np.int32), n_kf=np.zeros(F, np.int32),
               T=np.zeros((F, n_kf, 7), np.float64), xyz=np.zeros((lm_capacity, 3), np.float32), reliable=np.zeros(lm_capacity, np.uint8),
               lm_inlier=np.zeros(lm_capacity, np.uint8), kf_idx=np.zeros(edge_capacity, np.int32), lm_idx=np.zeros(edge_capacity, np.int32),
               uv=np.zeros((edge_capacity, 2), np.float32))
    rc = lib().vo_build_windows(F, kp_cap, lr_cap, match_cap, pnp_cap, _p(kps), _p(lr), _p(nlr), _p(xyz), _p(valid), _p(reliable), _p(f2f), _p(nf2f),
                                _p(pose_inlier), _p(T_rel), int(n_kf), int(lm_capacity), int(edge_capacity), _p(out["lm_off"]), _p(out["edge_off"]),
                                _p(out["n_kf"]), _p(out["T"]), _p(out["xyz"]), _p(out["reliable"]), _p(out["lm_inlier"]), _p(out["kf_idx"]),
                                _p(out["lm_idx"]), _p(out["uv"]), _p(_d(K_KITTI if K is None else K, 4)), C.c_double(reproj_thr), int(track_rule))
    if rc < 0:
        raise RuntimeError("vo_build_windows: inconsistent input (%d)" % rc)
    out["status"] = rc
    return out


def pnp_motion_only(xyz, uv, T0, K=K_KITTI, iters=10, huber_delta=5.991, reproj_thr=4.0):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3); uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
    T = _d(T0, 7).copy(); inl = np.zeros(len(xyz), np.uint8); st = LmStats()
    n = lib().vo_pnp_motion_only(_p(xyz), _p(uv), len(xyz), _p(_d(K, 4)), _p(T), int(iters), C.c_double(huber_delta),
                                 C.c_double(reproj_thr), _p(inl), C.byref(st))
    return T, inl, n, _stats(st)
