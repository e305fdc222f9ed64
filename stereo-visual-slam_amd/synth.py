"""Seeded synthetic inputs shaped like the reference's workload (there is no KITTI data and no network).

Shapes follow SURVEY.md section 8(d): KITTI-00 grayscale stereo pairs 1241x376 with the intrinsics the
reference hard-codes (types_def.hpp:53-54, run_vslam.cpp:34-35), random 256-bit descriptor sets with planted
matches and engineered ties, 3D-2D motion-only problems, and 10-keyframe local-BA windows.

Pure numpy; used by tests/ and bench.py.  Nothing here is on the product's compute path.
"""
import numpy as np

W_KITTI, H_KITTI = 1241, 376
FX, FY, CX, CY, BASELINE = 718.856, 718.856, 607.1928, 185.2157, 0.573
CAM = np.array([FX, FY, CX, CY, BASELINE])
K4 = CAM[:4].copy()


# --------------------------------------------------------------------------- SE3 helpers (numpy, f64)
def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def so3_exp(w):
    th = np.linalg.norm(w)
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Wx
    return np.eye(3) + np.sin(th) / th * Wx + (1 - np.cos(th)) / th ** 2 * Wx @ Wx


def quat_from_R(R):
    """unit quaternion (x,y,z,w) from a rotation matrix"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q /= np.linalg.norm(q)
    if q[3] < 0:
        q = -q
    return q


def se3_from_Rt(R, t):
    return np.concatenate([quat_from_R(R), np.asarray(t, float)])


def R_from_quat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def project(T_c_w, pw, K=K4):
    R = R_from_quat(T_c_w[:4]); t = T_c_w[4:]
    pc = pw @ R.T + t
    return np.stack([K[0] * pc[:, 0] / pc[:, 2] + K[2], K[1] * pc[:, 1] / pc[:, 2] + K[3]], 1), pc[:, 2]


# --------------------------------------------------------------------------- images
def _hash_u32(ix, iy, pid, seed):
    h = (ix.astype(np.uint32) * np.uint32(73856093)) ^ (iy.astype(np.uint32) * np.uint32(19349663)) \
        ^ np.uint32((pid * 83492791 + seed * 2654435761) & 0xFFFFFFFF)
    h ^= h >> np.uint32(16); h *= np.uint32(0x7FEB352D); h ^= h >> np.uint32(15)
    h *= np.uint32(0x846CA68B); h ^= h >> np.uint32(16)
    return h


def _texture(a, b, pid, seed, px_per_m=None):
    """blocky multi-octave texture in plane coordinates (metres) -> grey level float.  px_per_m (optional, per pixel): image
    pixels per metre of plane along its most compressed direction; an octave whose cells project to less than ~4 px is faded
    out (analytic anti-aliasing: point-sampling sub-pixel cells gives noise that differs between the two views of a stereo pair
    and between consecutive frames, which no camera produces)."""
    val = np.full(a.shape, 110.0)
    for cell, amp, k in ((0.9, 70.0, 1), (0.22, 50.0, 2), (0.06, 30.0, 3)):
        ix = np.floor(a / cell).astype(np.int64); iy = np.floor(b / cell).astype(np.int64)
        h = _hash_u32(ix, iy, pid * 4 + k, seed)
        o = amp * (((h >> np.uint32(8)) & np.uint32(0xFF)).astype(np.float64) / 255.0 - 0.5)
        if px_per_m is not None:
            o = o * np.clip((cell * px_per_m - 2.0) / 2.0, 0.0, 1.0)
        val += o
    return val


class Scene:
    """ground plane + fronto-parallel textured walls + far backdrop, in the world (= first camera) frame.
    Camera convention: x right, y down, z forward (KITTI)."""

    def __init__(self, seed=0, n_walls=10, z_far=80.0, antialias=False, path=None):
        """path (optional): (n, 2) camera centres (x, z) of the sequence that will be rendered; walls then keep at least 2 m of
        lateral clearance from it (a camera driving through a wall loses every feature at once)."""
        rng = np.random.default_rng(seed)
        self.seed = int(seed)
        self.antialias = bool(antialias)
        self.walls = []
        for i in range(n_walls):
            z = rng.uniform(6.0, z_far)
            side = rng.choice([-1.0, 1.0])
            x0 = side * rng.uniform(2.0, 12.0) + rng.uniform(-2, 2)
            wdt = rng.uniform(3.0, 10.0); hgt = rng.uniform(2.0, 6.0)
            if path is not None:
                xc = float(np.interp(z, path[:, 1], path[:, 0]))
                x0 = xc + side * (rng.uniform(2.0, 9.0) + wdt / 2)
            self.walls.append((z, x0 - wdt / 2, x0 + wdt / 2, 1.65 - hgt, 1.65))
        self.ground_y = 1.65
        self.back_z = z_far + 40.0

    def render(self, T_c_w, w=W_KITTI, h=H_KITTI, cam=CAM, x_offset=0.0):
        """render the view of a camera with pose T_c_w (7: quat xyzw + t); x_offset shifts the optical centre
        along the camera x axis (right camera: x_offset = baseline)."""
        fx, fy, cx, cy = cam[:4]
        R = R_from_quat(T_c_w[:4]); t = T_c_w[4:]
        Rwc = R.T
        o = -Rwc @ t + Rwc @ np.array([x_offset, 0, 0])
        u, v = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
        d = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], -1) @ Rwc.T
        best_t = np.full(u.shape, np.inf)
        img = np.full(u.shape, 90.0)
        pid = np.zeros(u.shape, np.int32)          # winning plane per pixel (0 = none)
        A = np.zeros(u.shape); Bc = np.zeros(u.shape)  # plane coordinates of the hit
        with np.errstate(divide="ignore", invalid="ignore"):
            # backdrop
            tt = (self.back_z - o[2]) / d[..., 2]
            ok = tt > 0
            best_t = np.where(ok, tt, best_t); pid = np.where(ok, 1, pid)
            A = np.where(ok, (o[0] + tt * d[..., 0]) * 0.25, A); Bc = np.where(ok, (o[1] + tt * d[..., 1]) * 0.25, Bc)
            # ground
            tt = (self.ground_y - o[1]) / d[..., 1]
            ok = (tt > 0) & (tt < best_t)
            best_t = np.where(ok, tt, best_t); pid = np.where(ok, 2, pid)
            A = np.where(ok, o[0] + tt * d[..., 0], A); Bc = np.where(ok, o[2] + tt * d[..., 2], Bc)
            for i, (z, x0, x1, y0, y1) in enumerate(self.walls):
                # only the pixels inside the projected bounding box of the wall can hit it (all four corners in front of the
                # camera: the image of the quad is the convex hull of its corners); walls behind the camera are skipped
                cw = np.array([[x0, y0, z], [x1, y0, z], [x0, y1, z], [x1, y1, z]]) - o
                cc = cw @ Rwc  # camera frame (rows: corner . columns of Rwc = R rows)
                if cc[:, 2].max() <= 0.05:
                    continue
                sl = (slice(0, h), slice(0, w))
                if cc[:, 2].min() > 0.05:
                    uu = fx * cc[:, 0] / cc[:, 2] + cx; vv = fy * cc[:, 1] / cc[:, 2] + cy
                    u0, u1 = int(np.floor(uu.min())) - 1, int(np.ceil(uu.max())) + 2
                    v0, v1 = int(np.floor(vv.min())) - 1, int(np.ceil(vv.max())) + 2
                    if u1 <= 0 or v1 <= 0 or u0 >= w or v0 >= h:
                        continue
                    sl = (slice(max(v0, 0), min(v1, h)), slice(max(u0, 0), min(u1, w)))
                ds = d[sl]
                tt = (z - o[2]) / ds[..., 2]
                X = o[0] + tt * ds[..., 0]; Y = o[1] + tt * ds[..., 1]
                ok = (tt > 0) & (tt < best_t[sl]) & (X >= x0) & (X <= x1) & (Y >= y0) & (Y <= y1)
                best_t[sl] = np.where(ok, tt, best_t[sl]); pid[sl] = np.where(ok, 3 + i, pid[sl])
                A[sl] = np.where(ok, X, A[sl]); Bc[sl] = np.where(ok, Y, Bc[sl])
        # texture lookup once per pixel, for the winning plane only
        for p in np.unique(pid):
            if p == 0:
                continue
            m = pid == p
            ppm = None
            if self.antialias:  # pixels per metre of plane: fronto-parallel planes f / Z; the ground is compressed along z: f * height / Z^2
                Zm = np.maximum(best_t[m], 1e-3)
                ppm = fx / Zm * (0.25 if p == 1 else 1.0) if p != 2 else fy * abs(self.ground_y - o[1]) / (Zm * Zm)
            img[m] = _texture(A[m], Bc[m], int(p), self.seed, ppm)
        depth = best_t * 1.0  # z-depth along camera axis since d_cam.z == 1
        return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def trajectory(n_frames, seed=0):
    """forward motion 0.8-1.2 m/frame, yaw U(-0.03,0.03) rad/frame. returns T_c_w list."""
    rng = np.random.default_rng(seed + 1000)
    Rwc = np.eye(3); p = np.zeros(3)
    out = []
    for i in range(n_frames):
        out.append(se3_from_Rt(Rwc.T, -Rwc.T @ p))
        yaw = rng.uniform(-0.03, 0.03); step = rng.uniform(0.8, 1.2)
        Rwc = Rwc @ rot_y(yaw)
        p = p + Rwc @ np.array([0, 0, step])
    return out


def _sequence_scene(n_frames, seed):
    traj = trajectory(n_frames, seed)
    centres = np.array([-R_from_quat(T[:4]).T @ T[4:] for T in traj])
    far = centres[-1] + 200.0 * (R_from_quat(traj[-1][:4]).T @ np.array([0, 0, 1.0]))  # the last heading extended beyond the walls
    path = np.vstack([centres[:, [0, 2]], far[[0, 2]]])
    return traj, Scene(seed, n_walls=10 + n_frames // 2, z_far=80.0 + 1.2 * n_frames, antialias=True, path=path)


def _render_frames(job):
    """pool worker of stereo_sequence: frames `idx` of the (n_frames, seed) sequence; every frame depends only on the scene and its pose"""
    n_frames, seed, w, h, idx = job
    traj, sc = _sequence_scene(n_frames, seed)
    out = []
    for i in idx:
        L, depth = sc.render(traj[i], w, h)
        Rr, _ = sc.render(traj[i], w, h, x_offset=BASELINE)
        out.append((i, L, Rr, depth))
    return out


def stereo_sequence(n_frames, seed=0, w=W_KITTI, h=H_KITTI, workers=0):
    """list of (left u8, right u8, T_c_w, depth_left).  Walls are spread over the whole path (about 1 m per frame) so that
    the last frames still see structure within the 10-40 m reliable-depth range (visual_odometry.cpp:194,201).
    workers > 1: the frames are rendered by a pool of spawned processes (numpy only); the result does not depend on it."""
    traj, sc = _sequence_scene(n_frames, seed)
    workers = int(min(workers, n_frames))
    if workers > 1:
        import multiprocessing as mp
        jobs = [(n_frames, seed, w, h, list(range(k, n_frames, 4 * workers))) for k in range(min(4 * workers, n_frames))]
        frames = [None] * n_frames
        with mp.get_context("spawn").Pool(workers) as pool:
            for part in pool.imap_unordered(_render_frames, jobs):
                for i, L, Rr, depth in part:
                    frames[i] = (L, Rr, traj[i], depth)
        return frames
    out = []
    for T in traj:
        L, depth = sc.render(T, w, h)
        Rr, _ = sc.render(T, w, h, x_offset=BASELINE)
        out.append((L, Rr, T, depth))
    return out


def noise_image(seed, w=W_KITTI, h=H_KITTI):
    """cheap corner-rich image (blocky noise at three scales + smooth ramp), for ORB parity tests"""
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w))
    for cell, amp in ((37, 80), (11, 60), (4, 40)):
        g = rng.uniform(-0.5, 0.5, (h // cell + 2, w // cell + 2))
        img += amp * np.kron(g, np.ones((cell, cell)))[:h, :w]
    yy, xx = np.mgrid[0:h, 0:w]
    img += 120 + 20 * np.sin(xx / 90.0) + 10 * np.cos(yy / 40.0) + rng.normal(0, 2.0, (h, w))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


# --------------------------------------------------------------------------- descriptors
def random_descriptors(nq, nt, seed=1, planted=0.7, flip_p=0.06, tie_frac=0.05):
    """q, t ~ U{0,1}^256; `planted` of the query rows get a train partner with Binomial(256, flip_p) bit flips;
    `tie_frac` of the train rows are exact duplicates of other train rows (engineered distance ties)."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    n_pl = int(min(nq, nt) * planted)
    qi = rng.permutation(nq)[:n_pl]; ti = rng.permutation(nt)[:n_pl]
    flips = rng.random((n_pl, 256)) < flip_p
    t[ti] = q[qi] ^ np.packbits(flips, axis=1, bitorder="little")
    n_tie = int(nt * tie_frac)
    if n_tie and nt > 1:
        src = rng.integers(0, nt, n_tie); dst = rng.integers(0, nt, n_tie)
        t[dst] = t[src]
    if nq > 2:  # and a few duplicated query rows
        q[rng.integers(0, nq, max(1, nq // 50))] = q[rng.integers(0, nq, max(1, nq // 50))]
    return q, t


# --------------------------------------------------------------------------- motion-only / BA problems
def perturb_pose(T, rng, sigma):
    d = rng.normal(0, sigma, 6)
    R = so3_exp(d[3:]) @ R_from_quat(T[:4])
    t = so3_exp(d[3:]) @ T[4:] + d[:3]
    return se3_from_Rt(R, t)


def pnp_problem(M=500, seed=3, sigma_px=0.5, outlier_frac=0.15, guess_sigma=0.02):
    """M 3D-2D pairs, sigma 0.5 px, 15 % gross outliers U(+-50 px)."""
    rng = np.random.default_rng(seed)
    T_true = se3_from_Rt(rot_y(rng.uniform(-0.05, 0.05)), rng.normal(0, 0.3, 3) + np.array([0, 0, -1.0]))
    Z = rng.uniform(10, 60, M)
    u = rng.uniform(40, W_KITTI - 40, M); v = rng.uniform(40, H_KITTI - 40, M)
    pc = np.stack([(u - CX) / FX * Z, (v - CY) / FY * Z, Z], 1)
    R = R_from_quat(T_true[:4])
    pw = ((pc - T_true[4:]) @ R).astype(np.float32)
    uv, _ = project(T_true, pw.astype(np.float64))
    uv = uv + rng.normal(0, sigma_px, uv.shape)
    is_out = rng.random(M) < outlier_frac
    uv[is_out] += rng.uniform(-50, 50, (int(is_out.sum()), 2))
    T0 = perturb_pose(T_true, rng, guess_sigma)
    return dict(xyz=pw, uv=uv.astype(np.float32), T_true=T_true, T0=T0, outlier=is_out)


def ba_window(n_kf=10, n_lm=3000, seed=2, sigma_px=0.5, outlier_frac=0.05, pose_sigma=0.02, min_obs=2, max_obs=5, ordered=True):
    """10 poses on a forward arc (1 m spacing, 0.02 rad yaw/KF), landmarks in the frustum
    with Z in [10,40] m of some keyframe, each seen by 2-5 consecutive keyframes; edges sorted by landmark."""
    rng = np.random.default_rng(seed)
    Rwc = np.eye(3); p = np.zeros(3)
    T_true = []
    for k in range(n_kf):
        T_true.append(se3_from_Rt(Rwc.T, -Rwc.T @ p))
        Rwc = Rwc @ rot_y(0.02); p = p + Rwc @ np.array([0, 0, 1.0])
    T_true = np.array(T_true)
    xyz = np.zeros((n_lm, 3), np.float32)
    kf_idx, lm_idx, uvs = [], [], []
    # landmark ids follow creation order (curr_landmark_id_++ at the keyframe that first sees the point,
    # visual_odometry.cpp:411-417): draw the observation spans first, then number the landmarks by first keyframe
    spans = []
    for l in range(n_lm):
        nobs = int(rng.integers(min_obs, max_obs + 1))
        spans.append((int(rng.integers(0, n_kf - nobs + 1)), nobs))
    if ordered:
        spans.sort(key=lambda s: s[0])
    for l in range(n_lm):
        k0, nobs = spans[l]
        # place in the frustum of the middle observing keyframe
        km = k0 + nobs // 2
        Z = rng.uniform(10, 40); u = rng.uniform(150, W_KITTI - 150); v = rng.uniform(60, H_KITTI - 60)
        pc = np.array([(u - CX) / FX * Z, (v - CY) / FY * Z, Z])
        R = R_from_quat(T_true[km][:4])
        xyz[l] = ((pc - T_true[km][4:]) @ R).astype(np.float32)
        for k in range(k0, k0 + nobs):
            uv, z = project(T_true[k], xyz[l:l + 1].astype(np.float64))
            if z[0] < 1.0:
                continue
            kf_idx.append(k); lm_idx.append(l); uvs.append(uv[0])
    kf_idx = np.array(kf_idx, np.int32); lm_idx = np.array(lm_idx, np.int32)
    uv = np.array(uvs) + rng.normal(0, sigma_px, (len(uvs), 2))
    is_out = rng.random(len(uv)) < outlier_frac
    uv[is_out] += rng.uniform(-30, 30, (int(is_out.sum()), 2))
    T0 = np.array([perturb_pose(T, rng, pose_sigma) for T in T_true])
    return dict(T_true=T_true, T0=T0, xyz=xyz, kf_idx=kf_idx, lm_idx=lm_idx, uv=uv.astype(np.float32), outlier=is_out)


def ba_window_fast(n_kf=10, n_lm=3000, seed=2, sigma_px=0.5, outlier_frac=0.05, pose_sigma=0.02, min_obs=2, max_obs=5):
    """same construction as ba_window, vectorised over the landmarks so that a bench can afford one
    UNIQUE window per batch item; a different random stream than ba_window (not interchangeable seed for seed)."""
    rng = np.random.default_rng(seed)
    Rwc = np.eye(3); p = np.zeros(3)
    Rs, ts = [], []
    for k in range(n_kf):
        Rs.append(Rwc.T.copy()); ts.append(-Rwc.T @ p)
        Rwc = Rwc @ rot_y(0.02); p = p + Rwc @ np.array([0, 0, 1.0])
    Rs = np.array(Rs); ts = np.array(ts)
    T_true = np.array([se3_from_Rt(Rs[k], ts[k]) for k in range(n_kf)])
    nobs = rng.integers(min_obs, max_obs + 1, n_lm)
    k0 = (rng.random(n_lm) * (n_kf - nobs + 1)).astype(np.int64)
    order = np.argsort(k0, kind="stable")  # landmark ids follow creation order (first observing keyframe)
    nobs, k0 = nobs[order], k0[order]
    km = k0 + nobs // 2
    Z = rng.uniform(10, 40, n_lm); u = rng.uniform(150, W_KITTI - 150, n_lm); v = rng.uniform(60, H_KITTI - 60, n_lm)
    pc = np.stack([(u - CX) / FX * Z, (v - CY) / FY * Z, Z], 1)
    xyz = np.einsum("nij,nj->ni", np.transpose(Rs[km], (0, 2, 1)), pc - ts[km]).astype(np.float32)  # R^T (pc - t)
    lm_idx = np.repeat(np.arange(n_lm), nobs)
    kf_idx = (np.repeat(k0, nobs) + (np.arange(len(lm_idx)) - np.repeat(np.cumsum(nobs) - nobs, nobs))).astype(np.int64)
    pw = xyz[lm_idx].astype(np.float64)
    pcam = np.einsum("nij,nj->ni", Rs[kf_idx], pw) + ts[kf_idx]
    keep = pcam[:, 2] >= 1.0
    kf_idx, lm_idx, pcam = kf_idx[keep], lm_idx[keep], pcam[keep]
    uv = np.stack([FX * pcam[:, 0] / pcam[:, 2] + CX, FY * pcam[:, 1] / pcam[:, 2] + CY], 1)
    uv = uv + rng.normal(0, sigma_px, uv.shape)
    is_out = rng.random(len(uv)) < outlier_frac
    uv[is_out] += rng.uniform(-30, 30, (int(is_out.sum()), 2))
    T0 = np.array([perturb_pose(T, rng, pose_sigma) for T in T_true])
    return dict(T_true=T_true, T0=T0, xyz=xyz, kf_idx=kf_idx.astype(np.int32), lm_idx=lm_idx.astype(np.int32), uv=uv.astype(np.float32), outlier=is_out)


# --------------------------------------------------------------------------- dataset on disk (C++ driver)
def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def write_png(path, img, filters=True):
    """8-bit grayscale PNG (the KITTI odometry file type).  With `filters`, scanlines cycle through the five PNG filter
    types so that a reader has to implement all of them."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    rows = []
    prev = np.zeros(w, np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32)
        ft = (y % 5) if filters else 0
        a = np.concatenate([[0], cur[:-1]]); b = prev; c = np.concatenate([[0], prev[:-1]])
        if ft == 0: pred = np.zeros(w, np.int32)
        elif ft == 1: pred = a
        elif ft == 2: pred = b
        elif ft == 3: pred = (a + b) >> 1
        else:
            p = a + b - c; pa = np.abs(p - a); pb = np.abs(p - b); pc = np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        rows.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(b"".join(rows), 6)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)))
        for o in range(0, len(z), 32768):  # several IDAT chunks, like real encoders
            f.write(chunk(b"IDAT", z[o:o + 32768]))
        f.write(chunk(b"IEND", b""))


def write_pgm_sequence(root, n_frames, seed=0, w=W_KITTI, h=H_KITTI, fmt="pgm"):
    """KITTI-like layout: root/image_0/%06d.<fmt> (left), root/image_1/%06d.<fmt> (right), fmt = "pgm" (binary P5) or "png"
    (8-bit gray, what KITTI ships).  Returns the ground-truth T_c_w list."""
    import os
    os.makedirs(os.path.join(root, "image_0"), exist_ok=True); os.makedirs(os.path.join(root, "image_1"), exist_ok=True)
    seq = stereo_sequence(n_frames, seed, w, h)
    wr = write_png if fmt == "png" else write_pgm
    for i, (L, R, T, _) in enumerate(seq):
        wr(os.path.join(root, "image_0", "%06d.%s" % (i, fmt)), L)
        wr(os.path.join(root, "image_1", "%06d.%s" % (i, fmt)), R)
    return [s[2] for s in seq]
