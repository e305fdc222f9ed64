"""CPU known-answer tests pinning the ORB part of the oracle (oracle/orb.c).

PARITY UNPINNED against real OpenCV (absent from the image and from the reference repo: no golden vectors exist).  These
tests pin the restatement against (a) constants derivable by hand from the published algorithm, (b) independent numpy
restatements written from the definitions, (c) structural properties."""
import os

import numpy as np
import pytest

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def test_layout_known_answers(oracle):
    L = oracle.orb_layout(1241, 376, 3000)
    assert L["w"] == [1241, 1034, 862, 718, 598, 499, 416, 346]   # SURVEY.md section 4 (computed from cvRound(W / 1.2^l))
    assert L["h"] == [376, 313, 261, 218, 181, 151, 126, 105]
    assert L["nfeat"] == [652, 543, 452, 377, 314, 262, 218, 182] and sum(L["nfeat"]) == 3000
    assert np.allclose(L["scale"], [1.2 ** l for l in range(8)], rtol=1e-6)


def test_gaussian_kernel_fixed_point(oracle):
    # exp(-x^2/8)/sum for x=-3..3 is 0.0702 0.1311 0.1907 0.2161 ...; x256 rounded
    assert oracle.gaussian_kernel7_fixed() == [18, 34, 49, 55, 49, 34, 18]
    k = np.exp(-np.arange(-3, 4) ** 2 / 8.0); k /= k.sum()
    assert [int(round(v * 256)) for v in k] == [18, 34, 49, 55, 49, 34, 18]


def test_blur_matches_numpy_fixed_point(oracle):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 53), dtype=np.uint8)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(k[i] * pad[:, i:i + img.shape[1]] for i in range(7))
    cols = sum(k[i] * rows[i:i + img.shape[0], :] for i in range(7))
    want = np.clip((cols + 32768) >> 16, 0, 255).astype(np.uint8)
    assert np.array_equal(oracle.gaussian_blur7(img), want)


def test_blur_close_to_scipy_gaussian(oracle):
    """third-party pin of kernel shape and border mode: scipy.ndimage.gaussian_filter(sigma=2, radius 3, mode="mirror" = reflect-101)
    in float64.  cv::GaussianBlur's 8-bit path uses the taps cvRound(256 k) = {18,34,49,55,49,34,18}, which add up to 257: the
    fixed-point result is the float Gaussian times (257/256)^2, rounded -- within one grey level after that gain"""
    from scipy.ndimage import gaussian_filter, uniform_filter
    rng = np.random.default_rng(5)
    img = uniform_filter(rng.integers(0, 256, (97, 131)).astype(np.float64), 3).astype(np.uint8)
    got = oracle.gaussian_blur7(img).astype(np.float64)
    want = gaussian_filter(img.astype(np.float64), sigma=2.0, truncate=1.5, mode="mirror") * (257.0 / 256.0) ** 2
    diff = np.abs(got - want)
    assert diff.max() < 0.9 and diff.mean() < 0.35, (diff.max(), diff.mean())


def test_resize_matches_numpy_fixed_point(oracle):
    """independent vectorised restatement of the 11-bit fixed-point INTER_LINEAR path"""
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (61, 97), dtype=np.uint8)
    for dw, dh in ((81, 51), (97, 61), (33, 20)):
        sw, sh = src.shape[1], src.shape[0]

        def tab(dn, sn):
            scale = 1.0 / (dn / sn)
            f = ((np.arange(dn) + 0.5) * scale - 0.5).astype(np.float32)
            s = np.floor(f).astype(np.int64)
            f = (f - s.astype(np.float32)).astype(np.float32)
            return s, f
        sx, fx = tab(dw, sw); sy, fy = tab(dh, sh)
        fx = np.where((sx < 0) | (sx >= sw - 1), np.float32(0), fx); sx = np.clip(sx, 0, sw - 1)
        a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64); a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
        b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64); b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
        S = src.astype(np.int64)
        sx1 = np.minimum(sx + 1, sw - 1)
        H = S[:, sx] * a0 + S[:, sx1] * a1
        y0 = np.clip(sy, 0, sh - 1); y1 = np.clip(sy + 1, 0, sh - 1)
        want = ((((b0[:, None] * (H[y0] >> 4)) >> 16) + ((b1[:, None] * (H[y1] >> 4)) >> 16) + 2) >> 2).astype(np.uint8)
        assert np.array_equal(oracle.resize_linear(src, dw, dh), want)
    const = np.full((30, 40), 77, np.uint8)
    assert (oracle.resize_linear(const, 33, 25) == 77).all()


def test_resize_geometry_matches_scikit_image(oracle):
    """third-party pin of the geometry (pixel-centre alignment, replicated edge): scikit-image's float64 bilinear resize at the ORB
    scale 1.2 (tests/golden/skimage_resize.npz, generator tests/golden/make_skimage_resize.py); the 8-bit fixed-point path must
    stay within one grey level of it"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_resize.npz"))
    got = oracle.resize_linear(g["img"], int(g["dw"]), int(g["dh"])).astype(np.float64)
    diff = np.abs(got - g["out"].astype(np.float64))
    assert diff.max() < 0.9 and diff.mean() < 0.35, (diff.max(), diff.mean())


def test_pyramid_chain_is_level_from_previous_level(oracle, synth):
    img = synth.noise_image(3, 320, 200)
    lv = oracle.build_pyramid(img)
    assert np.array_equal(lv[0], img)
    for l in range(1, 8):
        assert np.array_equal(lv[l], oracle.resize_linear(lv[l - 1], lv[l].shape[1], lv[l].shape[0]))


def _fast_numpy(img, t):
    """FAST-9/16 from the definition: >= 9 contiguous ring pixels all > v+t or all < v-t; score = largest t' that still passes"""
    h, w = img.shape
    I = img.astype(np.int64)
    ring = np.stack([I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # 16 x (h-6) x (w-6)
    c = I[3:h - 3, 3:w - 3]
    d = c[None] - ring
    ext = np.concatenate([d, d[:8]])
    mins = np.stack([ext[i:i + 9].min(0) for i in range(16)]).max(0)       # best dark arc: all d >= mins
    maxs = np.stack([(-ext[i:i + 9]).min(0) for i in range(16)]).max(0)    # best bright arc
    best = np.maximum(mins, maxs)
    corner = best > t
    return corner, np.maximum(best, t) - 1


def test_fast_matches_definition(oracle, synth):
    rng = np.random.default_rng(2)
    img = synth.noise_image(5, 200, 120)
    img[rng.random(img.shape) < 0.02] = 255
    corner, score = _fast_numpy(img, 20)
    kps = oracle.fast9_16(img, 20, nonmax=False)
    got = np.zeros_like(corner)
    got[kps["y"].astype(int) - 3, kps["x"].astype(int) - 3] = True
    assert np.array_equal(got, corner) and corner.sum() > 50
    ys, xs = np.nonzero(corner)
    for y, x in list(zip(ys, xs))[:300]:
        assert oracle.fast_corner_score(img, x + 3, y + 3, 20) == score[y, x]
    # with 3x3 NMS: strict maximum over the 8 neighbours' scores (0 where not a corner), rows/cols 3..n-4
    sc = np.where(corner, score, 0)
    full = np.zeros(img.shape, np.int64); full[3:-3, 3:-3] = sc
    keep = np.zeros_like(full, bool)
    for y in range(3, img.shape[0] - 3):
        for x in range(3, img.shape[1] - 3):
            if corner[y - 3, x - 3]:
                nb = full[y - 1:y + 2, x - 1:x + 2].copy(); nb[1, 1] = -1
                keep[y, x] = full[y, x] > nb.max()
    kn = oracle.fast9_16(img, 20, nonmax=True)
    got = np.zeros_like(keep); got[kn["y"].astype(int), kn["x"].astype(int)] = True
    assert np.array_equal(got, keep)
    assert (kn["response"] == full[kn["y"].astype(int), kn["x"].astype(int)]).all()
    order = np.lexsort((kn["x"], kn["y"]))
    assert (order == np.arange(len(kn))).all()  # raster order


def test_fast_corner_set_matches_scikit_image(oracle):
    """third-party pin: the corner set of scikit-image's corner_fast(n=9, threshold=20) on three seeded images
    (tests/golden/skimage_fast9.npz, generator tests/golden/make_skimage_fast9.py -- scikit-image itself is not needed here)"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_fast9.npz"))
    for k in range(3):
        img = g["img%d" % k]
        h, w = img.shape
        want = np.unpackbits(g["mask%d" % k])[:h * w].reshape(h, w).astype(bool)
        assert want[3:h - 3, 3:w - 3].sum() > 500 and not want[:3].any() and not want[:, :3].any()
        c = oracle.fast9_16(img, 20, nonmax=False)
        got = np.zeros((h, w), bool)
        got[c["y"].astype(int), c["x"].astype(int)] = True
        assert np.array_equal(got, want), "image %d: %d differing pixels" % (k, int((got != want).sum()))


def test_fast_hand_made_patches(oracle):
    base = np.full((7, 7), 100, np.uint8)
    for n_bright, expect in ((8, False), (9, True), (12, True), (16, True)):
        p = base.copy()
        for k in range(n_bright):
            dx, dy = RING[k]
            p[3 + dy, 3 + dx] = 150
        kp = oracle.fast9_16(np.pad(p, 3, mode="edge"), 20, nonmax=False)
        is_corner = any((k["x"] == 6 and k["y"] == 6) for k in kp)
        assert is_corner == expect, n_bright
    p = base.copy()
    for k in range(10):
        dx, dy = RING[(k + 5) % 16]
        p[3 + dy, 3 + dx] = 30 + k  # darker by 70-k (70..61): the best 9-arc is k=0..8 with min 62 -> largest passing t is 61
    assert oracle.fast_corner_score(np.pad(p, 3, mode="edge"), 6, 6, 20) == 61


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0, 1) == 0.0 and oracle.fast_atan2(1, 0) == 90.0 and oracle.fast_atan2(0, -1) == 180.0 and oracle.fast_atan2(-1, 0) == 270.0
    rng = np.random.default_rng(3)
    for _ in range(2000):
        y, x = rng.normal(0, 100, 2)
        ref = np.degrees(np.arctan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        assert 0 <= got <= 360 and min(abs(got - ref), 360 - abs(got - ref)) < 0.3  # the polynomial is a ~0.3 deg approximation


def test_ic_angle_matches_scikit_image(oracle):
    """third-party pin: scikit-image's corner_orientations on its OFAST_MASK (the same 749-pixel circular patch) at 200 seeded points
    (tests/golden/skimage_ic_angle.npz, generator tests/golden/make_skimage_ic_angle.py); scikit-image uses atan2 in double
    precision, OpenCV the fastAtan2 polynomial (~0.01 degree): same angle within 0.05 degree"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "skimage_ic_angle.npz"))
    img, pts, want = g["img"], g["pts"], np.degrees(g["angle_rad"]) % 360.0
    got = np.array([oracle.ic_angle(img, int(c), int(r)) for r, c in pts])
    diff = (got - want + 180.0) % 360.0 - 180.0
    assert np.abs(diff).max() < 0.05, np.abs(diff).max()


def test_ic_angle_on_ramps(oracle):
    yy, xx = np.mgrid[0:64, 0:64]
    for gx, gy, want in ((1, 0, 0.0), (0, 1, 90.0), (-1, 0, 180.0), (0, -1, 270.0), (1, 1, 45.0)):
        img = np.clip(128 + 2 * (gx * (xx - 32) + gy * (yy - 32)), 0, 255).astype(np.uint8)
        got = oracle.ic_angle(img, 32, 32)
        assert min(abs(got - want), 360 - abs(got - want)) < 0.5, (gx, gy, got)
    # the disc: moments over |u| <= umax[|v|] with umax = 15 15 15 15 14 14 14 13 13 12 11 10 9 8 6 3
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (64, 64)).astype(np.uint8)
    m10 = m01 = 0
    for v in range(-15, 16):
        for u in range(-umax[abs(v)], umax[abs(v)] + 1):
            m10 += u * int(img[32 + v, 32 + u]); m01 += v * int(img[32 + v, 32 + u])
    assert oracle.ic_angle(img, 32, 32) == oracle.fast_atan2(np.float32(m01), np.float32(m10))


def test_harris_matches_numpy(oracle):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (40, 40)).astype(np.uint8)
    I = img.astype(np.int64)
    x0, y0 = 20, 17
    a = b = c = 0
    for y in range(y0 - 3, y0 + 4):
        for x in range(x0 - 3, x0 + 4):
            Ix = (I[y, x + 1] - I[y, x - 1]) * 2 + (I[y - 1, x + 1] - I[y - 1, x - 1]) + (I[y + 1, x + 1] - I[y + 1, x - 1])
            Iy = (I[y + 1, x] - I[y - 1, x]) * 2 + (I[y + 1, x - 1] - I[y - 1, x - 1]) + (I[y + 1, x + 1] - I[y - 1, x + 1])
            a += Ix * Ix; b += Iy * Iy; c += Ix * Iy
    f = np.float32
    scale = f(1.0) / (f(4) * f(7) * f(255)); s4 = scale * scale * scale * scale
    want = (f(a) * f(b) - f(c) * f(c) - f(0.04) * (f(a) + f(b)) * (f(a) + f(b))) * s4
    assert oracle.harris_response(img, x0, y0) == pytest.approx(float(want), rel=1e-6)


def test_harris_matches_scipy_sobel_and_box(oracle):
    """the same response composed from third-party building blocks: scipy.ndimage.sobel (the 3 x 3 Sobel derivative, its sign and
    smoothing convention) and a 7 x 7 box sum, in float64; cv::cornerHarris' scale 1 / (4 * blockSize * 255) for 8-bit input to the
    fourth power, k = 0.04.  The oracle works in int32 sums and float32 like orb.cpp's HarrisResponses: relative 1e-5"""
    from scipy.ndimage import sobel, uniform_filter
    rng = np.random.default_rng(9)
    img = uniform_filter(rng.integers(0, 256, (60, 70)).astype(np.float64), 3).astype(np.uint8)
    I = img.astype(np.float64)
    Ix, Iy = sobel(I, axis=1, mode="nearest"), sobel(I, axis=0, mode="nearest")
    box = lambda A: uniform_filter(A, size=7, mode="nearest") * 49.0
    a, b, c = box(Ix * Ix), box(Iy * Iy), box(Ix * Iy)
    R = (a * b - c * c - 0.04 * (a + b) ** 2) * (1.0 / (4 * 7 * 255)) ** 4
    for _ in range(50):
        x, y = int(rng.integers(8, 62)), int(rng.integers(8, 52))
        assert oracle.harris_response(img, x, y) == pytest.approx(R[y, x], rel=2e-5, abs=1e-12)


def test_retain_best_keeps_ties(oracle):
    kps = np.zeros(10, oracle.KEYPOINT_DTYPE)
    kps["response"] = [5, 9, 7, 7, 7, 1, 8, 7, 2, 3]; kps["x"] = np.arange(10)
    out = oracle.retain_best(kps, 3)            # 3rd best is 7 -> all four 7s stay, input order preserved
    assert list(out["x"]) == [1, 2, 3, 4, 6, 7]
    assert len(oracle.retain_best(kps, 10)) == 10 and len(oracle.retain_best(kps, 0)) == 0


def _anms_numpy(kps, num):
    """line-by-line numpy/python restatement of visual_odometry.cpp:96-157"""
    if len(kps) < num:
        return kps
    order = np.argsort(-kps["response"], kind="stable")
    k = kps[order]
    rad = np.full(len(k), np.finfo(np.float64).max)
    for i in range(len(k)):
        thr = np.float32(k["response"][i]) * np.float32(1.11)
        j = 0
        while j < i and k["response"][j] > thr:
            dx = np.float32(k["x"][i] - k["x"][j]); dy = np.float32(k["y"][i] - k["y"][j])
            rad[i] = min(rad[i], np.sqrt(np.float64(dx) * np.float64(dx) + np.float64(dy) * np.float64(dy)))
            j += 1
    final = np.sort(rad)[::-1][num - 1]
    return k[rad >= final]


def test_anms_matches_reference_restated_in_numpy(oracle, synth):
    img = synth.noise_image(7, 400, 240)
    kps = oracle.orb_detect(img, 800)
    for num in (100, 300):
        got = oracle.anms(kps, num)
        want = _anms_numpy(kps, num)
        assert len(got) >= num and len(got) == len(want)
        assert np.array_equal(got["x"], want["x"]) and np.array_equal(got["y"], want["y"])
    assert len(oracle.anms(kps, len(kps) + 1)) == len(kps)  # :100 no-op


def _pattern_from_header():
    import os, re
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "orb_pattern.h")).read()
    nums = [int(v) for v in re.findall(r"-?\d+", txt.split("= {", 1)[1])]
    return np.array(nums[:1024]).reshape(256, 4)


def test_pattern_table_first_rows():
    p = _pattern_from_header()
    # the published ORB learned pattern starts 8,-3,9,5 / 4,2,7,-12 / -11,9,-8,2 / 7,-12,12,-13 / 2,-13,2,12 ... and ends -1,-6,0,-11
    assert p[:5].tolist() == [[8, -3, 9, 5], [4, 2, 7, -12], [-11, 9, -8, 2], [7, -12, 12, -13], [2, -13, 2, 12]]
    assert p[-1].tolist() == [-1, -6, 0, -11] and np.abs(p).max() == 13


def test_rbrief_matches_numpy(oracle, synth):
    """independent restatement of computeOrbDescriptors on level 0 and a coarse level"""
    img = synth.noise_image(9, 400, 240)
    kps = oracle.orb_detect(img, 600)
    kps = kps[np.isin(kps["octave"], (0, 3))][:40]
    out_k, desc = oracle.orb_compute(img, kps)
    pat = _pattern_from_header().astype(np.float32)
    lv = oracle.build_pyramid(img, nfeatures=500)
    L = oracle.orb_layout(400, 240, 500)
    f = np.float32
    for kp, d in zip(out_k, desc):
        l = int(kp["octave"])
        blur = oracle.gaussian_blur7(lv[l])
        s = f(1.0) / f(L["scale"][l])
        cx = int(np.rint(f(kp["x"]) * s)); cy = int(np.rint(f(kp["y"]) * s))
        ang = f(kp["angle"]) * f(np.pi / 180.0)
        a = f(np.cos(np.float64(ang))); b = f(np.sin(np.float64(ang)))
        bits = []
        for x0, y0, x1, y1 in pat:
            ix0 = int(np.rint(x0 * a - y0 * b)); iy0 = int(np.rint(x0 * b + y0 * a))
            ix1 = int(np.rint(x1 * a - y1 * b)); iy1 = int(np.rint(x1 * b + y1 * a))
            bits.append(int(blur[cy + iy0, cx + ix0]) < int(blur[cy + iy1, cx + ix1]))
        want = np.packbits(np.array(bits, np.uint8), bitorder="little")
        assert np.array_equal(d, want)


def test_compute_regroups_by_octave_stably(oracle, synth):
    img = synth.noise_image(9, 400, 240)
    kps = oracle.anms(oracle.orb_detect(img, 600), 200)          # response order: octaves interleaved
    assert (np.diff(kps["octave"]) < 0).any()
    out, desc = oracle.orb_compute(img, kps)
    assert (np.diff(out["octave"]) >= 0).all()
    for l in range(8):
        a = kps[kps["octave"] == l]; b = out[out["octave"] == l]
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["y"], b["y"])   # stable inside a level


def test_detect_structure(oracle, synth):
    img = synth.noise_image(0)
    kps = oracle.orb_detect(img)
    L = oracle.orb_layout(1241, 376, 3000)
    cnt = np.bincount(kps["octave"], minlength=8)
    assert (cnt >= np.minimum(cnt, L["nfeat"])).all() and (cnt <= np.array(L["nfeat"]) + 8).all()
    assert (np.diff(kps["octave"]) >= 0).all()
    for l in range(8):
        k = kps[kps["octave"] == l]
        s = np.float32(L["scale"][l])
        x = np.rint(k["x"] / s); y = np.rint(k["y"] / s)
        assert (x >= 31).all() and (x < L["w"][l] - 31).all() and (y >= 31).all() and (y < L["h"][l] - 31).all()
        assert (np.lexsort((x, y)) == np.arange(len(k))).all()          # raster order inside a level
        assert np.allclose(k["size"], 31 * s)
