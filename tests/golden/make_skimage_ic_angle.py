#!/opt/conda/bin/python3.9
"""Third-party pin for the intensity-centroid orientation of ORB (cv::ORB's IC_Angle: umax table, half-patch 15): scikit-image
0.18.3's `corner_orientations` with its own `OFAST_MASK` (the same 31 x 31 circular patch, 749 pixels) on 200 seeded points of one
seeded image.  scikit-image returns atan2(m01, m10) in double precision (radians, (-pi, pi]); OpenCV returns fastAtan2 (degrees,
[0, 360), a polynomial good to ~0.01 degree), so the check in tests/test_oracle_orb.py is: same angle within 0.05 degree.  What this
pins: the patch geometry (the umax table), the moments and the axis / sign convention.

    /opt/conda/bin/python3.9 tests/golden/make_skimage_ic_angle.py     # writes tests/golden/skimage_ic_angle.npz

Inputs and expected outputs only; scikit-image is NOT needed to run the tests.
"""
import os

import numpy as np
from scipy.ndimage import uniform_filter
from skimage.feature import corner_orientations
from skimage.feature.orb import OFAST_MASK


def main():
    rng = np.random.default_rng(11)
    h, w = 90, 130
    img = uniform_filter(rng.integers(0, 256, (h, w)).astype(np.float64), 5).astype(np.uint8)  # (smoothed: well-conditioned centroids)
    pts = np.stack([rng.integers(16, h - 16, 200), rng.integers(16, w - 16, 200)], 1)          # (row, col)
    ang = corner_orientations(img.astype(np.float64), pts, OFAST_MASK)
    assert OFAST_MASK.shape == (31, 31) and int(OFAST_MASK.sum()) == 749
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_ic_angle.npz"), img=img, pts=pts.astype(np.int32), angle_rad=ang)
    print("200 points, angles %.3f .. %.3f rad" % (ang.min(), ang.max()))


if __name__ == "__main__":
    main()
