#!/opt/conda/bin/python3.9
"""Third-party pin for the GEOMETRY of the pyramid's resize (cv::resize INTER_LINEAR: pixel-centre alignment
src = (dst + 0.5) * scale - 0.5, replicated edge): scikit-image 0.18.3's `transform.resize(order=1, mode="edge",
anti_aliasing=False)` in float64 on one seeded image, scale 1.2 (the ORB scale factor).  OpenCV's 8-bit path interpolates with
11-bit fixed-point coefficients and rounds to integers, so the oracle must stay within one grey level of the float result
(|rounding| <= 0.5 + coefficient quantisation); the arithmetic itself is pinned separately (tests/test_oracle_orb.py).

    /opt/conda/bin/python3.9 tests/golden/make_skimage_resize.py       # writes tests/golden/skimage_resize.npz

Inputs and expected outputs only; scikit-image is NOT needed to run the tests.
"""
import os

import numpy as np
from scipy.ndimage import uniform_filter
from skimage.transform import resize


def main():
    rng = np.random.default_rng(3)
    img = uniform_filter(rng.integers(0, 256, (120, 173)).astype(np.float64), 3).astype(np.uint8)
    dh, dw = int(round(120 / 1.2)), int(round(173 / 1.2))
    out = resize(img.astype(np.float64), (dh, dw), order=1, mode="edge", anti_aliasing=False, preserve_range=True)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_resize.npz"), img=img, out=out.astype(np.float32), dw=dw, dh=dh)
    print("%dx%d -> %dx%d" % (173, 120, dw, dh))


if __name__ == "__main__":
    main()
