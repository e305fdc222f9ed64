#!/opt/conda/bin/python3.9
"""Third-party pin for the FAST-9/16 corner predicate: scikit-image's `corner_fast` (0.18.3, the Anaconda Python 3.9 of the build
container; an implementation of Rosten & Drummond's segment test that shares no code with OpenCV or with this repository) on three
seeded images.  A pixel is a corner when `corner_fast(image, n=9, threshold=20) > 0`, i.e. when 9 contiguous pixels of the 16-pixel
Bresenham circle are all > I + 20 or all < I - 20 -- the definition cv::FAST(threshold=20, TYPE_9_16) implements.  The image is passed
as float64 with integer values, so every comparison is exact.

    /opt/conda/bin/python3.9 tests/golden/make_skimage_fast9.py        # writes tests/golden/skimage_fast9.npz

The fixture holds the images and the bit-packed corner masks (inputs and expected outputs only); tests/test_oracle_orb.py checks the
oracle's FAST corner set (before non-maximum suppression) against it.  scikit-image is NOT needed to run the tests.
"""
import os

import numpy as np
from skimage.feature import corner_fast


def main():
    rng = np.random.default_rng(7)
    out = {}
    for k, (h, w) in enumerate([(64, 96), (80, 120), (57, 83)]):
        # a gradient with random rectangles and +-3 noise: plenty of corners, and exact ties (p == v +- 20) do occur
        img = (np.add.outer(np.arange(h), np.arange(w)) % 256 // 4 + 60).astype(np.int32)
        for _ in range(60):
            x0, y0 = rng.integers(0, w - 4), rng.integers(0, h - 4)
            img[y0:y0 + rng.integers(2, 20), x0:x0 + rng.integers(2, 30)] += rng.integers(-80, 80)
        img += rng.integers(-3, 4, img.shape)
        img = np.clip(img, 0, 255).astype(np.uint8)
        mask = corner_fast(img.astype(np.float64), n=9, threshold=20.0) > 0
        out["img%d" % k] = img
        out["mask%d" % k] = np.packbits(mask)
        print("image %d: %dx%d, %d corners" % (k, w, h, int(mask.sum())))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_fast9.npz"), **out)


if __name__ == "__main__":
    main()
