#!/usr/bin/env python3
"""Two 1241x376 8-bit gray PNGs written by an INDEPENDENT encoder (Pillow / its bundled zlib: adaptive per-row filter choice, the stream cut
into many IDAT chunks) -- until round 6 the host reader `ImageSource::read_png` (visual_odometry.cpp:49-50's cv::imread of KITTI's
image_0/%06d.png) had only ever seen PNGs written by the repo's own synth.write_png.  The pixels are a deterministic function (`scene`)
so the fixtures stay small (smooth shading + shapes + two noisy bands) and the test regenerates the expected array instead of storing it.
usage: python tests/golden/make_png_pil.py   (writes pil_kitti_a.png / pil_kitti_b.png next to this file)"""
import os

import numpy as np

W, H = 1241, 376


def scene(k):
    """deterministic 376 x 1241 u8 image number k (0 | 1)"""
    rng = np.random.default_rng(1000 + k)
    y, x = np.mgrid[0:H, 0:W].astype(np.float64)
    img = 96 + 60 * np.sin(x / (37.0 + 11 * k)) * np.cos(y / 29.0) + 0.05 * x - 0.08 * y
    for _ in range(60):                                   # rectangles and discs with sharp edges (Sub / Up / Paeth rows win there)
        cx, cy = rng.integers(0, W), rng.integers(0, H)
        rw, rh = rng.integers(6, 90), rng.integers(4, 60)
        val = float(rng.integers(0, 256))
        if rng.random() < 0.5:
            img[max(cy - rh, 0):cy + rh, max(cx - rw, 0):cx + rw] = val
        else:
            img[(x - cx) ** 2 + (y - cy) ** 2 < float(min(rw, rh)) ** 2] = val
    band = slice(40 + 150 * k, 80 + 150 * k)             # a band of white noise (filter None / a stored-ish deflate block) and a dithered band
    img[band] += rng.integers(-40, 41, (band.stop - band.start, W))
    img[300:330, 200:900] += 25 * ((x[300:330, 200:900] + y[300:330, 200:900]) % 2)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def main():
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 4096                              # IDAT chunks of at most 4 KB: dozens per file
    here = os.path.dirname(os.path.abspath(__file__))
    for k, name in enumerate(("pil_kitti_a.png", "pil_kitti_b.png")):
        Image.fromarray(scene(k), mode="L").save(os.path.join(here, name), format="PNG", optimize=bool(k), compress_level=9 if k else 6)
        print(name, os.path.getsize(os.path.join(here, name)), "bytes")


if __name__ == "__main__":
    main()
