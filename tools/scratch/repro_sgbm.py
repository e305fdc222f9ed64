import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle as O
import stereo_visual_slam_amd as pkg
from stereo_visual_slam_amd import synth
w, h, seed, sh = 100, 124, 880364543, 17
vo = pkg.VO(device=0, max_batch=1)
for noise in (False, True):
    L = synth.noise_image(seed % 1000, w + 40, h)
    Lc = np.ascontiguousarray(L[:, :w]); R = np.ascontiguousarray(L[:, sh:sh + w]).copy()
    if noise:
        rng = np.random.default_rng(1); R = np.clip(R.astype(int) + rng.integers(-15, 16, R.shape), 0, 255).astype(np.uint8)
    gf, gi, graw = vo.disparity_map(Lc, R, return_i16=True)
    wi, wraw = O.sgbm_compute(Lc, R, return_raw=True)
    print("noise", noise, "raw diff", (graw != wraw).sum(), "final diff", (gi != wi).sum())
    bad = np.argwhere(graw != wraw)
    print(bad[:10], [(int(graw[y, x]), int(wraw[y, x])) for y, x in bad[:10]])
    bad = np.argwhere(gi != wi)
    print(bad[:10], [(int(gi[y, x]), int(wi[y, x])) for y, x in bad[:10]])
for ww in (97, 98, 99, 100, 101, 104, 112):
    L = synth.noise_image(5, ww + 40, 60); Lc = np.ascontiguousarray(L[:, :ww]); R = np.ascontiguousarray(L[:, 9:9 + ww])
    gf, gi, graw = vo.disparity_map(Lc, R, return_i16=True); wi, wraw = O.sgbm_compute(Lc, R, return_raw=True)
    print(ww, (graw != wraw).sum(), (gi != wi).sum())
