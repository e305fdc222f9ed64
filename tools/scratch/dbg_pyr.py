import os, sys, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import stereo_visual_slam_amd as pkg
    from stereo_visual_slam_amd import synth
    import oracle as O
    img = synth.noise_image(0)
    vo = pkg.VO(device=0, max_batch=1)
    kps = O.orb_detect(img)[:64].copy()
    kps["x"] = np.linspace(31, img.shape[1] - 32, len(kps)).astype(np.float32)
    kps["y"] = np.where(np.arange(len(kps)) % 2 == 0, 31.0, img.shape[0] - 32.0).astype(np.float32)
    kps["octave"] = np.arange(len(kps)) % 8
    kps["angle"] = np.linspace(0, 359, len(kps)).astype(np.float32)
    gk, gd = vo.orb_compute(img, kps)
    wk, wd = O.orb_compute(img, kps)
    bad = np.nonzero((gd != wd).any(1))[0]
    print("env", os.environ.get("VSLAM_ORB_UNFUSED"), "mismatching keypoints:", [(int(i), float(gk["x"][i]), float(gk["y"][i]), int(gk["octave"][i]), int(np.unpackbits(gd[i] ^ wd[i]).sum())) for i in bad])
else:
    subprocess.check_call([sys.executable, __file__, "x"])
    subprocess.check_call([sys.executable, __file__, "x"], env=dict(os.environ, VSLAM_ORB_UNFUSED="1"))
