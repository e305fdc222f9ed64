import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stereo_visual_slam_amd import synth
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HOST = os.path.join(ROOT, "stereo-visual-slam_amd", "host")
n = 40
with tempfile.TemporaryDirectory() as d:
    synth.write_pgm_sequence(d + "/", n, seed=5, fmt="png")
    for depth, pnp in ((0, 0), (1, 1)):
        t0 = time.perf_counter()
        out = subprocess.run([os.path.join(HOST, "run_vslam"), d + "/", str(n), "1", "500", os.path.join(d, "t.txt"), "0", str(depth), str(pnp)], capture_output=True, text=True)
        dt = time.perf_counter() - t0
        print("depth=%d pnp=%d: %.1f ms/frame wall incl. process start + PNG decode (%s)" % (depth, pnp, dt / n * 1e3, out.stdout.strip().split("\n")[-2][:100]))
