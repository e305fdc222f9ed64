"""Experiment: two pipelines in flight, free-running vs anti-phase (a pipeline's front end starts when the OTHER pipeline's BA starts)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from importlib import import_module
KeyframePipeline = import_module("stereo-visual-slam_amd.pipeline").KeyframePipeline
synth = import_module("stereo-visual-slam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seq = synth.stereo_sequence(64, seed=0, w=1241, h=376, workers=0)
pipes = [KeyframePipeline(B, device=0, anms_num=1500, unique_frames=64, sequence=seq, ba_windows="tracks") for _ in range(2)]
ev = [None, None]
def step(k, mode):
    p, o = pipes[k % 2], (k + 1) % 2
    if mode == "anti" and ev[o] is not None:
        p.stream.wait_event(ev[o])
    p.stage_orb(); p.stage_stereo_match(); p.stage_track()
    e = torch.cuda.Event(); e.record(p.stream); ev[k % 2] = e
    p.stage_ba()
def run(n, mode):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n): step(k, mode)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
for mode in ("free", "anti"):
    ev[0] = ev[1] = None; run(4, mode)
for rep in range(3):
    for mode in ("free", "anti"):
        ev[0] = ev[1] = None
        ms = run(steps, mode); print("B=%d %s: %.3f ms/step %.0f keyframes/s" % (B, mode, ms, B / ms * 1e3), flush=True)
