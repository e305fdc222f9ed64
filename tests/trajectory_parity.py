#!/usr/bin/env python3
"""Trajectory-level parity: the reference's loop (run_vslam.cpp:40-82) through the GPU path and through the CPU path.

Both drivers are the SAME C++ host mirror (stereo-visual-slam_amd/host/): `host/run_vslam` links libvslam_hip.so (HIP kernels),
`oracle/run_vslam_cpu` links the oracle shim (oracle/cpu_shim.c -> libvo_oracle.so).  They run the same rendered KITTI-shaped
sequence and write one trace line per frame and per BA run; this tool compares the traces:
  * integers must be identical: tracking state, keyframe decisions, number of detections, number of gated frame-to-frame
    matches and an FNV hash over their (queryIdx, trainIdx, distance), PnP inlier count, map sizes, BA inlier-landmark count,
    keyframe ids in the window;
  * poses (per-frame T_c_w, all window poses after every BA schedule) within rtol 1e-4 / atol 1e-6 (north_star: 1e-4 rel).
Test infrastructure (it executes oracle/run_vslam_cpu); used by tests/test_gpu_host_driver.py and by hand:
    python tests/trajectory_parity.py --frames 50 --out profiles/r02_trajectory_parity.json
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HOST = os.path.join(ROOT, "stereo-visual-slam_amd", "host")
GPU_DRIVER = os.path.join(HOST, "run_vslam")
CPU_DRIVER = os.path.join(ROOT, "oracle", "run_vslam_cpu")


def build():
    subprocess.check_call(["make", "-C", HOST, "-s", "-j8"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


def parse_trace(path):
    """-> list of (kind, {int fields}, poses ndarray (n, 7))"""
    out = []
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        kind = tok[0]
        ints, poses = {kind: tok[1]}, []
        i = 2
        while i < len(tok):
            key = tok[i]
            if key == "T":
                poses.append([float(x) for x in tok[i + 1:i + 8]]); i += 8
            elif key == "kf" and kind == "ba":
                ints.setdefault("kf_ids", []).append(tok[i + 1])
                poses.append([float(x) for x in tok[i + 2:i + 9]]); i += 9
            else:
                ints[key] = tok[i + 1]; i += 2
        out.append((kind, ints, np.array(poses, np.float64).reshape(-1, 7)))
    return out


def compare(gpu_trace, cpu_trace, rtol=1e-4, atol=1e-6):
    g, c = parse_trace(gpu_trace), parse_trace(cpu_trace)
    res = dict(lines_gpu=len(g), lines_cpu=len(c), first_integer_mismatch=None, first_pose_mismatch=None, n_frames=0, n_ba=0, n_keyframes=0)
    sq_t, sq_r, n_pose = 0.0, 0.0, 0
    max_rel = 0.0
    for i, ((kg, ig, pg), (kc, ic, pc)) in enumerate(zip(g, c)):
        if (kg, ig) != (kc, ic) and res["first_integer_mismatch"] is None:
            res["first_integer_mismatch"] = dict(line=i, gpu={k: v for k, v in ig.items() if ic.get(k) != v}, cpu={k: v for k, v in ic.items() if ig.get(k) != v})
        if pg.shape == pc.shape and len(pg):
            if not np.allclose(pg, pc, rtol=rtol, atol=atol) and res["first_pose_mismatch"] is None:
                res["first_pose_mismatch"] = dict(line=i, max_abs_diff=float(np.abs(pg - pc).max()))
            d = pg - pc
            sq_t += float((d[:, 4:] ** 2).sum()); sq_r += float((d[:, :4] ** 2).sum()); n_pose += len(pg)
            max_rel = max(max_rel, float((np.abs(d) / np.maximum(np.abs(pc), 1e-2)).max()))
        elif pg.shape != pc.shape and res["first_pose_mismatch"] is None:
            res["first_pose_mismatch"] = dict(line=i, shapes=[list(pg.shape), list(pc.shape)])
        if kg == "frame":
            res["n_frames"] += 1; res["n_keyframes"] += int(ig.get("kf", "0"))
        else:
            res["n_ba"] += 1
    res["pose_rmse_translation_m"] = float(np.sqrt(sq_t / max(n_pose, 1)))
    res["pose_rmse_quaternion"] = float(np.sqrt(sq_r / max(n_pose, 1)))
    res["pose_max_rel_diff"] = max_rel
    res["poses_compared"] = n_pose
    res["identical_integers"] = res["first_integer_mismatch"] is None and len(g) == len(c)
    res["poses_within_tol"] = res["first_pose_mismatch"] is None and len(g) == len(c)
    return res


def run_config(data_dir, n_frames, anms, q1, depth, pnp, workdir, tag):
    """runs both drivers; returns (comparison dict, timings)"""
    out = {}
    for name, exe in (("gpu", GPU_DRIVER), ("cpu", CPU_DRIVER)):
        trace = os.path.join(workdir, "trace_%s_%s.txt" % (tag, name)); traj = os.path.join(workdir, "traj_%s_%s.txt" % (tag, name))
        t0 = time.perf_counter()
        r = subprocess.run([exe, data_dir, str(n_frames), "1", str(anms), traj, str(q1), str(depth), str(pnp), trace], capture_output=True, text=True, timeout=1800)
        out[name] = dict(seconds=time.perf_counter() - t0, rc=r.returncode, stdout_tail=r.stdout[-300:], stderr_tail=r.stderr[-300:], trace=trace, traj=traj)
        if r.returncode != 0:
            raise RuntimeError("%s driver failed: %s %s" % (name, r.stdout[-500:], r.stderr[-500:]))
    cmp_ = compare(out["gpu"]["trace"], out["cpu"]["trace"])
    cmp_.update(config=dict(frames=n_frames, anms=anms, q1=q1, depth="sgbm" if depth else "lr_match_dlt", pnp="ransac" if pnp else "motion_only_lm"),
                gpu_seconds=round(out["gpu"]["seconds"], 2), cpu_seconds=round(out["cpu"]["seconds"], 2),
                gpu_summary=out["gpu"]["stdout_tail"].strip().splitlines()[-2:] if out["gpu"]["stdout_tail"].strip() else [])
    # the two trajectory files (estimated_traj.txt format, map.cpp:168-196) must list the same keyframes
    tg, tc = np.loadtxt(out["gpu"]["traj"], ndmin=2), np.loadtxt(out["cpu"]["traj"], ndmin=2)
    cmp_["traj_rows"] = [len(tg), len(tc)]
    cmp_["traj_same_frames"] = bool(tg.shape == tc.shape and np.array_equal(tg[:, 0], tc[:, 0]))
    cmp_["traj_max_abs_diff"] = float(np.abs(tg - tc).max()) if tg.shape == tc.shape and len(tg) else None
    return cmp_


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--out", default="")
    ap.add_argument("--configs", default="1500:1:1:1,500:0:1:1,1500:1:0:0,500:0:0:0", help="comma list of anms:q1:depth:pnp")
    a = ap.parse_args()
    from stereo_visual_slam_amd import synth
    build()
    results = []
    with tempfile.TemporaryDirectory() as d:
        synth.write_pgm_sequence(d + "/", a.frames, seed=a.seed, fmt="png")
        for spec in a.configs.split(","):
            anms, q1, depth, pnp = (int(x) for x in spec.split(":"))
            r = run_config(d + "/", a.frames, anms, q1, depth, pnp, d, spec.replace(":", "_"))
            print(json.dumps(r), flush=True)
            results.append(r)
    ok = all(r["identical_integers"] and r["poses_within_tol"] for r in results)
    if a.out:
        json.dump(dict(sequence="synth.write_pgm_sequence(seed=%d, %d frames, png)" % (a.seed, a.frames), all_ok=ok, results=results), open(a.out, "w"), indent=1)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
