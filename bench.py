#!/usr/bin/env python3
"""bench.py -- stereo keyframes/s through the MI355X hot path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over a batch of B synthetic KITTI-00-shaped stereo keyframes per GPU, inputs
resident in HBM (stereo-visual-slam_amd/pipeline.py): ORB detect+ANMS(1500)+describe on 2B images, L/R Hamming match,
DLT triangulation, frame-to-frame match, motion-only LM pose, local BA (10 KF x ~3000 landmarks, schedule 5+5+10 LM +
10 pose-only).  Keyframes shard across ranks with no data-path collective ("weak" scaling); the only collective is the
RCCL all-gather of the per-keyframe poses (56 B each) once per step.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with N ranks
(one per GPU, rendezvous on 127.0.0.1); under a launcher, --gpus must equal WORLD_SIZE.  It refuses to run on fewer GPUs.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, timed live with HIP events on
the launch stream through the library's stage profiler) and `cpu_baseline` (the CPU oracle -- a "port", the reference
itself is unbuildable here -- timed on this host, rank 0 at N=1, on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable with a float4 copy


def algorithmic_bytes(kernel, pipe, anms_num):
    """SURVEY.md 8(d) compulsory bytes moved by one bracket of `kernel` at this batch size"""
    B = pipe.B
    w, h = pipe.w, pipe.h
    if kernel.startswith("orb_"):
        # ORB per image: image read once + N x (32 B descriptor + 28 B keypoint) written once; 2B images per launch set
        return 2 * B * (w * h + anms_num * 60), "2B images x (w*h + N*60) B"
    if kernel.startswith("match_"):
        # (Nq+Nt)*32 + Nq*8 per item
        return B * ((anms_num + anms_num) * 32 + anms_num * 8), "B x ((Nq+Nt)*32 + Nq*8) B"
    if kernel.startswith("lm_window_kernel<pnp>") or kernel.startswith("pnp_wave_kernel"):
        return (B - 1) * 10 * 20 * 500, "(B-1) x 10 its x 20 B/point x ~500 points"
    if kernel.startswith("lm_window_kernel"):
        # LM linearisations a window's schedule EXECUTES: 10 of the pose-only pass + 20 of the three optimize_map passes -- or 10 / 15 when the
        # adaptive schedule continued the window's first / second pass as the last one (pipe.ba_passes, vslam_ba_schedule_passes_dev)
        passes = getattr(pipe, "ba_passes", None)
        lin = 10.0 + (np.array([0, 10, 15, 20], np.float64)[np.clip(np.asarray(passes), 0, 3)] if passes is not None else 20.0)
        lin_txt = "LM linearisations executed (10 pose-only + 10 / 15 / 20 of the optimize_map passes: mean %.1f)" % float(np.mean(lin))
        if getattr(pipe, "ba_shape", None) is not None:  # windows built from the step's tracks: their actual sizes
            E, L, K = (np.asarray(x, np.float64) for x in pipe.ba_shape)
            per_it = E * 16 + L * 12 + K * 56 + (6 * K) ** 2 * 8 + 6 * K * 8 + L * 12
            return float((per_it * lin).sum()), "sum over the B built windows of %s x (E*16 + L*12 + K*56 + (6K)^2*8 + 6K*8 + L*12) B" % lin_txt
        E, L, K = pipe.edges_per_window, pipe.lms_per_window, pipe.n_kf
        per_it = E * 16 + L * 12 + K * 56 + (6 * K) ** 2 * 8 + 6 * K * 8 + L * 12
        return float(np.sum(np.broadcast_to(lin, (B,)) * per_it)), "sum over the B windows of %s x (E*16 + L*12 + K*56 + (6K)^2*8 + 6K*8 + L*12) B" % lin_txt
    if kernel.startswith("triangulate"):
        return B * anms_num * 30, "B x N x 30 B"
    return 0, "n/a"


def schedule_stats(passes):
    """what the adaptive BA schedule did (vslam_ba_schedule_passes_dev): windows by passes executed, LM iterations run against the plain 20"""
    p = np.clip(np.asarray(passes), 0, 3)
    its = np.array([0, 10, 15, 20], np.float64)[p]
    return {"windows_by_passes_executed": {"1": int((p == 1).sum()), "2": int((p == 2).sum()), "3": int((p == 3).sum())},
            "lm_iterations_per_window_mean": round(float(its.mean()), 2), "lm_iterations_of_the_plain_schedule": 20,
            "rule": "all three optimize_map passes start from the same poses and landmarks (run_vslam.cpp:61-64: the first two do not write back) and differ only "
                    "by the landmark flags the previous pass left; a pass that flags nothing new is continued to the last pass's 10 iterations instead of "
                    "being repeated -- bit-identical results (tests/test_gpu_lm.py::test_adaptive_schedule_is_bit_identical; --ba-plain-schedule runs every pass)"}


def _sha16(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    except Exception:
        return None


def load_counter_json(name):
    """a counter summary under profiles/ that was measured offline (rocprofv3 PMC passes).  It names the kernel sources it was measured on
    (`source_sha16`: {file under csrc/: first 16 hex digits of its sha256}); if any of them has changed since, the numbers describe another
    kernel and are NOT reported (returns (None, reason))."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        tj = json.load(open(path))
    except Exception as e:
        return None, "profiles/%s unreadable (%r)" % (name, e)
    shas = tj.get("source_sha16")
    if not shas:
        return None, "profiles/%s carries no source_sha16 (measured before the kernels were hashed): not reported" % name
    for f, h in shas.items():
        cur = _sha16(os.path.join(ROOT, "stereo-visual-slam_amd", "csrc", f))
        if cur != h:
            return None, "profiles/%s is stale: %s changed since it was measured (%s -> %s); re-run tools/profile_round.sh" % (name, f, h, cur)
    return tj, tj.get("source")


def other_rooflines(prof, pipe, args, n_steps, copy_gbs):
    """every other kernel family of the step against the roofline that bounds it; informative (the dominant kernel has `roofline`)"""
    out = []
    # ORB family: compulsory bytes of the whole family per step (image read once + 60 B per keypoint written once) over each
    # kernel's own time, and over the family's time
    orb_alg, _ = algorithmic_bytes("orb_", pipe, args.anms)
    orb_total_ms = 0.0
    for name in ("orb_resize_kernel", "orb_pyrblur_kernel", "orb_fast_kernel", "orb_select_kernel", "orb_anms_kernel", "orb_orient_kernel", "orb_blur_kernel", "orb_describe_kernel"):
        k = prof.get(name)
        if not k or k[0] <= 0:
            continue
        ms = k[0] / n_steps
        orb_total_ms += ms
        gbs = orb_alg / (ms / 1e3) / 1e9
        out.append({"kernel": name, "bound": "hbm", "ms_per_step": round(ms, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "note": "ORB-family algorithmic bytes per step over THIS kernel's time"})
    if orb_total_ms > 0:
        gbs = orb_alg / (orb_total_ms / 1e3) / 1e9
        # counter traffic of the family per step (profiles/traffic_orb.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes on these sources, same batch)
        oj, osrc = load_counter_json("traffic_orb.json")
        n_img = pipe.B if pipe.depth == "sgbm" else 2 * pipe.B
        otraffic = None
        if oj is not None:
            if oj.get("batch") == pipe.B:
                otraffic = int(oj["hbm_bytes_per_launch_set"])
            else:
                osrc = "profiles/traffic_orb.json was measured at batch %s, not %d" % (oj.get("batch"), pipe.B)
        out.append({"kernel": "orb_* (family)", "bound": "hbm", "ms_per_step": round(orb_total_ms, 4), "ms_per_1024_images": round(orb_total_ms * 1024.0 / n_img, 4),
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": int(orb_alg),
                    "traffic": otraffic, "traffic_per_1024_images": None if otraffic is None else int(otraffic * 1024.0 / n_img), "traffic_source": osrc,
                    "frac_of_copy_ceiling": round(gbs / copy_gbs, 5) if copy_gbs > 0 else None})
    # ORB is bound by VALU issue, not by bytes: the right ruler is wave-instructions per second.  SQ_INSTS_VALU per image
    # from the counter pass (profiles/orb_valu.json, tools/profile_sq.sh); a wave64 integer / packed-16 instruction holds its SIMD ~4 cycles.
    oj, _ = load_counter_json("orb_valu.json")
    if oj:
        n_img = pipe.B if pipe.depth == "sgbm" else 2 * pipe.B
        for name, per_img in oj.get("valu_wave_insts_per_image", {}).items():
            k = prof.get(name)
            if not k or k[0] <= 0 or oj.get("anms") != args.anms:
                continue
            t = k[0] / 1e3 / n_steps
            rate = per_img * n_img / t
            peak = 256 * 4 * 2.4e9 / 4.0
            out.append({"kernel": name, "bound": "valu-issue", "achieved": round(rate / 1e12, 4), "peak": round(peak / 1e12, 4), "unit": "T wave-instructions/s",
                        "frac": round(rate / peak, 4), "note": "SQ_INSTS_VALU per image (%s) x images per step over this kernel's time; peak = 1024 SIMDs x 2.4 GHz / 4 cycles" % oj.get("source")})
    k = prof.get("match_train_nearest_kernel")
    if k and k[0] > 0:
        n = float(args.anms)
        items = pipe.B + max(pipe.B - 1, 0)                      # L/R call + frame-to-frame call per step
        macs = items * n * n * 256.0 * n_steps                  # +-1 byte products per step
        tops = 2.0 * macs / (k[0] / 1e3) / 1e12
        out.append({"kernel": "match_train_nearest_kernel", "bound": "mfma", "achieved": round(tops, 1), "peak": 5000.0, "unit": "TOP/s (int8)",
                    "frac": round(tops / 5000.0, 4), "note": "v_mfma_i32_32x32x32_i8; 4250 TOP/s sustained in tools/scratch/mfma_rate.hip"})
    # what actually bounds the dominant kernel: VALU issue.  Wave-instructions per window and schedule come from the SQ counter
    # pass (tools/profile_sq.sh -> profiles/traffic.json); every VALU op, f64 or not, takes a 4-cycle issue slot of its SIMD.
    k = prof.get("lm_window_kernel")
    tj, _ = load_counter_json("traffic.json")
    wi = float(tj.get("valu_wave_insts_per_window_schedule", 0)) if tj else 0.0
    config4 = getattr(pipe, "lms_per_window", None) == 3000 and pipe.n_kf == 10 and pipe.ba_windows == "synthetic"
    if k and k[0] > 0 and wi > 0 and config4:
        sets = k[2] if len(k) > 2 and k[2] else n_steps
        t = k[0] / 1e3 / sets                                   # seconds per schedule batch
        slots = 256 * 4 * 2.4e9 * t / 4.0                       # 256 CUs x 4 SIMDs, one VALU issue per 4 cycles at 2.4 GHz
        used = wi * pipe.B
        out.append({"kernel": "lm_window_kernel", "bound": "valu-issue", "achieved": round(used / t / 1e12, 3), "peak": round(256 * 4 * 2.4e9 / 4.0 / 1e12, 3),
                    "unit": "T wave-instructions/s", "frac": round(used / slots, 4),
                    "note": "SQ_INSTS_VALU of the BA schedule (profiles/r02_ba_sq_issue_stall_summary.txt) over the live kernel time; two waves per SIMD (256 VGPRs)"})
    # ... and in FP64 vector terms (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of the BA schedule, tools/profile_f64.sh): the f64 pipe is the
    # unit the two big phases of the kernel saturate at two waves per SIMD (tools/scratch/valu_rate.hip: a wave64 f64 op holds its SIMD
    # ~4 cycles, an f32 / integer op ~2)
    fj = tj.get("f64_per_window_schedule") if tj else None
    if k and k[0] > 0 and fj and config4:
        sets = k[2] if len(k) > 2 and k[2] else n_steps
        t = k[0] / 1e3 / sets
        tflops = fj["f64_flop"] * pipe.B / t / 1e12
        simd_cycles = (fj["f64_total"] * 4.0 + (fj["valu"] - fj["f64_total"]) * 2.0) * pipe.B / 1024.0  # per SIMD
        out.append({"kernel": "lm_window_kernel", "bound": "fp64-vector", "achieved": round(tflops, 2), "peak": 78.6, "unit": "TFLOP/s (f64, FMA = 2)",
                    "frac": round(tflops / 78.6, 4), "valu_pipe_busy_frac_at_2.4GHz": round(simd_cycles / (t * 2.4e9), 3),
                    "note": "f64 ops are 66 % of the kernel's VALU wave-instructions (profiles/r02_ba_valu_f64.json); pipe occupancy = (4 x f64 + 2 x other) cycles per SIMD over the live kernel time"})
    return out


def _pose_diff(a, b):
    """(translation RMSE [m], quaternion RMSE, max |diff| relative to max(|value|, 1e-2)) between two (n, 7) pose arrays"""
    a = np.asarray(a, np.float64).reshape(-1, 7); b = np.asarray(b, np.float64).reshape(-1, 7)
    q = np.where((np.sum(a[:, :4] * b[:, :4], 1) < 0)[:, None], -b[:, :4], b[:, :4])  # q and -q are the same rotation
    d = np.concatenate([a[:, :4] - q, a[:, 4:] - b[:, 4:]], 1)
    return (float(np.sqrt((d[:, 4:] ** 2).sum(1).mean())), float(np.sqrt((d[:, :4] ** 2).sum(1).mean())),
            float((np.abs(d) / np.maximum(np.abs(a), 1e-2)).max()))


def cpu_baseline(pipe, out, anms_num, n_single=24, per_core=2, chunk_len=8):
    """The CPU oracle (a "port": the reference itself is unbuildable here) on bounded samples of the same workload, on this
    host: one thread in-process (and, on the way, the GPU's poses / counts are checked against the oracle's on that sample),
    then all cores through a pool of worker processes (oracle/cpu_keyframe.py)."""
    import multiprocessing as mp
    import tempfile
    from oracle import cpu_keyframe as W
    B, U = pipe.B, pipe.unique_frames
    tmp = tempfile.NamedTemporaryFile(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False)
    tmp.close()
    uniq = np.concatenate([pipe.h_imgs_unique_left, pipe.h_imgs_unique_right])
    np.save(tmp.name, uniq)
    init_args = (tmp.name, pipe.w, anms_num, pipe.n_kf, int(pipe.lms_per_window), pipe.window_seed0)
    try:
        # ---- one thread, and parity of the GPU step on the same keyframes
        W.init(*init_args)
        n1 = min(n_single, B)
        t0 = time.perf_counter()
        prev, pnp_o, pnp_g, ba_o, ba_g, int_mismatch = None, [], [], [], [], 0
        stage_s = {"orb_lr_match_triangulate": 0.0, "f2f_match_motion_only_lm": 0.0, "local_ba_schedule": 0.0}
        for b in range(n1):
            ts = time.perf_counter()
            cur = W.front_end(pipe.frame_of[b])
            stage_s["orb_lr_match_triangulate"] += time.perf_counter() - ts
            int_mismatch += int(out["cnt"][b] != len(cur[0])) + int(out["cnt"][B + b] != cur[5]) + int(out["nlr"][b] != len(cur[2]))
            if prev is not None:
                ts = time.perf_counter()
                T, npts, ninl, nf = W.track(prev, cur)
                stage_s["f2f_match_motion_only_lm"] += time.perf_counter() - ts
                int_mismatch += int(out["nf2f"][b - 1] != nf) + int(out["pn"][b - 1] != npts) + int(out["ninl"][b - 1] != ninl)
                if npts >= 6:
                    pnp_o.append(T); pnp_g.append(out["Tpnp"][b - 1])
            ts = time.perf_counter()
            Tb, inl = W.ba_schedule(pipe.h_windows[b % pipe.unique_windows])
            stage_s["local_ba_schedule"] += time.perf_counter() - ts
            ba_o.append(Tb); ba_g.append(out["ba_T"][b])
            lo, hi = pipe.h_lm_off[b], pipe.h_lm_off[b + 1]
            int_mismatch += int((out["ba_inl"][lo:hi] != inl).sum())
            prev = cur
        dt1 = time.perf_counter() - t0
        pt, pq, pr = _pose_diff(np.array(pnp_g), np.array(pnp_o)) if pnp_o else (None, None, None)
        bt, bq, br = _pose_diff(np.array(ba_g), np.array(ba_o))
        parity = dict(keyframes_checked=n1, integer_mismatches=int_mismatch, pnp_translation_rmse_m=pt, pnp_quaternion_rmse=pq, pnp_max_rel_diff=pr,
                      ba_translation_rmse_m=bt, ba_quaternion_rmse=bq, ba_max_rel_diff=br,
                      note="GPU step vs oracle on the same inputs: counts of keypoints / LR / f2f matches / PnP points / PnP inliers and BA landmark "
                           "flags must be identical (integer_mismatches = 0); poses: motion-only LM of keyframe b-1 -> b, and the 10 window poses "
                           "after the 5+5+10+10 schedule")
        # ---- all cores
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        all_cores = None
        if cores > 1:
            n_all = int(min(max(per_core * cores, 64), 2048))
            n_all = (n_all + chunk_len - 1) // chunk_len * chunk_len
            period = max(2 * (U - 1), 1)
            frames = [(t if t < U else period - t) for t in (b % period for b in range(n_all))]
            tasks = [(b0, min(b0 + chunk_len, n_all), frames, pipe.unique_windows) for b0 in range(0, n_all, chunk_len)]
            ctx = mp.get_context("spawn")  # workers import numpy + the oracle only; nothing of this process's HIP state is forked
            with ctx.Pool(min(cores, len(tasks)), initializer=W.init, initargs=init_args) as pool:
                pool.map(W.warm, range(4 * min(cores, len(tasks))), chunksize=1)  # every worker started and initialised
                t0 = time.perf_counter()
                done = sum(pool.map(W.chunk, tasks, chunksize=1))
                dta = time.perf_counter() - t0
            all_cores = dict(value=done / dta, unit="keyframes/s", cores=min(cores, len(tasks)), keyframes=done, seconds=round(dta, 2))
    finally:
        os.unlink(tmp.name)
    res = dict(unit="keyframes/s", kind="port",
               single_thread=dict(value=n1 / dt1, cores=1, keyframes=n1, seconds=round(dt1, 2),
                                  stage_ms_per_keyframe={k: round(1e3 * v / n1, 2) for k, v in stage_s.items()}),
               sample=("oracle/libvo_oracle.so per stereo keyframe: 2 ORB images (3000 -> ANMS %d -> rBRIEF), L/R + frame-to-frame match, DLT, "
                       "motion-only LM, BA schedule 5+5+10+10 on one 10x%d window; single thread: %d keyframes in %.1f s" % (anms_num, int(pipe.lms_per_window), n1, dt1)))
    if all_cores:
        res.update(value=all_cores["value"], cores=all_cores["cores"])
        res["sample"] += "; all cores: %d keyframes in %.1f s on %d worker processes (chunks of %d with a 1-frame front-end halo); host has %d cores" % (
            all_cores["keyframes"], all_cores["seconds"], all_cores["cores"], chunk_len, os.cpu_count())
    else:
        res.update(value=n1 / dt1, cores=1)
    return res, parity


def usable_cores():
    """CPUs this process may really use: the affinity mask, cut by a cgroup CPU quota if there is one (a box that shows 256 logical CPUs may
    grant 64), and by 128 (the workers are numpy / C single-threaded compute: SMT siblings add nothing but memory pressure)"""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, min(n, 128))


def cpu_baseline_tracks(pipe, out, anms_num, n_single=24, per_core=2):
    """tracks mode (the default step): the CPU oracle on the SAME pipeline -- front end per frame, frame-to-frame stage per pair, the map
    bookkeeping of oracle/windows.c, the BA schedule on every built window -- one thread in-process on the first n_single keyframes of the
    batch (window b only depends on frames <= b, so these are exactly the GPU's first windows: parity is checked on the way), then all
    cores with the stages as parallel maps over a process pool (no halo recomputation: a stage's results travel through the parent)."""
    import multiprocessing as mp
    import tempfile
    import oracle as O
    from oracle import cpu_keyframe as W
    B, U, cap, n_kf = pipe.B, pipe.unique_frames, pipe.cap, pipe.n_kf
    tmp = tempfile.NamedTemporaryFile(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False)
    tmp.close()
    np.save(tmp.name, np.concatenate([pipe.h_imgs_unique_left, pipe.h_imgs_unique_right]))
    init_args = (tmp.name, pipe.w, anms_num, n_kf, 0, 0)
    try:
        W.init(*init_args)
        n1 = min(n_single, B)
        stage_s = {"orb_lr_match_triangulate": 0.0, "f2f_match_motion_only_lm": 0.0, "build_windows": 0.0, "local_ba_schedule": 0.0}
        t0 = time.perf_counter()
        fronts, tracks, int_mismatch, pnp_o, pnp_g = [], [], 0, [], []
        for b in range(n1):
            ts = time.perf_counter()
            fronts.append(W.front_end_full(pipe.frame_of[b]))
            stage_s["orb_lr_match_triangulate"] += time.perf_counter() - ts
            cur = fronts[-1]
            int_mismatch += int(out["cnt"][b] != len(cur[0])) + int(out["cnt"][B + b] != cur[5]) + int(out["nlr"][b] != len(cur[2]))
            if b > 0:
                ts = time.perf_counter()
                tracks.append(W.track_full(fronts[b - 1], cur))
                stage_s["f2f_match_motion_only_lm"] += time.perf_counter() - ts
                T, f, il = tracks[-1]
                int_mismatch += int(out["nf2f"][b - 1] != len(f)) + int(out["pn"][b - 1] != len(il)) + int(out["ninl"][b - 1] != int(il.sum()))
                if len(il) >= 6:
                    pnp_o.append(T); pnp_g.append(out["Tpnp"][b - 1])
        ts = time.perf_counter()
        w = O.build_windows(*W.pack_tracks(fronts, tracks, cap), n_kf=n_kf)
        stage_s["build_windows"] += time.perf_counter() - ts
        int_mismatch += int((w["lm_off"][:n1 + 1] != out["ba_lm_off"][:n1 + 1]).sum()) + int((w["edge_off"][:n1 + 1] != out["ba_e_off"][:n1 + 1]).sum())
        ba_o, ba_g = [], []
        for b in range(n1):
            ts = time.perf_counter()
            Tb, inl = W.ba_schedule_built(W.window_slice(w, b))
            stage_s["local_ba_schedule"] += time.perf_counter() - ts
            nk = int(w["n_kf"][b])
            ba_o.append(Tb); ba_g.append(out["ba_T"][b][:nk])
            lo, hi = int(out["ba_lm_off"][b]), int(out["ba_lm_off"][b + 1])
            int_mismatch += int((out["ba_inl"][lo:hi] != inl).sum()) if hi - lo == len(inl) else len(inl)
        dt1 = time.perf_counter() - t0
        pt, pq, pr = _pose_diff(np.array(pnp_g), np.array(pnp_o)) if pnp_o else (None, None, None)
        bt, bq, br = _pose_diff(np.concatenate(ba_g), np.concatenate(ba_o))
        parity = dict(keyframes_checked=n1, integer_mismatches=int_mismatch, pnp_translation_rmse_m=pt, pnp_quaternion_rmse=pq, pnp_max_rel_diff=pr,
                      ba_translation_rmse_m=bt, ba_quaternion_rmse=bq, ba_max_rel_diff=br,
                      note="GPU step vs oracle on the same inputs: counts of keypoints / LR / f2f matches / pose inputs / pose inliers, the landmark and "
                           "edge offsets of the built windows and the BA landmark flags must be identical (integer_mismatches = 0); poses: motion-only LM "
                           "of keyframe b-1 -> b, and every window's poses after the 5+5+10+10 schedule")
        cores = usable_cores()
        all_cores = None
        if cores > 1:
            n_all = int(min(max(4 * cores, 64), 1024))
            period = max(2 * (U - 1), 1)
            frames = [(t if t < U else period - t) for t in (b % period for b in range(n_all))]
            ctx = mp.get_context("spawn")
            nproc = min(cores, n_all)
            with ctx.Pool(nproc, initializer=W.init, initargs=init_args) as pool:
                pool.map(W.warm, range(4 * nproc), chunksize=1)
                t0 = time.perf_counter()
                fr = pool.map(W.front_end_full, frames, chunksize=1)
                tr = pool.map(W.track_pair, [(fr[i], fr[i + 1]) for i in range(n_all - 1)], chunksize=1)
                wa = O.build_windows(*W.pack_tracks(fr, tr, cap), n_kf=n_kf)
                pool.map(W.ba_schedule_built, [W.window_slice(wa, b) for b in range(n_all)], chunksize=1)
                dta = time.perf_counter() - t0
            all_cores = dict(value=n_all / dta, cores=nproc, keyframes=n_all, seconds=round(dta, 2))
    finally:
        os.unlink(tmp.name)
    res = dict(unit="keyframes/s", kind="port",
               single_thread=dict(value=n1 / dt1, cores=1, keyframes=n1, seconds=round(dt1, 2),
                                  stage_ms_per_keyframe={k: round(1e3 * v / n1, 2) for k, v in stage_s.items()}),
               sample=("oracle/libvo_oracle.so on the same pipeline as the GPU step: per stereo keyframe 2 ORB images (3000 -> ANMS %d -> rBRIEF), L/R + "
                       "frame-to-frame match, DLT, motion-only LM; the map bookkeeping (oracle/windows.c) and the BA schedule 5+5+10+10 on the window built "
                       "for every keyframe (up to %d keyframes); single thread: %d keyframes in %.1f s" % (anms_num, n_kf, n1, dt1)))
    if all_cores:
        res.update(value=all_cores["value"], cores=all_cores["cores"])
        res["sample"] += "; all cores: %d keyframes in %.1f s, the stages as parallel maps over %d worker processes; host has %d cores" % (
            all_cores["keyframes"], all_cores["seconds"], all_cores["cores"], os.cpu_count())
    else:
        res.update(value=n1 / dt1, cores=1)
    return res, parity


def ba_config4_measure(args, local, torch):
    """the BA schedule alone on B canned windows of the BASELINE config-4 shape (10 keyframes x 3000 landmarks, ~10 k edges: SURVEY 8d) -- the
    shape lm_window_kernel's roofline figures have been quoted on since round 1; the default step's windows come from real tracks and are smaller"""
    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    B = 256   # (one window per CU: the shape and batch every earlier round quoted, whatever --batch is)
    pipe = KeyframePipeline(B, device=local, anms_num=500, n_lm=3000, unique_frames=2, seed=0, ba_windows="synthetic")
    try:
        for _ in range(2):
            pipe.stage_ba()
        torch.cuda.synchronize(pipe.dev)
        pipe.vo.profile_enable(True); pipe.vo.profile_read()
        n = max(args.steps, 3)
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.stage_ba()
        torch.cuda.synchronize(pipe.dev)
        wall = (time.perf_counter() - t0) / n
        prof = pipe.vo.profile_read(); pipe.vo.profile_enable(False)
        if int((pipe.vo.ba_status(B) != 0).sum()):
            return {"error": "windows rejected"}
        pipe.ba_passes = pipe.vo.ba_schedule_passes(B)
        nd = int(pipe.vo.ba_deferred(B).sum())   # windows ba_resident_kernel left to lm_window_kernel (config 4's 3.5 observations per landmark: all of them)
        ran = "lm_window_kernel+pose_only_wave_kernel" if nd == B else ("ba_resident_kernel+pose_only_wave_kernel" if nd == 0 else "ba_resident_kernel+lm_window_kernel+pose_only_wave_kernel")
        pipe.vo.set_tuning(ba_adaptive=0)   # ... and every pass for every window, three launches (the schedule of rounds 1-3)
        pipe.stage_ba(); pipe.vo.profile_read()
        pipe.vo.profile_enable(True)
        for _ in range(n):
            pipe.stage_ba()
        prof_plain = pipe.vo.profile_read(); pipe.vo.profile_enable(False)
        pipe.vo.set_tuning(ba_adaptive=-1)
        ms = prof["lm_window_kernel"][0] / n
        alg, formula = algorithmic_bytes("lm_window_kernel", pipe, 500)
        achieved = alg / (ms / 1e3) / 1e9
        tj, src = load_counter_json("traffic.json")
        traffic = int(tj["hbm_bytes_per_launch_set"]) if tj and tj.get("batch") == B and tj.get("kernel") == "lm_window_kernel" else None
        res = {"workload": "BA schedule (5+5+10 LM + 10 pose-only) on %d unique synthetic windows, 10 KF x 3000 landmarks x %.0f edges" % (B, pipe.edges_per_window),
               "ms_per_schedule_batch": round(ms, 4), "wall_ms_per_schedule_batch": round(1e3 * wall, 4), "windows_per_s": round(B / (ms / 1e3), 1),
               "schedule": schedule_stats(pipe.ba_passes), "plain_schedule_ms_per_batch": round(prof_plain["lm_window_kernel"][0] / n, 4),
               "roofline": {"bound": "hbm", "kernel": ran, "stage_family": "lm_window_kernel", "windows_left_to_lm_window_kernel": nd,
                            "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": src,
                            "algorithmic_bytes_per_launch_set": int(alg), "formula": formula},
               "other_rooflines": [r for r in other_rooflines(prof, pipe, args, n, 0.0) if r["kernel"] == "lm_window_kernel"]}
        return res
    finally:
        pipe.close()


def inflight_overlap(ring, args, timed_region, one_step):
    """`steps` steps with every pipeline in flight and every pipeline's stage profiler on: the hipEvent brackets of all of them on one time axis.
    overlap_share = part of the busy span with >= 2 kernel families running; sum_kernel_ms_over_span_ms = their durations added up over the span
    (1.0 = nothing overlaps); per family: its share of the added-up durations."""
    for p_ in ring.pipes:
        p_.vo.profile_enable(True); p_.vo.profile_read()
    el = timed_region(one_step)
    iv = []
    for i, p_ in enumerate(ring.pipes):
        iv += [(n, a, b, i) for n, a, b in p_.vo.profile_intervals() if b > a]
        p_.vo.profile_enable(False)
    if not iv:
        return None
    t0 = min(a for _, a, _, _ in iv); t1 = max(b for _, _, b, _ in iv)
    ev = sorted([(a, 1) for _, a, _, _ in iv] + [(b, -1) for _, _, b, _ in iv])
    busy = over = 0.0; depth = 0; last = t0
    for t, d in ev:
        if depth >= 1: busy += t - last
        if depth >= 2: over += t - last
        depth += d; last = t
    total = sum(b - a for _, a, b, _ in iv)
    fam = {}
    for n, a, b, _ in iv:
        fam[n] = fam.get(n, 0.0) + (b - a)
    return {"steps": args.steps, "pipelines": len(ring.pipes), "span_ms": round(t1 - t0, 3), "busy_ms": round(busy, 3), "ms_per_step_with_brackets": round(1e3 * el / args.steps, 4),
            "overlap_share": round(over / busy, 4) if busy > 0 else None, "sum_kernel_ms_over_span_ms": round(total / (t1 - t0), 4) if t1 > t0 else None,
            "family_ms_per_step_in_flight": {k_: round(v / args.steps, 4) for k_, v in sorted(fam.items(), key=lambda kv: -kv[1])},
            "how": "vslam_profile_intervals of every pipeline (hipEvent brackets around each kernel family on its stream, one time origin per process); a bracket "
                   "covers a family's launches back to back, so gaps inside a family count as busy"}


def config4_step_measure(args, local, torch, seqs):
    """the SAME step (front end on the rendered frames, pose stage) with the BA schedule on canned windows of the BASELINE config-4 shape (10 keyframes x
    3000 landmarks, ~10.5 k edges) instead of the windows the step's own tracks give (3.5 k landmarks, 4.6 k edges): keyframes/s at the config-4 BA shape"""
    from stereo_visual_slam_amd.pipeline import PipelineRing
    n_flight = max(1, args.in_flight)
    ring = PipelineRing(n_flight, args.batch, sequences=seqs[:n_flight], device=local, anms_num=args.anms, n_lm=3000, unique_frames=len(seqs[0]), ba_windows="synthetic",
                        unique_windows=min(args.batch, 256))
    pipe = ring.pipes[0]
    try:
        for _ in range(2 * n_flight):
            ring.step()
        torch.cuda.synchronize(pipe.dev)
        n = max(args.steps // 2, 4)
        t0 = time.perf_counter()
        for _ in range(n):
            ring.step()
        torch.cuda.synchronize(pipe.dev)
        el = time.perf_counter() - t0
        bad = sum(int((p_.vo.orb_status(2 * p_.B) != 0).sum()) + int((p_.vo.ba_status(p_.B) != 0).sum()) for p_ in ring.pipes)
        if bad:
            return {"error": "status words non-zero (%d)" % bad}
        return {"value": round(args.batch * n / el, 2), "unit": "keyframes/s", "ms_per_step": round(1e3 * el / n, 4), "steps": n, "batch": args.batch, "batches_in_flight": n_flight,
                "workload": "the default step with its BA schedule run on canned synthetic windows of the config-4 shape (10 KF x 3000 landmarks x %.0f edges, %d unique windows) "
                            "instead of the windows built from the step's tracks" % (pipe.edges_per_window, pipe.unique_windows)}
    finally:
        ring.close()


def live_dropin_measure(n_frames=50, anms=500, with_cpu=True, cpu_frames=12):
    """The drop-in itself: the C++ host mirror of the reference's node loop (host/run_vslam: VO::pipeline per frame, the BA schedule per keyframe, run_vslam.cpp:40-82) on 50
    rendered stereo pairs read from PNG files, reference-faithful configuration (ANMS 500, SGBM depth, solvePnPRansac pose, Q1 quirk on) -- one frame at a time through the
    host-buffer tier of the C-ABI, i.e. live-SLAM latency, not batch throughput.  Beside it (part of the cpu_baseline leg) the same host code on the CPU oracle
    (oracle/run_vslam_cpu) over the first `cpu_frames` pairs."""
    import subprocess
    import tempfile
    from stereo_visual_slam_amd import synth
    gpu_exe = os.path.join(ROOT, "stereo-visual-slam_amd", "host", "run_vslam"); cpu_exe = os.path.join(ROOT, "oracle", "run_vslam_cpu")
    if not os.path.exists(gpu_exe):
        return {"error": "host/run_vslam not built (python -c 'import __graft_entry__ as g; g.build()')"}

    def run(exe, d, n):
        t0 = time.perf_counter()
        r = subprocess.run([exe, d + "/", str(n), "0", str(anms), os.path.join(d, "traj.txt"), "1", "1", "1"], capture_output=True, text=True, timeout=1500)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stdout[-200:] + r.stderr[-200:]).strip()}
        line = [l for l in r.stdout.splitlines() if l.startswith("timing ")]
        summ = [l for l in r.stdout.splitlines() if l.startswith("frames ")]
        out = {"process_wall_s": round(wall, 2), "summary": summ[-1] if summ else None}
        if line:
            t = line[-1].split()
            kv = {t[i]: float(t[i + 1]) for i in range(1, len(t) - 1, 2)}
            out.update(loop_s=kv.get("loop_s"), frames=int(kv.get("frames", 0)), frames_per_s=kv.get("frames_per_s"), keyframes_per_s=kv.get("keyframes_per_s"),
                       s_per_keyframe=kv.get("s_per_keyframe"))
        return out

    with tempfile.TemporaryDirectory() as d:
        synth.write_pgm_sequence(d + "/", n_frames, seed=5, fmt="png")
        res = {"frames": n_frames, "config": "ANMS %d, SGBM depth + find_3d, solvePnPRansac(100, 4.0, 0.99), BA schedule 5+5+10+10 per keyframe on the 10-keyframe map, "
                                             "Q1 quirk on; PNG pairs read from disk per frame; the frame loop only (context creation excluded)" % anms,
               "reference_figure": "the reference's README.md:90 quotes 0.18 s per keyframe on its authors' CPU (not reproducible here: OpenCV / g2o absent)",
               "gpu": run(gpu_exe, d, n_frames)}
        if with_cpu and os.path.exists(cpu_exe):
            res["cpu"] = run(cpu_exe, d, min(cpu_frames, n_frames))
            res["cpu"]["note"] = "oracle/run_vslam_cpu: the same host code over the CPU oracle behind the same C-ABI, first %d pairs, one thread" % min(cpu_frames, n_frames)
        return res


def kitti_measure(n_frames=50, anms=500):
    """Real-data hook: when KITTI_ROOT names a KITTI odometry tree on this box ($KITTI_ROOT/sequences/00/image_0/%06d.png, or the
    sequence directory itself), the first `n_frames` pairs of sequence 00 go through the drop-in host/run_vslam exactly as visual_odometry.cpp:37-68
    reads them, and the per-frame trace is condensed to detection / match / inlier statistics.  No KITTI on the box: {"available": False}."""
    import subprocess
    import tempfile
    root = os.environ.get("KITTI_ROOT")
    if not root:
        return {"available": False, "reason": "KITTI_ROOT not set"}
    cands = [os.path.join(root, "sequences", "00"), os.path.join(root, "dataset", "sequences", "00"), os.path.join(root, "00"), root]
    seq = next((c for c in cands if os.path.exists(os.path.join(c, "image_0", "000000.png")) and os.path.exists(os.path.join(c, "image_1", "000000.png"))), None)
    if seq is None:
        return {"available": False, "reason": "no image_0/000000.png + image_1/000000.png under KITTI_ROOT=%s (tried sequences/00, dataset/sequences/00, 00, .)" % root}
    exe = os.path.join(ROOT, "stereo-visual-slam_amd", "host", "run_vslam")
    if not os.path.exists(exe):
        return {"available": True, "error": "host/run_vslam not built"}
    n = 0
    while n < n_frames and os.path.exists(os.path.join(seq, "image_0", "%06d.png" % n)):
        n += 1
    with tempfile.TemporaryDirectory() as d:
        trace = os.path.join(d, "trace.txt")
        r = subprocess.run([exe, seq + "/", str(n), "0", str(anms), os.path.join(d, "traj.txt"), "1", "1", "1", trace], capture_output=True, text=True, timeout=1500)
        if r.returncode != 0:
            return {"available": True, "sequence_dir": seq, "error": (r.stdout[-300:] + r.stderr[-300:]).strip()}
        det, mat, inl, lms = [], [], [], []
        for line in open(trace):
            t = line.split()
            if t and t[0] == "frame":
                kv = {t[i]: t[i + 1] for i in range(0, min(len(t) - 1, 18), 2) if t[i] in ("det", "matches", "inliers", "landmarks")}
                det.append(int(kv.get("det", 0))); mat.append(int(kv.get("matches", 0))); inl.append(int(kv.get("inliers", 0))); lms.append(int(kv.get("landmarks", 0)))
        out = {"available": True, "sequence_dir": seq, "frames": n, "config": "ANMS %d, SGBM depth, solvePnPRansac pose, BA schedule per keyframe (the reference's configuration)" % anms,
               "summary": [l for l in r.stdout.splitlines() if l.startswith("frames ")][-1:], "final_position": [l for l in r.stdout.splitlines() if l.startswith("final_position")][-1:],
               "timing": [l for l in r.stdout.splitlines() if l.startswith("timing ")][-1:]}
        if det:
            out.update(keypoints_per_frame_mean=float(np.mean(det)), keypoints_per_frame_min=int(min(det)), f2f_matches_mean=float(np.mean(mat[1:] or [0])),
                       pnp_inliers_mean=float(np.mean(inl[1:] or [0])), pnp_inliers_min=int(min(inl[1:] or [0])), landmarks_in_map_last=int(lms[-1]))
        return out


def reference_pipeline_measure(args, local, torch, seq, B=256):
    """The REFERENCE's own stages in throughput mode, measured in the default run so that the driver records it: depth from StereoSGBM +
    Frame::find_3d on the left keypoints (visual_odometry.cpp:159-217) instead of L/R match + DLT, pose from cv::solvePnPRansac(..., 100, 4.0,
    0.99) (:277) instead of the motion-only LM, then the same device-built windows and BA schedule.  B keyframes per step (256: SGBM is at 0.14 ms per
    pair there against 0.17 at 64, and the BA has one window per CU), a few steps."""
    from stereo_visual_slam_amd.pipeline import PipelineRing
    B = min(B, len(seq))
    n_flight = max(1, args.in_flight)
    ring = PipelineRing(n_flight, B, device=local, anms_num=args.anms, unique_frames=B, sequence=seq[:B], depth="sgbm", pose="ransac", ba_windows="tracks")
    pipe = ring.pipes[0]
    try:
        for _ in range(2 * n_flight):
            ring.step()
        torch.cuda.synchronize(pipe.dev)
        n = max(args.steps // 2, 3)
        t0 = time.perf_counter()
        for _ in range(n):
            ring.step()
        torch.cuda.synchronize(pipe.dev)
        el = time.perf_counter() - t0
        el1 = None
        if n_flight > 1:   # one batch in flight (the figure of the earlier rounds)
            t0 = time.perf_counter()
            for _ in range(n):
                pipe.step()
            torch.cuda.synchronize(pipe.dev)
            el1 = time.perf_counter() - t0
        pipe.vo.profile_enable(True); pipe.vo.profile_read()
        for _ in range(n):
            pipe.step()
        prof = pipe.vo.profile_read(); pipe.vo.profile_enable(False)
        bad = sum(int((p_.vo.orb_status(B) != 0).sum()) + int((p_.vo.ba_status(B) != 0).sum()) + int(p_.ba_build_status.item()) + int(p_.vo.sgbm_status() != 0)
                  for p_ in ring.pipes)
        if bad:
            return {"error": "status words non-zero (%d)" % bad}
        out = pipe.download()
        kern = sorted(prof.items(), key=lambda kv: -kv[1][0])
        fam_ms = sum(v[0] for k, v in kern if k.startswith("sgbm_")) / n
        cv = (pipe.w - 96) * pipe.h * 96 * 2
        alg = B * (cv + 2 * pipe.w * pipe.h + 4 * pipe.w * pipe.h)
        gbs = alg / (fam_ms / 1e3) / 1e9 if fam_ms > 0 else 0.0
        tj, src = load_counter_json("traffic_sgbm.json")
        nt = max(B - 1, 1)
        return {"workload": "reference stages: ORB(3000)->ANMS(%d)->rBRIEF on the left image, StereoSGBM(0,96,9,648,2592,1,63,10,100,32) + find_3d depth, "
                            "frame-to-frame match, solvePnPRansac(100, 4.0, 0.99) pose (EPnP hypotheses, OpenCV 3.2.0 return value), device-built BA windows, "
                            "BA schedule 5+5+10+10; %d keyframes per step" % (args.anms, B),
                "value": round(B * n / el, 2), "unit": "keyframes/s", "ms_per_step": round(1e3 * el / n, 4), "steps": n, "batch": B,
                "batches_in_flight": n_flight,
                "one_batch_in_flight": None if el1 is None else {"value": round(B * n / el1, 2), "ms_per_step": round(1e3 * el1 / n, 4)},
                "kernels_ms_per_step": {k: round(v[0] / n, 4) for k, v in kern[:14]},
                "kernels_note": "per-kernel durations and the roofline: HIP-event brackets in a repeat with ONE batch in flight",
                "ba_schedule": schedule_stats(pipe.vo.ba_schedule_passes(B)),
                "roofline": {"bound": "hbm", "kernel": "sgbm_* (family)", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 5),
                             "ms_per_pair": round(fam_ms / B, 4), "algorithmic_bytes_per_launch_set": int(alg),
                             "formula": "B pairs x ((w-96)*h*96*2 B cost volume once + 2*w*h B images in + 4*w*h B f32 disparity out)",
                             "traffic": int(tj["hbm_bytes_per_launch_set"]) if tj and tj.get("batch") == B else None, "traffic_source": src},
                "stats": {"keypoints_per_image": float(out["cnt"][:B].mean()), "valid_depth_per_image": float(out["valid"].sum(1).mean()),
                          "f2f_matches": float(out["nf2f"][:nt].mean()), "pose_inputs": float(out["pn"][:nt].mean()), "ransac_inliers": float(out["ninl"][:nt].mean()),
                          "landmarks_per_window_mean": float(np.diff(out["ba_lm_off"]).mean()), "edges_per_window_mean": float(np.diff(out["ba_e_off"]).mean())}}
    finally:
        ring.close()


def host_input_region(ring, args, timed_region, torch):
    """--inputs host: the same steps with every step's 2B images arriving from pinned host memory: every pipeline owns a ring of two
    device batches, the hipMemcpyAsync of the pipeline's NEXT step goes onto a copy stream as soon as its current step is queued and is
    overlapped with the steps in flight (the reference reads its pairs from disk per frame, visual_odometry.cpp:37-68).
    Returns keyframes/s with the upload inside the timed region, and the H2D time per step alone."""
    pipe0 = ring.pipes[0]
    dev = pipe0.dev
    copy_stream = torch.cuda.Stream(dev)
    st = []
    for pipe in ring.pipes:
        st.append({"pipe": pipe, "k": 0,
                   "h": [torch.from_numpy(pipe.h_imgs).pin_memory(), torch.from_numpy(pipe.h_imgs.copy()).pin_memory()],
                   "d": [pipe.d_imgs, torch.empty_like(pipe.d_imgs)],
                   "copied": [torch.cuda.Event(), torch.cuda.Event()], "consumed": [torch.cuda.Event(), torch.cuda.Event()]})
    # H2D alone: one batch, synchronous bracket
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(copy_stream):
        st[0]["d"][1].copy_(st[0]["h"][1], non_blocking=True)
        e0.record(copy_stream)
        for _ in range(3):
            st[0]["d"][1].copy_(st[0]["h"][1], non_blocking=True)
        e1.record(copy_stream)
    torch.cuda.synchronize(dev)
    h2d_ms = e0.elapsed_time(e1) / 3.0
    nbytes = pipe0.h_imgs.nbytes
    for q in st:
        with torch.cuda.stream(copy_stream):       # prologue: every pipeline's first batch in flight
            q["d"][0].copy_(q["h"][0], non_blocking=True)
            q["copied"][0].record(copy_stream)
        for ev in q["consumed"]:
            ev.record(q["pipe"].stream)
    state = {"k": 0}

    def step_from_host():
        q = st[state["k"] % len(st)]
        pipe = q["pipe"]
        cur = q["k"] & 1
        nxt = cur ^ 1
        with torch.cuda.stream(copy_stream):   # upload of this pipeline's NEXT batch, once the step that last read that buffer has finished its ORB stage
            copy_stream.wait_event(q["consumed"][nxt])
            q["d"][nxt].copy_(q["h"][nxt], non_blocking=True)
            q["copied"][nxt].record(copy_stream)
        pipe.stream.wait_event(q["copied"][cur])
        pipe.d_imgs = q["d"][cur]
        pipe.stage_orb()
        if pipe.depth == "sgbm":
            pipe.stage_stereo_match(); q["consumed"][cur].record(pipe.stream)
        else:
            q["consumed"][cur].record(pipe.stream); pipe.stage_stereo_match()
        pipe.stage_track()
        pipe.stage_ba()
        q["k"] += 1
        state["k"] += 1

    for _ in range(2 * len(st)):
        step_from_host()
    el = timed_region(step_from_host)
    torch.cuda.synchronize(dev)
    for q in st:
        q["pipe"].d_imgs = q["d"][0]
    return {"value": round(pipe0.B * args.steps / el, 3), "unit": "keyframes/s", "ms_per_step": round(1e3 * el / args.steps, 4),
            "h2d_ms_per_step": round(h2d_ms, 4), "h2d_gbs": round(nbytes / (h2d_ms * 1e-3) / 1e9, 2), "h2d_bytes_per_step": int(nbytes),
            "batches_in_flight": len(st),
            "how": "pinned host ring of 2 batches per pipeline; hipMemcpyAsync of a pipeline's next batch on a copy stream overlapped with the steps in flight; "
                   "the upload is inside the timed region"}


def _rccl_version(torch):
    try:
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        return "unknown (%r)" % (e,)


def units_per_step(args, seq_mode, world, B):
    return args.sequence if seq_mode else world * B


def host_info():
    """lscpu-style identification of the host the cpu_baseline ran on + run-time probe for the reference's own libraries"""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except Exception:
        pass
    info = {"cpu_model": model, "logical_cpus": os.cpu_count()}
    try:
        info["cpus_available_to_this_process"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # the real cv::ORB / BFMatcher / g2o path, if and only if the libraries exist on this host
    import ctypes.util
    import importlib.util
    have_cv2 = importlib.util.find_spec("cv2") is not None
    g2o = ctypes.util.find_library("g2o_core")
    ocv = ctypes.util.find_library("opencv_features2d")
    info["reference_libs"] = {"cv2_python": have_cv2, "libopencv_features2d": ocv, "libg2o_core": g2o}
    return info, have_cv2


def reference_lib_timing(pipe, anms_num, n=8):
    """real cv::ORB(3000) detect + compute and BFMatcher(HAMMING, crossCheck) through cv2 -- only if cv2 exists on this host"""
    import cv2
    orb = cv2.ORB_create(3000); bf = cv2.BFMatcher(cv2.NORM_HAMMING, True)
    t0 = time.perf_counter()
    for u in range(n):
        L = np.ascontiguousarray(pipe.h_imgs_unique_left[u % pipe.unique_frames][:, :pipe.w]); R = np.ascontiguousarray(pipe.h_imgs_unique_right[u % pipe.unique_frames][:, :pipe.w])
        kL, dL = orb.detectAndCompute(L, None); kR, dR = orb.detectAndCompute(R, None)
        bf.match(dL, dR)
    dt = time.perf_counter() - t0
    return {"opencv_version": cv2.__version__, "orb_lr_match_ms_per_keyframe": round(1e3 * dt / n, 2), "keyframes": n,
            "note": "cv2.ORB_create(3000).detectAndCompute on L and R + BFMatcher(NORM_HAMMING, crossCheck=True); no ANMS (reference C++ only), no g2o"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=3, help="the K-step timed region is repeated this many times; the median is reported")
    ap.add_argument("--batch", type=int, default=1024, help="stereo keyframes per GPU per step.  Round 5: 1024 (two rounds of two BA windows per CU: the windows of a round finish at "
                                                             "different times and the next round fills the tail): 56.2 k keyframes/s against 53.1 k at 512, 55.7 k at 768, 56.6 k at 2048 with two "
                                                             "batches in flight (profiles/r05_experiments.log); rounds 1-4 ran 256 / 512")
    ap.add_argument("--in-flight", type=int, default=2, metavar="P",
                    help="batches in flight per GPU: P pipelines (context + HIP stream + buffers each), step k goes to pipeline k mod P without waiting for "
                         "step k - 1 (pipeline.PipelineRing): 41.6 k keyframes/s at 2 x 512 against 38.6 k at 1 x 512; per-kernel durations and the roofline "
                         "always come from a repeat with ONE batch in flight (kernels of two batches sharing the chip have no duration of their own)")
    ap.add_argument("--anms", type=int, default=1500, help="keypoints per image after ANMS (BASELINE config 2: ~1500; reference: 500)")
    ap.add_argument("--landmarks", type=int, default=3000)
    ap.add_argument("--unique-frames", type=int, default=0, help="rendered stereo keyframes per pipeline (one sequence, laid over the batch as a ping-pong); 0 = the batch size: "
                                                                  "every keyframe of a batch is a different rendered frame, and every pipeline in flight renders its own sequence")
    ap.add_argument("--no-config4-step", action="store_true", help="skip the extra timed steps with canned config-4-shaped BA windows (`value_config4_windows`)")
    ap.add_argument("--no-live-dropin", action="store_true", help="skip the C++ host driver run (host/run_vslam on 50 rendered pairs: the drop-in's own frames/s)")
    ap.add_argument("--inputs", choices=["resident", "host"], default="host",
                    help="host: additionally time the same steps with the images arriving from pinned host memory (ring of 2 batches, "
                         "hipMemcpyAsync on a copy stream overlapped with the previous step) and report it under `inputs_from_host`; "
                         "the headline `value` is always the HBM-resident one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--depth", choices=["match", "sgbm"], default="match",
                    help="stereo depth stage: north_star L/R match + DLT (default, BASELINE metric) or the reference's SGBM + find_3d")
    ap.add_argument("--ba-windows", choices=["tracks", "synthetic"], default="tracks",
                    help="tracks (default): the BA windows are built on the device from the step's own matches and poses (one pipeline: window b = "
                         "keyframes [b-9, b] of the batch); synthetic: B canned windows of the BASELINE config-4 shape (10 KF x --landmarks)")
    ap.add_argument("--ba-plain-schedule", action="store_true",
                    help="run all three optimize_map passes for every window (three launches) instead of continuing a pass that flags nothing new "
                         "(vslam_set_tuning ba_adaptive = 0); the default run reports this figure beside the headline")
    ap.add_argument("--no-plain-schedule", action="store_true", help="skip the extra timed region with the plain BA schedule (profiling runs: it would mix its launches into the counters)")
    ap.add_argument("--no-config4", action="store_true", help="skip the extra BA-only measurement on the config-4 shape")
    ap.add_argument("--pose", choices=["lm", "ransac"], default="lm",
                    help="pose stage: north_star motion-only LM (default, BASELINE metric) or the reference's solvePnPRansac(100, 4.0, 0.99), batched on the device")
    ap.add_argument("--no-reference-pipeline", action="store_true", help="skip the extra measurement of the reference's own stages (SGBM depth + RANSAC pose)")
    ap.add_argument("--sequence", type=int, default=0, metavar="F",
                    help="BASELINE config 5: one F-frame sequence split into contiguous chunks with a 1-frame halo across the ranks, relative poses "
                         "gathered over RCCL and chained on rank 0 (stereo-visual-slam_amd/sharding.py); 0 = independent batches per rank")
    ap.add_argument("--render-workers", type=int, default=-1, help="processes that render the synthetic frames (-1: one per available core up to 32; 0: in-process, "
                                                                     "e.g. under rocprofv3, which would otherwise attach to every worker)")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")

    import torch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become N ranks (one process per GPU over RCCL) instead of silently timing one GPU
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible; refusing to run (no single-GPU stand-in for an N-GPU line)" % (args.gpus, n_dev))
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE %d (launch with --nproc-per-node %d, or drop the launcher and let "
                         "bench.py start the ranks itself)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")
    if torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        assert dist.get_world_size() == args.gpus
    host_cores = usable_cores()   # (affinity cut by the cgroup quota: a box that shows 256 CPUs may grant 16)
    render_workers = max(1, min(host_cores // max(world, 1), 32)) if args.render_workers < 0 else args.render_workers

    from stereo_visual_slam_amd.pipeline import KeyframePipeline, PipelineRing
    from stereo_visual_slam_amd import sharding
    B = args.batch
    seq_mode = args.sequence > 0
    if args.unique_frames <= 0:
        args.unique_frames = args.sequence if seq_mode else B
    n_flight = 1 if seq_mode else max(1, args.in_flight)   # (sequence mode: a step is one pass over THE sequence; kept at one pass in flight)
    ba_windows = args.ba_windows  # (the BA results are not fed back into the gathered trajectory in either mode)
    if seq_mode:  # config 5: this rank's contiguous chunk of ONE sequence, plus its halo: the frame before it for the pose stage, nine frames when the BA
                  # windows are built from the tracks (window b = keyframes [b - 9, b]: with them, and the tracks carried across the chunk boundary, a
                  # rank's windows are the unsharded run's windows bit for bit -- sharding.sequence_windows_and_ba)
        assert args.sequence >= 2 * world, "--sequence needs at least two frames per rank"
        lo, hi = sharding.shard_range(args.sequence, rank, world)
        seq_exact_ba = (not args.no_ba) and ba_windows == "tracks"
        h_lo = sharding.halo_start(lo, sharding.WINDOW_HALO if seq_exact_ba else 1)
        B = hi - h_lo
        ring = PipelineRing(1, B, device=local, anms_num=args.anms, n_lm=args.landmarks, seed=0, verbose=args.verbose and rank == 0,
                            with_ba=not args.no_ba, depth=args.depth, unique_frames=args.unique_frames, frame_range=(h_lo, hi, args.sequence),
                            render_workers=render_workers, ba_windows=ba_windows, pose=args.pose)
    else:
        ring = PipelineRing(n_flight, B, distinct=True, device=local, anms_num=args.anms, n_lm=args.landmarks, seed=1000 * rank, verbose=args.verbose and rank == 0,
                            with_ba=not args.no_ba, depth=args.depth, unique_frames=args.unique_frames, render_workers=render_workers,
                            ba_windows=ba_windows, pose=args.pose)
    pipe = ring.pipes[0]
    dev = pipe.dev
    chained = [None]
    if args.ba_plain_schedule:
        for p_ in ring.pipes:
            p_.vo.set_tuning(ba_adaptive=0)

    def one_step(serial=False):
        """one pass of the hot path over one batch; step k runs on pipeline k mod P (serial: always on pipeline 0, i.e. one batch in flight)"""
        pipe = ring.pipes[0] if serial else ring.next_pipe()
        if seq_mode:
            # front end + pose stage on the chunk; ragged gather of the OWNED relative poses (56 B each, every rank gets all of them); the trajectory is their
            # chain; then the windows of the owned frames -- poses in the sequence's world, tracks continued across the chunk start -- and their BA schedule
            pipe.stage_orb(); pipe.stage_stereo_match(); pipe.stage_track()
            with torch.cuda.stream(pipe.stream):
                p_lo, p_hi = sharding.owned_pose_range(args.sequence, rank, world)
                rel = pipe.d_Tpnp[p_lo - h_lo - 1: p_hi - h_lo - 1]   # item i of d_Tpnp = T_{h_lo+i+1, h_lo+i}
                rel_all = sharding.gather_relative_poses(rel, args.sequence, dist, world)
                if rank == 0:
                    chained[0] = sharding.chain_poses(rel_all)
            if seq_exact_ba:
                sharding.sequence_windows_and_ba(pipe, args.sequence, rank, world, dist, rel_all)
            else:
                pipe.stage_ba()
            return
        # throughput mode: the step on pipeline k mod P, then -- N > 1 -- the pose gather (RCCL over xGMI, 56 B per keyframe), ordered after the step on its stream
        sharding.throughput_step(ring, dist, world, serial)

    for p_ in ring.pipes[1:]:   # (every pipeline has run once before the W warmup steps: none of them meets the timed region cold)
        p_.step()
    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    copy_probe = pipe.vo.hbm_copy_probe_best(1 << 30, 5) if rank == 0 else {"gbs": 0.0}
    copy_gbs = copy_probe["gbs"]

    def timed_region(step_fn):
        """EXACTLY `steps` steps between barrier + synchronize brackets; max over ranks"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_fn()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    # the timed repeats run WITHOUT the stage profiler (its hipEvent brackets sit between the kernels) ...
    rep_s = [timed_region(one_step) for _ in range(max(args.repeats, 1))]
    one_step_serial = lambda: one_step(serial=True)
    serial_s = timed_region(one_step_serial) if n_flight > 1 else None   # the same steps with ONE batch in flight (the figure of the earlier rounds)
    plain_s = None
    if pipe.with_ba and not args.ba_plain_schedule and not args.no_plain_schedule:   # ... and with the plain BA schedule (every pass for every window: the schedule of the earlier rounds)
        for p_ in ring.pipes:
            p_.vo.set_tuning(ba_adaptive=0)
        one_step(); one_step()
        plain_s = timed_region(one_step)
        for p_ in ring.pipes:
            p_.vo.set_tuning(ba_adaptive=-1)
        one_step(); one_step()    # (the state every later measurement sees is the default schedule's again)
    # ... and one extra, untimed-for-the-headline repeat of the same `steps` steps WITH it, one batch in flight: per-kernel durations from HIP events
    # on the launch stream (with two batches in flight the kernels of both share the CUs and an event bracket measures the mix)
    pipe.vo.profile_enable(True)
    pipe.vo.profile_read()
    profiled_s = timed_region(one_step_serial)
    prof = pipe.vo.profile_read()
    pipe.vo.profile_enable(False)
    n_prof_steps = args.steps
    # ... and the attribution of the HEADLINE configuration: the same steps with all pipelines in flight and every pipeline's brackets on ONE time
    # axis (vslam_profile_intervals): how much of the busy span has two or more kernel families running, and the sum of their durations over the span
    overlap = None
    if n_flight > 1 and world == 1:
        try:
            overlap = inflight_overlap(ring, args, timed_region, one_step)
        except Exception as e:
            overlap = {"error": repr(e)}
    # the pipelines in flight are handed DIFFERENT frames; that sharing the chip changes nobody's bits is checked by stepping each pipeline once more
    # alone and comparing with what it produced in flight
    flight_same = None
    if n_flight > 1 and world == 1 and not seq_mode:
        keys = ["kps", "desc", "lr", "f2f", "Tpnp", "inl"] + (["ba_T", "ba_inl"] if pipe.with_ba else [])
        one_step(); one_step()
        torch.cuda.synchronize(dev)
        inflight_out = [p_.download() for p_ in ring.pipes]
        flight_same = True
        for p_, o1 in zip(ring.pipes, inflight_out):
            p_.step(); torch.cuda.synchronize(dev)
            o2 = p_.download()
            flight_same = flight_same and all(np.array_equal(o1[k_], o2[k_]) for k_ in keys)
            del o2
        del inflight_out
        if not flight_same:
            raise SystemExit("bench invalid: a pipeline produced different results in flight beside another one than alone, on the same inputs")
    host_inputs = None
    if args.inputs == "host" and not seq_mode and world == 1:
        try:
            host_inputs = host_input_region(ring, args, timed_region, torch)
        except Exception as e:  # the extra measurement must never cost the headline
            host_inputs = {"error": repr(e)}
    elapsed = float(np.median(rep_s))
    # a step that overflowed an ORB capacity or whose BA windows were rejected must not count as processed keyframes
    n_img = pipe.B if args.depth == "sgbm" else 2 * pipe.B
    orb_bad = sum(int((p_.vo.orb_status(n_img) != 0).sum()) for p_ in ring.pipes)
    n_ba_windows = (hi - lo) if (seq_mode and seq_exact_ba) else pipe.B     # (sequence mode: the BA ran on the windows of the owned frames)
    ba_bad = sum(int((p_.vo.ba_status(n_ba_windows) != 0).sum()) for p_ in ring.pipes) if pipe.with_ba else 0
    build_bad = sum(int(p_.ba_build_status.item()) for p_ in ring.pipes) if (pipe.with_ba and pipe.ba_windows == "tracks") else 0
    if orb_bad or ba_bad or build_bad:
        raise SystemExit("bench invalid: %d images overflowed an ORB capacity, %d BA windows were rejected, window builder status %d" % (orb_bad, ba_bad, build_bad))

    if rank == 0:
        out = pipe.download()
        pipe.ba_passes = pipe.vo.ba_schedule_passes(n_ba_windows) if pipe.with_ba else None
        other_seqs = [p_.h_seq for p_ in ring.pipes[1:]]
        for p_ in ring.pipes[1:]:
            p_.close()
        pipe.ba_shape = None
        win_stats = None
        if pipe.with_ba and pipe.ba_windows == "tracks":
            nl_w, ne_w = np.diff(out["ba_lm_off"]), np.diff(out["ba_e_off"])
            pipe.ba_shape = (ne_w[pipe.B - n_ba_windows:], nl_w[pipe.B - n_ba_windows:], out["ba_nkf"][pipe.B - n_ba_windows:])
            n_l = int(out["ba_lm_off"][-1])
            win_stats = {"landmarks_per_window_mean": float(nl_w.mean()), "landmarks_per_window_max": int(nl_w.max()), "edges_per_window_mean": float(ne_w.mean()),
                         "edges_per_window_max": int(ne_w.max()), "reliable_fraction": float(out["ba_rel"][:n_l].mean()) if n_l else 0.0,
                         "keyframes_per_window": "1..%d for the first %d windows (the growing map), then %d" % (pipe.n_kf, pipe.n_kf - 1, pipe.n_kf),
                         "full_windows": int((out["ba_nkf"] == pipe.n_kf).sum()), "builder_status": build_bad,
                         "observations_per_landmark": float(ne_w.sum() / max(nl_w.sum(), 1))}
            # [r6] the track rule (vslam_build_windows_dev, Tuning::track_rule): the step runs the reference's -- a frame-to-frame match continues a track whenever
            # its last-frame keypoint is a feature (visual_odometry.cpp:568-599) -- and, outside the timed region, the same front-end results are built once
            # more under the convention of rounds 4-5 (only keypoints with a depth of their own pass a track on) for the before / after figures
            if not seq_mode:
                try:
                    pipe.vo.set_tuning(track_rule=0)
                    pipe.stage_build_windows(); pipe.vo.sync(); torch.cuda.synchronize(dev)
                    lo_ = pipe.ba_lm_off.cpu().numpy(); eo_ = pipe.ba_e_off.cpu().numpy()
                    pipe.vo.set_tuning(track_rule=-1)
                    pipe.stage_build_windows(); pipe.vo.sync(); torch.cuda.synchronize(dev)
                    win_stats["track_rule"] = {
                        "this_run": "1: the reference's tracking() rule (every feature of the last frame is a query; a feature without its own depth is judged by the pose "
                                    "stage's 4 px rule on its landmark's map position through the chained pose)",
                        "pose_inputs_per_frame": float(out["pn"][:max(B - 1, 1)].mean()),
                        "pose_inputs_note": "inputs of the pose stage = frame-to-frame matches whose last-frame keypoint owns a depth (triangulated in that frame); the same under both rules",
                        "observations_per_landmark": win_stats["observations_per_landmark"],
                        "round5_rule": {"landmarks_per_window_mean": float(np.diff(lo_).mean()), "edges_per_window_mean": float(np.diff(eo_).mean()),
                                        "observations_per_landmark": float(eo_[-1] / max(lo_[-1], 1))}}
                except Exception as e:
                    win_stats["track_rule"] = {"error": repr(e)}
        units = args.sequence if seq_mode else world * B   # keyframes all ranks processed per step (halo frames are not counted twice)
        value = units * args.steps / elapsed
        kern = sorted(prof.items(), key=lambda kv: -kv[1][0])
        dom, (dom_ms, dom_launches, dom_calls) = kern[0]
        alg, formula = algorithmic_bytes(dom, pipe, args.anms)
        alg_plain = None
        if dom == "lm_window_kernel" and getattr(pipe, "ba_passes", None) is not None:   # the same formula over the plain schedule's 30 linearisations per window
            keep = pipe.ba_passes; pipe.ba_passes = None
            alg_plain = algorithmic_bytes(dom, pipe, args.anms)[0]
            pipe.ba_passes = keep
        if args.depth == "sgbm" and any(k.startswith("sgbm_") for k, _ in kern):
            # --depth sgbm: the roofline line is the SGBM FAMILY (one bracket = one vslam_disparity_map_dev call = B pairs): compulsory bytes of a
            # fully fused design = the (w - 96) x h x 96 x i16 cost volume once + both images in + the f32 map out, per pair
            fam = [(k, v) for k, v in kern if k.startswith("sgbm_")]
            dom = "sgbm_* (family: %s)" % ", ".join(k for k, _ in fam)
            dom_ms = sum(v[0] for _, v in fam); dom_launches = sum(v[1] for _, v in fam); dom_calls = n_prof_steps
            cv = (pipe.w - 96) * pipe.h * 96 * 2
            alg = B * (cv + 2 * pipe.w * pipe.h + 4 * pipe.w * pipe.h)
            formula = "B pairs x ((w-96)*h*96*2 B cost volume once + 2*w*h B images in + 4*w*h B f32 disparity out)"
        per_bracket_s = dom_ms / 1e3 / max(dom_calls, 1)
        achieved = alg / per_bracket_s / 1e9 if per_bracket_s > 0 else 0.0
        # `traffic`: HBM bytes per launch set from rocprofv3 PMC passes made offline (tools/profile_round.sh).  Valid only for the same kernel
        # SOURCES (sha256 recorded in the file), the same kernel, batch and window kind; otherwise null with the reason in traffic_source.
        sg = args.depth == "sgbm" and dom.startswith("sgbm_*")
        tname = "traffic_sgbm.json" if sg else ("traffic_tracks.json" if (pipe.with_ba and pipe.ba_windows == "tracks") else "traffic.json")
        traffic = None
        tj, traffic_src = load_counter_json(tname)
        if tj is not None:
            if (sg and tj.get("batch") == B) or (tj.get("kernel") == dom and tj.get("batch") == B):
                traffic = int(tj["hbm_bytes_per_launch_set"])
            else:
                traffic_src = "profiles/%s was measured for kernel %s at batch %s, not %s at %d" % (tname, tj.get("kernel"), tj.get("batch"), dom, B)
        nt = max(B - 1, 1)
        # the roofline line is named after the kernels that ran inside the dominant stage bracket (the BA schedule's bracket keeps the family name
        # `lm_window_kernel` of rounds 1-4): vslam_ba_deferred_dev says which windows ba_resident_kernel took
        dom_kernels = dom; res_deferred = None
        if dom == "lm_window_kernel" and pipe.with_ba:
            try:
                nd = int(pipe.vo.ba_deferred(n_ba_windows).sum())
            except Exception:
                nd = n_ba_windows
            if nd == 0:
                dom_kernels = "ba_resident_kernel+pose_only_wave_kernel"
            elif nd == n_ba_windows:
                dom_kernels = "lm_window_kernel+pose_only_wave_kernel"
            else:
                dom_kernels = "ba_resident_kernel+lm_window_kernel+pose_only_wave_kernel"
            res_deferred = nd
        pose_txt = ("motion-only LM pose (10 its)" if args.pose == "lm" else
                    "solvePnPRansac(100, 4.0, 0.99) pose [the reference's own pose stage; not the BASELINE metric]")
        win_txt = ("BA windows built on the device from this step's own tracks (window b = keyframes [b-9, b]: poses = chained pose-stage estimates, "
                   "landmarks / observations as VO::insert_key_frame records them)" if pipe.ba_windows == "tracks" else
                   "%d canned synthetic windows of 10 KF x %d landmarks (config-4 shape; NOT fed by the front end of the step)" % (pipe.unique_windows, args.landmarks))
        workload = ("stereo keyframe hot path, north_star stages (NOT the reference's SGBM depth / solvePnPRansac pose, which are built and measured "
                    "under `reference_pipeline`; --depth sgbm / --pose ransac select them here): ORB(3000)->ANMS(%d)->rBRIEF on L+R 1241x376, L/R + "
                    "frame-to-frame BF-Hamming cross-check match, epipolar-gated DLT triangulation, %s, %s, local BA schedule 5+5+10 LM + 10 pose-only "
                    "per window%s%s" % (args.anms, pose_txt, win_txt, "" if not args.no_ba else " [BA disabled]",
                                        " [depth stage swapped for the reference's own: SGBM disparity + find_3d; not the BASELINE metric]" if args.depth == "sgbm" else ""))
        res = {
            "metric": "stereo keyframes/sec (ORB+match+tri+local-BA), KITTI-00 1241x376",
            "value": round(value, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "world_size": (dist.get_world_size() if dist is not None else 1),
            "collective": ("RCCL %s over torch.distributed backend nccl" % _rccl_version(torch)) if dist is not None else "none (1 rank)",
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "strong" if seq_mode else "weak", "vs_baseline": None,
            "dtype": "u8+f64", "data": "synthetic",
            "config": {"workload": workload,
                       "batch_keyframes_per_gpu": B, "image": "1241x376 u8",
                       "batches_in_flight_per_gpu": len(ring),
                       "ba_windows": pipe.ba_windows if pipe.with_ba else None,
                       "unique_inputs": "%d rendered stereo keyframes per pipeline (one sequence each, ping-pong over the batch when shorter than it; %s), %d BA windows (%s)" % (
                           pipe.unique_frames, "every pipeline in flight its own sequence" if getattr(ring, "distinct", False) else "the pipelines share the frames",
                           pipe.unique_windows if pipe.with_ba else 0, "built from the step's tracks" if pipe.ba_windows == "tracks" else "canned"),
                       "parallelism": ("one %d-frame sequence in %d contiguous chunks with a %d-frame halo, relative poses gathered and chained, track state carried across the "
                                       "chunk boundaries rank by rank, BA on the windows of the owned frames (the unsharded run's windows bit for bit)" % (
                                           args.sequence, world, sharding.WINDOW_HALO if seq_exact_ba else 1))
                                      if seq_mode else "%d independent replicas, sharded keyframes" % world},
            "timing": {"repeats": len(rep_s), "ms_per_step_each": [round(1e3 * x / args.steps, 4) for x in rep_s], "reported": "median",
                       "spread_pct": round(100.0 * (max(rep_s) - min(rep_s)) / elapsed, 2),
                       "in_flight": {"batches": len(ring),
                                     "how": "step k is queued on pipeline k mod %d (own context, HIP stream and buffers) without waiting for step k - 1; every step is "
                                            "the whole hot path over one batch of %d keyframes; the timed region ends with a device-wide synchronize" % (len(ring), B),
                                     "one_batch_in_flight": None if serial_s is None else {"ms_per_step": round(1e3 * serial_s / args.steps, 4),
                                                                                            "value": round(units_per_step(args, seq_mode, world, B) * args.steps / serial_s, 3)},
                                     "results_identical_in_flight_and_alone": flight_same,
                                     "overlap": overlap},
                       "ba_schedule": None if not pipe.with_ba else dict(
                           schedule_stats(pipe.ba_passes), mode="plain (--ba-plain-schedule)" if args.ba_plain_schedule else "adaptive (default)",
                           plain_schedule=None if plain_s is None else {"ms_per_step": round(1e3 * plain_s / args.steps, 4),
                                                                         "value": round(units_per_step(args, seq_mode, world, B) * args.steps / plain_s, 3),
                                                                         "how": "the same steps, same batches in flight, with every optimize_map pass run for every window "
                                                                                "(three launches; vslam_set_tuning ba_adaptive = 0)"}),
                       "stage_profiler": "off in the timed repeats; on in one extra repeat of the same %d steps with ONE batch in flight (%.4f ms/step) that feeds "
                                         "`roofline` and `kernels_ms_per_step`" % (args.steps, 1e3 * profiled_s / args.steps)},
            "roofline": {"bound": "hbm", "kernel": dom_kernels, "stage_family": dom,
                         "windows_left_to_lm_window_kernel": (res_deferred if dom == "lm_window_kernel" and pipe.with_ba else None),
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "copy_ceiling_gbs": round(copy_gbs, 1), "copy_probe": copy_probe, "copy_ceiling_guide_gbs": 6290.0, "frac_of_copy_ceiling": round(achieved / copy_gbs, 6) if copy_gbs > 0 else None,
                         "algorithmic_bytes_per_launch_set": int(alg), "formula": formula,
                         "plain_schedule_equivalent": None if alg_plain is None or per_bracket_s <= 0 else {
                             "algorithmic_bytes_per_launch_set": int(alg_plain), "achieved": round(alg_plain / per_bracket_s / 1e9, 3),
                             "frac": round(alg_plain / per_bracket_s / 1e9 / HBM_PEAK_GBS, 6),
                             "note": "the bytes of the 30 linearisations per window the reference's schedule performs, over the same time: the adaptive schedule delivers "
                                     "that result with fewer of them; `achieved` / `frac` above count only what was executed"},
                         "avg_ms_per_launch_set": round(1e3 * per_bracket_s, 4), "kernel_launches_per_set": dom_launches // max(dom_calls, 1),
                         "launch_set": (("one BA schedule = 3 x lm_window_kernel (optimize_map, 5 + 5 + 10 iterations) + 1 x pose_only_wave_kernel (10 iterations)"
                                         if args.ba_plain_schedule else
                                         "one BA schedule (stage-profiler family `lm_window_kernel`) = 1 x ba_resident_kernel (optimize_map passes 5 + 5 + 10 back to back for every "
                                         "window that fits its LDS budget: all of them here; a pass that flags nothing new is continued as the last one) + 1 x lm_window_kernel (the "
                                         "windows ba_resident_kernel deferred: none here, its workgroups return at once) + 1 x pose_only_wave_kernel (10 iterations)")
                                        if dom == "lm_window_kernel" else "one stage bracket")},
            "kernels_ms_per_step": {k: round(v[0] / n_prof_steps, 4) for k, v in kern},
            "other_rooflines": other_rooflines(prof, pipe, args, n_prof_steps, copy_gbs),
            "stats": {"keypoints_per_image": float(out["cnt"].mean()), "lr_matches": float(out["nlr"].mean()), "lr_matches_min": int(out["nlr"].min()),
                      "f2f_matches": float(out["nf2f"][:nt].mean()), "pnp_points": float(out["pn"][:nt].mean()), "pnp_points_min": int(out["pn"][:nt].min()),
                      "pnp_inliers": float(out["ninl"][:nt].mean()), "pnp_inliers_min": int(out["ninl"][:nt].min()),
                      "orb_status_nonzero": orb_bad, "ba_status_nonzero": ba_bad},
        }
        if win_stats is not None:
            res["stats"]["ba_windows"] = win_stats
        if host_inputs is not None:
            res["inputs_from_host"] = host_inputs
        if seq_mode and chained[0] is not None:
            res["trajectory"] = {"frames": int(len(chained[0])), "final_position": [float(x) for x in sharding.camera_centre(chained[0][-1])]}
        extras_ok = world == 1 and not args.no_ba and args.depth == "match" and args.pose == "lm" and not seq_mode
        if extras_ok and pipe.ba_windows == "tracks":
            pipe.close()    # (host arrays stay; the GPU memory goes back before the next pipeline is built)
            if not args.no_reference_pipeline:
                try:
                    res["reference_pipeline"] = reference_pipeline_measure(args, local, torch, pipe.h_seq)
                except Exception as e:  # an extra must never cost the headline
                    res["reference_pipeline"] = {"error": repr(e)}
            if not args.no_config4:
                try:
                    res["ba_config4"] = ba_config4_measure(args, local, torch)
                except Exception as e:
                    res["ba_config4"] = {"error": repr(e)}
            if not args.no_config4_step:
                try:
                    res["value_config4_windows"] = config4_step_measure(args, local, torch, [pipe.h_seq] + other_seqs)
                except Exception as e:
                    res["value_config4_windows"] = {"error": repr(e)}
            if not args.no_live_dropin:
                try:
                    res["live_dropin"] = live_dropin_measure(with_cpu=not args.no_cpu_baseline)
                except Exception as e:
                    res["live_dropin"] = {"error": repr(e)}
                try:   # real KITTI pairs, when the box has them (KITTI_ROOT); otherwise a one-line "not available"
                    res["kitti_seq00"] = kitti_measure()
                except Exception as e:
                    res["kitti_seq00"] = {"available": None, "error": repr(e)}
        if extras_ok and not args.no_cpu_baseline:
            res["cpu_baseline"], res["pose_rmse_vs_oracle"] = (cpu_baseline_tracks if pipe.ba_windows == "tracks" else cpu_baseline)(pipe, out, args.anms)
            info, have_cv2 = host_info()
            res["cpu_baseline"]["host"] = info
            if have_cv2:
                try:   # ... and, the day a box has OpenCV, the oracle is compared with it (tests/test_reference_libs_pin.py) and the outcome recorded here
                    sys.path.insert(0, os.path.join(ROOT, "tests"))
                    import oracle as O
                    from stereo_visual_slam_amd import synth as S
                    from test_reference_libs_pin import compare_all
                    res["cpu_baseline"]["host"]["reference_libs_pin"] = compare_all(O, S, n_images=2)
                except Exception as e:
                    res["cpu_baseline"]["host"]["reference_libs_pin"] = "cv2 present but the comparison failed: %r" % (e,)
                try:
                    res["cpu_baseline"]["reference_libs_timing"] = reference_lib_timing(pipe, args.anms)
                except Exception as e:
                    res["cpu_baseline"]["reference_libs_timing"] = "cv2 present but failed: %r" % (e,)
            else:
                res["cpu_baseline"]["reference_libs_timing"] = "unavailable on this host (no cv2 / OpenCV / g2o found at run time)"
                res["cpu_baseline"]["host"]["reference_libs_pin"] = "skipped: no cv2 on this host (tests/test_reference_libs_pin.py)"
        # scalars of the extra measurements, repeated inside `config` (a record that keeps only the contract's keys still carries them)
        def _g(d, *ks):
            for k_ in ks:
                d = d.get(k_) if isinstance(d, dict) else None
            return d
        # [r6] FLAT scalar keys of `config` (round 5 nested them under config.extras, which a parser that keeps scalar config keys dropped)
        orb_fam = [r for r in res["other_rooflines"] if r.get("kernel") == "orb_* (family)"]
        res["config"].update({
            "one_in_flight_kfps": _g(res, "timing", "in_flight", "one_batch_in_flight", "value"),
            "plain_ba_schedule_kfps": _g(res, "timing", "ba_schedule", "plain_schedule", "value"),
            "overlap_share_two_or_more_kernels": _g(res, "timing", "in_flight", "overlap", "overlap_share"),
            "sum_kernel_ms_over_span_ms": _g(res, "timing", "in_flight", "overlap", "sum_kernel_ms_over_span_ms"),
            "inputs_from_host_kfps": _g(res, "inputs_from_host", "value"),
            "value_config4_windows_kfps": _g(res, "value_config4_windows", "value"),
            "reference_pipeline_kfps": _g(res, "reference_pipeline", "value"),
            "reference_pipeline_sgbm_ms_per_pair": _g(res, "reference_pipeline", "roofline", "ms_per_pair"),
            "ba_config4_ms_per_256": _g(res, "ba_config4", "ms_per_schedule_batch"),
            "ba_built_windows_ms_per_schedule_batch": _g(res, "roofline", "avg_ms_per_launch_set"),
            "orb_family_ms_per_1024_images": orb_fam[0].get("ms_per_1024_images") if orb_fam else None,
            "orb_family_traffic_bytes_per_1024_images": orb_fam[0].get("traffic_per_1024_images") if orb_fam else None,
            "kitti_seq00_available": _g(res, "kitti_seq00", "available"),
            "live_dropin_fps": _g(res, "live_dropin", "gpu", "frames_per_s"),
            "live_dropin_kfps": _g(res, "live_dropin", "gpu", "keyframes_per_s"),
            "live_dropin_cpu_path_fps": _g(res, "live_dropin", "cpu", "frames_per_s"),
            "pose_rmse_vs_oracle_m": _g(res, "pose_rmse_vs_oracle", "ba_translation_rmse_m"),
        })
        print(json.dumps(res), flush=True)
    ring.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
