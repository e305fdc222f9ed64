#!/usr/bin/env python3
"""bench.py -- stereo keyframes/s through the MI355X hot path (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one pass of the hot path over a batch of B synthetic KITTI-00-shaped stereo keyframes per GPU, inputs
resident in HBM (stereo-visual-slam_amd/pipeline.py): ORB detect+ANMS(1500)+describe on 2B images, L/R Hamming match,
DLT triangulation, frame-to-frame match, motion-only LM pose, local BA (10 KF x ~3000 landmarks, schedule 5+5+10 LM +
10 pose-only).  Keyframes shard across ranks with no data-path collective ("weak" scaling); the only collective is the
RCCL all-gather of the per-keyframe poses (56 B each) once per step.

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
    if kernel.startswith("lm_window_kernel<pnp>"):
        return (B - 1) * 10 * 20 * 500, "(B-1) x 10 its x 20 B/point x ~500 points"
    if kernel.startswith("lm_window_kernel"):
        E, L, K = pipe.edges_per_window, pipe.lms_per_window, pipe.n_kf
        per_it = E * 16 + L * 12 + K * 56 + (6 * K) ** 2 * 8 + 6 * K * 8 + L * 12
        return B * per_it * 30, "B windows x 30 LM linearisations x (E*16 + L*12 + K*56 + (6K)^2*8 + 6K*8 + L*12) B"
    if kernel.startswith("triangulate"):
        return B * anms_num * 30, "B x N x 30 B"
    return 0, "n/a"


def other_rooflines(prof, pipe, args):
    """the one MFMA kernel of the path (the matcher's Hamming table) against the dense int8 peak; informative only"""
    out = []
    k = prof.get("match_train_nearest_kernel")
    if k and k[0] > 0:
        n = float(args.anms)
        items = pipe.B + max(pipe.B - 1, 0)                      # L/R call + frame-to-frame call per step
        macs = items * n * n * 256.0 * args.steps               # +-1 byte products per step
        tops = 2.0 * macs / (k[0] / 1e3) / 1e12
        out.append({"kernel": "match_train_nearest_kernel", "bound": "mfma", "achieved": round(tops, 1), "peak": 5000.0, "unit": "TOP/s (int8)",
                    "frac": round(tops / 5000.0, 4), "note": "v_mfma_i32_32x32x32_i8; 4250 TOP/s sustained in tools/scratch/mfma_rate.hip"})
    # what actually bounds the dominant kernel: VALU issue.  Wave-instructions per window and schedule come from the SQ counter
    # pass (tools/profile_sq.sh -> profiles/traffic.json); every VALU op, f64 or not, takes a 4-cycle issue slot of its SIMD.
    k = prof.get("lm_window_kernel")
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        wi = float(tj.get("valu_wave_insts_per_window_schedule", 0))
    except Exception:
        wi = 0.0
    if k and k[0] > 0 and wi > 0 and pipe.lms_per_window == 3000 and pipe.n_kf == 10:
        sets = k[2] if len(k) > 2 and k[2] else args.steps
        t = k[0] / 1e3 / sets                                   # seconds per schedule batch
        slots = 256 * 4 * 2.4e9 * t / 4.0                       # 256 CUs x 4 SIMDs, one VALU issue per 4 cycles at 2.4 GHz
        used = wi * pipe.B
        out.append({"kernel": "lm_window_kernel", "bound": "valu-issue", "achieved": round(used / t / 1e12, 3), "peak": round(256 * 4 * 2.4e9 / 4.0 / 1e12, 3),
                    "unit": "T wave-instructions/s", "frac": round(used / slots, 4),
                    "note": "SQ_INSTS_VALU of the BA schedule (profiles/r01c_ba_sq_issue_stall_summary.txt) over the live kernel time; two waves per SIMD (256 VGPRs)"})
    return out


def cpu_baseline(pipe, anms_num, n_keyframes=64):  # ~14 s of single-thread CPU work
    """the CPU oracle (single thread) on a bounded sample of the same workload: `n_keyframes` stereo keyframes + windows"""
    import oracle as O
    from stereo_visual_slam_amd import synth
    B, w = pipe.B, pipe.w
    imgs = pipe.h_imgs
    n_keyframes = min(n_keyframes, B)
    win = synth.ba_window(n_kf=pipe.n_kf, n_lm=int(pipe.lms_per_window), seed=100)
    t0 = time.perf_counter()
    prev = None
    for b in range(n_keyframes):
        kL, dL = O.feature_detection(imgs[b][:, :w], 3000, anms_num)
        kR, dR = O.feature_detection(imgs[B + b][:, :w], 3000, anms_num)
        m = O.feature_matching(dL, dR, 1.0)
        uvL = np.stack([kL["x"][m["queryIdx"]], kL["y"][m["queryIdx"]]], 1)
        uvR = np.stack([kR["x"][m["trainIdx"]], kR["y"][m["trainIdx"]]], 1)
        ident = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)
        xyz, valid, rel = O.triangulate_dlt(uvL, uvR, ident)
        if prev is not None:
            pk, pd, pm, pxyz, pvalid = prev
            f = O.feature_matching(pd, dL, 1.0)
            kp2lr = -np.ones(len(pk), np.int64); kp2lr[pm["queryIdx"]] = np.arange(len(pm))
            li = kp2lr[f["queryIdx"]]
            ok = (li >= 0) & (pvalid[np.maximum(li, 0)] != 0)
            if ok.sum() >= 4:
                O.pnp_motion_only(pxyz[li[ok]], np.stack([kL["x"][f["trainIdx"][ok]], kL["y"][f["trainIdx"][ok]]], 1), ident, iters=10)
        prev = (kL, dL, m, xyz, valid)
        # local BA schedule (run_vslam.cpp:58-71)
        T = win["T0"].copy(); inl = np.ones(len(win["xyz"]), np.uint8)
        for iters, upd in ((5, False), (5, False), (10, True)):
            act = inl.astype(bool)[win["lm_idx"]]
            T2, _, chi2, _ = O.local_ba(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=iters)
            _, inl, _, _ = O.chi2_classify(chi2, win["lm_idx"][act], inl)
            if upd:
                T = T2
        act = inl.astype(bool)[win["lm_idx"]]
        O.pose_only_window(T, win["xyz"], win["kf_idx"][act], win["lm_idx"][act], win["uv"][act], iters=10)
    dt = time.perf_counter() - t0
    return dict(value=n_keyframes / dt, unit="keyframes/s", cores=1, kind="port",
                sample="%d stereo keyframes (2 ORB images, L/R + frame-to-frame match, DLT, motion-only LM, BA schedule on one "
                       "10x%d window) through oracle/libvo_oracle.so, single thread, %.1f s; host has %d cores"
                       % (n_keyframes, int(pipe.lms_per_window), dt, os.cpu_count()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="stereo keyframes per GPU per step")
    ap.add_argument("--anms", type=int, default=1500, help="keypoints per image after ANMS (BASELINE config 2: ~1500; reference: 500)")
    ap.add_argument("--landmarks", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ba", action="store_true")
    ap.add_argument("--depth", choices=["match", "sgbm"], default="match",
                    help="stereo depth stage: north_star L/R match + DLT (default, BASELINE metric) or the reference's SGBM + find_3d")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP-only (no CPU fallback)")

    from stereo_visual_slam_amd.pipeline import KeyframePipeline
    B = args.batch
    pipe = KeyframePipeline(B, device=local, anms_num=args.anms, n_lm=args.landmarks, seed=1000 * rank, verbose=args.verbose and rank == 0,
                            with_ba=not args.no_ba, depth=args.depth)
    dev = pipe.dev
    from stereo_visual_slam_amd.sharding import gather_poses

    def one_step():
        pipe.step()
        if world > 1:  # throughput-mode pose gather (RCCL over xGMI), 56 B per keyframe; ordered after the step on its stream
            with torch.cuda.stream(pipe.stream):
                gather_poses(pipe.d_Tpnp, dist)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize(dev)
    pipe.vo.profile_enable(True)
    pipe.vo.profile_read()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    prof = pipe.vo.profile_read()
    pipe.vo.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        out = pipe.download()
        value = world * B * args.steps / elapsed
        kern = sorted(prof.items(), key=lambda kv: -kv[1][0])
        dom, (dom_ms, dom_launches, dom_calls) = kern[0]
        alg, formula = algorithmic_bytes(dom, pipe, args.anms)
        per_bracket_s = dom_ms / 1e3 / max(dom_calls, 1)
        achieved = alg / per_bracket_s / 1e9 if per_bracket_s > 0 else 0.0
        traffic = None
        try:  # measured offline with rocprofv3 PMC passes (tools/profile_round.sh); only valid for the same kernel and batch
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("kernel") == dom and tj.get("batch") == B:
                traffic = int(tj["hbm_bytes_per_launch_set"])
        except Exception:
            traffic = None
        res = {
            "metric": "stereo keyframes/sec (ORB+match+tri+local-BA), KITTI-00 1241x376",
            "value": round(value, 3), "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8+f64", "data": "synthetic",
            "config": {"workload": ("stereo keyframe hot path: ORB(3000)->ANMS(%d)->rBRIEF on L+R 1241x376, L/R + frame-to-frame BF-Hamming "
                                    "cross-check match, DLT triangulation, motion-only LM pose (10 its), local BA 10 KF x %d landmarks "
                                    "(5+5+10 LM + 10 pose-only)%s" % (args.anms, args.landmarks, "" if not args.no_ba else " [BA disabled]"))
                                   + (" [depth stage swapped for the reference's own: SGBM disparity + find_3d; not the BASELINE metric]" if args.depth == "sgbm" else ""),
                       "batch_keyframes_per_gpu": B, "image": "1241x376 u8", "parallelism": "%d independent replicas, sharded keyframes" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch_set": int(alg), "formula": formula,
                         "avg_ms_per_launch_set": round(1e3 * per_bracket_s, 4), "kernel_launches_per_set": dom_launches // max(dom_calls, 1)},
            "kernels_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in kern},
            "other_rooflines": other_rooflines(prof, pipe, args),
            "stats": {"keypoints_per_image": float(out["cnt"].mean()), "lr_matches": float(out["nlr"].mean()),
                      "f2f_matches": float(out["nf2f"][:max(B - 1, 1)].mean()), "pnp_points": float(out["pn"][:max(B - 1, 1)].mean()),
                      "pnp_inliers": float(out["ninl"][:max(B - 1, 1)].mean())},
        }
        if world == 1 and not args.no_cpu_baseline and not args.no_ba and args.depth == "match":
            res["cpu_baseline"] = cpu_baseline(pipe, args.anms)
        print(json.dumps(res), flush=True)
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
