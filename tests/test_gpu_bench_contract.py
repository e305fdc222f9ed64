"""GPU: bench.py prints ONE JSON line that carries the driver's contract fields, `roofline` and `cpu_baseline` (small batch so that
the whole run takes well under a minute); `--depth sgbm` reports the SGBM family as its roofline line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]        # exactly one JSON line
    return json.loads(lines[0])


def test_bench_line_contract():
    r = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "16", "--unique-frames", "8")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["world_size"] == 1 and r["steps"] == 2 and r["warmup"] == 1
    assert r["unit"] == "keyframes/s" and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 16 * 2 / (r["ms_per_step"] * 2e-3)) / r["value"] < 1e-3           # value = units / time
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    # [r6] the roofline line is named after the kernels that ran in the BA bracket (the windows built from tracks all fit ba_resident_kernel), the stage
    # profiler's family name stays beside it
    assert rf["kernel"] == "ba_resident_kernel+pose_only_wave_kernel" and rf["stage_family"] == "lm_window_kernel" and rf["windows_left_to_lm_window_kernel"] == 0
    assert rf["copy_ceiling_gbs"] > 3000
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and "cpu_model" in cb["host"] and "reference_libs_timing" in cb
    assert r["pose_rmse_vs_oracle"]["integer_mismatches"] == 0
    assert r["inputs_from_host"]["value"] > 0 and r["inputs_from_host"]["h2d_ms_per_step"] > 0
    assert r["stats"]["orb_status_nonzero"] == 0 and r["stats"]["ba_status_nonzero"] == 0
    # round 4: the step is one pipeline -- the BA consumes windows built on the device from the step's own tracks, and the line says so
    assert r["config"]["ba_windows"] == "tracks" and "built on the device" in r["config"]["workload"]
    w = r["stats"]["ba_windows"]
    assert w["builder_status"] == 0 and w["landmarks_per_window_mean"] > 100 and w["edges_per_window_mean"] >= w["landmarks_per_window_mean"]
    assert "build_windows_kernels" in r["kernels_ms_per_step"]
    # ... the same run measures the BA schedule on the config-4 shape and the reference's own stages (SGBM depth + RANSAC pose)
    c4 = r["ba_config4"]
    assert c4["ms_per_schedule_batch"] > 0 and c4["roofline"]["kernel"] == "lm_window_kernel+pose_only_wave_kernel" and c4["roofline"]["windows_left_to_lm_window_kernel"] == 256 and "traffic" in c4["roofline"]
    rp = r["reference_pipeline"]
    assert rp["value"] > 0 and rp["unit"] == "keyframes/s" and rp["roofline"]["kernel"].startswith("sgbm_*") and rp["stats"]["ransac_inliers"] > 10
    assert "pnp_epnp_kernels" in rp["kernels_ms_per_step"] and "sgbm_down_kernel" in rp["kernels_ms_per_step"] or rp["batch"] < 8
    assert "reference_libs_pin" in cb["host"]
    # two batches in flight by default (pipeline.PipelineRing): the line says so, carries the one-batch figure beside it, and the run itself
    # checked that the pipelines produced the same bits
    fl = r["timing"]["in_flight"]
    assert fl["batches"] == 2 and r["config"]["batches_in_flight_per_gpu"] == 2 and fl["results_identical_in_flight_and_alone"] is True
    # round 5: the pipelines in flight show different frames; the headline configuration has its own attribution (brackets of both pipelines on one time axis)
    assert "its own sequence" in r["config"]["unique_inputs"]
    ov = fl["overlap"]
    assert ov["pipelines"] == 2 and 0.0 <= ov["overlap_share"] <= 1.0 and ov["sum_kernel_ms_over_span_ms"] > 0.5 and "lm_window_kernel" in ov["family_ms_per_step_in_flight"]
    # ... the same step at the config-4 BA shape, the drop-in's own frames/s, and the scalars of every extra measurement inside `config`
    v4 = r["value_config4_windows"]
    assert v4["value"] > 0 and v4["unit"] == "keyframes/s" and v4["batch"] == 16
    ld = r["live_dropin"]
    assert ld["gpu"]["frames"] == 50 and ld["gpu"]["frames_per_s"] > 1 and ld["gpu"]["keyframes_per_s"] > 0 and ld["cpu"]["frames_per_s"] > 0
    # [r6] ... as FLAT scalar keys of `config` (no nested dict: a parser that keeps the scalar keys of `config` keeps them)
    ex = r["config"]
    assert "extras" not in ex and all(not isinstance(v, (dict, list)) for v in ex.values())
    assert ex["value_config4_windows_kfps"] == v4["value"] and ex["reference_pipeline_kfps"] == rp["value"]
    assert ex["ba_config4_ms_per_256"] == c4["ms_per_schedule_batch"] and ex["one_in_flight_kfps"] == fl["one_batch_in_flight"]["value"]
    assert ex["live_dropin_fps"] == ld["gpu"]["frames_per_s"] and ex["overlap_share_two_or_more_kernels"] == ov["overlap_share"]
    fam = [o for o in r["other_rooflines"] if o["kernel"] == "orb_* (family)"][0]
    assert "traffic" in fam and ex["orb_family_ms_per_1024_images"] == fam["ms_per_1024_images"] > 0
    assert fl["one_batch_in_flight"]["value"] > 0 and r["inputs_from_host"]["batches_in_flight"] == 2
    assert rp["batches_in_flight"] == 2 and rp["one_batch_in_flight"]["value"] > 0
    # the BA schedule continues a pass that flags nothing new instead of repeating it: the line says what it did and carries the plain schedule's figure
    bs = r["timing"]["ba_schedule"]
    assert bs["mode"].startswith("adaptive") and sum(bs["windows_by_passes_executed"].values()) == 16 and 10 <= bs["lm_iterations_per_window_mean"] <= 20
    assert bs["plain_schedule"]["value"] > 0 and c4["plain_schedule_ms_per_batch"] > 0 and sum(c4["schedule"]["windows_by_passes_executed"].values()) == 256


def test_bench_synthetic_windows_and_ransac_pose():
    """--ba-windows synthetic keeps the config-4 shape of rounds 1-3; --pose ransac swaps in the reference's pose stage"""
    r = _bench("--ba-windows", "synthetic", "--landmarks", "600", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "8", "--unique-frames", "8",
               "--no-cpu-baseline", "--inputs", "resident")
    assert r["config"]["ba_windows"] == "synthetic" and "ba_windows" not in r["stats"] and "reference_pipeline" not in r
    r1 = _bench("--in-flight", "1", "--ba-windows", "synthetic", "--landmarks", "600", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "8",
                "--unique-frames", "8", "--no-cpu-baseline", "--inputs", "resident")
    assert r1["timing"]["in_flight"]["batches"] == 1 and r1["timing"]["in_flight"]["one_batch_in_flight"] is None
    r = _bench("--pose", "ransac", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "8", "--unique-frames", "8", "--no-cpu-baseline", "--inputs", "resident")
    assert "solvePnPRansac" in r["config"]["workload"] and "pnp_epnp_kernels" in r["kernels_ms_per_step"] and r["stats"]["pnp_inliers"] > 10


def test_bench_sgbm_depth_line():
    r = _bench("--depth", "sgbm", "--steps", "2", "--warmup", "1", "--repeats", "1", "--batch", "8", "--unique-frames", "8", "--no-cpu-baseline", "--inputs", "resident")
    assert r["roofline"]["kernel"].startswith("sgbm_* (family") and r["roofline"]["frac"] > 0
    assert "sgbm_down_kernel" in r["kernels_ms_per_step"]        # 8 pairs per call: the fused top-down kernel
    assert "SGBM" in r["config"]["workload"]


def test_kitti_root_hook(tmp_path, monkeypatch):
    """bench.py's real-data hook: a KITTI-layout tree (sequences/00/image_{0,1}/%06d.png) under KITTI_ROOT goes through host/run_vslam and comes back as
    keypoint / match / inlier statistics (no KITTI in this image: the tree holds eight rendered pairs)"""
    import importlib.util
    sys.path.insert(0, ROOT)
    from stereo_visual_slam_amd import synth
    seq = tmp_path / "sequences" / "00"
    seq.mkdir(parents=True)
    synth.write_pgm_sequence(str(seq) + "/", 8, seed=5, fmt="png")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    monkeypatch.delenv("KITTI_ROOT", raising=False)
    assert b.kitti_measure()["available"] is False
    monkeypatch.setenv("KITTI_ROOT", str(tmp_path))
    r = b.kitti_measure(n_frames=50)
    assert r["available"] is True and "error" not in r, r
    assert r["frames"] == 8 and r["keypoints_per_frame_mean"] >= 400 and r["pnp_inliers_min"] >= 10 and r["summary"]
