"""CPU: the launcher contract of bench.py (ADVICE r2 / VERDICT r2 weak #3): `--gpus N` is never silently ignored.
With N > 1 and no WORLD_SIZE it starts N ranks itself, or refuses when fewer GPUs are visible (here: none); under a launcher a
WORLD_SIZE that contradicts --gpus is an error; N = 1 without a GPU fails loudly (there is no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    env["HIP_VISIBLE_DEVICES"] = ""   # make "no GPU" explicit even on a GPU box
    env["CUDA_VISIBLE_DEVICES"] = ""
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_n_without_devices_refuses():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "--gpus 8 but only 0 GPU(s) visible" in (r.stderr + r.stdout), r.stderr[-400:]
    assert '"n_gpus"' not in r.stdout      # no line of any kind was printed


def test_gpus_must_match_world_size():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "does not match WORLD_SIZE 4" in (r.stderr + r.stdout), r.stderr[-400:]


def test_single_gpu_without_device_fails_loudly():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout), r.stderr[-400:]
