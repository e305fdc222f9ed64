"""CPU: the oracle's C sources under -fsanitize=address,undefined.  oracle/asan_check.c runs
ORB detect + ANMS + rBRIEF, the matcher, SGBM, EPnP + RANSAC + the motion-only LM on generated inputs; any out-of-bounds access,
signed overflow or misaligned access aborts with a sanitizer report."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan_check"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "asan_check")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("orb ") and int(lines[0].split()[1]) > 50          # the detector found keypoints on both views
    assert "sgbm rc 0" in lines[1] and int(lines[1].split()[4]) > 1000             # the disparity map has valid pixels
    assert int(lines[2].split()[2]) > 80                                           # RANSAC kept the inliers
