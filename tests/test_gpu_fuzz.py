"""A fixed-seed slice of tests/fuzz_parity.py inside `pytest -m gpu`: random sizes / seeds / outlier rates for all nine kinds
(matcher, SGBM, ORB, local BA, motion-only pose, RANSAC pose, the device window builder, batched RANSAC), HIP path vs oracle case by case.  The stand-alone tool runs for
minutes to hours; this slice is bounded to about a minute and prints its case counts into the pytest log."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_parity  # noqa: E402


def test_fuzz_slice_all_stages(pkg):
    ctx = pkg.VO(device=0, max_batch=2)
    try:
        # round-robin over the six stages so that each gets its share of the budget (ORB cases build a context per image size and
        # run the CPU oracle on up to 1400 x 700 pixels: they dominate the wall clock)
        n = fuzz_parity.run(seconds=55.0, seed=20260929, vo=ctx, schedule=["match", "sgbm", "ba", "pnp", "ransac", "windows", "ransac_dev", "ba_schedule", "ba_resident", "match", "orb"])
    finally:
        ctx.close()
    print("fuzz slice (seed 20260929):", n, "total", sum(n.values()))
    assert all(n[k] >= 2 for k in fuzz_parity.KINDS), n
