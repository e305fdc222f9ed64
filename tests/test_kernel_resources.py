"""Build invariants of the gfx950 code objects inside libvslam_hip.so, read from their metadata notes (no GPU needed).

Two measured rules of the MI355X (tools/scratch/occupancy_probe.hip) cost whole kernels their designed occupancy without any
compiler diagnostic, so they are pinned here:
  * a SIMD takes eight waves only from kernels with sgpr_count <= 80 (the compiler's own occupancy figure allows 96): kernels that were written for
    eight waves per SIMD -- at most 64 VGPRs -- must stay at or below 80 SGPRs (orb_anms_kernel ran ONE workgroup per CU instead of two at 83);
  * a device function that is not inlined makes the callee's registers and stack the kernel's (the ANMS kernel's sort): a kernel that has a private
    segment although the compiler reports no spill to it is the tell, so the kernels with a stack are an explicit, reviewed list."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "stereo-visual-slam_amd", "libvslam_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"

# kernels whose register allocation spills to scratch by design (256-VGPR LM / EPnP kernels); everything else must have no private segment
STACK_ALLOWED = ("lm_window_kernel", "ba_resident_kernel", "pose_only_wave_kernel", "epnp_front_kernel", "epnp_back_kernel")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not (os.path.exists(SO) and os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("libvslam_hip.so or the LLVM tools are missing")
    d = str(tmp_path_factory.mktemp("codeobj"))
    so = os.path.join(d, "lib.so")
    shutil.copy(SO, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, blk).group(1))
            out[name.group(1)] = dict(sgpr=g("sgpr_count"), vgpr=g("vgpr_count"), stack=g("private_segment_fixed_size"), wg=g("max_flat_workgroup_size"))
    assert len(out) > 60, "expected the library's ~90 kernels, found %d" % len(out)
    return out


def test_eight_wave_kernels_fit_the_sgpr_budget(kernels):
    """<= 64 VGPRs means the kernel was meant to run eight waves per SIMD: that needs sgpr_count <= 80 on this device."""
    over = {k: v for k, v in kernels.items() if v["vgpr"] <= 64 and v["sgpr"] > 80 and "window_rank_kernel" not in k and "track_walk_kernel" not in k}
    # (window_rank_kernel: one 1024-slot pass of 60 us per step, 94 SGPRs of ballot bookkeeping; it does not depend on the eighth wave.
    #  track_walk_kernel [r6]: one thread per candidate path, a chain of dependent loads per hop with thirteen table pointers and the camera in SGPRs (90);
    #  most threads return at once, the walkers wait on memory: seven waves per SIMD lose nothing)
    assert not over, "kernels written for 8 waves/SIMD above 80 SGPRs (they run at 7): %s" % {k[:60]: v for k, v in over.items()}


def test_anms_keeps_two_workgroups_per_cu(kernels):
    hits = {k: v for k, v in kernels.items() if "orb_anms_kernel" in k}
    assert len(hits) >= 2
    for k, v in hits.items():
        assert v["vgpr"] <= 64 and v["sgpr"] <= 80 and v["stack"] == 0, (k, v)


def test_only_the_reviewed_kernels_have_a_stack(kernels):
    bad = {k[:70]: v for k, v in kernels.items() if v["stack"] > 0 and not any(a in k for a in STACK_ALLOWED)}
    assert not bad, "kernels with a private segment (a non-inlined call or an unexpected spill): %s" % bad
