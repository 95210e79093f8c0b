"""The conservative box-test filter the wide traversal prunes with (include/pbrs_numeric.h, pn_slab_filter): whenever the
reference's box test (geometry/src/bvh.rs:84-99, correctly rounded f32 divisions) passes, the filter passes.  Plain f32
arithmetic, compiled like every other side of the numeric contract (-ffp-contract=off), so the check runs on the CPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_filter_passes_whenever_the_reference_test_passes(tmp_path):
    exe = tmp_path / "slab_filter_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "slab_filter_check.c"), "-lm"])
    out = subprocess.run([str(exe), "10000000"], capture_output=True, text=True)
    n, exact, filt, violations = (int(x) for x in out.stdout.strip().split("\n")[-1].split())
    assert out.returncode == 0 and violations == 0, out.stdout
    assert n == 10_000_000 and exact > 1_000_000 and filt >= exact  # random, flat-box, corner, on-face and extent-on-plane cases


def test_unused_slot_of_a_wide_node_never_passes(tmp_path):
    """device/wide.h: a slot not in use carries the inverted box (2^60, -2^60) instead of a mask bit; no ray of the guarded
    range may pass it (and none may overflow into a NaN that would)."""
    exe = tmp_path / "slab_filter_check"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "slab_filter_check.c"), "-lm"])
    out = subprocess.run([str(exe), "5000000", "unused"], capture_output=True, text=True)
    n, passes = (int(x) for x in out.stdout.strip().split())
    assert out.returncode == 0 and n == 5_000_000 and passes == 0, out.stdout
