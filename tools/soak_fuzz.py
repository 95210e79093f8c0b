"""Developer tool (GPU box): the randomised-scene parity tests of tests/test_gpu_fuzz.py over a range of seeds far beyond the ones
the suite runs: images of both integrators (also with two objects re-covered by Fourier BSDFs; scenes that hold a ParallelQuad next to a mesh also as
their twin with disks for quads, which takes the regular kernels) and hit records / occlusion ray by ray, GPU vs oracle, bit for bit.

usage: python tools/soak_fuzz.py [first_seed [end_seed]] [--budget SECONDS]     (default 48 3000; about 15 seeds per second)

The run ends ITSELF: at end_seed or when the wall-clock budget is spent (default 240 s), whichever comes first, and always prints
its verdict line `seeds a .. b failures: [...]` and exits 0 (no failure) or 1 — size the budget to the GPU minutes at hand instead of
letting the run's time limit cut it short without a verdict."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrs_amd  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402

args = [a for a in sys.argv[1:]]
budget = 240.0
if "--budget" in args:
    k = args.index("--budget")
    budget = float(args[k + 1])
    del args[k:k + 2]
first = int(args[0]) if len(args) > 0 else 48
end = int(args[1]) if len(args) > 1 else 3000
t0 = time.perf_counter()
ctx = pbrs_amd.Context(0)
bad = []
seed = first
while seed < end and time.perf_counter() - t0 < budget:
    try:
        F.test_random_scene_rays_match_oracle(ctx, seed)
        F.test_random_scene_matches_oracle(ctx, seed)
        F.test_random_scene_with_fourier_materials_matches_oracle(ctx, seed)
        if F.takes_the_exact_extent_walk(F.random_scene(seed)):  # its twin without ParallelQuads: through the regular kernels
            F.test_random_scene_without_parallel_quads_matches_oracle(ctx, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:100]))
    if seed % 256 == 0:
        print("seed", seed, "failures so far", len(bad), "elapsed %.0f s" % (time.perf_counter() - t0), flush=True)
    seed += 1
print("seeds", first, "..", seed - 1, "failures:", bad, "(%.0f s of a %.0f s budget, library sources %s)" % (
    time.perf_counter() - t0, budget, __import__("pbrs_amd.roofline", fromlist=["source_hash"]).source_hash()), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
