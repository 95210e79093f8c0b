"""Developer tool (GPU box): the two randomised-scene parity tests of tests/test_gpu_fuzz.py over a range of seeds far
beyond the ones the suite runs: images of both integrators (also with two objects re-covered by Fourier BSDFs) and hit
records / occlusion ray by ray, GPU vs oracle, bit for bit.
usage: python tools/soak_fuzz.py [first_seed [end_seed]]   (default 48 3000; about 40 seeds per second)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pbrs_amd
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 48
end = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
ctx = pbrs_amd.Context(0)
bad = []
for seed in range(first, end):
    try:
        F.test_random_scene_rays_match_oracle(ctx, seed)
        F.test_random_scene_matches_oracle(ctx, seed)
        F.test_random_scene_with_fourier_materials_matches_oracle(ctx, seed)
    except AssertionError as e:
        bad.append((seed, str(e)[:100]))
    if seed % 256 == 0:
        print("seed", seed, "failures so far", len(bad), flush=True)
print("seeds", first, "..", end - 1, "failures:", bad)
sys.exit(1 if bad else 0)
