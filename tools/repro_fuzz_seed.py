"""Developer tool (GPU box): one seed of the randomised parity scenes (tests/test_gpu_fuzz.py, tools/soak_fuzz.py) at several depths, both
integrators, instrumented and timed kernels, GPU against the oracle: differing values and the first differing pixels (then
tools/repro_fuzz_pixel.py SEED ROW COL).   python tools/repro_fuzz_seed.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrs_amd
import test_gpu_fuzz as F
from oracle.binding import OracleScene
seed = int(sys.argv[1])
sb = F.random_scene(seed)
hs = pbrs_amd.HostScene(sb)
d = hs.desc
print("scene: inst", d.n_instances, "tris", d.n_triangles, "mats", d.n_materials, "bxdfs", d.n_bxdfs, "area", d.n_area_lights, "delta", d.n_delta_lights, "tex", d.n_textures, "bytes", hs.nbytes)
ctx = pbrs_amd.Context(0)
ctx.upload(hs)
osc = OracleScene(sb)
depth = 5
for integrator in ("path", "direct"):
    for dd in (5, 6, 8, 3):
        ref, ost = osc.render(2, 2, dd, 11 + seed, integrator=integrator)
        for counters in (True, False):
            img, st = ctx.render(2, 2, dd, 11 + seed, integrator=integrator, counters=counters)
            nan = np.isnan(ref)
            diff = (np.isnan(img) != nan) | ((img.view(np.uint32) != ref.view(np.uint32)) & ~nan)
            print(integrator, "depth", dd, "counters", counters, "differing values", int(diff.sum()), "pixels", int(diff.any(axis=-1).sum()), "ties", ost.get("tlas_ties"), "rays", st["closest_rays"], ost["closest_rays"], st["shadow_rays"], ost["shadow_rays"], "invalid", st["invalid_samples"], ost["nonfinite_samples"], "feat", st["kernel_features_extend"], st["kernel_features_shadow"])
            if diff.any():
                ys, xs, cs = np.nonzero(diff)
                for k in range(min(4, len(ys))):
                    print("    pixel", ys[k], xs[k], cs[k], img[ys[k], xs[k]], ref[ys[k], xs[k]])
