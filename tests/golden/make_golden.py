"""Regenerates tests/golden/*.npz from the CPU oracle.

The Rust reference cannot be built or run in this environment (no cargo/rustc, un-vendored crates —
SURVEY.md §8c), and it holds no golden images or integrator-level tests, so these fixtures are outputs
of the ORACLE (oracle/, the C++ restatement pinned by the reference's own known-answer tests), not of
the reference.  They pin (a) the oracle against drift between rounds and (b) the HIP path on the GPU
box, where they are compared bit for bit.   Usage:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.binding import OracleScene  # noqa: E402
from pbrs_amd import scenes  # noqa: E402

# name -> (config, width, height, strata_x, strata_y, depth, scene overrides)
CASES = {
    "c1_sphere_light": ("c1", 48, 48, 2, 2, 4, {}),
    "c2_cornell_diffuse": ("c2", 40, 40, 2, 2, 8, {}),
    "c3_cornell_specular": ("c3", 40, 40, 2, 2, 8, {}),
    "c4_terrain_8k_tris": ("c4", 48, 27, 2, 2, 8, {"nx": 64, "nz": 64}),
    "c5_many_lights": ("c5", 48, 27, 2, 2, 8, {}),
}
SEED = 1


def build(name):
    cfg, w, h, sx, sy, depth, kw = CASES[name]
    sb, _ = scenes.build_config(cfg, width=w, height=h, **kw)
    return sb, (w, h, sx, sy, depth)


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        sb, (w, h, sx, sy, depth) = build(name)
        osc = OracleScene(sb)
        image, stats = osc.render(sx, sy, depth, SEED, nthreads=4)
        o, d = osc.camera_rays(0, sx, sy, SEED)
        hits, occ, _ = osc.intersect(o, d, np.full(len(o), np.inf, dtype=np.float32))
        keys = ("closest_rays", "shadow_rays", "tlas_nodes", "blas_nodes", "instances", "instance_hits", "triangles", "tri_shading",
                "shade_events", "samples", "panics", "tlas_ties", "sphere_inside")
        np.savez_compressed(os.path.join(here, name + ".npz"), image=image, ray_o=o, ray_d=d, hit_t=hits["t"], hit_inst=hits["inst"],
                            hit_prim=hits["prim"], hit_b1=hits["b1"], hit_b2=hits["b2"], occluded=occ,
                            counters=np.array([stats[k] for k in keys], dtype=np.uint64), counter_names=np.array(keys))
        print(name, image.shape, "mean", image.mean(axis=(0, 1)), {k: stats[k] for k in ("closest_rays", "shadow_rays", "panics", "tlas_ties")})


if __name__ == "__main__":
    main()
