"""Developer tool (GPU box): the BASELINE scene FAMILIES with random seeds, sizes and cameras — what tools/soak_fuzz.py's small scenes do
not reach: deep BLASes (C4's terrain at 8 K .. 260 K triangles: several node steps per round, lean steps, the four-wide any-hit walk and
its slow list, the queue split) and TLASes too large to scan (C5's object / light grids at 20 .. 130 instances: the tree walk with the
TLAS in LDS, class-major shading), next to the Cornell boxes under random cameras.  Per seed: a small frame through the timed kernels
and through the instrumented ones against the oracle, bit for bit, plus hit records / occlusion of random rays (with axis-parallel,
denormal and huge components among them).

usage: python tools/soak_configs.py [first_seed [end_seed]] [--budget SECONDS]   (default 0 100000, 240 s; about 18 seeds per second)
Ends itself at end_seed or when the budget is spent; prints `seeds a .. b failures: [...]`, exits 0 / 1."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pbrs_amd  # noqa: E402
from oracle.binding import OracleScene  # noqa: E402
from pbrs_amd import roofline, scenes  # noqa: E402
from pbrs_amd.spec import deg  # noqa: E402


def scene_of(seed):
    rs = np.random.RandomState(seed)
    family = seed % 4
    w, h = int(rs.randint(40, 97)), int(rs.randint(24, 57))
    if family in (0, 1):  # terrain: 8 K .. 260 K triangles, the camera somewhere over it
        nx, nz = int(rs.choice([64, 96, 128, 192, 256])), int(rs.choice([64, 128, 256, 512]))
        sb = scenes.terrain_scene(width=w, height=h, nx=nx, nz=nz, seed=int(rs.randint(1, 1 << 30)))
        sb.set_camera(w, h, deg(rs.uniform(30, 70)), (float(rs.uniform(-80, 80)), float(rs.uniform(25, 90)), float(rs.uniform(-90, 100))),
                      (float(rs.uniform(-60, 60)), float(rs.uniform(-5, 10)), float(rs.uniform(120, 300))))
        if rs.rand() < 0.3:  # shadow rays outside the guarded range of the division-free box test (tests/test_gpu_render.py)
            sb.distant_light((0.0, -1.0, 0.0), (1.5, 1.4, 1.3), 700.0)
        name = "terrain %dx%d" % (nx, nz)
    elif family == 2:  # many lights: 20 .. 130 instances
        n_obj, n_l = int(rs.randint(8, 65)), int(rs.randint(8, 65))
        sb = scenes.many_lights_scene(width=w, height=h, n_objects=n_obj, n_lights=n_l, seed=int(rs.randint(1, 1 << 30)))
        sb.set_camera(w, h, deg(rs.uniform(35, 60)), (float(rs.uniform(-30, 30)), float(rs.uniform(15, 45)), float(rs.uniform(-45, -10))),
                      (float(rs.uniform(-20, 20)), float(rs.uniform(0, 8)), float(rs.uniform(40, 80))))
        name = "many lights %d+%d" % (n_obj, n_l)
    else:
        variant = "diffuse" if rs.rand() < 0.5 else "specular"
        sb = scenes.cornell_scene(width=w, height=h, variant=variant)
        sb.set_camera(w, h, deg(rs.uniform(40, 75)), (float(rs.uniform(120, 440)), float(rs.uniform(100, 450)), float(rs.uniform(10, 200))),
                      (float(rs.uniform(150, 400)), float(rs.uniform(100, 400)), 555.0))
        name = "cornell " + variant
    return sb, name, rs


def check(ctx, seed):
    sb, name, rs = scene_of(seed)
    osc = OracleScene(sb)
    ctx.upload(pbrs_amd.HostScene(sb))
    depth = int(rs.randint(3, 9))
    integrator = "path" if rs.rand() < 0.8 else "direct"
    ref, ost = osc.render(2, 2, depth, 5 + seed, integrator=integrator)
    if ost["tlas_ties"]:
        return name + " (skipped: coincident geometry)"
    for counters in (False, True):
        img, st = ctx.render(2, 2, depth, 5 + seed, integrator=integrator, counters=counters)
        nan = np.isnan(ref)
        assert (nan == np.isnan(img)).all() and (img.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all(), (name, integrator, depth, "counters" if counters else "timed")
        if counters:
            assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"], (name, "ray counts")
    # rays: from the camera and from points along them, some with awkward components
    o, d = osc.camera_rays(0, 2, 2, 5 + seed)
    o2 = (o + d * rs.uniform(0.5, 60, (len(o), 1))).astype(np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, rs.standard_normal(o2.shape).astype(np.float32)])
    d[::7, 0] = 0.0
    d[::11, 2] = 0.0
    d[5::13] = (0.0, -1.0, 0.0)
    d[3::17, 0] = 1e-42
    d[4::19, 1] *= np.float32(2.0 ** 50)
    tmax = np.where(rs.uniform(size=len(o)) < 0.5, np.inf, rs.uniform(5, 300, len(o))).astype(np.float32)
    h_ref, occ_ref, ist = osc.intersect(o, d, tmax)
    h_gpu, occ_gpu = ctx.intersect(o, d, tmax)
    keep = ~ist["tie_mask"]
    assert (h_ref["t"].view(np.uint32) == h_gpu["t"].view(np.uint32)).all(), (name, "t")
    assert (h_ref["inst"][keep] == h_gpu["inst"][keep]).all() and (h_ref["prim"][keep] == h_gpu["prim"][keep]).all(), (name, "hit")
    assert (occ_ref == occ_gpu).all(), (name, "occlusion")
    return name


args = list(sys.argv[1:])
budget = 240.0
if "--budget" in args:
    k = args.index("--budget")
    budget = float(args[k + 1])
    del args[k:k + 2]
first = int(args[0]) if len(args) > 0 else 0
end = int(args[1]) if len(args) > 1 else 100000
t0 = time.perf_counter()
ctx = pbrs_amd.Context(0)
bad, families = [], {}
seed = first
while seed < end and time.perf_counter() - t0 < budget:
    try:
        name = check(ctx, seed)
        key = "skipped (coincident geometry)" if "skipped" in name else name.split()[0]
        families[key] = families.get(key, 0) + 1
    except AssertionError as e:
        bad.append((seed, str(e)[:120]))
    if seed % 64 == 0:
        print("seed", seed, "failures so far", len(bad), "elapsed %.0f s" % (time.perf_counter() - t0), flush=True)
    seed += 1
print("seeds", first, "..", seed - 1, "scenes", families, "failures:", bad, "(%.0f s of a %.0f s budget, library sources %s)" % (
    time.perf_counter() - t0, budget, roofline.source_hash()), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
