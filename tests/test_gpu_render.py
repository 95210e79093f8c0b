"""Whole hot path (raygen -> extend -> shade -> shadow -> accumulate) through pbrs_render_tile against the oracle,
the committed fixtures, and size-independent properties at BASELINE.json's full sizes.

Tolerance: BASELINE.json asks per-pixel L2 radiance error < 1e-4 vs the CPU reference at matched seeds.  The f32
contract (include/pbrs_numeric.h, -ffp-contract=off, IEEE divide/sqrt) makes the two paths agree bit for bit, so
the tests assert equality of the bit patterns and report the L2 bound as the fallback criterion."""
import numpy as np
import pytest

import pbrs_amd
from common import GOLDEN_NAMES, SEED, bits, golden_case, load_golden
from oracle.binding import OracleScene
from pbrs_amd import scenes, tiling

pytestmark = pytest.mark.gpu
L2_TOL = 1e-4


def l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum(axis=-1)).max())


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_image_matches_golden_and_counters(gpu_ctx, name):
    g = load_golden(name)
    sb, (w, h, sx, sy, depth) = golden_case(name)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, st = gpu_ctx.render(sx, sy, depth, SEED, counters=True)
    assert l2(img, g["image"]) < L2_TOL
    assert (bits(img) == bits(g["image"])).all(), "radiance differs in the last bits from the oracle fixture"
    want = dict(zip(list(g["counter_names"]), g["counters"].tolist()))
    # identical traversal => identical work counts (the oracle lumps closest + shadow traversal together)
    assert st["closest_rays"] == want["closest_rays"] and st["shadow_rays"] == want["shadow_rays"]
    assert st["shade_events"] == want["shade_events"] and st["samples"] == want["samples"]
    hs = pbrs_amd.HostScene(sb)
    if not 2 <= hs.desc.n_instances <= 32:
        assert st["tlas_nodes"] + st["shadow_tlas_nodes"] == want["tlas_nodes"]
    else:
        # a TLAS of 2..16 instances is not walked: the wave tests every leaf box for each new ray (FlatScan, rays on the
        # division-free box test) and the walk visits the leaves that passed, in pre-order; the box-test count is then
        # leaves x rays, not the reference's.  The instances entered — what the boxes decide — are compared below.
        assert st["tlas_nodes"] + st["shadow_tlas_nodes"] <= hs.desc.n_instances * (st["closest_rays"] + st["shadow_rays"]) + want["tlas_nodes"]
    assert st["instances"] + st["shadow_instances"] == want["instances"]
    # any-hit visits BLAS children near-first (order-free for a boolean), so only its ray and TLAS counts are
    # comparable with the reference's left-first recursion; closest-hit counts are comparable in full
    assert st["blas_nodes"] <= want["blas_nodes"] and st["triangles"] <= want["triangles"]
    assert st["tri_shading"] == want["tri_shading"]


@pytest.mark.parametrize("cfg,w,h,sx,sy,depth,kw", [
    ("c1", 128, 128, 4, 4, 4, {}),
    ("c2", 96, 96, 3, 3, 8, {}),
    ("c3", 96, 96, 3, 3, 8, {}),
    ("c3", 64, 64, 2, 2, 12, {}),   # deeper than BASELINE: Russian roulette over many bounces
    ("c4", 80, 45, 4, 2, 8, {"nx": 96, "nz": 160}),  # non-square strata (32x16 at full size)
    ("c5", 96, 54, 2, 2, 8, {}),
])
def test_image_matches_oracle(gpu_ctx, cfg, w, h, sx, sy, depth, kw):
    sb, _ = scenes.build_config(cfg, width=w, height=h, **kw)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, _ = gpu_ctx.render(sx, sy, depth, 7)
    ref, ost = OracleScene(sb).render(sx, sy, depth, 7)
    assert ost["panics"] == 0 and ost["tlas_ties"] == 0
    assert l2(img, ref) < L2_TOL
    assert (bits(img) == bits(ref)).all()


@pytest.mark.parametrize("cfg,w,h,sx,sy,kw", [("c4", 80, 45, 4, 2, {"nx": 96, "nz": 160}), ("c3", 64, 64, 2, 2, {})])
def test_shading_check_kernels_give_the_same_image(gpu_ctx, cfg, w, h, sx, sy, kw):
    """The host proves, per mesh, that the tangent check of blas.rs:193-200 cannot fail (PBRS_MESH_*_SHADING_OK) and the
    traversal kernels then skip the shading frame of every candidate.  With the proofs cleared the same scene runs through
    the kernel variants that evaluate it for each candidate (helper lanes of the shared triangle step, four-triangle leaves
    on C4's mesh): same image, same counters, and both equal the oracle's."""
    import ctypes as C
    sb, _ = scenes.build_config(cfg, width=w, height=h, **kw)
    hs = pbrs_amd.HostScene(sb)
    gpu_ctx.upload(hs)
    img, st = gpu_ctx.render(sx, sy, 8, 7, counters=True)
    d = hs.desc
    meshes = np.ctypeslib.as_array(C.cast(d.meshes, C.POINTER(C.c_uint32)), shape=(d.n_meshes, 8))       # pbrs_mesh.flags: word 5
    insts = np.ctypeslib.as_array(C.cast(d.instances, C.POINTER(C.c_uint32)), shape=(d.n_instances, 32))  # pbrs_instance.mesh_flags: word 29
    assert (meshes[:, 5] & 3).any()
    meshes[:, 5] &= ~np.uint32(3)
    insts[:, 29] &= ~np.uint32(3)
    gpu_ctx.upload(hs)
    img2, st2 = gpu_ctx.render(sx, sy, 8, 7, counters=True)
    assert (bits(img) == bits(img2)).all()
    for k in ("closest_rays", "shadow_rays", "tlas_nodes", "blas_nodes", "triangles", "tri_shading", "instances", "instance_hits"):
        assert st[k] == st2[k], k
    ref, _ = OracleScene(sb).render(sx, sy, 8, 7)
    assert (bits(img2) == bits(ref)).all()


def test_materials_and_lights_outside_the_baseline_scenes(gpu_ctx):
    """Uber (5 lobes, swap_remove order), Substrate, Glossy, anisotropic Beckmann, point + distant lights (Q6 guard),
    disk + quad area lights, constant environment."""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    for variant in range(3):
        sb = SceneBuilder()
        floor = sb.substrate((0.4, 0.5, 0.3), (0.1, 0.1, 0.1))
        sb.instance(scenes.quad_mesh(sb, (-8, 0, -8), (8, 0, -8), (-8, 0, 8), (8, 0, 8), (0, 1, 0)), floor)
        uber = sb.uber((0.3, 0.2, 0.1), (0.4, 0.4, 0.4), kr=(0.5, 0.5, 0.5), kt=(0.3, 0.3, 0.3), rough=(0.05, 0.2), eta=1.4, opacity=0.7)
        sb.instance(sb.sphere((0, 0, 0), 1.0), uber, Transform.translater((-2.2, 1.0, 0)))
        sb.instance(sb.sphere((0, 0, 0), 1.0), sb.glossy((0.9, 0.8, 0.7), 0.01), Transform.translater((0, 1.0, 0.5)))
        sb.instance(sb.cuboid((-0.7, 0, -0.7), (0.7, 1.6, 0.7)), sb.uber((0.5, 0.1, 0.1), (0, 0, 0), rough=(0.3, 0.3), opacity=1.0),
                    Transform().rotate_y(deg(30)).translate((2.3, 0, 0)))
        if variant == 0:
            sb.env = (0.3, 0.4, 0.6)
            e = (6.0, 5.0, 4.0)
            disk = sb.disk((0, 5, 0), (0, -1, 0), (1.2, 0, 0))
            sb.instance(disk, sb.diffuse_light(e))
            sb.area_light(e, disk)
        elif variant == 1:
            sb.point_light((1, 4, -2), (30, 30, 25))
            sb.distant_light((0.3, -1.0, 0.4), (1.5, 1.5, 1.2), 12.0)
            e = (4.0, 4.0, 8.0)
            for k in range(3):  # 2 delta + 3 area: the Q6 guard sends the last two area picks to the env branch
                s = sb.sphere((-3 + 3 * k, 4.5, 1), 0.4)
                sb.instance(s, sb.diffuse_light(e))
                sb.area_light(e, s)
        else:
            e = (8.0, 8.0, 8.0)
            q = sb.quad((-1, 5, -1), (2, 0, 0), (0, 0, 2))
            sb.area_light(e, q)  # sampled as a light only (a ParallelQuad instance would trip D1)
            sb.env = (0.05, 0.05, 0.05)
        sb.set_camera(72, 48, deg(55.0), (0, 2.5, -7), (0, 1, 0))
        gpu_ctx.upload(pbrs_amd.HostScene(sb))
        img, _ = gpu_ctx.render(2, 2, 6, 11)
        ref, ost = OracleScene(sb).render(2, 2, 6, 11)
        nan_ref, nan_gpu = np.isnan(ref), np.isnan(img)
        assert (nan_ref == nan_gpu).all(), variant
        assert (bits(img)[~nan_ref] == bits(ref)[~nan_ref]).all(), variant


def _parallel_quad_scene(textured):
    """ParallelQuad INSTANCES in front of the camera (shape/src/simple.rs:120-163): the Interaction that k_shade rebuilds
    for them (`quad_isect`: pos = origin + u a + v b with the unsigned u, v of defect D1, uv = (u, v), dpdu = side_u) is
    what every shaded pixel of a quad goes through.  The camera sees the quads' own quadrant and, past their origins, the
    mirrored quadrants D1 creates; the reference would panic there (its `accurate_hit` assert) — the oracle counts
    those and carries on with the arithmetic result, which is what the device computes."""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    sb = SceneBuilder()
    floor = sb.lambertian((0.5, 0.5, 0.45))
    sb.instance(scenes.quad_mesh(sb, (-8, 0, -8), (8, 0, -8), (-8, 0, 8), (8, 0, 8), (0, 1, 0)), floor)
    tex = sb.checker((0.1, 0.1, 0.3), (0.9, 0.8, 0.6)) if textured else None  # reads the quad's uv
    sb.instance(sb.quad((-0.5, 0.6, 1.0), (1.6, 0.0, 0.3), (0.0, 1.4, 0.2)), sb.lambertian(tex if textured else (0.7, 0.3, 0.2)))
    sb.instance(sb.quad((0.0, 0.0, 0.0), (1.2, 0.0, 0.0), (0.0, 0.0, 1.2)), sb.plastic((0.2, 0.6, 0.3), (0.4, 0.4, 0.4), 0.15, True),
                Transform().rotate_y(deg(25.0)).rotate_x(deg(-20.0)).translate((-2.4, 1.2, 0.5)))
    sb.instance(sb.quad((2.0, 0.4, 0.0), (0.0, 1.5, 0.0), (0.9, 0.0, 0.9)), sb.mirror((0.9, 0.9, 0.9)))
    e = (9.0, 8.0, 7.0)
    light = sb.sphere((0.5, 5.0, -1.0), 0.7)
    sb.instance(light, sb.diffuse_light(e))
    sb.area_light(e, light)
    sb.env = (0.2, 0.25, 0.3)
    sb.set_camera(80, 56, deg(50.0), (0.2, 2.2, -6.0), (0, 1.0, 0.8))
    return sb


@pytest.mark.parametrize("lights", ["mixed", "disks", "delta_only", "env_only"])
def test_lambert_only_scenes_with_other_lights_than_the_baseline(gpu_ctx, lights):
    """The Lambert variant of k_shade without a light-shape cut (PBRS_SHADE_LAMBERT alone): every material one Lambertian lobe,
    area lights of several shapes — or of one shape that has no variant of its own, or none at all.  (C1 / C4 run the
    Lambert + sphere variant, C2 the Lambert + triangle one, everything else in this file the general kernel.)"""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    sb = SceneBuilder()
    sb.instance(scenes.quad_mesh(sb, (-8, 0, -8), (8, 0, -8), (-8, 0, 8), (8, 0, 8), (0, 1, 0)), sb.lambertian((0.5, 0.5, 0.45)))
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.lambertian((0.7, 0.3, 0.2)), Transform.translater((-2.0, 1.0, 0.5)))
    sb.instance(sb.cuboid((-0.7, 0, -0.7), (0.7, 1.5, 0.7)), sb.lambertian((0.2, 0.6, 0.3)), Transform().rotate_y(deg(30)).translate((1.8, 0, 0.3)))
    sb.instance(scenes.box_mesh(sb, (-0.5, 0, -0.5), (0.5, 0.8, 0.5)), sb.lambertian((0.3, 0.3, 0.8)), Transform().rotate_y(deg(-20)).translate((0, 0, -1.5)))

    def light(shape, e):
        sb.instance(shape, sb.diffuse_light(e))
        sb.area_light(e, shape)
    if lights == "mixed":
        light(sb.sphere((2.5, 4.5, -1.0), 0.5), (9.0, 8.0, 7.0))
        light(sb.disk((-2.0, 5.0, 1.0), (0, -1, 0), (0.9, 0, 0)), (6.0, 6.0, 9.0))
        light(sb.triangle((1, 5.5, 2), (1, 5.5, 0), (-1, 5.5, 1)), (8.0, 8.0, 8.0))
        sb.area_light((5.0, 5.0, 5.0), sb.quad((-1, 6, -1), (2, 0, 0), (0, 0, 2)))
        sb.point_light((0, 3, -3), (10, 10, 10))
    elif lights == "disks":
        light(sb.disk((-2.0, 5.0, 1.0), (0, -1, 0), (0.9, 0, 0)), (6.0, 6.0, 9.0))
        light(sb.disk((2.0, 4.0, -1.0), (0, -1, 0), (0.6, 0, 0)), (9.0, 7.0, 5.0))
    elif lights == "delta_only":
        sb.point_light((0, 3, -3), (20, 20, 20))
        sb.distant_light((0.3, -1.0, 0.4), (1.5, 1.5, 1.2), 12.0)
    else:
        sb.env = (0.4, 0.5, 0.7)
    sb.set_camera(72, 48, deg(55.0), (0, 2.5, -7), (0, 1, 0))
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, st = gpu_ctx.render(2, 2, 6, 23, counters=True)
    ref, ost = OracleScene(sb).render(2, 2, 6, 23)
    assert ost["tlas_ties"] == 0 and st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    nan = np.isnan(ref)
    assert (nan == np.isnan(img)).all() and (bits(img)[~nan] == bits(ref)[~nan]).all()


@pytest.mark.parametrize("n_objects,n_lights", [(9, 9), (10, 9), (15, 15), (16, 15)])
def test_scan_limits_of_the_two_traversal_kernels(gpu_ctx, n_objects, n_lights):
    """C5's scene family at 20, 21, 32 and 33 instances: k_extend scans the TLAS leaves up to 20 instances, k_shadow up to 32,
    beyond that each walks the tree — the image is the oracle's at every size."""
    sb = scenes.many_lights_scene(width=72, height=40, n_objects=n_objects, n_lights=n_lights)
    hs = pbrs_amd.HostScene(sb)
    assert hs.desc.n_instances == n_objects + n_lights + 2
    gpu_ctx.upload(hs)
    img, st = gpu_ctx.render(2, 2, 8, 5, counters=True)
    ref, ost = OracleScene(sb).render(2, 2, 8, 5)
    assert ost["tlas_ties"] == 0
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert (bits(img) == bits(ref)).all()


@pytest.mark.parametrize("textured", [False, True])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_parallel_quad_instances_match_oracle(gpu_ctx, textured, integrator):
    sb = _parallel_quad_scene(textured)
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = osc.render(2, 2, 6, 17, integrator=integrator)
    img, st = gpu_ctx.render(2, 2, 6, 17, integrator=integrator, counters=True)
    assert st["quads"] > 0 and ost["tlas_ties"] == 0
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    nan = np.isnan(ref)
    assert (nan == np.isnan(img)).all()
    assert (bits(img)[~nan] == bits(ref)[~nan]).all()
    # quads fill a good part of the frame: their pixels differ from a render without them
    if integrator == "path" and not textured:
        h0, _, _ = osc.intersect(*osc.camera_rays(0, 1, 1, 17), np.full(80 * 56, np.inf, dtype=np.float32))
        kinds = sb.build()
        quad_insts = [i for i in range(kinds.n_instances) if kinds.shapes[kinds.instances[i].shape].kind == 1]
        assert np.isin(h0["inst"], quad_insts).mean() > 0.05


def _specular_scene(variant):
    """Mirror sphere, glass sphere, mirror quad mesh and a dielectric-coated (uber: transmit + reflect) box in front of the
    camera, so that direct_lighting_integrator's specular arm (src/directlighting.rs:31-41) is taken for many pixels."""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    sb = SceneBuilder()
    floor = sb.lambertian((0.5, 0.45, 0.4))
    sb.instance(scenes.quad_mesh(sb, (-8, 0, -8), (8, 0, -8), (-8, 0, 8), (8, 0, 8), (0, 1, 0)), floor)
    sb.instance(scenes.quad_mesh(sb, (-8, 0, 6), (8, 0, 6), (-8, 8, 6), (8, 8, 6), (0, 0, -1)), sb.lambertian((0.2, 0.5, 0.7)))
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.mirror((0.9, 0.9, 0.9)), Transform.translater((-2.4, 1.0, 0.5)))
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.dielectric(1.5), Transform.translater((0.0, 1.0, -0.5)))
    sb.instance(scenes.quad_mesh(sb, (1.5, 0.2, 2), (4.5, 0.2, 1), (1.5, 3.2, 2), (4.5, 3.2, 1), (-0.316, 0, -0.949)), sb.mirror((0.8, 0.85, 0.9)))
    sb.instance(sb.cuboid((-0.6, 0, -0.6), (0.6, 1.2, 0.6)), sb.uber((0.3, 0.1, 0.1), (0.2, 0.2, 0.2), kr=(0.6, 0.6, 0.6), kt=(0.5, 0.5, 0.5),
                                                                       rough=(0.1, 0.1), eta=1.3, opacity=0.6),
                Transform().rotate_y(deg(25)).translate((2.2, 0, -1.5)))
    e = (9.0, 8.0, 7.0)
    if variant == 0:      # one triangle light: lone shadow rays
        t = sb.triangle((-1, 5, -1), (1, 5, 1), (1, 5, -1))
        sb.instance(t, sb.diffuse_light(e))
        sb.area_light(e, t)
    elif variant == 1:    # sphere lights: two-ray (both MIS terms) estimates behind the specular lobe
        for k in range(2):
            s = sb.sphere((-2.5 + 5 * k, 4.5, 0.5), 0.7)
            sb.instance(s, sb.diffuse_light(e))
            sb.area_light(e, s)
    else:                 # delta lights + environment (no emitter surfaces: the env branch asserts non-empty lobes, :81)
        sb.point_light((1, 4, -2), (40, 40, 35))
        sb.distant_light((0.3, -1.0, 0.4), (1.5, 1.5, 1.2), 12.0)
        sb.env = (0.2, 0.3, 0.5)
    sb.set_camera(96, 64, deg(55.0), (0.3, 2.2, -7), (0, 1, 0))
    return sb


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_direct_lighting_integrator_matches_oracle(gpu_ctx, variant):
    """The reference's other integrator behind the same seam (src/main.rs:160-163): emission, or the one-light estimate
    plus one level of perfect specular reflection / refraction; same kernels, PBRS_INTEGRATOR_DIRECT."""
    sb = _specular_scene(variant)
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = osc.render(2, 2, 5, 3, integrator="direct")
    img, st = gpu_ctx.render(2, 2, 5, 3, integrator="direct", counters=True)
    assert ost["panics"] == 0 and ost["tlas_ties"] == 0
    n_primary = 96 * 64 * 4
    assert ost["closest_rays"] > n_primary + n_primary // 20, "the specular arm is taken"
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"] and st["shade_events"] == ost["shade_events"]
    assert l2(img, ref) < L2_TOL
    assert (bits(img) == bits(ref)).all()
    # depth only gates it (directlighting.rs:15-17); it is a different estimator from the path integrator
    img1, _ = gpu_ctx.render(2, 2, 1, 3, integrator="direct")
    assert (bits(img1) == bits(img)).all()
    img0, _ = gpu_ctx.render(2, 2, 0, 3, integrator="direct")
    assert not img0.any()
    path, _ = gpu_ctx.render(2, 2, 5, 3)
    assert (bits(path) != bits(img)).any()


@pytest.mark.parametrize("cfg,w,h", [("c1", 96, 96), ("c2", 96, 96), ("c5", 96, 54)])
def test_direct_lighting_integrator_on_baseline_scenes(gpu_ctx, cfg, w, h):
    sb, _ = scenes.build_config(cfg, width=w, height=h)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = OracleScene(sb).render(2, 2, 5, 9, integrator="direct")
    img, _ = gpu_ctx.render(2, 2, 5, 9, integrator="direct", samples_per_pass=3)  # uneven passes
    assert ost["panics"] == 0
    assert (bits(img) == bits(ref)).all()


def _textured_scene(env):
    """Checker floor, Perlin-marble sphere, an Uber sphere with an image kd / checker kr (lobes that appear and vanish per
    hit), an image-textured rotated box, one sphere light; environment: None, "image" or a sky closure."""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    sb = SceneBuilder()
    ck = sb.checker((0.0, 0.0, 0.0), (0.9, 0.9, 0.9))  # odd squares are black: Uber drops the lobe there
    pl = sb.perlin(4.0, seed=3)
    rs = np.random.RandomState(0)
    img = sb.image(rs.rand(16, 32, 3))
    sb.instance(scenes.quad_mesh(sb, (-8, 0, -8), (8, 0, -8), (-8, 0, 8), (8, 0, 8), (0, 1, 0)), sb.lambertian(sb.checker((0.1, 0.1, 0.1), (0.9, 0.8, 0.7))))
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.lambertian(pl), Transform.translater((-2.2, 1, 0)))
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.uber(img, (0.2, 0.2, 0.2), kr=ck, rough=(0.1, 0.1), eta=1.4), Transform.translater((0.3, 1, 0.5)))
    sb.instance(sb.cuboid((-0.7, 0, -0.7), (0.7, 1.6, 0.7)), sb.lambertian(img), Transform().rotate_y(deg(30)).translate((2.6, 0, 0)))
    sb.instance(sb.disk((0, 0, 0), (0, 0, -1), (0.8, 0, 0)), sb.uber(ck, img, rough=(0.2, 0.1), eta=1.5), Transform.translater((-0.8, 1.2, 2.5)))
    if env == "image":
        sb.env_image(sb.image(rs.rand(8, 16, 3)), (0.8, 0.9, 1.0))
    elif env is not None:
        sb.env_sky(env)
    s = sb.sphere((0, 5, 0), 0.7)
    e = (8.0, 8.0, 8.0)
    sb.instance(s, sb.diffuse_light(e))
    sb.area_light(e, s)
    sb.set_camera(96, 72, deg(55.0), (0, 2.5, -7), (0, 1, 0))
    return sb


def _material_zoo():
    """One instance of every material kind in front of nothing, so that the visualiser's checker shows too."""
    from pbrs_amd.spec import SceneBuilder, Transform, deg
    sb = SceneBuilder()
    mats = [sb.lambertian((0.6, 0.5, 0.4)), sb.metal((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), 0.1), sb.glossy((0.7, 0.7, 0.7), 0.2),
            sb.mirror((0.9, 0.9, 0.9)), sb.plastic((0.3, 0.5, 0.2), (0.4, 0.4, 0.4), 0.1), sb.dielectric(1.5),
            sb.diffuse_light((4, 4, 4)), sb.uber(kd=(0.3, 0.3, 0.5), ks=(0.2, 0.2, 0.2)), sb.substrate((0.4, 0.2, 0.2), (0.3, 0.3, 0.3))]
    for k, m in enumerate(mats):
        sb.instance(sb.sphere((0, 0, 0), 0.45), m, Transform.translater((-2.0 + 1.0 * (k % 5), 0.6 - 1.2 * (k // 5), 0.0)))
    sb.point_light((0, 4, -4), (30, 30, 30))
    sb.set_camera(120, 72, deg(50.0), (0.0, 0.0, -6.0), (0, 0, 0))
    return sb


@pytest.mark.parametrize("cfg", ["zoo", "c3", "c5"])
def test_normal_visualizer_matches_oracle(gpu_ctx, cfg):
    """`--visualize-normals` (src/directlighting.rs:273-289): (the colour `Material::scatter` returns + the hit's normal) / 2
    per pixel, the environment where nothing is hit.  Covers every `scatter` that returns (a textured Lambertian among
    them), the Dielectric's random choice between its two colours (drawn from the pixel's RNG stream), and the four
    materials whose `scatter` is `todo!()` (their colour counts as black)."""
    if cfg == "zoo":
        sb = _material_zoo()
        from pbrs_amd.spec import Transform
        sb.instance(sb.sphere((0, 0, 0), 0.45), sb.lambertian(sb.checker((0.9, 0.2, 0.2), (0.1, 0.1, 0.8))), Transform.translater((2.0, -0.6, 0.0)))
        sb.env_sky(pbrs_amd.spec.ENV_DUSK)
    else:
        sb = scenes.build_config(cfg, width=96, height=54)[0]
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, st = gpu_ctx.render(1, 1, 0, 5, integrator="normals", counters=True)
    ref, ost = OracleScene(sb).render(1, 1, 0, 5, integrator="normals")
    assert (bits(img) == bits(ref)).all()
    assert st["closest_rays"] == ost["closest_rays"] == img.shape[0] * img.shape[1] and st["shadow_rays"] == 0
    if cfg == "zoo":
        assert ost["panics"] > 100 and len(np.unique(img.reshape(-1, 3), axis=0)) > 1000
    with pytest.raises(pbrs_amd.PbrsError):
        gpu_ctx.render(1, 2, 0, 1, integrator="normals")


@pytest.mark.parametrize("cfg", ["zoo", "c3", "c5"])
def test_material_visualizer_matches_oracle(gpu_ctx, cfg):
    """`--visualize-materials` (src/main.rs:166-187, src/directlighting.rs:234-271): one un-jittered ray per pixel, a
    palette colour per kind of material, a grey checker of the ray direction where nothing is hit."""
    sb = _material_zoo() if cfg == "zoo" else scenes.build_config(cfg, width=96, height=54)[0]
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, st = gpu_ctx.render(1, 1, 0, 1, integrator="materials", counters=True)
    ref, ost = OracleScene(sb).render(1, 1, 0, 1, integrator="materials")
    assert (bits(img) == bits(ref)).all()
    assert st["closest_rays"] == ost["closest_rays"] == img.shape[0] * img.shape[1] and st["shadow_rays"] == 0
    colours = {tuple(np.round(c * 255).astype(int)) for c in img.reshape(-1, 3)}
    if cfg == "zoo":  # all nine kinds (Glossy has no arm: black; Lambertian shares its grey with the checker) and both checker greys
        assert len(colours) == 10 and (0, 0, 0) in colours and {(230, 230, 230), (178, 178, 178)} <= colours
    with pytest.raises(pbrs_amd.PbrsError):
        gpu_ctx.render(2, 2, 0, 1, integrator="materials")


@pytest.mark.parametrize("env", [None, "image", 2, 3, 4])
@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_textures_and_environment_lights_match_oracle(gpu_ctx, env, integrator):
    """texture/src/lib.rs (Checker, Perlin marble, nearest-neighbour Image) feeding Lambertian and Uber, and the Image / Fn
    environment lights (scene/src/lib.rs:105-117, scene/src/preset.rs:25-53): k_shade<.., TEX = true> against the oracle."""
    sb = _textured_scene(env)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = OracleScene(sb).render(2, 2, 6, 4, integrator=integrator)
    img, st = gpu_ctx.render(2, 2, 6, 4, integrator=integrator, counters=True)
    assert ost["tlas_ties"] == 0
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    nan_ref, nan_gpu = np.isnan(ref), np.isnan(img)
    assert (nan_ref == nan_gpu).all()
    assert (bits(img)[~nan_ref] == bits(ref)[~nan_ref]).all()
    assert np.nanstd(ref) > 0.01


def test_tiles_passes_and_bands_do_not_change_the_image(gpu_ctx):
    """The RNG is keyed by film pixel and sample index: any tiling, any samples_per_pass and any GPU count give
    the same bits (the multi-GPU correctness argument)."""
    sb, _ = scenes.build_config("c3", width=64, height=48)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    full, _ = gpu_ctx.render(3, 2, 8, 5)
    for spp_pass in (1, 4, 6):
        img, _ = gpu_ctx.render(3, 2, 8, 5, samples_per_pass=spp_pass)
        assert (bits(img) == bits(full)).all()
    tiles = np.empty_like(full)
    for (x0, y0, w, h) in ((0, 0, 33, 17), (33, 0, 31, 17), (0, 17, 64, 31)):
        tiles[y0:y0 + h, x0:x0 + w] = gpu_ctx.render(3, 2, 8, 5, tile=(x0, y0, w, h))[0]
    assert (bits(tiles) == bits(full)).all()
    for world in (2, 3, 8):
        shares = [tiling.render_share(lambda tile, bands: gpu_ctx.render(3, 2, 8, 5, tile=tile, bands=bands)[0], 64, 48, world, r)
                  for r in range(world)]
        assert (bits(tiling.assemble(shares, 64, 48, world)) == bits(full)).all()


@pytest.mark.parametrize("w,h", [(104, 72), (100, 70), (128, 64)])
def test_slot_order_of_a_pass_does_not_change_the_image(gpu_ctx, w, h):
    """Pass slots go by chunks of 4 096 pixels (each with its K samples) over 8 x 8 pixel blocks when the tile's sides are
    multiples of 8 (kernels.h, sample_of_slot / pixel_of_order): more than one chunk with a shorter last one, sides that are
    and are not multiples of 8, passes of uneven size — the image is the oracle's bit for bit, whatever the slot order."""
    sb, c = scenes.build_config("c2", width=w, height=h)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = OracleScene(sb).render(3, 3, 5, 7)
    for spp_pass in (0, 4, 9):
        img, st = gpu_ctx.render(3, 3, 5, 7, samples_per_pass=spp_pass, counters=True)
        assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert (bits(img) == bits(ref)).all(), (w, h, spp_pass)
    tile = (8, 8, w - 16, h - 8)  # an offset tile of the same film
    img, _ = gpu_ctx.render(3, 3, 5, 7, tile=tile)
    assert (bits(img) == bits(ref[8:, 8:w - 8])).all()


def test_api_errors(gpu_ctx):
    sb, _ = golden_case("c1_sphere_light")
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    with pytest.raises(pbrs_amd.PbrsError, match="outside the film"):
        gpu_ctx.render(1, 1, 2, 1, tile=(40, 40, 16, 16))
    with pytest.raises(pbrs_amd.PbrsError, match="zero strata"):
        gpu_ctx.render(0, 1, 2, 1)
    fresh = pbrs_amd.Context(0)
    fresh.scene = gpu_ctx.scene
    with pytest.raises(pbrs_amd.PbrsError, match="no scene"):
        fresh.render(1, 1, 2, 1)
    fresh.close()


def test_failed_allocation_leaves_a_usable_context():
    """A pass whose path state cannot be allocated fails with PBRS_E_DEVICE and leaves the context without a working set
    (no stale capacities, no pointers into freed memory): the next, smaller render allocates afresh and is still bit-exact."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")  # the runtime the library itself runs on
    sb, (w, h, sx, sy, depth) = golden_case("c2_cornell_diffuse")
    ctx = pbrs_amd.Context(0)
    hog = ctypes.c_void_p()
    try:
        ctx.upload(pbrs_amd.HostScene(sb))
        img0, _ = ctx.render(sx, sy, depth, SEED)  # a working set exists before the failure
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        assert hip.hipMalloc(ctypes.byref(hog), ctypes.c_size_t(max(free.value - (3 << 30), 1 << 20))) == 0  # leave about 3 GiB
        big = pbrs_amd.HostScene(scenes.build_config("c2", width=2048, height=2048)[0])
        ctx.upload(big)
        with pytest.raises(pbrs_amd.PbrsError, match="hipMalloc"):
            ctx.render(8, 8, depth, SEED, samples_per_pass=60)  # 2048 x 2048 x 60 paths x ~280 B = 70 GB
        assert hip.hipFree(hog) == 0
        hog = ctypes.c_void_p()
        ctx.upload(pbrs_amd.HostScene(sb))
        img1, _ = ctx.render(sx, sy, depth, SEED)
        assert (bits(img0) == bits(img1)).all()
        assert (bits(img1) == bits(load_golden("c2_cornell_diffuse")["image"])).all()
    finally:
        if hog:
            hip.hipFree(hog)
        ctx.close()


# ---- BASELINE.json full sizes: properties that need no CPU reference -------------------------------------------------------------

def test_full_size_c2_properties(gpu_ctx):
    """Cornell diffuse at 1024x1024 (spp reduced to 16 to keep the test short): determinism, linearity in the
    emitted radiance (x2 is exact in binary floating point), energy sanity, and agreement of a sub-tile with
    the oracle."""
    sb, cfg = scenes.build_config("c2")
    hs = pbrs_amd.HostScene(sb)
    gpu_ctx.upload(hs)
    a, st = gpu_ctx.render(4, 4, cfg["depth"], 1, counters=True)
    b, _ = gpu_ctx.render(4, 4, cfg["depth"], 1)
    assert (bits(a) == bits(b)).all()
    assert np.isfinite(a).all() and (a >= 0).all()
    assert st["samples"] == 1024 * 1024 * 16 and st["closest_rays"] >= st["samples"]
    # doubling every emitter doubles every pixel exactly
    sb2, _ = scenes.build_config("c2")
    for m in sb2.materials:
        if m.kind == pbrs_amd.spec.MTL_DIFFUSE_LIGHT:
            for i in range(3):
                m.p[i] *= 2.0
    for al in sb2.area_lights:
        for i in range(3):
            al.emit[i] *= 2.0
    gpu_ctx.upload(pbrs_amd.HostScene(sb2))
    c, _ = gpu_ctx.render(4, 4, cfg["depth"], 1)
    assert (bits(c) == bits(a * np.float32(2.0))).all()
    # a 64x16 window of the full-size frame against the oracle
    ref, _ = OracleScene(sb).render(4, 4, cfg["depth"], 1, tile=(480, 500, 64, 16))
    assert (bits(a[500:516, 480:544]) == bits(ref)).all()


def test_full_size_c4_million_triangle_window(gpu_ctx):
    """The 1 048 576-triangle scene at 1920x1080: a 96x8 window at 4 spp against the oracle (deep BLAS, LDS
    stack depth from the host), and determinism of the same window."""
    sb, cfg = scenes.build_config("c4")
    hs = pbrs_amd.HostScene(sb)
    assert hs.desc.n_triangles == 1048576 + 2
    gpu_ctx.upload(hs)
    tile = (900, 600, 96, 8)
    a, st = gpu_ctx.render(2, 2, cfg["depth"], 1, tile=tile, counters=True)
    b, _ = gpu_ctx.render(2, 2, cfg["depth"], 1, tile=tile)
    assert (bits(a) == bits(b)).all()
    ref, ost = OracleScene(sb).render(2, 2, cfg["depth"], 1, tile=tile)
    assert ost["panics"] == 0 and ost["tlas_ties"] == 0
    assert (bits(a) == bits(ref)).all()
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]


@pytest.mark.parametrize("cfg,windows", [
    ("c3", [(300, 700, 64, 8), (40, 200, 64, 8)]),           # glass sphere / plastic box region, mirror on the red wall
    ("c4", [(900, 600, 64, 8)]),
    ("c5", [(1800, 1400, 64, 8), (600, 300, 64, 8)]),        # objects on the floor, lights
])
def test_full_size_frames_against_oracle_windows(gpu_ctx, cfg, windows):
    """BASELINE sizes (1024x1024, 1920x1080 with 1 M triangles, 3840x2160 with 130 instances) as whole frames at 4 spp:
    finite, deterministic across pass sizes, and equal to the oracle bit for bit wherever the oracle is evaluated."""
    sb, c = scenes.build_config(cfg)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    a, st = gpu_ctx.render(2, 2, c["depth"], 3, counters=True)
    assert a.shape == (c["height"], c["width"], 3)
    assert np.isfinite(a).all()
    assert st["samples"] == c["width"] * c["height"] * 4
    b, _ = gpu_ctx.render(2, 2, c["depth"], 3, samples_per_pass=1)
    assert (bits(a) == bits(b)).all()
    osc = OracleScene(sb)
    for (x0, y0, w, h) in windows:
        ref, ost = osc.render(2, 2, c["depth"], 3, tile=(x0, y0, w, h))
        assert ost["panics"] == 0
        assert (bits(a[y0:y0 + h, x0:x0 + w]) == bits(ref)).all(), (cfg, x0, y0)


# ---- every BASELINE config at its OWN strata and spp ---------------------------------------------------------------------------
# The stratum of sample i is (i / strata_y, i % strata_y) (src/main.rs:198-201 generalised): sample indices beyond 15, the 16 x 16 /
# 32 x 32 / 32 x 16 / 64 x 64 mappings and the sum of hundreds to thousands of samples per pixel in sample order are compared here
# with the oracle bit for bit, window by window (a window is a tile of the full-size film: same camera, same RNG keys).

FULL_STRATA_WINDOWS = [
    ("c2", (480, 500, 32, 8)),    # 16 x 16 = 256 spp
    ("c3", (300, 700, 16, 8)),    # 32 x 32 = 1024 spp: glass sphere / plastic box region
    ("c4", (900, 600, 32, 8)),    # 32 x 16 = 512 spp on the 1 M-triangle mesh
    ("c5", (1800, 1400, 8, 4)),   # 64 x 64 = 4096 spp, 130 instances, 64 lights
]


@pytest.mark.parametrize("cfg,window", FULL_STRATA_WINDOWS)
def test_configs_at_their_own_strata_match_oracle(gpu_ctx, cfg, window):
    sb, c = scenes.build_config(cfg)
    sx, sy, depth = c["strata_x"], c["strata_y"], c["depth"]
    assert (sx, sy) == {"c2": (16, 16), "c3": (32, 32), "c4": (32, 16), "c5": (64, 64)}[cfg]
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    img, st = gpu_ctx.render(sx, sy, depth, 1, tile=window, counters=True)  # automatic samples_per_pass
    ref, ost = OracleScene(sb).render(sx, sy, depth, 1, tile=window)
    assert ost["panics"] == 0 and ost["tlas_ties"] == 0
    assert st["samples"] == ost["samples"] == window[2] * window[3] * sx * sy
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert (bits(img) == bits(ref)).all(), cfg
    assert float(ref.mean()) > 0.0
    # the same window in several passes of an odd size: the pass boundaries fall inside strata rows
    img2, _ = gpu_ctx.render(sx, sy, depth, 1, tile=window, samples_per_pass=103)
    assert (bits(img2) == bits(ref)).all(), cfg


def test_c4_at_512_spp_in_chunk_ordered_passes(gpu_ctx):
    """C4 at its 32 x 16 strata on a tile of more than one 4096-pixel chunk (the second one partial), in passes of the size the
    full frame gets (K = 103): slot order = chunk, sample index, pixel in 8 x 8 blocks (kernels.h, sample_of_slot) — every
    sample must still land in its own pixel's sum, in sample order."""
    sb, c = scenes.build_config("c4")
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    tile = (840, 560, 128, 40)  # 5120 pixels: chunks of 4096 + 1024
    img, st = gpu_ctx.render(c["strata_x"], c["strata_y"], c["depth"], 1, tile=tile, samples_per_pass=103, counters=True)
    assert st["passes"] == 5
    ref, ost = OracleScene(sb).render(c["strata_x"], c["strata_y"], c["depth"], 1, tile=tile)
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    assert (bits(img) == bits(ref)).all()


# ---- the walks over four-wide nodes (device/wide.h) and the rays they refuse --------------------------------------------------

def test_wide_walk_kernels_and_their_slow_list_match_oracle(gpu_ctx):
    """A terrain mesh deep enough for k_shadow's wide walk (32 768 triangles, a scanned TLAS), lit by the scene's sphere lights
    AND by a distant light that shines straight down: `target.pos - world_radius * 2 * casting_dir - target.pos` is exactly zero
    in x and z (light/src/lib.rs:77-81), so every shadow ray towards it is outside the guarded range of the division-free box
    test — the wide-walk kernel hands those to the binary-walk kernel through its slow list, the others it traces itself.  Image,
    ray counts and invalid samples equal the oracle's; so do hit records and occlusion of synthetic rays with zero components,
    denormal components and origins outside the guarded range: the harness sends occlusion queries through the four-wide any-hit
    walk k_shadow runs for this scene (asserted through pbrs_last_intersect_info) and counts the rays that walk refused."""
    sb, c = scenes.build_config("c4", width=192, height=96, nx=128, nz=128)
    sb.distant_light((0.0, -1.0, 0.0), (1.5, 1.4, 1.3), 700.0)
    hs = pbrs_amd.HostScene(sb)
    assert hs.desc.n_triangles == 2 * 128 * 128 + 2
    gpu_ctx.upload(hs)
    osc = OracleScene(sb)
    ref, ost = osc.render(3, 3, c["depth"], 9)
    img, st = gpu_ctx.render(3, 3, c["depth"], 9)                    # the timed kernels: wide k_shadow + slow list
    cnt, stc = gpu_ctx.render(3, 3, c["depth"], 9, counters=True)    # the instrumented kernels: binary walks
    assert ost["tlas_ties"] == 0 and np.isfinite(ref).all()  # (the delta-light estimate has assert sites of its own: panics are not asserted 0)
    assert (bits(img) == bits(ref)).all()
    assert (bits(cnt) == bits(ref)).all()
    assert stc["closest_rays"] == ost["closest_rays"] and stc["shadow_rays"] == ost["shadow_rays"]
    assert st["invalid_samples"] == ost["nonfinite_samples"]
    # rays through the parity harness (every query through the walk its stage runs in the pipeline): axis-parallel, denormal and
    # huge direction components and tiny origin components (outside the guarded range: the wide walk's slow hand-off) among random ones
    rng = np.random.default_rng(4)
    n = 4096
    o = np.stack([rng.uniform(-90, 90, n), rng.uniform(20, 80, n), rng.uniform(20, 380, n)], axis=1).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[::7, 0] = 0.0
    d[::11, 2] = 0.0
    d[::5] = (0.0, -1.0, 0.0)
    d[3::13, 0] = 1e-42          # denormal
    d[4::17, 1] *= np.float32(2.0 ** 50)   # beyond 2^40
    o[6::19, 0] = 1e-30          # tiny but non-zero origin component
    refused = (d == 0).any(axis=1) | (np.abs(d) < 2.0 ** -40).any(axis=1) | (np.abs(d) > 2.0 ** 40).any(axis=1) | ((o != 0) & (np.abs(o) < 2.0 ** -60)).any(axis=1)
    tmax = np.where(rng.uniform(size=n) < 0.5, np.inf, rng.uniform(10, 300, n)).astype(np.float32)
    gh, gocc = gpu_ctx.intersect(o, d, tmax)
    info = gpu_ctx.last_intersect_info()
    assert info["wide_any"] == 1, info                      # occlusion went through AnyWalkW (tlas_any_wide), as in k_shadow
    assert info["slow_any"] >= int(refused.sum()) > 500, (info, int(refused.sum()))  # ... and its hand-off took at least the rays outside the range
    assert info["slow_any"] < n // 2, info                  # ... while most rays stayed on the wide walk
    oh, oocc, _ = osc.intersect(o, d, tmax)
    assert (gocc == oocc).all()
    assert 0.05 < gocc.mean() < 0.95
    assert (gh["inst"] == oh["inst"]).all() and (bits(gh["t"]) == bits(oh["t"])).all() and (gh["prim"] == oh["prim"]).all()
    assert (gh["inst"] != 0xffffffff).mean() > 0.3


def test_scene_outside_the_guarded_range_takes_the_full_step_kernels(gpu_ctx):
    """A round's further node steps are lean ones (kernels.h): they carry the division-free box test only.  A scene whose node
    coordinates leave that test's guarded range walks every ray on the literal divisions — the lean steps would sit idle — and
    gets the kernels with full further steps (PBRS_FEAT_FULL_STEPS, bit 5 of pbrs_stats::kernel_features_*); the same deep terrain
    with the offending height (2^-61: below the 2^-60 a coordinate may have) moved to zero gets the lean ones.  Both frames equal
    the oracle's bit for bit."""
    FULL = 32
    frames = {}
    for name, height in (("outside", 2.0 ** -61), ("inside", 0.0)):
        sb, c = scenes.build_config("c4", width=160, height=96, nx=128, nz=128)
        pos = sb.meshes[0][0]  # the terrain's vertex positions (pbrs_amd/spec.py, SceneBuilder.mesh): one height replaced
        pos[len(pos) // 2, 1] = height
        hs = pbrs_amd.HostScene(sb)
        gpu_ctx.upload(hs)
        img, st = gpu_ctx.render(2, 2, c["depth"], 21)
        ref, ost = OracleScene(sb).render(2, 2, c["depth"], 21)
        assert np.isfinite(ref).all() and (bits(img) == bits(ref)).all(), name
        assert st["invalid_samples"] == ost["nonfinite_samples"]
        assert st["kernel_features_extend"] & 8, st  # several node steps per round: the further steps exist
        frames[name] = st
    assert frames["outside"]["kernel_features_extend"] & FULL and frames["outside"]["kernel_features_shadow"] & FULL, frames["outside"]
    assert not (frames["outside"]["kernel_features_shadow"] & 16)  # no four-wide k_shadow without the division-free test
    assert not (frames["inside"]["kernel_features_extend"] & FULL) and not (frames["inside"]["kernel_features_shadow"] & FULL), frames["inside"]
    assert frames["inside"]["kernel_features_shadow"] & 16


def test_small_scenes_are_walked_from_lds_and_say_so(gpu_ctx):
    """Round 4: the traversal kernels of a scene of a few KB read it from the block's LDS (PBRS_FEAT_LDS_SCENE, bit 6 of
    pbrs_stats::kernel_features_*), those of a scene whose TLAS is too large for the wave's scan read the TLAS from there
    (PBRS_FEAT_LDS_TOP, bit 7); a big scene gets neither.  Each against the oracle, bit for bit, with the counters' variant
    (which stages nothing) agreeing."""
    cases = [("c2", dict(width=96, height=96), 64, 0), ("c5", dict(width=128, height=72), 128, 64),
             ("c4", dict(width=96, height=64, nx=64, nz=64), 0, 64 | 128)]
    for name, kw, want, not_want in cases:
        sb, c = scenes.build_config(name, **kw)
        hs = pbrs_amd.HostScene(sb)
        gpu_ctx.upload(hs)
        img, st = gpu_ctx.render(2, 2, c["depth"], 17)
        cnt, stc = gpu_ctx.render(2, 2, c["depth"], 17, counters=True)
        ref, ost = OracleScene(sb).render(2, 2, c["depth"], 17)
        for key in ("kernel_features_extend", "kernel_features_shadow"):
            assert st[key] & want == want and not (st[key] & not_want), (name, key, st[key])
            assert stc[key] & 0x80000000, (name, key, stc[key])
        assert (bits(img) == bits(ref)).all() and (bits(cnt) == bits(ref)).all(), name
        assert stc["closest_rays"] == ost["closest_rays"] and stc["shadow_rays"] == ost["shadow_rays"], name


def test_passes_overlapped_on_two_streams_render_the_same_frame(gpu_ctx):
    """A render of several passes hands each pass's late bounces to a second stream, beside the next pass's first bounces
    (pbrs_set_pass_overlap, on by default; two sets of per-pass memory).  The passes' samples reach the pixel sums in pass order
    whatever the streams do: five passes of two samples (and a last one of one) give the oracle's frame bit for bit, overlapped or
    not, on a scene that hands over at bounce 2 (C3) and on one that does at bounce 4 (a terrain beyond one XCD's L2), for both
    integrators; per-stage times are filled either way."""
    for name, kw in (("c3", dict(width=96, height=64)), ("c4", dict(width=96, height=64, nx=192, nz=192))):
        sb, c = scenes.build_config(name, **kw)
        hs = pbrs_amd.HostScene(sb)
        gpu_ctx.upload(hs)
        osc = OracleScene(sb)
        for integrator in ("path", "direct"):
            ref, _ = osc.render(3, 3, c["depth"], 23, integrator=integrator)
            try:
                on, st_on = gpu_ctx.render(3, 3, c["depth"], 23, samples_per_pass=2, timing=True, integrator=integrator)
                gpu_ctx.set_pass_overlap(False)
                off, st_off = gpu_ctx.render(3, 3, c["depth"], 23, samples_per_pass=2, timing=True, integrator=integrator)
            finally:
                gpu_ctx.set_pass_overlap(True)
            one, _ = gpu_ctx.render(3, 3, c["depth"], 23, integrator=integrator)  # a single pass
            assert st_on["passes"] == st_off["passes"] == 5
            assert (bits(on) == bits(ref)).all() and (bits(off) == bits(ref)).all() and (bits(one) == bits(ref)).all(), (name, integrator)
            assert st_on["ms_total"] > 0 and st_off["ms_extend"] > 0 and st_on["invalid_samples"] == st_off["invalid_samples"]
    # consecutive overlapped frames through the same two sets of memory
    a, _ = gpu_ctx.render(3, 3, c["depth"], 5, samples_per_pass=4)
    b, _ = gpu_ctx.render(3, 3, c["depth"], 5, samples_per_pass=4)
    assert (bits(a) == bits(b)).all()
