"""Synthetic scenes for BASELINE.json's five configs (SURVEY.md §8d "Concrete synthetic inputs").

Everything is procedural and seeded (SplitMix64 stream); no asset files.  Geometry/material
constants follow scene/src/preset.rs (`cornell_box` :194-257, `plates` :259-358, metal IORs
:467-493) and the triangle-light pattern of scene/src/loader.rs:411-430.  ParallelQuad instances
are avoided because of the reference defects D1/D2: walls are
two-triangle `TriangleMesh` quads in the pattern of preset.rs:296-301.
"""
import numpy as np

from .spec import SceneBuilder, Transform, deg, f32

GOLD = ((0.143176, 0.373096, 1.443834), (3.982675, 2.387439, 1.602465))  # preset.rs:481-486
SILVER = ((0.155184, 0.116681, 0.138360), (4.828131, 3.122411, 2.147082))  # :467-472
COPPER = ((0.195470, 0.925682, 1.102186), (3.910869, 2.451263, 2.142653))  # :488-493


class SplitMix64:
    def __init__(self, seed):
        self.s = np.uint64(seed)

    def next_u64(self):
        with np.errstate(over="ignore"):
            self.s = self.s + np.uint64(0x9E3779B97F4A7C15)
            z = self.s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, lo=0.0, hi=1.0):
        u = float(int(self.next_u64()) >> 40) * (1.0 / (1 << 24))
        return lo + (hi - lo) * u


def quad_mesh(sb, t00, t01, t10, t11, normal):
    """Two-triangle TriangleMesh quad, pattern of scene/src/preset.rs:296-301."""
    return sb.mesh([t00, t01, t10, t11], [normal] * 4, [(0, 0), (0, 1), (1, 0), (1, 1)], [(0, 1, 2), (2, 1, 3)])


def box_mesh(sb, p0, p1):
    """Axis-aligned box [p0, p1] as 6 quad faces (24 vertices, 12 triangles), outward normals."""
    (x0, y0, z0), (x1, y1, z1) = p0, p1
    faces = [
        ((x0, y0, z0), (x0, y1, z0), (x0, y0, z1), (x0, y1, z1), (-1, 0, 0)),
        ((x1, y0, z0), (x1, y0, z1), (x1, y1, z0), (x1, y1, z1), (1, 0, 0)),
        ((x0, y0, z0), (x0, y0, z1), (x1, y0, z0), (x1, y0, z1), (0, -1, 0)),
        ((x0, y1, z0), (x1, y1, z0), (x0, y1, z1), (x1, y1, z1), (0, 1, 0)),
        ((x0, y0, z0), (x1, y0, z0), (x0, y1, z0), (x1, y1, z0), (0, 0, -1)),
        ((x0, y0, z1), (x0, y1, z1), (x1, y0, z1), (x1, y1, z1), (0, 0, 1)),
    ]
    pos, nrm, uv, idx = [], [], [], []
    for f, (a, b, c, d, n) in enumerate(faces):
        pos += [a, b, c, d]
        nrm += [n] * 4
        uv += [(0, 0), (0, 1), (1, 0), (1, 1)]
        idx += [(4 * f, 4 * f + 1, 4 * f + 2), (4 * f + 2, 4 * f + 1, 4 * f + 3)]
    return sb.mesh(pos, nrm, uv, idx)


def sphere_light_scene(width=256, height=256):
    """C1: one Lambertian sphere + one spherical area light (pattern of preset.rs:165-174)."""
    sb = SceneBuilder()
    grey = sb.lambertian((0.5, 0.5, 0.5))
    emit = (10.0, 10.0, 10.0)
    light_mtl = sb.diffuse_light(emit)
    sb.instance(sb.sphere((0, 0, 0), 1.0), grey)
    light_sphere = sb.sphere((0, 3, 0), 0.5)
    sb.instance(light_sphere, light_mtl)
    sb.area_light(emit, light_sphere)
    sb.set_camera(width, height, deg(40.0), (0, 1, -6), (0, 0, 0))
    return sb


def _cornell_shell(sb):
    """Walls, boxes and the two-triangle ceiling light of the Cornell box (preset.rs:196-244)."""
    red = sb.lambertian((0.65, 0.05, 0.05))
    white = sb.lambertian((0.73, 0.73, 0.73))
    green = sb.lambertian((0.12, 0.45, 0.15))
    light_color = (15.0, 15.0, 15.0)
    light = sb.diffuse_light(light_color)
    S = 555.0
    sb.instance(quad_mesh(sb, (S, 0, 0), (S, S, 0), (S, 0, S), (S, S, S), (-1, 0, 0)), green)   # x = 555
    sb.instance(quad_mesh(sb, (0, 0, 0), (0, 0, S), (0, S, 0), (0, S, S), (1, 0, 0)), red)       # x = 0
    sb.instance(quad_mesh(sb, (0, 0, 0), (S, 0, 0), (0, 0, S), (S, 0, S), (0, 1, 0)), white)     # floor
    sb.instance(quad_mesh(sb, (0, S, 0), (0, S, S), (S, S, 0), (S, S, S), (0, -1, 0)), white)    # ceiling
    sb.instance(quad_mesh(sb, (0, 0, S), (S, 0, S), (0, S, S), (S, S, S), (0, 0, -1)), white)    # back
    sb.instance(quad_mesh(sb, (0, 0, 0), (0, S, 0), (S, 0, 0), (S, S, 0), (0, 0, 1)), white)     # closes the box (z = 0)
    # ceiling light, x in [213,343], z in [227,332], y = 554: two IsolatedTriangles whose
    # ((p0-p1) x (p2-p1)) normal points down (light/src/sample_shape.rs:277-287, lib.rs:127-133).
    A, B, Cc, D = (213, 554, 227), (343, 554, 227), (343, 554, 332), (213, 554, 332)
    for tri in ((Cc, B, A), (A, D, Cc)):
        t = sb.triangle(*tri)
        sb.instance(t, light)
        sb.area_light(light_color, t)
    return red, white, green


def cornell_scene(width=1024, height=1024, variant="diffuse"):
    """C2 (variant='diffuse') and C3 (variant='specular': glass + gold spheres, mirror, plastic box)."""
    sb = SceneBuilder()
    red, white, green = _cornell_shell(sb)
    short_xf = Transform().rotate_y(deg(15.0)).translate((265.0, 0.0, 105.0))   # preset.rs:239-241
    tall_xf = Transform().rotate_y(deg(-18.0)).translate((130.0, 0.0, 225.0))   # preset.rs:242-244
    if variant == "diffuse":
        sb.instance(box_mesh(sb, (0, 0, 0), (165, 165, 165)), white, short_xf)
        sb.instance(box_mesh(sb, (0, 0, 0), (165, 330, 165)), white, tall_xf)
    else:
        plastic = sb.plastic((0.2, 0.3, 0.7), (0.5, 0.5, 0.5), 0.1, True)
        sb.instance(box_mesh(sb, (0, 0, 0), (165, 330, 165)), plastic, tall_xf)
        glass = sb.dielectric(1.5)
        gold = sb.metal(GOLD[0], GOLD[1], 0.05)
        mirror = sb.mirror((0.9, 0.9, 0.9))
        sb.instance(sb.sphere((0, 0, 0), 80.0), glass, Transform.translater((370.0, 80.5, 160.0)))
        sb.instance(sb.sphere((0, 0, 0), 70.0), gold, Transform.translater((150.0, 400.0, 330.0)))
        # mirror quad leaning on the red wall
        sb.instance(quad_mesh(sb, (2, 100, 150), (2, 400, 150), (2, 100, 450), (2, 400, 450), (1, 0, 0)), mirror)
    sb.set_camera(width, height, deg(65.0), (278, 278, 20), (278, 278, 555))
    return sb


def heightfield_mesh(sb, nx, nz, size, amp, seed):
    """(nx x nz) grid -> 2*nx*nz triangles; vertex normals by `compute_normals`
    (geometry/src/lib.rs:16-32, f32 accumulation in index order); uv = grid coordinates."""
    xs = np.linspace(0.0, size[0], nx + 1, dtype=np.float64)
    zs = np.linspace(0.0, size[1], nz + 1, dtype=np.float64)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    rng = np.random.RandomState(seed)
    Y = np.zeros_like(X)
    for _ in range(6):
        fx, fz = rng.uniform(2, 14, size=2)
        ph = rng.uniform(0, 2 * np.pi, size=2)
        Y += rng.uniform(0.3, 1.0) * np.sin(2 * np.pi * fx * X / size[0] + ph[0]) * np.sin(2 * np.pi * fz * Z / size[1] + ph[1])
    Y = amp * Y / 3.0
    pos = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(f32)
    i, j = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    v00 = (i * (nz + 1) + j).ravel()
    v01 = v00 + 1
    v10 = v00 + (nz + 1)
    v11 = v10 + 1
    idx = np.empty((2 * nx * nz, 3), dtype=np.uint32)
    idx[0::2] = np.stack([v00, v01, v10], axis=-1)
    idx[1::2] = np.stack([v10, v01, v11], axis=-1)
    # compute_normals: n = (p1-p0) x (p2-p0) added to the three vertices, then hat().
    p0, p1, p2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    fn = np.cross(p1 - p0, p2 - p0).astype(f32)
    nrm = np.zeros_like(pos)
    for k in range(3):
        np.add.at(nrm, idx[:, k], fn)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(f32)
    uv = np.stack([(X / size[0]).ravel(), (Z / size[1]).ravel()], axis=-1).astype(f32)
    return sb.mesh(pos, nrm, uv, idx)


def terrain_scene(width=1920, height=1080, nx=512, nz=1024, seed=4):
    """C4: one 2*nx*nz-triangle mesh (default 1 048 576) + floor mesh + 4 spherical lights."""
    sb = SceneBuilder()
    ground = sb.lambertian((0.55, 0.5, 0.4))
    floor_m = sb.lambertian((0.4, 0.4, 0.4))
    sx, sz = 200.0, 400.0
    sb.instance(heightfield_mesh(sb, nx, nz, (sx, sz), 12.0, seed), ground, Transform.translater((-sx / 2, 0.0, 0.0)))
    sb.instance(quad_mesh(sb, (-400, -14, -100), (400, -14, -100), (-400, -14, 600), (400, -14, 600), (0, 1, 0)), floor_m)
    rs = SplitMix64(seed)
    for k in range(4):
        c = (rs.uniform(-80, 80), rs.uniform(60, 90), rs.uniform(40, 360))
        emit = tuple(rs.uniform(8, 20) for _ in range(3))
        s = sb.sphere(c, rs.uniform(6, 12))
        sb.instance(s, sb.diffuse_light(emit))
        sb.area_light(emit, s)
    sb.set_camera(width, height, deg(45.0), (0, 60, -90), (0, 0, 160))
    return sb


def many_lights_scene(width=3840, height=2160, n_objects=64, n_lights=64, seed=5):
    """C5: mixed BSDFs on n_objects (spheres, cuboids, small meshes) + n_lights area lights
    (spheres and triangles, Le uniform in [2,20] per channel), NEE heavy."""
    sb = SceneBuilder()
    rs = SplitMix64(seed)
    floor_m = sb.lambertian((0.5, 0.5, 0.5))
    sb.instance(quad_mesh(sb, (-60, 0, -20), (60, 0, -20), (-60, 0, 140), (60, 0, 140), (0, 1, 0)), floor_m)
    sb.instance(quad_mesh(sb, (-60, 0, 140), (60, 0, 140), (-60, 60, 140), (60, 60, 140), (0, 0, -1)), floor_m)

    def rand_color(lo=0.05, hi=0.9):
        return tuple(rs.uniform(lo, hi) for _ in range(3))

    def rand_material(k):
        kind = k % 6
        if kind == 0:
            return sb.lambertian(rand_color())
        if kind == 1:
            eta, kk = (GOLD, SILVER, COPPER)[(k // 6) % 3]
            return sb.metal(eta, kk, rs.uniform(0.02, 0.3))
        if kind == 2:
            return sb.plastic(rand_color(), rand_color(0.2, 0.6), rs.uniform(0.05, 0.3), True)
        if kind == 3:
            return sb.dielectric(rs.uniform(1.3, 1.7))
        if kind == 4:
            return sb.mirror(rand_color(0.6, 0.95))
        return sb.glossy(rand_color(0.5, 0.9), rs.uniform(0.001, 0.05))

    for k in range(n_objects):
        gx, gz = k % 8, k // 8
        cx = -49.0 + 14.0 * gx + rs.uniform(-2, 2)
        cz = 6.0 + 15.0 * gz + rs.uniform(-2, 2)
        r = rs.uniform(2.5, 4.5)
        mtl = rand_material(k)
        which = k % 3
        if which == 0:
            sb.instance(sb.sphere((0, 0, 0), r), mtl, Transform.translater((cx, r + 0.01, cz)))
        elif which == 1:
            xf = Transform().rotate_y(deg(rs.uniform(0, 90))).translate((cx, 0.01, cz))
            sb.instance(sb.cuboid((-r, 0, -r), (r, 2 * r, r)), mtl, xf)
        else:
            xf = Transform().rotate_y(deg(rs.uniform(0, 90))).translate((cx, 0.01, cz))
            sb.instance(box_mesh(sb, (-r, 0, -r), (r, 1.5 * r, r)), mtl, xf)
    for k in range(n_lights):
        emit = tuple(rs.uniform(2, 20) for _ in range(3))
        lx = -52.0 + 104.0 * ((k % 8) + rs.uniform(0.1, 0.9)) / 8.0
        lz = 0.0 + 130.0 * ((k // 8) + rs.uniform(0.1, 0.9)) / 8.0
        ly = rs.uniform(22, 40)
        lm = sb.diffuse_light(emit)
        if k % 2 == 0:
            s = sb.sphere((lx, ly, lz), rs.uniform(0.5, 1.5))
        else:
            e = rs.uniform(1.0, 2.5)
            # normal ((p0-p1) x (p2-p1)) points down
            s = sb.triangle((lx + e, ly, lz + e), (lx + e, ly, lz), (lx, ly, lz))
        sb.instance(s, lm)
        sb.area_light(emit, s)
    sb.set_camera(width, height, deg(50.0), (0, 38, -45), (0, 4, 60))
    return sb


CONFIGS = {
    # name: (builder, kwargs, width, height, strata_x, strata_y, depth)
    "c1": (sphere_light_scene, {}, 256, 256, 4, 4, 4),
    "c2": (cornell_scene, {"variant": "diffuse"}, 1024, 1024, 16, 16, 8),
    "c3": (cornell_scene, {"variant": "specular"}, 1024, 1024, 32, 32, 8),
    "c4": (terrain_scene, {}, 1920, 1080, 32, 16, 8),
    "c5": (many_lights_scene, {}, 3840, 2160, 64, 64, 8),
    # not a BASELINE config: C4's generator on a 2048 x 4096 grid (16.8 M triangles, ~2.4 GB flattened: ten times the Infinity
    # Cache) at 64 spp — what the traversal kernels do when node fetches really go to HBM (bench.py --config c4xl)
    "c4xl": (terrain_scene, {"nx": 2048, "nz": 4096}, 1920, 1080, 8, 8, 8),
}


def build_config(name, width=None, height=None, **overrides):
    fn, kwargs, w, h, sx, sy, depth = CONFIGS[name]
    kw = dict(kwargs)
    kw.update(overrides)
    sb = fn(width=width or w, height=height or h, **kw)
    return sb, dict(width=width or w, height=height or h, strata_x=sx, strata_y=sy, depth=depth)
