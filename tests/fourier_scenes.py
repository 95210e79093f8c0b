"""Scenes with Fourier BSDF materials (geometry/src/fourier.rs) shared by the CPU and GPU tests."""
import numpy as np

from pbrs_amd import fourier, scenes
from pbrs_amd.spec import SceneBuilder, Transform, deg

TABLES = {}


def table(name):
    if name not in TABLES:
        TABLES[name] = {
            "rgb": lambda: fourier.project(fourier.lambert_like((0.7, 0.5, 0.3)), n_mu=16, m=10, n_channels=3),
            "mono": lambda: fourier.project(fourier.lambert_like((0.6, 0.6, 0.6)), n_mu=12, m=6, n_channels=1),
            "fine": lambda: fourier.project(fourier.lambert_like((0.4, 0.6, 0.8)), n_mu=40, m=24, n_channels=3, tol=1e-6),
            "translucent": lambda: fourier.project(fourier.translucent((0.5, 0.5, 0.5)), n_mu=16, m=8, n_channels=1),
            "lambert": lambda: fourier.project(lambda mi, mo, c: tuple(np.full_like(c, (0.5 / np.pi) if mi * mo < 0 else 0.0) for _ in range(3)),
                                               n_mu=32, m=2, n_channels=3),
        }[name]()
    return TABLES[name]


def scene(tables=("rgb",), lights="area", textured=False, size=(48, 32)):
    """A floor, a sphere, a box and a rotated quad mesh carrying the given tables' materials in turn."""
    sb = SceneBuilder()
    mats = [sb.fourier(sb.fourier_table(table(t))) for t in tables]
    floor = sb.lambertian(sb.checker((0.1, 0.1, 0.1), (0.8, 0.8, 0.8))) if textured else sb.lambertian((0.5, 0.5, 0.5))
    sb.instance(scenes.quad_mesh(sb, (-9, 0, -9), (9, 0, -9), (-9, 0, 9), (9, 0, 9), (0, 1, 0)), floor)
    sb.instance(sb.sphere((0, 1, 0), 1.0), mats[0])
    sb.instance(sb.cuboid((-0.5, 0, -0.5), (0.5, 1.2, 0.5)), mats[1 % len(mats)], Transform().rotate_y(deg(30)).translate((2.2, 0, 0.5)))
    sb.instance(scenes.quad_mesh(sb, (-1, 0, 0), (1, 0, 0), (-1, 1.5, 0.4), (1, 1.5, 0.4), (0, -0.26, 0.97)), mats[2 % len(mats)],
                Transform().rotate_y(deg(-25)).translate((-2.4, 0.0, 0.8)))
    sb.instance(sb.sphere((1.0, 0.5, -2.0), 0.5), sb.mirror((0.9, 0.9, 0.9)))
    if "area" in lights:
        e = (12.0, 11.0, 10.0)
        s = sb.sphere((0, 5, -1), 0.7)
        sb.instance(s, sb.diffuse_light(e))
        sb.area_light(e, s)
    if "point" in lights:
        sb.point_light((2, 5, -3), (40, 40, 40))
    if "env" in lights:
        sb.env = (0.4, 0.5, 0.7)
    sb.set_camera(size[0], size[1], deg(45), (0, 2.5, -7), (0, 1, 0))
    return sb
