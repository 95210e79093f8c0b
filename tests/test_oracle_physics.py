"""The oracle's INTEGRATORS against a closed form.  The reference holds no integrator-level fixture, so the light estimate,
its MIS weights and the bounce loop are pinned by their parts only — except where radiometry itself gives the answer: a Lambertian plane
under a spherical area light.  A point x of the plane sees the sphere (radius r, centre c, radiance Le, wholly above the horizon) under
irradiance E = pi Le (r / |c - x|)^2 cos(theta), theta between the plane's normal and c - x, and leaves L = rho / pi * E toward the
camera; the plane does not see itself, so the path integrator at any depth and the direct-lighting integrator must both converge to it.
They do, to the noise of 4 096 samples a pixel — which pins `uniform_sample_one_light`, `estimate_direct_area_light` (both MIS terms,
pdfs in solid angle, the cone sampling of light/src/sample_shape.rs:197-250), the Lambertian lobe and the bounce loop's bookkeeping
(emission only at the first or a specular vertex) as a whole.

The same scene under a TRIANGLE light converges to |c - x| times the closed form: the reference's default `pdf_at`
(light/src/sample_shape.rs:28-33) divides by the distance where the solid-angle measure has its square (Q4).  The
oracle keeps it, as the GPU side does: what is pinned here is that the checker follows the reference, not radiometry.  No GPU needed."""
import numpy as np
import pytest

from oracle.binding import OracleScene
from pbrs_amd import scenes
from pbrs_amd.spec import SceneBuilder, deg

RHO = (0.6, 0.4, 0.8)
LE = (5.0, 7.0, 3.0)
C = np.array((0.3, 3.0, -0.4))
R = 0.5


def _scene(light, target):
    sb = SceneBuilder()
    sb.instance(scenes.quad_mesh(sb, (-50, 0, -50), (50, 0, -50), (-50, 0, 50), (50, 0, 50), (0, 1, 0)), sb.lambertian(RHO))
    if light == "sphere":
        s = sb.sphere(tuple(C), R)
    else:
        s = sb.triangle((C[0] + 1, C[1], C[2] + 1), (C[0] + 1, C[1], C[2] - 1), (C[0] - 1, C[1], C[2]))
    sb.instance(s, sb.diffuse_light(LE))
    sb.area_light(LE, s)
    sb.set_camera(5, 5, deg(0.3), (target[0] + 0.5, 1.5, target[2] - 4.0), target)  # 25 pixels on a centimetre of the plane around the target
    return sb


def _mean_radiance(sb, integrator, depth):
    img, st = OracleScene(sb).render(64, 64, depth, 7, integrator=integrator)
    assert st["panics"] == 0 and st["nonfinite_samples"] == 0
    return img.reshape(-1, 3).astype(np.float64).mean(0)


@pytest.mark.parametrize("target", [(0.3, 0.0, -0.4), (2.0, 0.0, 1.0), (-3.0, 0.0, 0.5)])
def test_a_plane_under_a_sphere_light_converges_to_the_closed_form(target):
    d = C - np.array(target)
    dist = np.linalg.norm(d)
    want = np.array(RHO) / np.pi * (np.pi * np.array(LE) * (R / dist) ** 2 * (d[1] / dist))
    for integrator, depth in (("direct", 3), ("path", 1), ("path", 5)):
        got = _mean_radiance(_scene("sphere", target), integrator, depth)
        assert np.allclose(got / want, 1.0, atol=4e-3), (integrator, depth, got / want)  # measured 0.9999 .. 1.0025


def test_a_flat_light_carries_the_distance_of_the_reference_pdf():
    target = (0.3, 0.0, -0.4)  # straight below the triangle's plane, 3 away
    sb = _scene("triangle", target)
    # closed form by quadrature: E = Le * integral of cos cos' / d^2 over the triangle
    p = [np.array(v, dtype=np.float64) for v in ((C[0] + 1, C[1], C[2] + 1), (C[0] + 1, C[1], C[2] - 1), (C[0] - 1, C[1], C[2]))]
    n = 400
    u, v = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n)
    inside = u + v < 1
    pts = p[0] + (p[1] - p[0]) * u[inside][:, None] + (p[2] - p[0]) * v[inside][:, None]
    area = 0.5 * np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
    dv = pts - np.array(target)
    dist = np.linalg.norm(dv, axis=1)
    E = (dv[:, 1] / dist) ** 2 / dist ** 2  # light and plane are parallel: both cosines are dy / d
    want = np.array(RHO) / np.pi * np.array(LE) * E.mean() * area
    for integrator, depth in (("direct", 3), ("path", 5)):
        ratio = _mean_radiance(sb, integrator, depth) / want
        assert np.allclose(ratio, 3.0, atol=0.06), (integrator, ratio)  # Q4: measured 2.9994; radiometry says 1


def test_a_plane_under_a_constant_environment_returns_albedo_times_radiance():
    """The environment arm of `uniform_sample_one_light` (src/directlighting.rs:80-96): a cosine-sampled direction, f |cos| / pdf = rho for a
    Lambertian lobe whatever the sample, so every sample of every pixel is rho * Le up to rounding, at any depth (the continuation ray
    escapes at a non-specular vertex and adds nothing, src/pathintegrator.rs:19-22)."""
    env = (0.7, 0.9, 1.1)
    sb = SceneBuilder()
    sb.instance(scenes.quad_mesh(sb, (-50, 0, -50), (50, 0, -50), (-50, 0, 50), (50, 0, 50), (0, 1, 0)), sb.lambertian(RHO))
    sb.env = env
    sb.set_camera(8, 8, deg(20), (0.5, 2.0, -4.0), (0, 0, 0))
    for integrator, depth in (("direct", 3), ("path", 1), ("path", 6)):
        img, st = OracleScene(sb).render(4, 4, depth, 3, integrator=integrator)
        assert st["panics"] == 0
        assert np.allclose(img, np.array(RHO) * np.array(env), rtol=2e-6), (integrator, depth)


def test_a_plane_under_a_point_light_follows_the_inverse_square_law():
    """`estimate_direct_delta_light` (src/directlighting.rs:101-153) with DeltaLight::Point (light/src/lib.rs:29-103): no randomness but the
    film jitter; L = rho / pi * I cos(theta) / d^2 at the point the camera looks at."""
    inten = np.array((30.0, 20.0, 10.0))
    lp = np.array((1.0, 2.5, -0.5))
    for target in ((1.0, 0.0, -0.5), (-1.5, 0.0, 1.0)):
        sb = SceneBuilder()
        sb.instance(scenes.quad_mesh(sb, (-50, 0, -50), (50, 0, -50), (-50, 0, 50), (50, 0, 50), (0, 1, 0)), sb.lambertian(RHO))
        sb.point_light(tuple(lp), tuple(inten))
        sb.set_camera(5, 5, deg(0.05), (target[0] + 0.5, 1.5, target[2] - 4.0), target)
        d = lp - np.array(target)
        dist = np.linalg.norm(d)
        want = np.array(RHO) / np.pi * inten * (d[1] / dist) / dist ** 2
        for integrator, depth in (("direct", 3), ("path", 4)):
            got = _mean_radiance(sb, integrator, depth)
            assert np.allclose(got / want, 1.0, atol=2e-3), (target, integrator, got / want)
