"""The oracle's restatement of texture/src/lib.rs and of the environment lights (scene/src/lib.rs:105-117,
scene/src/preset.rs:25-53).  The reference holds no tests for these ("parity unpinned"): the checks below are the
properties its code implies, evaluated on literal inputs."""
import numpy as np

from oracle.binding import OracleScene, texture_value
from pbrs_amd import spec
from pbrs_amd.spec import SceneBuilder, deg

f32 = np.float32


def test_checker_follows_the_sign_of_the_three_sines():
    sb = SceneBuilder()
    t = sb.checker((0.1, 0.2, 0.3), (0.9, 0.8, 0.7))
    rs = np.random.RandomState(0)
    p = (rs.rand(2000, 3) * 4 - 2).astype(f32)
    out, panics = texture_value(sb.textures[t], np.zeros((2000, 2)), p)
    s = np.sin(10.0 * p.astype(np.float64)).prod(axis=1)
    clear = np.abs(s) > 1e-4  # away from the sign change the f32 kernels and libm agree
    odd = (out == f32([0.1, 0.2, 0.3])).all(axis=1)
    assert panics == 0 and (odd[clear] == (s[clear] < 0)).all()
    assert ((out == f32([0.1, 0.2, 0.3])).all(axis=1) | (out == f32([0.9, 0.8, 0.7])).all(axis=1)).all()


def test_image_is_nearest_texel_with_clamped_uv_and_wraps_at_one():
    sb = SceneBuilder()
    img = np.arange(4 * 8 * 3, dtype=f32).reshape(4, 8, 3)
    t = sb.image(img)
    uv = np.array([[0.0, 0.0], [0.124, 0.0], [0.126, 0.0], [0.99, 0.99], [1.0, 1.0], [-3.0, 0.5], [7.0, 0.26], [np.nan, 0.0]], dtype=f32)
    out, _ = texture_value(sb.textures[t], uv, np.zeros((len(uv), 3)))
    # col = (u * 8) as usize % 8, row = (v * 4) as usize % 4; u = 1 -> 8 % 8 = 0 (texture/src/lib.rs:211-223); NaN -> clamp keeps NaN -> 0
    want = [img[0, 0], img[0, 0], img[0, 1], img[3, 7], img[0, 0], img[2, 0], img[1, 0], img[0, 0]]
    assert (out == np.array(want)).all()


def test_perlin_marble_is_grey_in_unit_range_and_reproducible_from_its_tables():
    sb = SceneBuilder()
    t = sb.perlin(4.0, seed=7)
    rs = np.random.RandomState(1)
    p = (rs.rand(3000, 3) * 6 - 3).astype(f32)
    out, panics = texture_value(sb.textures[t], np.zeros((3000, 2)), p)
    assert panics == 0, "the reference asserts -1 <= noise <= 1 (texture/src/lib.rs:133-134)"
    assert (out[:, 0] == out[:, 1]).all() and (out[:, 1] == out[:, 2]).all()
    assert (out >= 0).all() and (out <= 1).all() and out.std() > 0.1
    sb2 = SceneBuilder()
    out2, _ = texture_value(sb2.textures[sb2.perlin(4.0, seed=7)], np.zeros((3000, 2)), p)
    assert (out.view(np.uint32) == out2.view(np.uint32)).all()
    # at lattice points every corner weight but one vanishes and that corner's offset is zero: noise = 0, marble = sin(freq z) / 2 + 1/2
    lattice = np.array([[0.25, 0.5, 0.75], [1.0, -0.5, 0.25]], dtype=f32)  # * freq 4 = integers
    out3, _ = texture_value(sb.textures[t], np.zeros((2, 2)), lattice)
    assert np.allclose(out3[:, 0], np.sin(4.0 * lattice[:, 2]) * 0.5 + 0.5, atol=1e-6)


def _env_scene(kind=None, image=None):
    sb = SceneBuilder()
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.lambertian((0.5, 0.5, 0.5)))
    if image is not None:
        sb.env_image(sb.image(image), (2.0, 1.0, 0.5))
    elif kind is not None:
        sb.env_sky(kind)
    sb.set_camera(8, 8, deg(40.0), (0, 0, -5), (0, 0, 0))
    return sb


def test_sky_closures():
    up, down, side = [0, 2, 0], [0, -3, 0], [1, 0, 0]
    e = OracleScene(_env_scene(spec.ENV_BLUE_SKY)).env_eval([up, down, side])
    assert np.allclose(e, [[0.5, 0.7, 1.0], [1, 1, 1], [0.75, 0.85, 1.0]], atol=1e-6)
    e = OracleScene(_env_scene(spec.ENV_DARK_ROOM)).env_eval([up, down, side])
    assert np.allclose(e, 0.1, atol=1e-6)
    e = OracleScene(_env_scene(spec.ENV_DUSK)).env_eval([up, down, side, [1, 1, 0]])
    dome, horizon = np.array([109, 150, 204]) / 255.0, np.array([245, 174, 82]) / 255.0
    assert np.allclose(e[0], 0.2, atol=1e-6), "tilt == 0 falls through both `>` tests (preset.rs:43-51)"
    assert np.allclose(e[1], dome, atol=1e-6) and np.allclose(e[2], dome, atol=1e-6)
    assert np.allclose(e[3], dome, atol=1e-5), "tilt = pi/4: not `> pi/4` only by rounding; the blend at t = 1 is the dome too"


def test_image_environment_is_a_lat_long_lookup():
    img = np.zeros((4, 8, 3), dtype=f32)
    img[:, :, 0] = np.arange(8)[None, :]   # column index in red
    img[:, :, 1] = np.arange(4)[:, None]   # row index in green
    img[:, :, 2] = 1.0
    osc = OracleScene(_env_scene(image=img))
    # phi = atan2(z, x); u = fract(phi / (2 pi) + 1); v = acos(y / |d|) / pi   (scene/src/lib.rs:108-113)
    dirs = np.array([[1, 0, 0.001], [0, 0, 1], [-1, 0, 0.001], [0, 0, -1], [0.001, 1, 0], [0.001, -1, 0]], dtype=f32)
    e = osc.env_eval(dirs)
    assert e[:, 0].tolist() == [0.0, 2.0 * 2, 2.0 * 3, 2.0 * 6, 0.0, 0.0]  # columns 0, 2, 3 (just short of 4), 6; red scaled by 2
    assert e[:, 1].tolist() == [2.0, 2.0, 2.0, 2.0, 0.0, 3.0]              # rows: equator 2, zenith 0, nadir 3 (y / |d| is just above -1, so v stays below 1)
    assert (e[:, 2] == 0.5).all()
