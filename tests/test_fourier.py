"""The Fourier BSDF layer on the CPU (geometry/src/fourier.rs, math/src/spline.rs:161-318): the `.bsdf` file format against the
reference's header vector, the host flattener's table records, the pbrt front-end's `Material "fourier"`, and the oracle's
restatement against what a table must do when it encodes a BSDF whose answer is known (a Lambertian)."""
import struct

import numpy as np
import pytest

import fourier_scenes
import pbrs_amd
from oracle.binding import OracleScene
from pbrs_amd import fourier, scenes
from pbrs_amd.spec import SceneBuilder, deg

f32 = np.float32


def test_header_vector_of_the_reference():
    """geometry/src/fourier.rs:510-530 read_header_test: the 64 header bytes of assets/paint.bsdf."""
    buffer = bytes([
        0x53, 0x43, 0x41, 0x54, 0x46, 0x55, 0x4e, 0x01, 0x01, 0x00, 0x00, 0x00, 0x54, 0x03, 0x00, 0x00, 0xd6, 0xb4, 0x73, 0x01,
        0x3f, 0x06, 0x00, 0x00, 0x03, 0x00, 0x00, 0x00, 0x01, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
        0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x80, 0x3f, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00, 0x00,
        0x00, 0x00, 0x00, 0x00])
    h = fourier.parse_header(buffer)
    assert h["n_mu"] == 0x0354 and h["n_coeffs"] == 0x0173b4d6 and h["m_max"] == 0x063f and h["n_channels"] == 3 and h["eta"] == 1.0
    with pytest.raises(ValueError):
        fourier.parse_header(b"SCATFUN\x02" + buffer[8:])  # version 2


def test_file_round_trip(tmp_path):
    t = fourier_scenes.table("rgb")
    fourier.write_bsdf(tmp_path / "t.bsdf", t)
    u = fourier.read_bsdf(tmp_path / "t.bsdf")
    assert u.n_channels == 3 and (u.mu == t.mu).all() and (u.cdf == t.cdf).all() and (u.a == t.a).all()
    assert (u.offset_and_length == t.offset_and_length).all() and u.m_max == t.m_max
    lengths = t.offset_and_length[:, 1]
    assert len(set(lengths.tolist())) > 2 and (lengths == 0).any(), "pairs carry series of different lengths, some none"


def test_host_flattener_lays_the_table_into_the_pools():
    sb = fourier_scenes.scene(("rgb", "mono"), "area", textured=True)
    hs = pbrs_amd.HostScene(sb)
    d = hs.desc
    assert d.n_fourier_tables == 2
    kinds = [hs.array("bxdfs")[i]["kind"] for i in range(d.n_bxdfs)] if hasattr(hs, "array") else None
    t = fourier_scenes.table("rgb")
    n = len(t.mu)
    assert d.n_tex_floats >= n + 2 * n * n + len(t.a) + t.m_max and d.n_tex_words >= 2 * n * n
    assert kinds is None or 3 in kinds


@pytest.mark.parametrize("breakage,needle", [("descending", "ascending"), ("repeated", "strictly ascending"), ("nan", "finite"), ("series", "outside"),
                                             ("channels", "sizes"), ("index", "missing table")])
def test_malformed_tables_are_errors(breakage, needle):
    t = fourier_scenes.table("mono")
    t = fourier.FourierTable(t.mu.copy(), t.cdf.copy(), t.offset_and_length.copy(), t.a.copy(), t.n_channels)
    sb = SceneBuilder()
    if breakage == "descending":
        t.mu[3], t.mu[4] = t.mu[4], t.mu[3]
    if breakage == "repeated":  # an interval of zero width: catmull_rom_weights would divide 0 by 0 at that node
        t.mu[4] = t.mu[3]
    if breakage == "nan":
        t.mu[2] = np.nan
    if breakage == "series":
        t.offset_and_length[5, 0] = len(t.a)
        t.offset_and_length[5, 1] = 4
    ti = sb.fourier_table(t)
    if breakage == "channels":
        sb.fourier_tables[0].n_channels = 2
    sb.instance(sb.sphere((0, 0, 0), 1.0), sb.fourier(ti + (1 if breakage == "index" else 0)))
    sb.set_camera(8, 8, deg(40), (0, 0, -5), (0, 0, 0))
    with pytest.raises(pbrs_amd.PbrsError) as e:
        pbrs_amd.HostScene(sb)
    assert needle in str(e.value)


def _furnace(material_of):
    """A sphere in a uniform white environment, seen head on: with a reflector of albedo rho the radiance towards the camera is
    about rho / (1 - ...) of the environment's — enough to tell a wrong normalisation from a right one."""
    sb = SceneBuilder()
    sb.instance(sb.sphere((0, 0, 0), 1.0), material_of(sb))
    sb.env = (1.0, 1.0, 1.0)
    sb.set_camera(24, 24, deg(20), (0, 0, -6), (0, 0, 0))
    img, st = OracleScene(sb).render(6, 6, 6, 3)
    return img[8:16, 8:16].mean(axis=(0, 1)), st


def test_a_table_of_a_lambertian_renders_like_the_lambertian():
    """eval, pdf and sample together: a table that encodes f = 0.5 / pi must reflect like Lambertian(0.5) — the Monte-Carlo means
    agree within noise only if eval's 1 / |mu_i| scale, prob's rho normalisation and sample's two inversions are all right."""
    fo, st = _furnace(lambda sb: sb.fourier(sb.fourier_table(fourier_scenes.table("lambert"))))
    la, _ = _furnace(lambda sb: sb.lambertian((0.5, 0.5, 0.5)))
    assert st["panics"] == 0 and st["nonfinite_samples"] == 0
    assert np.abs(fo / la - 1.0).max() < 0.03, (fo, la)


def test_direct_lighting_of_a_point_light_uses_eval_alone():
    """estimate_direct_delta_light (src/directlighting.rs:101-153) only evaluates the BSDF: no sampling noise, so the table of the
    Lambertian must give the Lambertian's image up to the interpolation error of the table."""
    def scene(material_of):
        sb = SceneBuilder()
        sb.instance(scenes.quad_mesh(sb, (-4, 0, -4), (4, 0, -4), (-4, 0, 4), (4, 0, 4), (0, 1, 0)), material_of(sb))
        sb.point_light((0.5, 3.0, -0.5), (20, 20, 20))
        sb.set_camera(32, 24, deg(50), (0, 3, -5), (0, 0, 0))
        return OracleScene(sb).render(1, 1, 3, 1, integrator="direct")[0]
    fo = scene(lambda sb: sb.fourier(sb.fourier_table(fourier_scenes.table("lambert"))))
    la = scene(lambda sb: sb.lambertian((0.5, 0.5, 0.5)))
    lit = la[..., 0] > 0.05
    assert lit.mean() > 0.5 and np.abs(fo[lit] / la[lit] - 1.0).max() < 2e-3


def test_transmitted_samples_reach_the_reference_todo():
    sb = fourier_scenes.scene(("translucent",), "area")
    img, st = OracleScene(sb).render(2, 2, 6, 9)
    assert st["panics"] > 0 and st["nonfinite_samples"] == 0
    sb = fourier_scenes.scene(("rgb",), "area")
    assert OracleScene(sb).render(2, 2, 6, 9)[1]["panics"] == 0


def test_pbrt_front_end_reads_fourier_materials(tmp_path):
    """scene/src/loader.rs:705-710: `Material "fourier" "string bsdffile"`; the loaded scene renders bit for bit like the same
    scene built through the spec."""
    t = fourier_scenes.table("rgb")
    fourier.write_bsdf(tmp_path / "paintlike.bsdf", t)
    (tmp_path / "s.pbrt").write_text("""
LookAt 0 2 -6 0 1 0 0 1 0  Camera "perspective" "float fov" [45]  Film "image" "integer xresolution" [32] "integer yresolution" [24]
WorldBegin
LightSource "point" "point from" [2 5 -3] "color L" [40 40 40]
AttributeBegin Material "fourier" "string bsdffile" "paintlike.bsdf" Translate 0 1 0 Shape "sphere" "float radius" [1] AttributeEnd
WorldEnd
""")
    ls = pbrs_amd.load_pbrt(str(tmp_path / "s.pbrt"))
    s = ls.build()
    assert s.n_fourier_tables == 1 and s.fourier_tables[0].n_mu == len(t.mu) and s.fourier_tables[0].n_coeffs == len(t.a)
    a, _ = OracleScene(ls).render(2, 2, 4, 3)
    assert pbrs_amd.HostScene(ls).desc.n_fourier_tables == 1
    (tmp_path / "bad.pbrt").write_text((tmp_path / "s.pbrt").read_text().replace("paintlike.bsdf", "missing.bsdf"))
    with pytest.raises(pbrs_amd.PbrsError):
        pbrs_amd.load_pbrt(str(tmp_path / "bad.pbrt"))
    (tmp_path / "short.bsdf").write_bytes((tmp_path / "paintlike.bsdf").read_bytes()[:200])
    (tmp_path / "short.pbrt").write_text((tmp_path / "s.pbrt").read_text().replace("paintlike.bsdf", "short.bsdf"))
    with pytest.raises(pbrs_amd.PbrsError) as e:
        pbrs_amd.load_pbrt(str(tmp_path / "short.pbrt"))
    assert "truncated" in str(e.value)
    # a header that announces 2^31 - 1 coefficients on a file of a few KB must fail on the file size, before anything is sized from it
    raw = bytearray((tmp_path / "paintlike.bsdf").read_bytes())
    raw[16:20] = struct.pack("<i", 2**31 - 1)
    (tmp_path / "huge.bsdf").write_bytes(bytes(raw))
    (tmp_path / "huge.pbrt").write_text((tmp_path / "s.pbrt").read_text().replace("paintlike.bsdf", "huge.bsdf"))
    with pytest.raises(pbrs_amd.PbrsError) as e:
        pbrs_amd.load_pbrt(str(tmp_path / "huge.pbrt"))
    assert "truncated" in str(e.value)
    assert a.mean() > 0.005


def nan_light_scene():
    """Every vertex lit by a distant light whose direction is NaN (DeltaLight::Distant hands `-casting_dir` to the BSDF as it
    is, light/src/lib.rs:77-89) or by a sound point light: the Fourier lobe meets NaN directions at every other estimate."""
    sb = fourier_scenes.scene(("rgb", "mono", "fine"), "point", size=(40, 28))
    sb.distant_light((float("nan"), float("nan"), float("nan")), (3.0, 3.0, 3.0), 50.0)
    return sb


def test_nan_directions_end_in_black_not_in_a_wild_index():
    """A NaN direction passes `x < nodes[0] || x > nodes[n - 1]` and trips the reference's assert (math/src/spline.rs:214): the
    oracle counts the panic and the lobe answers black; no knot index leaves the table (run under ASan by tools/cpu_asan.sh).
    Head-on views of the sphere put mu in the table's edge intervals, where one knot of the four does not exist."""
    img, st = OracleScene(nan_light_scene()).render(2, 2, 5, 11)
    assert st["panics"] > 0
    assert np.isfinite(img).all() and img.mean() > 0.003
