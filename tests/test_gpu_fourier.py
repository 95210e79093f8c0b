"""The Fourier BSDF (geometry/src/fourier.rs:224-485, math/src/spline.rs:161-318) on the GPU against the oracle, bit for
bit: f64 series sums, the two Newton-bisection samplers, Catmull-Rom interpolation of the table — through both
integrators and both visualisers, with one- and three-channel tables, every kind of light, textures next to it."""
import numpy as np
import pytest

import pbrs_amd
from oracle.binding import OracleScene
import fourier_scenes

pytestmark = pytest.mark.gpu


def same(ref, img):
    nan = np.isnan(ref)
    return (nan == np.isnan(img)).all() and (img.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all()


@pytest.mark.parametrize("tables,lights,textured", [(("rgb",), "area", False), (("mono",), "point", False), (("rgb", "mono", "fine"), "area point", False),
                                                    (("fine",), "env", False), (("rgb", "mono"), "area env", True)])
def test_fourier_materials_match_oracle(gpu_ctx, tables, lights, textured):
    sb = fourier_scenes.scene(tables, lights, textured)
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    for integrator, depth in (("path", 6), ("direct", 3)):
        ref, ost = osc.render(2, 2, depth, 5, integrator=integrator)
        img, st = gpu_ctx.render(2, 2, depth, 5, integrator=integrator, counters=True)
        if lights == "area":
            assert ost["panics"] == 0  # reflection-only tables never reach the reference's `todo!()` (mixed light lists have panic sites of their own)
        assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"], integrator
        assert same(ref, img), integrator
        assert st["invalid_samples"] == ost["nonfinite_samples"] == 0
    assert ref.mean() > 0.01


def test_fourier_visualisers_match_oracle(gpu_ctx):
    sb = fourier_scenes.scene(("rgb", "mono"), "area")
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    for integrator in ("materials", "normals"):
        ref, _ = osc.render(1, 1, 1, 5, integrator=integrator)
        img, _ = gpu_ctx.render(1, 1, 1, 5, integrator=integrator)
        assert same(ref, img), integrator
    ref, _ = osc.render(1, 1, 1, 5, integrator="materials")
    assert (np.abs(ref - np.array([143, 112, 252], dtype=np.float32) / np.float32(255)).max(axis=2) == 0).any()  # "Fourier" => pal[6]


def test_transmitting_table_ends_paths_where_the_reference_stops(gpu_ctx):
    """FourierBSDF::sample is `todo!()` for a sampled direction on the far side (:423-428): the oracle counts a panic and
    the path ends black there; the GPU ends it the same way."""
    sb = fourier_scenes.scene(("translucent",), "area")
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = osc.render(2, 2, 6, 9)
    img, _ = gpu_ctx.render(2, 2, 6, 9)
    assert ost["panics"] > 0
    assert same(ref, img)


def test_more_samples_and_a_larger_frame(gpu_ctx):
    sb = fourier_scenes.scene(("fine", "rgb", "mono"), "area point env", textured=True, size=(96, 64))
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    ref, ost = osc.render(4, 4, 8, 21)
    img, st = gpu_ctx.render(4, 4, 8, 21, counters=True)
    assert st["closest_rays"] == ost["closest_rays"]
    assert same(ref, img)


def test_a_bsdf_file_through_the_pbrt_front_end(gpu_ctx, tmp_path):
    """`Material "fourier" "string bsdffile"` (scene/src/loader.rs:705-710): file -> host library -> flattener -> GPU."""
    from pbrs_amd import fourier
    fourier.write_bsdf(tmp_path / "coat.bsdf", fourier_scenes.table("fine"))
    (tmp_path / "s.pbrt").write_text("""
LookAt 0 2 -6 0 1 0 0 1 0  Camera "perspective" "float fov" [45]  Film "image" "integer xresolution" [48] "integer yresolution" [32]
WorldBegin
LightSource "point" "point from" [2 5 -3] "color L" [40 40 40]
LightSource "infinite" "rgb L" [.2 .3 .4]
AttributeBegin Material "fourier" "string bsdffile" "coat.bsdf" Translate 0 1 0 Shape "sphere" "float radius" [1] AttributeEnd
AttributeBegin Material "matte" "rgb Kd" [.5 .5 .5] Translate 0 -100 0 Shape "sphere" "float radius" [100] AttributeEnd
WorldEnd
""")
    ls = pbrs_amd.load_pbrt(str(tmp_path / "s.pbrt"))
    ref, ost = OracleScene(ls).render(3, 3, 5, 2)
    gpu_ctx.upload(pbrs_amd.HostScene(ls))
    img, st = gpu_ctx.render(3, 3, 5, 2, counters=True)
    assert st["closest_rays"] == ost["closest_rays"] and same(ref, img) and ref.mean() > 0.05


def test_nan_directions_match_the_oracle(gpu_ctx):
    """tests/test_fourier.py::test_nan_directions_end_in_black_not_in_a_wild_index on the device: the lobe refuses a NaN
    direction before any weight exists and skips knots outside the table, so no lane reads outside the pools."""
    from test_fourier import nan_light_scene
    sb = nan_light_scene()
    osc = OracleScene(sb)
    gpu_ctx.upload(pbrs_amd.HostScene(sb))
    for integrator, depth in (("path", 5), ("direct", 3)):
        ref, ost = osc.render(2, 2, depth, 11, integrator=integrator)
        img, st = gpu_ctx.render(2, 2, depth, 11, integrator=integrator, counters=True)
        assert ost["panics"] > 0
        assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"], integrator
        assert same(ref, img), integrator
        assert st["invalid_samples"] == ost["nonfinite_samples"]
