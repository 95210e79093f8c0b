"""A scene read by the pbrt-v3 front-end rendered on the GPU: same image as the oracle, bit for bit, for both integrators."""
import numpy as np
import pytest

import pbrs_amd
from oracle.binding import OracleScene
from test_pbrt_loader import scene_dir  # noqa: F401  (fixture: scene.pbrt + more.pbrt + tex.png + box.ply)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("integrator", ["path", "direct"])
def test_loaded_pbrt_scene_matches_oracle(gpu_ctx, scene_dir, integrator):  # noqa: F811
    d, _ = scene_dir
    ls = pbrs_amd.load_pbrt(str(d / "scene.pbrt"))
    gpu_ctx.upload(pbrs_amd.HostScene(ls))
    ref, ost = OracleScene(ls).render(2, 2, 6, 5, integrator=integrator)
    img, st = gpu_ctx.render(2, 2, 6, 5, integrator=integrator, counters=True)
    assert ost["tlas_ties"] == 0
    assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
    nan = np.isnan(ref)
    assert (nan == np.isnan(img)).all()
    assert (img.view(np.uint32)[~nan] == ref.view(np.uint32)[~nan]).all()
    assert np.nanstd(ref) > 0.05


def test_spd_metal_and_blackbody_lights_match_oracle(gpu_ctx, tmp_path):
    """`"spectrum eta" "Au.eta.spd"` / `"spectrum k"` (scene/src/loader.rs:548-570, :858-879) and `"blackbody L" [T scale]`
    (:763): file -> host library (spline + CIE integration) -> flattener -> GPU, against the oracle's render of the loaded scene."""
    from test_pbrt_loader import GOLD_ETA, GOLD_K, write_spd
    write_spd(tmp_path / "Au.eta.spd", GOLD_ETA)
    write_spd(tmp_path / "Au.k.spd", GOLD_K, shuffle=True)
    (tmp_path / "s.pbrt").write_text("""
LookAt 0 2 -6 0 1 0 0 1 0  Camera "perspective" "float fov" [45]  Film "image" "integer xresolution" [64] "integer yresolution" [48]
WorldBegin
AttributeBegin AreaLightSource "diffuse" "blackbody L" [6500 2.5] Translate 0 5 0 Shape "sphere" "float radius" [0.7] AttributeEnd
LightSource "point" "point from" [3 4 -3] "blackbody L" [2700 30]
AttributeBegin Material "metal" "spectrum eta" "Au.eta.spd" "spectrum k" "Au.k.spd" "float roughness" [0.05] Translate 0 1 0 Shape "sphere" "float radius" [1] AttributeEnd
AttributeBegin Material "matte" "rgb Kd" [.5 .5 .5] Translate 0 -100 0 Shape "sphere" "float radius" [100] AttributeEnd
WorldEnd
""")
    ls = pbrs_amd.load_pbrt(str(tmp_path / "s.pbrt"))
    gpu_ctx.upload(pbrs_amd.HostScene(ls))
    for integrator in ("path", "direct"):
        ref, ost = OracleScene(ls).render(3, 3, 6, 5, integrator=integrator)
        img, st = gpu_ctx.render(3, 3, 6, 5, integrator=integrator, counters=True)
        assert st["closest_rays"] == ost["closest_rays"] and st["shadow_rays"] == ost["shadow_rays"]
        assert (img.view(np.uint32) == ref.view(np.uint32)).all(), integrator
    # gold under a warm and a daylight source: the sphere's highlight is yellow (red > blue)
    sphere = ref[14:34, 22:42].reshape(-1, 3).mean(axis=0)
    assert sphere[0] > sphere[2]
