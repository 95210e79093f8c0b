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
