"""Developer tool: small PNG previews of the BASELINE scenes (gpurun_out/view_<cfg>.png) rendered on the GPU."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, pbrs_amd
from pbrs_amd import scenes
ctx = pbrs_amd.Context(0)
for name, w, h, s in (("c2", 320, 320, 8), ("c3", 320, 320, 8), ("c5", 480, 270, 6), ("c4", 480, 270, 4)):
    sb, cfg = scenes.build_config(name, width=w, height=h)
    ctx.upload(pbrs_amd.HostScene(sb))
    img, st = ctx.render(s, s, cfg["depth"], 1)
    img = np.nan_to_num(img) / max(1e-6, np.percentile(np.nan_to_num(img), 99)) * 0.9
    pbrs_amd.write_image(f"gpurun_out/view_{name}.png", img)
    print(name, "ok")
