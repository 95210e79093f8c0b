"""Developer tool (GPU box): lean against full further node steps on one scene, same box, same library — a developer build
(tools/ablate.sh "dev:-DPBRS_DEV_OVERRIDES", PBRS_GPU_LIB) whose per-scene choice is overridden through PBRS_FULL_STEPS=0 / 1.
    PBRS_GPU_LIB=$PWD/pbrs_amd/lib/abl_dev.so python tools/steps_ab.py [config]        (default c4xl)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pbrs_amd
from pbrs_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c4xl"
sb, cfg = scenes.build_config(name)
hs = pbrs_amd.HostScene(sb)
ctx = pbrs_amd.Context(0)
out = torch.empty((cfg["height"], cfg["width"], 3), dtype=torch.float32, device="cuda:0")
samples = cfg["width"] * cfg["height"] * cfg["strata_x"] * cfg["strata_y"]
for label, env in (("default", None), ("lean", "0"), ("full", "1"), ("default", None), ("lean", "0"), ("full", "1")):
    os.environ.pop("PBRS_FULL_STEPS", None)
    if env is not None:
        os.environ["PBRS_FULL_STEPS"] = env
    ctx.upload(hs)
    best = None
    for k in range(4):
        ctx.render_device(out.data_ptr(), cfg["strata_x"], cfg["strata_y"], cfg["depth"], 1, timing=True)
        st = ctx.collect_stats()
        if k and (best is None or st["ms_total"] < best["ms_total"]):
            best = st
    print("%s %-8s features extend %d shadow %d: %.1f Msamples/s  (extend %.1f shade %.1f shadow %.1f ms; %.1f ms per frame)" % (
        name, label, best["kernel_features_extend"], best["kernel_features_shadow"], samples / best["ms_total"] / 1e3, best["ms_extend"], best["ms_shade"],
        best["ms_shadow"], best["ms_total"]), flush=True)
ctx.close()
