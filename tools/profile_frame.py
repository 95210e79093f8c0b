#!/usr/bin/env python3
"""Developer tool (GPU box): renders full frames of one BASELINE config and nothing else — the program rocprofv3's counter
passes wrap (tools/round_artifacts.sh).  One untimed frame allocates the path state, then --frames frames run.
Prints one JSON line with the pass geometry the traffic tool needs (pixels, samples per pass)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--strata", type=int, nargs=2, default=None)
    ap.add_argument("--no-warm", action="store_true")
    a = ap.parse_args()
    import torch

    import pbrs_amd
    from pbrs_amd import roofline, scenes
    sb, cfg = scenes.build_config(a.config)
    if a.strata:
        cfg["strata_x"], cfg["strata_y"] = a.strata
    hs = pbrs_amd.HostScene(sb)
    ctx = pbrs_amd.Context(0)
    ctx.upload(hs)
    out = torch.empty((cfg["height"], cfg["width"], 3), dtype=torch.float32, device="cuda:0")
    st = None
    for k in range(a.frames + (0 if a.no_warm else 1)):
        ctx.render_device(out.data_ptr(), cfg["strata_x"], cfg["strata_y"], cfg["depth"], 1, timing=True)
        st = ctx.collect_stats()
    spp = cfg["strata_x"] * cfg["strata_y"]
    print(json.dumps({"config": a.config, "pixels": cfg["width"] * cfg["height"], "spp": spp, "passes": st["passes"],
                      "samples_per_pass": -(-spp // st["passes"]), "frames": a.frames + (0 if a.no_warm else 1),
                      "stages_ms": {k: v for k, v in st.items() if k.startswith("ms_")}, "scene_bytes": hs.nbytes,
                      # what the counters of this run were measured on (bench.py quotes them only for the same sources)
                      "source_hash": roofline.source_hash(),
                      # no .git travels to the GPU box: the commit is stamped from the builder side (PBRS_GIT_HEAD) or left out
                      **({"git_head": os.environ["PBRS_GIT_HEAD"]} if os.environ.get("PBRS_GIT_HEAD") else {})}))
    ctx.close()


if __name__ == "__main__":
    main()
