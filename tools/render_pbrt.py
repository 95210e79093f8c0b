#!/usr/bin/env python3
"""Renders a pbrt-v3 scene file on the GPU and writes an EXR (or PNG):  tools/render_pbrt.py scene.pbrt out.exr [msaa] [depth] [path|direct|materials|normals]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd

scene, out = sys.argv[1], sys.argv[2]
msaa = int(sys.argv[3]) if len(sys.argv) > 3 else 4
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 5  # src/main.rs:205
integrator = sys.argv[5] if len(sys.argv) > 5 else "path"
ls = pbrs_amd.load_pbrt(scene)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(ls))
if integrator in ("materials", "normals"):  # --visualize-materials / --visualize-normals (src/main.rs:180-185): one ray per pixel
    msaa = 1
img, st = ctx.render(msaa, msaa, depth, 1, integrator=integrator, timing=True)
pbrs_amd.write_image(out, img)
print(f"{img.shape[1]}x{img.shape[0]} at {msaa * msaa} spp in {st['ms_total']:.1f} ms -> {out}")
