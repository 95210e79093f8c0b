"""Developer tool (GPU box): stage times of a scene whose objects carry Fourier BSDFs (tests/fourier_scenes.py), for A/B runs of
the k_shade variants with the lobe (PBRS_GPU_LIB selects the library).     python tools/fourier_bench.py [W H SX SY]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pbrs_amd
import fourier_scenes

w, h, sx, sy = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (960, 640, 8, 8)
for tables, lights, textured in ((("rgb", "fine", "mono"), "area", False), (("fine",), "area+env", True)):
    sb = fourier_scenes.scene(tables=tables, lights=lights, textured=textured, size=(w, h))
    ctx = pbrs_amd.Context(0)
    ctx.upload(pbrs_amd.HostScene(sb))
    ctx.render(sx, sy, 8, 1)
    img, st = ctx.render(sx, sy, 8, 1, timing=True)
    print("tables", tables, "lights", lights, "textured", textured, {k: round(v, 2) for k, v in st.items() if k.startswith("ms_")}, "mean", float(np.mean(img)))
    ctx.close()
