"""Developer probe: where k_shade's wave time goes.  Needs a library built with -DPBRS_PROBE_SHADE
(tools/ablate.sh "sprobe:-DPBRS_PROBE_SHADE") selected through PBRS_GPU_LIB."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes, api

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
sb, cfg = scenes.build_config(name)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(sb))
L = api.gpu_lib() if hasattr(api, "gpu_lib") else ctx._L
buf = (C.c_ulonglong * 16)()
ctx.render(4, 4, cfg["depth"], 1)
L.pbrs_debug_shade_probe(buf)  # clear after warm-up
img, st = ctx.render(4, 4, cfg["depth"], 1, timing=True)
L.pbrs_debug_shade_probe(buf)
names = ["loads+emission", "interaction rebuild + frame", "NEE light sample + pdf", "NEE term1 eval/pdf/MIS", "NEE term2 sample+light isect",
         "NEE bookkeeping", "bounce sample/RR/stores", "compaction + writes"]
for off, what in ((0, "the Lambert variants"), (8, "the general / textured / Fourier variants")):
    tot = sum(buf[off:off + 8])
    if not tot:
        continue
    print(name, what, "ms_shade (all variants) %.3f" % st["ms_shade"], "wave-cycles of the sampled blocks %.3e" % tot)
    for k, n in enumerate(names):
        print("  %-32s %5.1f %%" % (n, 100.0 * buf[off + k] / tot))
