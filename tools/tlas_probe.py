"""Developer probe: stage times of the C5 scene family at a chosen instance count (tools/tlas_probe.py N_OBJECTS N_LIGHTS), to
compare the shared TLAS scan with the tree walk around PBRS_FLAT_TLAS_MAX (build variants through tools/ablate.sh and select
them with PBRS_GPU_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes

no, nl = int(sys.argv[1]), int(sys.argv[2])
sb = scenes.many_lights_scene(width=960, height=540, n_objects=no, n_lights=nl)
hs = pbrs_amd.HostScene(sb)
ctx = pbrs_amd.Context(0)
ctx.upload(hs)
ctx.render(8, 8, 8, 1)
img, st = ctx.render(8, 8, 8, 1, timing=True)
img, cs = ctx.render(8, 8, 8, 1, counters=True)
rays = cs["closest_rays"]
print(os.path.basename(os.environ.get("PBRS_GPU_LIB", "libpbrs_gpu.so")), "instances", hs.desc.n_instances, "tlas box tests/ray %.1f" % (cs["tlas_nodes"] / rays),
      {k: round(v, 2) for k, v in st.items() if k in ("ms_extend", "ms_shade", "ms_shadow", "ms_total")})
