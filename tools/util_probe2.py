"""Developer probe: where the lanes of a traversal wave are at the start of a loop round.  Needs a library built with
-DPBRS_PROBE_UTIL -DPBRS_PROBE_UTIL2 (tools/ablate.sh "util2:-DPBRS_PROBE_UTIL -DPBRS_PROBE_UTIL2") selected through PBRS_GPU_LIB;
run once more without PBRS_GPU_LIB for the scene's real tri_shading count (the probe borrows that counter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
sb, cfg = scenes.build_config(name)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(sb))
img, st = ctx.render(4, 4, 8, 1, counters=True)
print(name, {k: st[k] for k in ("closest_rays", "tlas_nodes", "blas_nodes", "instances", "triangles", "tri_shading", "quads", "cuboids", "disks")})
