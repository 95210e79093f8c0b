"""Developer probe: lane utilisation of the traversal steps.  Needs a library built with -DPBRS_PROBE_UTIL
(tools/ablate.sh "util:-DPBRS_PROBE_UTIL") selected through PBRS_GPU_LIB; the scene must have no cuboids."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sb, cfg = scenes.build_config(name)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(sb))
img, st = ctx.render(4, 4, depth, 1, counters=True, timing=True)
rays = st["closest_rays"]
nodes = st["tlas_nodes"] + st["blas_nodes"]
print(name, "depth", depth, "extend rays", rays, "nodes/ray %.2f (tlas %.2f blas %.2f) inst/ray %.2f tris/ray %.2f" % (
    nodes / rays, st["tlas_nodes"] / rays, st["blas_nodes"] / rays, st["instances"] / rays, st["triangles"] / rays))
wn, wl, wx, it = st["cuboids"], st["disks"], st["spheres"], st["quads"]
print("  loop rounds", it, "| rounds with node lanes", wn, "(lane utilisation %.3f)" % (nodes / (64.0 * wn) if wn else 0),
      "| boundary-step executions", wx, "(lane utilisation %.3f: one entry + one exit per instance)" % (2.0 * st["instances"] / (64.0 * wx) if wx else 0))
le = wl  # wave-level leaf-step executions
print("  triangles", st["triangles"], "instances", st["instances"], "| leaf-step executions", le,
      "(lane utilisation %.3f)" % (st["triangles"] / (64.0 * le) if le else 0))
print("  shadow rays", st["shadow_rays"], "nodes/ray %.2f tris/ray %.2f" % (
    (st["shadow_tlas_nodes"] + st["shadow_blas_nodes"]) / max(1, st["shadow_rays"]), st["shadow_triangles"] / max(1, st["shadow_rays"])))
print("  ms", {k: round(st[k], 3) for k in ("ms_extend", "ms_shade", "ms_shadow", "ms_total")})
