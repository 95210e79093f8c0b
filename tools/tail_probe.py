import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes
ctx = pbrs_amd.Context(0)
for name in ("c2", "c4", "c5"):
    sb, cfg = scenes.build_config(name)
    ctx.upload(pbrs_amd.HostScene(sb))
    _, st = ctx.render(2, 2, cfg["depth"], 1, counters=True)
    print(name, "sum lanes nodes", st["disks"], "wave max*64", st["quads"], "inner-loop utilisation bound %.3f" % (st["disks"] / max(st["quads"], 1)),
          "nodes/ray %.1f" % ((st["tlas_nodes"] + st["blas_nodes"]) / st["closest_rays"]), "tris/ray %.2f" % (st["triangles"] / st["closest_rays"]))
