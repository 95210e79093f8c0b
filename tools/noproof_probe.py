"""Developer probe: times a config with the host's mesh shading proofs cleared, i.e. through the k_extend variants
that run the Q22 tangent check per candidate (PBRS_FEAT_SHADING_CHECK).  The image is the same; only the work differs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrs_amd
from pbrs_amd import scenes

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
sb, cfg = scenes.build_config(name)
hs = pbrs_amd.HostScene(sb)
ctx = pbrs_amd.Context(0)
ctx.upload(hs)
ref, _ = ctx.render(4, 4, cfg["depth"], 1)
import ctypes as C
d = hs.desc
meshes = np.ctypeslib.as_array(C.cast(d.meshes, C.POINTER(C.c_uint32)), shape=(d.n_meshes, 8))       # pbrs_mesh: flags = word 5
insts = np.ctypeslib.as_array(C.cast(d.instances, C.POINTER(C.c_uint32)), shape=(d.n_instances, 32))  # pbrs_instance: mesh_flags = word 29
meshes[:, 5] &= ~np.uint32(3)
insts[:, 29] &= ~np.uint32(3)
ctx.upload(hs)
ctx.render(4, 4, cfg["depth"], 1)
img, st = ctx.render(4, 4, cfg["depth"], 1, timing=True)
print(name, "no proofs: same image", bool(np.array_equal(ref, img)), {k: round(st[k], 3) for k in ("ms_extend", "ms_shade", "ms_shadow", "ms_total")})
