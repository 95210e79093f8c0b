"""Developer probe: where a wave's cycles go in the traversal loops of k_extend / k_shadow (s_memtime at the boundaries of the
loop's regions; the stamps themselves cost about a tenth).  Needs a library built with -DPBRS_PROBE_TIME
(tools/ablate.sh "ttime:-DPBRS_PROBE_TIME") selected through PBRS_GPU_LIB.     python tools/trav_time.py c4 [sx sy]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes, api

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 8)
sb, cfg = scenes.build_config(name)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(sb))
L = api.gpu_lib()
buf = (C.c_ulonglong * 16)()
ctx.render(sx, sy, cfg["depth"], 1)
L.pbrs_debug_trav_time(buf)
img, st = ctx.render(sx, sy, cfg["depth"], 1, timing=True)
L.pbrs_debug_trav_time(buf)
print(name, "ms_extend %.2f ms_shadow %.2f" % (st["ms_extend"], st["ms_shadow"]))
for which, kern in enumerate(("k_extend", "k_shadow")):
    v = [buf[which * 8 + k] for k in range(4)]
    tot = max(sum(v), 1)
    print(f"{kern}: wave cycles {tot:.3e}: refill {100 * v[0] / tot:.1f} %, boundary step {100 * v[1] / tot:.1f} %, node steps {100 * v[2] / tot:.1f} %, leaf step {100 * v[3] / tot:.1f} %")
