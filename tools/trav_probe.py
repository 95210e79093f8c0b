"""Developer probe: what the traversal loops of k_extend / k_shadow execute, per ray.  Needs a library built with
-DPBRS_PROBE_TRAV (tools/ablate.sh "tprobe:-DPBRS_PROBE_TRAV -DPBRS_DEV_OVERRIDES") selected through PBRS_GPU_LIB; PBRS_WIDE
(bit 0 k_extend, bit 1 k_shadow) picks the walks.     python tools/trav_probe.py c4 [sx sy]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pbrs_amd
from pbrs_amd import scenes, api

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
sx, sy = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4, 4)
sb, cfg = scenes.build_config(name)
ctx = pbrs_amd.Context(0)
ctx.upload(pbrs_amd.HostScene(sb))
L = api.gpu_lib()
buf = (C.c_ulonglong * 48)()
ctx.render(sx, sy, cfg["depth"], 1)
L.pbrs_debug_trav_probe(buf)  # clear after warm-up (the first frame's instrumented pass is not a probed kernel anyway)
img, st = ctx.render(sx, sy, cfg["depth"], 1, timing=True)
L.pbrs_debug_trav_probe(buf)
print(name, "PBRS_WIDE=%s" % os.environ.get("PBRS_WIDE", "(default)"), "ms_extend %.2f ms_shadow %.2f" % (st["ms_extend"], st["ms_shadow"]))
for which, kern in enumerate(("k_extend", "k_shadow")):
    v = [buf[which * 24 + k] for k in range(24)]
    rays = max(v[2], 1)
    rounds = max(v[0], 1)
    print(f"{kern}: {v[2]} rays, {v[0]} wave rounds ({64.0 * v[0] / rays:.1f} lane-rounds per ray), live lanes at round start {v[10] / rounds:.1f}")
    print(f"  refills {v[1]} ({v[2] / max(v[1], 1):.1f} rays each, one per {v[0] / max(v[1], 1):.1f} rounds)")
    print(f"  boundary steps: {v[3]} executions ({v[3] / rounds:.3f} per round) at {v[4] / max(v[3], 1):.1f} lanes; per ray in {v[22] / rays:.2f} out {v[23] / rays:.2f}")
    print(f"  node steps: first {v[5] / max(v[11], 1):.1f} lanes in {v[11] / rounds:.3f} of rounds, second {v[6] / rounds:.1f}, third {v[7] / rounds:.1f} lanes per round")
    print(f"  per ray: node steps {v[16] / rays:.1f}, box tests {v[17] / rays:.1f} (failed {v[18] / rays:.1f}), leaves up {v[19] / rays:.2f} passing {v[20] / rays:.2f}, triangle tests {v[21] / rays:.2f}")
    print(f"  leaf steps: {v[8]} executions ({v[8] / rounds:.3f} per round) with {v[9] / max(v[8], 1):.1f} lanes holding a leaf, {v[21] / max(v[8], 1):.1f} helper lanes")
