"""Exploratory GPU-vs-oracle probe (developer tool, not a test)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pbrs_amd
from pbrs_amd import scenes
from oracle.binding import OracleScene, numeric_eval

ctx = pbrs_amd.Context(0)
rs = np.random.RandomState(0)
print("== numeric")
for fn, gen in [("sin", lambda: rs.uniform(-20, 20, 200000)), ("cos", lambda: rs.uniform(-20, 20, 200000)),
                ("tan", lambda: rs.uniform(-7, 7, 200000)), ("atan", lambda: rs.standard_normal(200000) * 10),
                ("acos", lambda: rs.uniform(-1, 1, 200000)), ("exp", lambda: rs.uniform(-90, 90, 200000)),
                ("ln", lambda: np.exp(rs.uniform(-80, 80, 200000))), ("sqrt", lambda: np.exp(rs.uniform(-80, 80, 200000))),
                ("asin", lambda: rs.uniform(-1, 1, 200000))]:
    x = gen().astype(np.float32)
    a = numeric_eval(fn, x); b = ctx.numeric_eval(fn, x)
    print(fn, "mismatch", int((a.view(np.uint32) != b.view(np.uint32)).sum()))
for fn in ("div", "hypot", "atan2"):
    x = (rs.standard_normal(200000) * np.exp(rs.uniform(-30, 30, 200000))).astype(np.float32)
    y = (rs.standard_normal(200000) * np.exp(rs.uniform(-30, 30, 200000))).astype(np.float32)
    a = numeric_eval(fn, x, y); b = ctx.numeric_eval(fn, x, y)
    print(fn, "mismatch", int((a.view(np.uint32) != b.view(np.uint32)).sum()))

def probe(name, w, h, sx, sy, depth, **kw):
    print("==", name, w, h)
    sb, cfg = scenes.build_config(name, width=w, height=h, **kw)
    t = time.time(); osc = OracleScene(sb); print("oracle build %.2fs" % (time.time() - t))
    t = time.time(); hs = pbrs_amd.HostScene(sb); print("host flatten %.2fs" % (time.time() - t))
    ctx.upload(hs)
    o1, d1 = osc.camera_rays(3, sx, sy, 1); o2, d2 = ctx.camera_rays(3, sx, sy, 1)
    print("camera rays mismatch", int((o1.view(np.uint32) != o2.view(np.uint32)).sum()), int((d1.view(np.uint32) != d2.view(np.uint32)).sum()))
    tmax = np.full(len(o1), np.inf, dtype=np.float32)
    h1, occ1, st = osc.intersect(o1, d1, tmax); h2, occ2 = ctx.intersect(o1, d1, tmax)
    for f in ("t", "inst", "prim", "b1", "b2"):
        a, b = h1[f], h2[f]
        neq = (a.view(np.uint32) != b.view(np.uint32))
        print("  hit", f, "mismatch", int(neq.sum()))
        if neq.sum():
            i = np.nonzero(neq)[0][0]; print("   first", i, h1[i], h2[i])
    print("  occluded mismatch", int((occ1 != occ2).sum()), "ties", st["tlas_ties"], "panics", st["panics"])
    t = time.time(); img1, st1 = osc.render(sx, sy, depth, 1); to = time.time() - t
    t = time.time(); img2, st2 = ctx.render(sx, sy, depth, 1, counters=True, timing=True); tg = time.time() - t
    diff = np.abs(img1 - img2)
    neq = (img1.view(np.uint32) != img2.view(np.uint32)).any(axis=2)
    print("  render oracle %.2fs gpu %.2fs; pixels differing %d / %d; max abs diff %g; nan o/g %d %d" % (
        to, tg, int(neq.sum()), neq.size, float(np.nanmax(diff)), int(np.isnan(img1).sum()), int(np.isnan(img2).sum())))
    print("  oracle stats", {k: st1[k] for k in ("closest_rays", "shadow_rays", "tlas_nodes", "blas_nodes", "instances", "triangles", "tri_shading", "shade_events", "panics", "tlas_ties", "sphere_inside")})
    print("  gpu    stats", {k: st2[k] for k in ("closest_rays", "shadow_rays", "tlas_nodes", "blas_nodes", "instances", "triangles", "tri_shading", "shade_events", "shadow_tlas_nodes", "shadow_blas_nodes", "shadow_triangles")})
    print("  gpu ms", {k: round(st2[k], 3) for k in ("ms_raygen", "ms_extend", "ms_shade", "ms_shadow", "ms_accumulate", "ms_total")})
    if neq.sum():
        ys, xs = np.nonzero(neq)
        y, x = ys[0], xs[0]
        print("  first differing pixel", y, x, img1[y, x], img2[y, x])
        for s in range(sx * sy):
            r = ctx.sample_radiance(s, sx, sy, depth, 1)[y, x]
            tr = osc.trace_sample(y, x, s, sx, sy, depth, 1)
            ro = np.array(list(tr.radiance), dtype=np.float32)
            if (r.view(np.uint32) != ro.view(np.uint32)).any():
                print("   sample", s, "gpu", r, "oracle", ro, "bounces", tr.n_bounces)
                for b in range(tr.n_bounces):
                    bt = tr.bounce[b]
                    print("    b", b, "hit", bt.hit, "t", bt.t, "inst", bt.inst, "prim", bt.prim, "Lnee", list(bt.radiance_after_nee), "f", list(bt.f), "pr", bt.pr, bt.pr_is_mass, "beta", list(bt.beta_after))
                break
    return osc, hs

probe("c1", 64, 64, 2, 2, 4)
probe("c2", 64, 64, 2, 2, 8)
probe("c3", 64, 64, 2, 2, 8)
probe("c4", 64, 48, 2, 2, 8, nx=64, nz=128)
probe("c5", 96, 54, 2, 2, 8)
