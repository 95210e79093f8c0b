"""Developer tool (GPU box): one pixel of a randomised parity scene (tests/test_gpu_fuzz.py) sample by sample and depth by depth, GPU
(pbrs_render_sample_radiance) against the oracle's path trace.   python tools/repro_fuzz_pixel.py SEED ROW COL"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pbrs_amd
import test_gpu_fuzz as F
from oracle.binding import OracleScene
seed, row, col = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sb = F.random_scene(seed)
hs = pbrs_amd.HostScene(sb)
ctx = pbrs_amd.Context(0)
ctx.upload(hs)
osc = OracleScene(sb)
rseed = 11 + seed
for s in range(4):
    for depth in range(1, 9):
        g = ctx.sample_radiance(s, 2, 2, depth, rseed, tile=(col, row, 1, 1))[0, 0]
        tr = osc.trace_sample(row, col, s, 2, 2, depth, rseed)
        o = np.array(list(tr.radiance), dtype=np.float32)
        same = (g.view(np.uint32) == o.view(np.uint32)).all()
        if not same:
            print("sample", s, "depth", depth, "GPU", g, "oracle", o, "n_bounces", tr.n_bounces, "panics", tr.panics)
            for b in range(tr.n_bounces):
                bt = tr.bounce[b]
                print("   bounce", b, "hit", bt.hit, "t", bt.t, "inst", bt.inst, "prim", bt.prim, "pos", list(bt.pos), "n", list(bt.normal))
                print("          L after nee", list(bt.radiance_after_nee), "f", list(bt.f), "wi", list(bt.wi), "pr", bt.pr, "mass", bt.pr_is_mass, "beta", list(bt.beta_after))
            f32 = np.float32
            for b in range(tr.n_bounces - 1):  # the ray each bounce spawns (interaction.rs:63-66), closest hit on both sides
                bt = tr.bounce[b]
                pos, nrm, wi = (np.array(list(x), dtype=f32) for x in (bt.pos, bt.normal, bt.wi))
                dn = f32(0)
                for k in range(3):  # dot as the contract does it: x*x' + y*y' + z*z' left to right
                    dn = f32(dn + f32(wi[k] * nrm[k])) if k else f32(wi[k] * nrm[k])
                sg = f32(1.0) if dn > 0 else (f32(-1.0) if dn < 0 else f32(0.0))
                org = (pos + (sg * nrm) * f32(0.001)).astype(f32)
                t = np.array([np.inf], dtype=f32)
                ho, _, _ = osc.intersect(org[None], wi[None], t)
                hg, _ = ctx.intersect(org[None], wi[None], t)
                print("   ray spawned at bounce", b, "o", org, "d", wi, "oracle", ho, "gpu", hg)
            break
    else:
        print("sample", s, "agrees at every depth")
spec_ = sb.build()
print("instances", spec_.n_instances)
for i in range(spec_.n_instances):
    ins = spec_.instances[i]
    print("  inst", i, "shape", ins.shape, "material", ins.material, "kind", spec_.shapes[ins.shape].kind)
