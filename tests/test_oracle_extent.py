"""The TLAS extent can RISE (tlas/src/bvh.rs:84-88 sets ray.t_max to the LEFT subtree's result; a mesh may return a hit beyond the extent it
was given, shape/src/blas.rs:468), and with a ParallelQuad in the scene the rise decides the closest hit: the quad's mirrored quadrants
(D1) report hits outside its own box.  The oracle restates the recursion as it stands; this pins what it answers for the ray of fuzz seed
211699 (round 4) and shows from the flattened TLAS why: the winning quad's box is entered BEHIND the best hit found before it.  The GPU
side of the same ray: tests/test_gpu_fuzz.py::test_a_raised_extent_reaches_a_mirrored_quad_hit.  No GPU needed."""
import numpy as np

import pbrs_amd
from oracle.binding import OracleScene
from pbrs_amd import spec
from test_gpu_fuzz import random_scene

LEAF = 0x80000000


def test_a_raised_extent_lets_the_oracle_reach_a_mirrored_quad_hit():
    sb = random_scene(211699)
    osc = OracleScene(sb)
    o = np.array([[1076550976, 1070745188, 3218686800]], dtype=np.uint32).view(np.float32)
    d = np.array([[0, 3212667273, 1041316617]], dtype=np.uint32).view(np.float32)
    h, _, st = osc.intersect(o, d, np.array([np.inf], dtype=np.float32))
    assert not st["tie_mask"].any()
    built = sb.build()
    quad = int(h["inst"][0])
    assert built.shapes[built.instances[quad].shape].kind == spec.SHAPE_QUAD
    t_quad = h["t"][0]
    assert t_quad.view(np.uint32) == np.float32(0.34959823).view(np.uint32)
    # the leaf boxes of the flattened TLAS (pre-order = the reference's visiting order), with the reference's slab arithmetic
    nodes = pbrs_amd.HostScene(sb).nodes("tlas")
    entered = {}
    with np.errstate(divide="ignore", invalid="ignore"):
        for n in nodes[(nodes[:, 7] & LEAF) != 0]:
            lo, hi = n[0:3].view(np.float32), n[4:7].view(np.float32)
            t0, t1 = (lo - o[0]) / d[0], (hi - o[0]) / d[0]
            t_low, t_high = max(np.minimum(t0, t1).max(), np.float32(0)), np.maximum(t0, t1).min()
            if t_low <= t_high:
                entered[int(n[3])] = float(t_low)
    assert quad in entered and entered[quad] > t_quad  # the hit lies in front of the quad's own box
    # some instance visited before the quad is hit nearer than the quad's box begins: with the extent at the best hit so far the box is pruned
    order = [int(n[3]) for n in nodes[(nodes[:, 7] & LEAF) != 0]]
    before = order[:order.index(quad)]
    t_inf = np.array([np.inf], dtype=np.float32)
    nearer = []
    for i in before:
        one = random_scene(211699)
        one.instances = [one.instances[i]]
        hi_, _, _ = OracleScene(one).intersect(o, d, t_inf)
        if hi_["inst"][0] != 0xFFFFFFFF:
            nearer.append(float(hi_["t"][0]))
    assert min(nearer) < entered[quad] < max(nearer)  # ... and one of them (a mesh, beyond the extent it was given) reaches past it
