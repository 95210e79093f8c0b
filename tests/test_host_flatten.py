"""Host flattener (pbrs_amd/csrc/host): tree shape follows the reference's builders and agrees with the oracle's
independent restatement; flattened records are well-formed.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

import pbrs_amd
from common import GOLDEN_NAMES, golden_case
from oracle.binding import OracleScene

LEAF = 0x80000000


def node_arrays(hs, which):
    raw = hs.nodes(which)
    boxes = raw[:, [0, 1, 2, 4, 5, 6]].copy().view(np.float32)
    return boxes, raw[:, 3], raw[:, 7]


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_trees_are_well_formed_and_match_the_oracle(name):
    sb, _ = golden_case(name)
    hs = pbrs_amd.HostScene(sb)
    osc = OracleScene(sb)
    d = hs.desc
    assert d.tlas_height == osc.tlas_height()
    assert d.n_tlas_nodes == 2 * d.n_instances - 1  # binary tree with one instance per leaf (tlas/src/bvh.rs:116-152)
    # tri_verts[] also holds one record per IsolatedTriangle shape (after the mesh triangles): those belong to no BLAS leaf
    n_mesh_tris = int(_meshes(hs)[:, 3].sum()) if d.n_meshes else 0
    inst = _instances(hs)
    n_isolated = len(set(inst[inst[:, 24] == 4, 28].tolist()))
    assert d.n_triangles >= n_mesh_tris + n_isolated
    for which, n_leaf_items in (("tlas", d.n_instances), ("blas", n_mesh_tris)):
        boxes, a, b = node_arrays(hs, which)
        if len(a) == 0:
            continue
        leaf = (b & LEAF) != 0
        inner = np.nonzero(~leaf)[0]
        # children of inner nodes are enclosed by the parent (BvhNode::geometric_sound, tlas/src/bvh.rs:62-71)
        for i in inner:
            for c in (i + 1, a[i]):
                assert (boxes[c, :3] >= boxes[i, :3]).all() and (boxes[c, 3:] <= boxes[i, 3:]).all()
        if which == "blas":
            counts = (b[leaf] & ~np.uint32(LEAF)).astype(np.int64)
            assert counts.sum() == n_leaf_items and counts.min() >= 1
            assert (b[inner] <= 2).all()  # split axis
            # leaves tile the triangle array without gaps, in pre-order
            starts = a[leaf].astype(np.int64)
            assert (np.sort(starts) == np.concatenate([[0], np.cumsum(counts[np.argsort(starts)])[:-1]])).all()
        else:
            assert sorted(a[leaf].tolist()) == list(range(d.n_instances))


def test_blas_leaf_size_and_height():
    """recursive_build: leaves hold <= 4 triangles unless the centroids coincide (shape/src/blas.rs:338, :354-360)."""
    from pbrs_amd import scenes
    sb, _ = scenes.build_config("c4", width=32, height=32, nx=32, nz=48)
    hs = pbrs_amd.HostScene(sb)
    _, a, b = node_arrays(hs, "blas")
    leaf = (b & LEAF) != 0
    assert ((b[leaf] & ~np.uint32(LEAF)) <= 4).all()
    assert hs.stack_depth >= hs.desc.tlas_height + 2


def test_q11_vertex_swap_is_baked_in():
    """`let (i, k, j) = tri.index_triple` (shape/src/blas.rs:162): the flattened triangle reads (v0, v2, v1)."""
    from pbrs_amd.spec import SceneBuilder, deg
    sb = SceneBuilder()
    m = sb.lambertian((0.5, 0.5, 0.5))
    pos = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    sb.instance(sb.mesh(pos, [(0, 0, 1)] * 3, [(0, 0), (1, 0), (0, 1)], [(0, 1, 2)]), m)
    sb.set_camera(8, 8, deg(40.0), (0, 0, -5), (0, 0, 0))
    hs = pbrs_amd.HostScene(sb)
    tv = np.ctypeslib.as_array(C.cast(hs.desc.tri_verts, C.POINTER(C.c_float)), shape=(1, 12))
    assert tv[0, 0:3].tolist() == [0, 0, 0] and tv[0, 4:7].tolist() == [0, 1, 0] and tv[0, 8:11].tolist() == [1, 0, 0]


def test_rejects_inconsistent_specs():
    from pbrs_amd.spec import SceneBuilder, deg
    sb = SceneBuilder()
    sb.set_camera(8, 8, deg(40.0), (0, 0, -5), (0, 0, 0))
    with pytest.raises(pbrs_amd.PbrsError, match="empty instances"):
        pbrs_amd.HostScene(sb)


def _instances(hs):
    n = hs.desc.n_instances
    return np.ctypeslib.as_array(C.cast(hs.desc.instances, C.POINTER(C.c_uint32)), shape=(n, 32)).copy()


def _meshes(hs):
    n = hs.desc.n_meshes
    return np.ctypeslib.as_array(C.cast(hs.desc.meshes, C.POINTER(C.c_uint32)), shape=(n, 8)).copy()


def test_precomputed_triangle_normals_and_flags():
    """tri_verts carries `(p0-p1).cross(p2-p1).try_hat()` (simple.rs:436); flat-shaded meshes whose triangles pass the
    tangent check of blas.rs:193-200 are flagged; identity instances are flagged; TLAS leaves carry the shape kind."""
    from pbrs_amd import scenes
    sb, _ = scenes.build_config("c2", width=16, height=16)
    hs = pbrs_amd.HostScene(sb)
    tv = np.ctypeslib.as_array(C.cast(hs.desc.tri_verts, C.POINTER(C.c_float)), shape=(hs.desc.n_triangles, 12)).copy()
    p0, p1, p2, n = tv[:, 0:3], tv[:, 4:7], tv[:, 8:11], tv[:, [3, 7, 11]]
    c = np.cross((p0 - p1).astype(np.float64), (p2 - p1).astype(np.float64))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    assert np.abs(n - c).max() < 1e-6
    meshes, inst = _meshes(hs), _instances(hs)
    assert (meshes[:, 5] & 1).all(), "every Cornell mesh is flat shaded and passes the Q22 check"
    is_mesh = inst[:, 24] == 5
    identity = (inst[:, 27] & 1) != 0
    assert identity.sum() == 8 and (~identity & is_mesh).sum() == 2  # 6 walls + 2 light triangles; the two rotated boxes are not
    assert (inst[is_mesh, 28] == meshes[inst[is_mesh, 25], 0]).all() and (inst[is_mesh, 29] == meshes[inst[is_mesh, 25], 5]).all()
    _, a, b = node_arrays(hs, "tlas")
    leaf = (b & LEAF) != 0
    assert (((b[leaf] >> 8) & 7) == inst[a[leaf], 24]).all()
    # smooth normals (terrain): no flat-shading shortcut
    sb, _ = scenes.build_config("c4", width=16, height=16, nx=8, nz=8)
    hs = pbrs_amd.HostScene(sb)
    assert (_meshes(hs)[0, 5] & 1) == 0
    # ... but the tangent check is provably passed by every hit (normals near the face normal, dpdu in the face plane)
    assert (_meshes(hs)[0, 5] & 2) != 0


def test_shading_flags_of_skewed_normals_and_isolated_triangle_records():
    """Vertex normals that lean into the tangent direction defeat the smooth-shading bound (the kernels then evaluate
    blas.rs:193-200 per candidate hit); an IsolatedTriangle shape gets a triangle record that its instance points at."""
    from pbrs_amd.spec import SceneBuilder, deg
    sb = SceneBuilder()
    m = sb.lambertian((0.5, 0.5, 0.5))
    pos = [(-1, 0, -1), (1, 0, -1), (-1, 0, 1), (1, 0, 1)]
    uv = [(0, 0), (1, 0), (0, 1), (1, 1)]
    sb.instance(sb.mesh(pos, [(0, 0.0001 + 0.0002 * k, 1) for k in range(4)], uv, [(0, 1, 2), (2, 1, 3)]), m)  # normals along dpdu
    sb.instance(sb.mesh(pos, [(0.3, 1, 0), (0, 1, 0.2), (0, 1, 0), (-0.1, 1, 0)], uv, [(0, 1, 2), (2, 1, 3)]), m)  # ordinary smooth normals
    sb.instance(sb.triangle((-1, 2, 0), (1, 2, 0.5), (0, 3.5, 0)), m)
    sb.set_camera(8, 8, deg(40.0), (0, 3, -5), (0, 0, 0))
    hs = pbrs_amd.HostScene(sb)
    meshes, inst = _meshes(hs), _instances(hs)
    assert (meshes[0, 5] & 3) == 0 and (meshes[1, 5] & 3) == 2
    assert hs.desc.n_triangles == 5 and inst[2, 24] == 4 and inst[2, 28] == 4
    tv = np.ctypeslib.as_array(C.cast(hs.desc.tri_verts, C.POINTER(C.c_float)), shape=(5, 12))
    assert (tv[4, [0, 1, 2]] == np.float32([-1, 2, 0])).all() and (tv[4, [8, 9, 10]] == np.float32([0, 3.5, 0])).all()


def test_degenerate_triangle_gets_nan_normal():
    from pbrs_amd.spec import SceneBuilder, deg
    sb = SceneBuilder()
    m = sb.lambertian((0.5, 0.5, 0.5))
    sb.instance(sb.mesh([(0, 0, 0), (1, 0, 0), (2, 0, 0), (0, 1, 0)], [(0, 0, 1)] * 4, [(0, 0), (1, 0), (0, 1), (1, 1)], [(0, 1, 2), (0, 1, 3)]), m)
    sb.set_camera(8, 8, deg(40.0), (0, 0, -5), (0, 0, 0))
    hs = pbrs_amd.HostScene(sb)
    tv = np.ctypeslib.as_array(C.cast(hs.desc.tri_verts, C.POINTER(C.c_float)), shape=(2, 12))
    nans = np.isnan(tv[:, 3])
    assert nans.sum() == 1  # the collinear triangle: try_hat() is None
