"""ctypes mirror of include/pbrs_scene_spec.h plus a small builder.

The spec is the plain-data form of what the reference's scene/src/preset.rs hands to
`tlas::build_bvh` and `Scene::new(..).with_lights(..)`: instances of
(shape, material, transform), area/delta lights, a constant environment colour and the camera.
It carries no acceleration structure; the host flattener (pbrs_amd/csrc/host) builds that.
"""
import ctypes as C

import numpy as np

SHAPE_SPHERE, SHAPE_QUAD, SHAPE_CUBOID, SHAPE_DISK, SHAPE_TRIANGLE, SHAPE_MESH = range(6)
(MTL_LAMBERTIAN, MTL_METAL, MTL_GLOSSY, MTL_MIRROR, MTL_PLASTIC, MTL_DIELECTRIC, MTL_DIFFUSE_LIGHT, MTL_UBER,
 MTL_SUBSTRATE, MTL_FOURIER) = range(10)
MTL_FLAG_REMAP_ROUGHNESS, MTL_FLAG_HAS_KR, MTL_FLAG_HAS_KT = 1, 2, 4
DELTA_POINT, DELTA_DISTANT = 0, 1
TEX_CHECKER, TEX_PERLIN, TEX_IMAGE = 1, 2, 3
ENV_CONSTANT, ENV_IMAGE, ENV_BLUE_SKY, ENV_DARK_ROOM, ENV_DUSK = range(5)

f32 = np.float32


class ShapeSpec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mesh", C.c_uint32), ("p", C.c_float * 9)]


class MeshSpec(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32), ("positions", C.POINTER(C.c_float)),
                ("normals", C.POINTER(C.c_float)), ("uvs", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32))]


class TextureSpec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("odd", C.c_float * 3), ("even", C.c_float * 3), ("freq", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("data", C.POINTER(C.c_float)), ("perm", C.POINTER(C.c_uint32))]


class FourierTableSpec(C.Structure):
    _fields_ = [("n_mu", C.c_uint32), ("n_channels", C.c_uint32), ("n_coeffs", C.c_uint32), ("eta", C.c_float),
                ("mu", C.POINTER(C.c_float)), ("cdf", C.POINTER(C.c_float)), ("offset_and_length", C.POINTER(C.c_int32)),
                ("a", C.POINTER(C.c_float))]


class MaterialSpec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("flags", C.c_uint32), ("p", C.c_float * 16), ("tex", C.c_uint32 * 4)]


class InstanceSpec(C.Structure):
    _fields_ = [("shape", C.c_uint32), ("material", C.c_uint32), ("forward", C.c_float * 16), ("inverse", C.c_float * 16)]


class AreaLightSpec(C.Structure):
    _fields_ = [("emit", C.c_float * 3), ("shape", ShapeSpec)]


class DeltaLightSpec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("v", C.c_float * 3), ("color", C.c_float * 3), ("world_radius", C.c_float)]


class CameraSpec(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("fov_y_rad", C.c_float), ("from_", C.c_float * 3),
                ("target", C.c_float * 3), ("up", C.c_float * 3)]


class SceneSpec(C.Structure):
    _fields_ = [("n_meshes", C.c_uint32), ("meshes", C.POINTER(MeshSpec)),
                ("n_shapes", C.c_uint32), ("shapes", C.POINTER(ShapeSpec)),
                ("n_materials", C.c_uint32), ("materials", C.POINTER(MaterialSpec)),
                ("n_instances", C.c_uint32), ("instances", C.POINTER(InstanceSpec)),
                ("n_area_lights", C.c_uint32), ("area_lights", C.POINTER(AreaLightSpec)),
                ("n_delta_lights", C.c_uint32), ("delta_lights", C.POINTER(DeltaLightSpec)),
                ("env_constant", C.c_float * 3), ("camera", CameraSpec),
                ("n_textures", C.c_uint32), ("textures", C.POINTER(TextureSpec)),
                ("env_kind", C.c_uint32), ("env_texture", C.c_uint32), ("env_scale", C.c_float * 3),
                ("n_fourier_tables", C.c_uint32), ("fourier_tables", C.POINTER(FourierTableSpec))]


# ---- AffineTransform (geometry/src/transform.rs:133-194) in f32, column-major Mat4 -----------------

def _mat4_mul(a, b):
    """math/src/hcm.rs:546-556: column c of the product is a * b.cols[c]; sums run left to right in f32."""
    out = np.zeros((4, 4), dtype=f32)  # out[col][row]
    for c in range(4):
        acc = (a[0] * b[c][0]).astype(f32)
        for k in range(1, 4):
            acc = (acc + (a[k] * b[c][k]).astype(f32)).astype(f32)
        out[c] = acc
    return out


class Transform:
    """AffineTransform{forward, inverse}; matrices indexed [col][row] like glam's Mat4.cols."""

    def __init__(self, forward=None, inverse=None):
        self.forward = np.eye(4, dtype=f32) if forward is None else forward.astype(f32)
        self.inverse = np.eye(4, dtype=f32) if inverse is None else inverse.astype(f32)

    @staticmethod
    def translater(t):  # transform.rs:140-145
        fwd = np.eye(4, dtype=f32)
        inv = np.eye(4, dtype=f32)
        fwd[3, :3] = np.asarray(t, dtype=f32)
        inv[3, :3] = -np.asarray(t, dtype=f32)
        return Transform(fwd, inv)

    @staticmethod
    def rotater(axis, angle_rad):  # transform.rs:146-152, hcm.rs:508-520
        axis = np.asarray(axis, dtype=f32)
        s, c = f32(np.sin(f32(angle_rad))), f32(np.cos(f32(angle_rad)))
        fwd = np.eye(4, dtype=f32)
        ahat = (axis / f32(np.sqrt(f32(np.dot(axis, axis))))).astype(f32)
        for i in range(3):
            base = np.zeros(3, dtype=f32)
            base[i] = 1.0
            vc = (f32(np.dot(base, axis)) * axis / f32(np.dot(axis, axis))).astype(f32)
            v1 = (base - vc).astype(f32)
            v2 = np.cross(v1, ahat).astype(f32)
            fwd[i, :3] = (vc + v1 * c + v2 * s).astype(f32)
        return Transform(fwd, fwd.T.copy())

    def __matmul__(self, rhs):  # transform.rs:185-194: self * rhs
        return Transform(_mat4_mul(self.forward, rhs.forward), _mat4_mul(rhs.inverse, self.inverse))

    def translate(self, t):  # :169-171  Translate(t) * self
        return Transform.translater(t) @ self

    def rotate_y(self, angle_rad):  # :177-179
        return Transform.rotater([0, 1, 0], angle_rad) @ self

    def rotate_x(self, angle_rad):
        return Transform.rotater([1, 0, 0], angle_rad) @ self

    def rotate_z(self, angle_rad):
        return Transform.rotater([0, 0, 1], angle_rad) @ self


def deg(d):
    """f32::to_radians (math/src/float.rs:239-243)."""
    return f32(f32(d) * f32(np.pi / 180.0))


class SceneBuilder:
    """Collects numpy-backed arrays and emits a SceneSpec whose pointers stay valid while the
    builder is alive."""

    def __init__(self):
        self.meshes = []  # (positions, normals, uvs, indices)
        self.shapes = []
        self.materials = []
        self.instances = []
        self.area_lights = []
        self.delta_lights = []
        self.env = (0.0, 0.0, 0.0)
        self.env_kind, self.env_texture, self.env_scale = ENV_CONSTANT, 0, (1.0, 1.0, 1.0)
        self.textures = []  # (TextureSpec, arrays kept alive)
        self.fourier_tables = []  # FourierTableSpec (arrays kept alive in _keep)
        self.camera = None
        self._keep = []

    # -- shapes
    def _shape(self, kind, p=(), mesh=0):
        s = ShapeSpec()
        s.kind, s.mesh = kind, mesh
        for i, v in enumerate(p):
            s.p[i] = float(f32(v))
        return s

    def add_shape(self, s):
        self.shapes.append(s)
        return len(self.shapes) - 1

    def sphere(self, center, radius):
        return self._shape(SHAPE_SPHERE, list(center) + [radius])

    def quad(self, origin, side_u, side_v):
        return self._shape(SHAPE_QUAD, list(origin) + list(side_u) + list(side_v))

    def cuboid(self, p0, p1):
        return self._shape(SHAPE_CUBOID, list(p0) + list(p1))

    def disk(self, center, normal, radial):
        return self._shape(SHAPE_DISK, list(center) + list(normal) + list(radial))

    def triangle(self, p0, p1, p2):
        return self._shape(SHAPE_TRIANGLE, list(p0) + list(p1) + list(p2))

    def mesh(self, positions, normals, uvs, indices):
        positions = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
        normals = np.ascontiguousarray(normals, dtype=f32).reshape(-1, 3)
        uvs = np.ascontiguousarray(uvs, dtype=f32).reshape(-1, 2)
        indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        assert len(positions) == len(normals) == len(uvs)
        assert indices.max() < len(positions)
        self.meshes.append((positions, normals, uvs, indices))
        return self._shape(SHAPE_MESH, mesh=len(self.meshes) - 1)

    # -- textures (texture/src/lib.rs); returned handles go where a colour is expected in lambertian() / uber()
    class Tex(int):
        """Index of a texture in the scene."""

    def _texture(self, t, *keep):
        self.textures.append(t)
        self._keep.extend(keep)
        return SceneBuilder.Tex(len(self.textures) - 1)

    def checker(self, odd, even):
        t = TextureSpec()
        t.kind = TEX_CHECKER
        for i in range(3):
            t.odd[i], t.even[i] = float(f32(odd[i])), float(f32(even[i]))
        return self._texture(t)

    def perlin(self, freq=1.0, seed=1):
        """Perlin::with_freq (texture/src/lib.rs:66-95) with tables drawn from a seeded stream instead of `rand::random`:
        256 uniform unit vectors (uniform_random_sphere, :6-13) and three swap-shuffled permutations (:85-95)."""
        rs = np.random.RandomState(seed)
        u, v = rs.rand(256).astype(f32), rs.rand(256).astype(f32)
        theta = (f32(2.0 * np.pi) * u).astype(f32)
        cos_phi = (f32(2.0) * v - f32(1.0)).astype(f32)
        sin_phi = np.sqrt(np.maximum(f32(1.0) - cos_phi * cos_phi, f32(0.0))).astype(f32)
        vec = np.ascontiguousarray(np.stack([sin_phi * np.sin(theta), sin_phi * np.cos(theta), cos_phi], axis=1), dtype=f32)
        perm = np.empty((3, 256), dtype=np.uint32)
        for a in range(3):
            pm = np.arange(256, dtype=np.uint32)
            for i in range(256):
                j = int(rs.randint(0, 256))
                pm[i], pm[j] = pm[j], pm[i]
            perm[a] = pm
        t = TextureSpec()
        t.kind, t.freq = TEX_PERLIN, float(f32(freq))
        t.data = vec.ctypes.data_as(C.POINTER(C.c_float))
        t.perm = perm.ctypes.data_as(C.POINTER(C.c_uint32))
        return self._texture(t, vec, perm)

    def image(self, rgb):
        """Image texture from an (h, w, 3) array of colours in [0, 1] (`Color::rgb(u8..)` already applied, :199-203)."""
        rgb = np.ascontiguousarray(rgb, dtype=f32)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        t = TextureSpec()
        t.kind, t.height, t.width = TEX_IMAGE, rgb.shape[0], rgb.shape[1]
        t.data = rgb.ctypes.data_as(C.POINTER(C.c_float))
        return self._texture(t, rgb)

    def env_image(self, tex, scale=(1.0, 1.0, 1.0)):
        self.env_kind, self.env_texture, self.env_scale = ENV_IMAGE, int(tex), tuple(scale)

    def env_sky(self, kind):
        """kind: ENV_BLUE_SKY / ENV_DARK_ROOM / ENV_DUSK (scene/src/preset.rs:25-53)."""
        self.env_kind = kind

    # -- materials
    def material(self, kind, p, flags=0, tex=()):
        m = MaterialSpec()
        m.kind, m.flags = kind, flags
        for i, v in enumerate(p):
            m.p[i] = float(f32(v))
        for i, t in enumerate(tex):
            m.tex[i] = 0 if t is None else int(t) + 1
        self.materials.append(m)
        return len(self.materials) - 1

    @staticmethod
    def _colour_or_tex(c):
        """(constant colour, texture handle or None) for a parameter that may be either."""
        return ((0.0, 0.0, 0.0), c) if isinstance(c, SceneBuilder.Tex) else (tuple(c), None)

    def lambertian(self, albedo):
        c, t = self._colour_or_tex(albedo)
        return self.material(MTL_LAMBERTIAN, c, tex=(t,))

    def metal(self, eta, k, fuzziness):
        return self.material(MTL_METAL, list(eta) + list(k) + [fuzziness])

    def glossy(self, albedo, roughness):
        return self.material(MTL_GLOSSY, list(albedo) + [roughness])

    def mirror(self, albedo):
        return self.material(MTL_MIRROR, albedo)

    def plastic(self, diffuse, specular, roughness, remap_roughness=True):
        return self.material(MTL_PLASTIC, list(diffuse) + list(specular) + [roughness],
                             MTL_FLAG_REMAP_ROUGHNESS if remap_roughness else 0)

    def dielectric(self, ior, reflect=(1, 1, 1), transmit=(1, 1, 1)):
        return self.material(MTL_DIELECTRIC, [ior] + list(reflect) + list(transmit))

    def diffuse_light(self, emit):
        return self.material(MTL_DIFFUSE_LIGHT, emit)

    def uber(self, kd, ks, kr=None, kt=None, rough=(0.1, 0.1), eta=1.5, opacity=1.0, remap_roughness=True):
        flags = (MTL_FLAG_REMAP_ROUGHNESS if remap_roughness else 0) | (MTL_FLAG_HAS_KR if kr is not None else 0) | (
            MTL_FLAG_HAS_KT if kt is not None else 0)
        cols, texs = zip(*(self._colour_or_tex(c if c is not None else (0, 0, 0)) for c in (kd, ks, kr, kt)))
        p = [x for c in cols for x in c] + list(rough) + [eta, opacity]
        return self.material(MTL_UBER, p, flags, tex=texs)

    def substrate(self, kd, ks):
        return self.material(MTL_SUBSTRATE, list(kd) + list(ks))

    def fourier_table(self, table):
        """Registers a pbrs_amd.fourier.FourierTable (the arrays of a `.bsdf` file, geometry/src/fourier.rs:167-221);
        returns its index for fourier()."""
        mu = np.ascontiguousarray(table.mu, dtype=f32)
        cdf = np.ascontiguousarray(table.cdf, dtype=f32).reshape(-1)
        ol = np.ascontiguousarray(table.offset_and_length, dtype=np.int32).reshape(-1)
        a = np.ascontiguousarray(table.a, dtype=f32)
        assert len(mu) >= 3 and len(cdf) == len(mu) ** 2 and len(ol) == 2 * len(mu) ** 2 and table.n_channels in (1, 3)
        t = FourierTableSpec()
        t.n_mu, t.n_channels, t.n_coeffs, t.eta = len(mu), table.n_channels, len(a), float(f32(table.eta))
        t.mu, t.cdf = mu.ctypes.data_as(C.POINTER(C.c_float)), cdf.ctypes.data_as(C.POINTER(C.c_float))
        t.offset_and_length = ol.ctypes.data_as(C.POINTER(C.c_int32))
        t.a = a.ctypes.data_as(C.POINTER(C.c_float))
        self._keep.extend([mu, cdf, ol, a])
        self.fourier_tables.append(t)
        return len(self.fourier_tables) - 1

    def fourier(self, table_index):
        """material::Fourier (material/src/lib.rs:451-475) over a registered table."""
        m = self.material(MTL_FOURIER, [])
        self.materials[m].tex[0] = int(table_index)
        return m

    # -- instances / lights / camera
    def instance(self, shape, material, transform=None):
        """`shape` is a ShapeSpec (added) or an existing shape index."""
        sid = shape if isinstance(shape, int) else self.add_shape(shape)
        t = transform or Transform()
        inst = InstanceSpec()
        inst.shape, inst.material = sid, material
        for c in range(4):
            for r in range(4):
                inst.forward[4 * c + r] = float(t.forward[c][r])
                inst.inverse[4 * c + r] = float(t.inverse[c][r])
        self.instances.append(inst)
        return len(self.instances) - 1

    def area_light(self, emit, shape):
        a = AreaLightSpec()
        for i in range(3):
            a.emit[i] = float(f32(emit[i]))
        a.shape = shape
        self.area_lights.append(a)

    def point_light(self, position, intensity):
        d = DeltaLightSpec()
        d.kind = DELTA_POINT
        for i in range(3):
            d.v[i] = float(f32(position[i]))
            d.color[i] = float(f32(intensity[i]))
        d.world_radius = 0.0
        self.delta_lights.append(d)

    def distant_light(self, casting_dir, radiance, world_radius):
        d = DeltaLightSpec()
        d.kind = DELTA_DISTANT
        for i in range(3):
            d.v[i] = float(f32(casting_dir[i]))
            d.color[i] = float(f32(radiance[i]))
        d.world_radius = float(f32(world_radius))
        self.delta_lights.append(d)

    def set_camera(self, width, height, fov_y_rad, from_, target, up=(0, 1, 0)):
        cam = CameraSpec()
        cam.width, cam.height, cam.fov_y_rad = width, height, float(f32(fov_y_rad))
        for i in range(3):
            cam.from_[i] = float(f32(from_[i]))
            cam.target[i] = float(f32(target[i]))
            cam.up[i] = float(f32(up[i]))
        self.camera = cam

    def build(self):
        spec = SceneSpec()

        def arr(items, ctype):
            a = (ctype * max(len(items), 1))()
            for i, it in enumerate(items):
                a[i] = it
            self._keep.append(a)
            return a

        mesh_structs = []
        for (pos, nrm, uv, idx) in self.meshes:
            m = MeshSpec()
            m.n_vertices, m.n_triangles = len(pos), len(idx)
            m.positions = pos.ctypes.data_as(C.POINTER(C.c_float))
            m.normals = nrm.ctypes.data_as(C.POINTER(C.c_float))
            m.uvs = uv.ctypes.data_as(C.POINTER(C.c_float))
            m.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
            mesh_structs.append(m)
        spec.n_meshes, spec.meshes = len(mesh_structs), arr(mesh_structs, MeshSpec)
        spec.n_shapes, spec.shapes = len(self.shapes), arr(self.shapes, ShapeSpec)
        spec.n_materials, spec.materials = len(self.materials), arr(self.materials, MaterialSpec)
        spec.n_instances, spec.instances = len(self.instances), arr(self.instances, InstanceSpec)
        spec.n_area_lights, spec.area_lights = len(self.area_lights), arr(self.area_lights, AreaLightSpec)
        spec.n_delta_lights, spec.delta_lights = len(self.delta_lights), arr(self.delta_lights, DeltaLightSpec)
        for i in range(3):
            spec.env_constant[i] = float(f32(self.env[i]))
        assert self.camera is not None, "camera not set"
        spec.camera = self.camera
        spec.n_textures, spec.textures = len(self.textures), arr(self.textures, TextureSpec)
        spec.env_kind, spec.env_texture = self.env_kind, self.env_texture
        for i in range(3):
            spec.env_scale[i] = float(f32(self.env_scale[i]))
        spec.n_fourier_tables, spec.fourier_tables = len(self.fourier_tables), arr(self.fourier_tables, FourierTableSpec)
        return spec
