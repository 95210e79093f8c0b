"""ctypes bindings of the product path: libpbrs_host.so (include/pbrs_host.h) and libpbrs_gpu.so
(include/pbrs_gpu.h).

`HostScene` mirrors the reference's `Scene::new(*tlas::build_bvh(instances), camera)
.with_lights(..)` (scene/src/lib.rs:36-63, :118-126); `Context.render` replaces the frame loop of
src/main.rs:192-231 for one tile.  There is no CPU fallback: if the HIP library is missing or the
device call fails, these raise.
"""
import ctypes as C
import os

import numpy as np

from .spec import SceneSpec

from . import spec as _spec

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")


class PbrsError(RuntimeError):
    pass


class Node(C.Structure):
    _fields_ = [("min", C.c_float * 3), ("a", C.c_uint32), ("max", C.c_float * 3), ("b", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [("n_tlas_nodes", C.c_uint32), ("tlas_nodes", C.c_void_p), ("tlas_height", C.c_uint32),
                ("n_instances", C.c_uint32), ("instances", C.c_void_p),
                ("n_shapes", C.c_uint32), ("shapes", C.c_void_p),
                ("n_meshes", C.c_uint32), ("meshes", C.c_void_p),
                ("n_blas_nodes", C.c_uint32), ("blas_nodes", C.c_void_p),
                ("n_triangles", C.c_uint32), ("tri_verts", C.c_void_p), ("tri_shade", C.c_void_p),
                ("n_materials", C.c_uint32), ("materials", C.c_void_p),
                ("n_bxdfs", C.c_uint32), ("bxdfs", C.c_void_p),
                ("n_area_lights", C.c_uint32), ("area_lights", C.c_void_p),
                ("n_delta_lights", C.c_uint32), ("delta_lights", C.c_void_p),
                ("env_constant", C.c_float * 3), ("env_kind", C.c_uint32),
                ("n_textures", C.c_uint32), ("textures", C.c_void_p),
                ("n_tex_floats", C.c_uint32), ("tex_floats", C.c_void_p),
                ("n_tex_words", C.c_uint32), ("tex_words", C.c_void_p),
                ("env_texture", C.c_uint32), ("env_scale", C.c_float * 3),
                ("n_fourier_tables", C.c_uint32), ("fourier_tables", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("width", C.c_uint32), ("c", C.c_float * 3), ("height", C.c_uint32),
                ("a", C.c_float * 3), ("pad0", C.c_float), ("b", C.c_float * 3), ("pad1", C.c_float)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in (
        "samples", "closest_rays", "shadow_rays", "shade_events", "tlas_nodes", "blas_nodes", "instances", "instance_hits",
        "triangles", "tri_shading", "spheres", "quads", "cuboids", "disks", "shadow_tlas_nodes", "shadow_blas_nodes",
        "shadow_instances", "shadow_triangles", "shadow_prims", "invalid_samples")] +
                [(n, C.c_float) for n in ("ms_raygen", "ms_extend", "ms_shade", "ms_shadow", "ms_accumulate", "ms_total")] +
                [(n, C.c_uint32) for n in ("launches_extend", "launches_shadow", "launches_shade", "passes", "kernel_features_extend", "kernel_features_shadow")] +
                [("paths_at_bounce", C.c_uint64 * 16), ("shadow_rays_at_bounce", C.c_uint64 * 16)])

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n.endswith("_at_bounce") else getattr(self, n)) for n, _ in self._fields_}


class RenderParams(C.Structure):
    _fields_ = [("x0", C.c_uint32), ("y0", C.c_uint32), ("w", C.c_uint32), ("h", C.c_uint32), ("strata_x", C.c_uint32),
                ("strata_y", C.c_uint32), ("max_depth", C.c_uint32), ("samples_per_pass", C.c_uint32), ("seed", C.c_uint64),
                ("collect_counters", C.c_uint32), ("time_stages", C.c_uint32),
                ("band_rows", C.c_uint32), ("band_count", C.c_uint32), ("band_index", C.c_uint32), ("integrator", C.c_uint32)]


HIT_DTYPE = np.dtype([("t", np.float32), ("inst", np.uint32), ("prim", np.uint32), ("b1", np.float32), ("b2", np.float32)])
NUMERIC_FNS = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "atan2": 4, "acos": 5, "exp": 6, "ln": 7, "hypot": 8, "div": 9,
               "sqrt": 10, "asin": 11, "powi": 12, "fract": 13, "floor": 14, "box_quotient": 15}

GPU_SYMBOLS = ["pbrs_create", "pbrs_destroy", "pbrs_last_error", "pbrs_set_stream", "pbrs_set_pass_overlap", "pbrs_upload_scene", "pbrs_render_tile",
               "pbrs_render_tile_device", "pbrs_collect_stats", "pbrs_intersect_rays", "pbrs_last_intersect_info", "pbrs_camera_rays",
               "pbrs_numeric_eval", "pbrs_render_sample_radiance"]
HOST_SYMBOLS = ["pbrs_host_scene_build", "pbrs_host_scene_free", "pbrs_host_scene_desc", "pbrs_host_scene_camera",
                "pbrs_host_scene_stack_depth", "pbrs_host_last_error",
                "pbrs_host_load_pbrt", "pbrs_loaded_scene_spec", "pbrs_loaded_scene_free", "pbrs_host_load_error",
                "pbrs_host_write_exr", "pbrs_host_write_png", "pbrs_host_io_error"]

_host = None
_gpu = None


def lib_paths():
    return os.path.join(_LIBDIR, "libpbrs_host.so"), os.path.join(_LIBDIR, "libpbrs_gpu.so")


def host_lib():
    global _host
    if _host is None:
        # PBRS_HOST_LIB: the sanitizer build of the host library (tools/cpu_asan.sh: make -C pbrs_amd/csrc host-asan)
        path = os.environ.get("PBRS_HOST_LIB") or lib_paths()[0]
        if not os.path.exists(path):
            raise PbrsError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (make -C pbrs_amd/csrc)")
        L = C.CDLL(path)
        L.pbrs_host_scene_build.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.pbrs_host_scene_free.argtypes = [C.c_void_p]
        L.pbrs_host_scene_desc.restype = C.POINTER(SceneDesc)
        L.pbrs_host_scene_desc.argtypes = [C.c_void_p]
        L.pbrs_host_scene_camera.restype = C.POINTER(Camera)
        L.pbrs_host_scene_camera.argtypes = [C.c_void_p]
        L.pbrs_host_scene_stack_depth.restype = C.c_uint32
        L.pbrs_host_scene_stack_depth.argtypes = [C.c_void_p]
        L.pbrs_host_last_error.restype = C.c_char_p
        L.pbrs_host_load_pbrt.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.pbrs_loaded_scene_spec.restype = C.POINTER(SceneSpec)
        L.pbrs_loaded_scene_spec.argtypes = [C.c_void_p]
        L.pbrs_loaded_scene_free.argtypes = [C.c_void_p]
        L.pbrs_host_load_error.restype = C.c_char_p
        L.pbrs_host_write_exr.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pbrs_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.pbrs_host_io_error.restype = C.c_char_p
        _host = L
    return _host


def gpu_lib():
    """Loads the HIP library. Raises (never falls back) when it is absent or cannot be loaded."""
    global _gpu
    if _gpu is None:
        # PBRS_GPU_LIB: developer override to A/B two builds of the HIP library in one session (tools/ablate.sh)
        path = os.environ.get("PBRS_GPU_LIB") or lib_paths()[1]
        if not os.path.exists(path):
            raise PbrsError(f"{path} is missing: the HIP extension must be built (make -C pbrs_amd/csrc); there is no CPU fallback")
        L = C.CDLL(path)
        L.pbrs_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.pbrs_destroy.argtypes = [C.c_void_p]
        L.pbrs_last_error.restype = C.c_char_p
        L.pbrs_last_error.argtypes = [C.c_void_p]
        L.pbrs_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.pbrs_set_pass_overlap.argtypes = [C.c_void_p, C.c_int]
        L.pbrs_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.pbrs_render_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pbrs_render_tile_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pbrs_collect_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.pbrs_intersect_rays.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
        L.pbrs_last_intersect_info.argtypes = [C.c_void_p, C.c_void_p]
        L.pbrs_camera_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.pbrs_numeric_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pbrs_render_sample_radiance.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        _gpu = L
    return _gpu


INTEGRATORS = {"path": 0, "direct": 1, "materials": 2, "normals": 3}  # PBRS_INTEGRATOR_*


class LoadedScene:
    """A scene read from a pbrt-v3 file by the host library (include/pbrs_host.h, pbrs_host_load_pbrt).  Stands where a
    SceneBuilder stands: `build()` returns the scene spec, so HostScene(...) and the oracle take it as they take a builder."""

    def __init__(self, path):
        h = C.c_void_p()
        rc = host_lib().pbrs_host_load_pbrt(os.fsencode(path), C.byref(h))
        if rc != 0:
            raise PbrsError(f"pbrs_host_load_pbrt({path}) failed ({rc}): {host_lib().pbrs_host_load_error().decode()}")
        self._h = h
        self.spec = host_lib().pbrs_loaded_scene_spec(h).contents

    def build(self):
        return self.spec

    def close(self):
        if getattr(self, "_h", None):
            host_lib().pbrs_loaded_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def load_pbrt(path):
    return LoadedScene(path)


def write_image(path, rgb):
    """`write_exr` (path ending in .exr: f32 RGB) or `write_image` (.png: 8-bit, sqrt-gamma) of src/main.rs:28-53 for an
    (h, w, 3) radiance array."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w, _ = rgb.shape
    fn = host_lib().pbrs_host_write_exr if str(path).lower().endswith(".exr") else host_lib().pbrs_host_write_png
    rc = fn(os.fsencode(path), rgb.ctypes.data, w, h)
    if rc != 0:
        raise PbrsError(f"writing {path} failed ({rc}): {host_lib().pbrs_host_io_error().decode()}")


class HostScene:
    """Flattened scene: TLAS/BLAS built with the reference's algorithms, linearised for HBM."""

    def __init__(self, scene_builder):
        self._sb = scene_builder
        self._spec = scene_builder.build()
        h = C.c_void_p()
        rc = host_lib().pbrs_host_scene_build(C.addressof(self._spec), C.byref(h))
        if rc != 0:
            raise PbrsError(f"pbrs_host_scene_build failed ({rc}): {host_lib().pbrs_host_last_error().decode()}")
        self._h = h
        self.desc = host_lib().pbrs_host_scene_desc(h).contents
        self.camera = host_lib().pbrs_host_scene_camera(h).contents
        self.width, self.height = self.camera.width, self.camera.height

    @property
    def stack_depth(self):
        return host_lib().pbrs_host_scene_stack_depth(self._h)

    @property
    def nbytes(self):
        """Bytes of the flattened scene as it sits in HBM (record sizes of include/pbrs_gpu.h)."""
        d = self.desc
        return (32 * (d.n_tlas_nodes + d.n_blas_nodes) + 128 * d.n_instances + 48 * d.n_shapes + 32 * d.n_meshes +
                (48 + 64) * d.n_triangles + 32 * d.n_materials + 64 * d.n_bxdfs + 64 * d.n_area_lights + 32 * d.n_delta_lights +
                48 * d.n_textures + 4 * (d.n_tex_floats + d.n_tex_words) + 48 * d.n_fourier_tables)

    def nodes(self, which="tlas"):
        n, p = (self.desc.n_tlas_nodes, self.desc.tlas_nodes) if which == "tlas" else (self.desc.n_blas_nodes, self.desc.blas_nodes)
        if n == 0 or not p:
            return np.zeros((0, 8), dtype=np.uint32)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n, 8)).copy()

    def close(self):
        if getattr(self, "_h", None):
            host_lib().pbrs_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


class Context:
    """One per GPU (the per-thread state of the reference's rayon row loop, src/main.rs:219-224)."""

    def __init__(self, device=0):
        self._L = gpu_lib()
        h = C.c_void_p()
        rc = self._L.pbrs_create(device, C.byref(h))
        if rc != 0:
            raise PbrsError(f"pbrs_create(device={device}) failed ({rc}): no usable HIP device; there is no CPU fallback")
        self._h = h
        self.scene = None

    def _check(self, rc, what):
        if rc != 0:
            raise PbrsError(f"{what} failed ({rc}): {self._L.pbrs_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._L.pbrs_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_stream(self, hip_stream_ptr):
        self._check(self._L.pbrs_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "pbrs_set_stream")

    def set_pass_overlap(self, enabled):
        """Late bounces of a pass beside the next pass's first bounces, on a second stream (default on; include/pbrs_gpu.h)."""
        self._check(self._L.pbrs_set_pass_overlap(self._h, int(bool(enabled))), "pbrs_set_pass_overlap")

    def upload(self, host_scene):
        self._check(self._L.pbrs_upload_scene(self._h, C.addressof(host_scene.desc)), "pbrs_upload_scene")
        self.scene = host_scene

    def _params(self, strata_x, strata_y, depth, seed, tile, samples_per_pass=0, counters=False, timing=False, bands=None,
                integrator="path"):
        x0, y0, w, h = tile or (0, 0, self.scene.width, self.scene.height)
        p = RenderParams()
        if bands:
            p.band_rows, p.band_count, p.band_index = bands
        p.x0, p.y0, p.w, p.h = x0, y0, w, h
        p.strata_x, p.strata_y, p.max_depth, p.samples_per_pass = strata_x, strata_y, depth, samples_per_pass
        p.seed, p.collect_counters, p.time_stages = seed, int(counters), int(timing)
        p.integrator = INTEGRATORS[integrator]
        return p

    def render(self, strata_x, strata_y, depth, seed, tile=None, samples_per_pass=0, counters=False, timing=False, bands=None,
               integrator="path"):
        """-> (h, w, 3) f32 radiance, stats dict.  Host output (one D2H copy at the end).
        bands = (band_rows, band_count, band_index): the tile's rows are interleaved row bands.
        integrator: "path" (src/pathintegrator.rs) or "direct" (direct_lighting_integrator, src/directlighting.rs:14-47)."""
        p = self._params(strata_x, strata_y, depth, seed, tile, samples_per_pass, counters, timing, bands, integrator)
        out = np.empty((p.h, p.w, 3), dtype=np.float32)
        st = Stats()
        self._check(self._L.pbrs_render_tile(self._h, C.addressof(self.scene.camera), C.addressof(p), out.ctypes.data, C.addressof(st)),
                    "pbrs_render_tile")
        return out, st.as_dict()

    def render_device(self, rgb_device_ptr, strata_x, strata_y, depth, seed, tile=None, samples_per_pass=0, counters=False,
                      timing=False, bands=None, integrator="path"):
        """Asynchronous: the result lands in caller-owned device memory on the context's own NON-BLOCKING stream.  It is valid
        after `collect_stats()` (which waits for that stream), not merely after work queued later on torch's or the default
        stream: those are not ordered against it (include/pbrs_gpu.h, "Stream ordering")."""
        p = self._params(strata_x, strata_y, depth, seed, tile, samples_per_pass, counters, timing, bands, integrator)
        self._check(self._L.pbrs_render_tile_device(self._h, C.addressof(self.scene.camera), C.addressof(p), C.c_void_p(rgb_device_ptr), None),
                    "pbrs_render_tile_device")

    def collect_stats(self):
        st = Stats()
        self._check(self._L.pbrs_collect_stats(self._h, C.addressof(st)), "pbrs_collect_stats")
        return st.as_dict()

    def intersect(self, origins, dirs, tmax, closest=True, anyhit=True):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(tmax)
        hits = np.empty(n, dtype=HIT_DTYPE) if closest else None
        occ = np.empty(n, dtype=np.uint8) if anyhit else None
        self._check(self._L.pbrs_intersect_rays(self._h, n, origins.ctypes.data, dirs.ctypes.data, tmax.ctypes.data,
                                                hits.ctypes.data if closest else None, occ.ctypes.data if anyhit else None),
                    "pbrs_intersect_rays")
        return hits, occ

    def last_intersect_info(self):
        """Which walks the last intersect() went through (include/pbrs_gpu.h, pbrs_intersect_info)."""
        v = (C.c_uint32 * 4)()
        self._check(self._L.pbrs_last_intersect_info(self._h, C.addressof(v)), "pbrs_last_intersect_info")
        return {"wide_any": int(v[0]), "wide_closest": int(v[1]), "slow_any": int(v[2]), "slow_closest": int(v[3])}

    def camera_rays(self, sample, strata_x, strata_y, seed, tile=None):
        p = self._params(strata_x, strata_y, 1, seed, tile)
        o = np.empty((p.w * p.h, 3), dtype=np.float32)
        d = np.empty((p.w * p.h, 3), dtype=np.float32)
        self._check(self._L.pbrs_camera_rays(self._h, C.addressof(self.scene.camera), C.addressof(p), sample, o.ctypes.data, d.ctypes.data),
                    "pbrs_camera_rays")
        return o, d

    def sample_radiance(self, sample, strata_x, strata_y, depth, seed, tile=None, integrator="path"):
        p = self._params(strata_x, strata_y, depth, seed, tile, integrator=integrator)
        out = np.empty((p.h, p.w, 3), dtype=np.float32)
        self._check(self._L.pbrs_render_sample_radiance(self._h, C.addressof(self.scene.camera), C.addressof(p), sample, out.ctypes.data),
                    "pbrs_render_sample_radiance")
        return out

    def numeric_eval(self, fn, x, y=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        yp = None
        if y is not None:
            y = np.ascontiguousarray(y, dtype=np.float32)
            yp = y.ctypes.data
        self._check(self._L.pbrs_numeric_eval(self._h, NUMERIC_FNS[fn], x.size, x.ctypes.data, yp, out.ctypes.data), "pbrs_numeric_eval")
        return out
