"""ctypes binding of oracle/libpbrs_oracle.so (see oracle/oracle_api.h). TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TRACE_MAX_BOUNCES = 16


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "closest_rays", "shadow_rays", "tlas_nodes", "blas_nodes", "instances", "instance_hits", "triangles", "spheres",
        "quads", "cuboids", "disks", "tri_shading", "shade_events", "samples", "panics", "tlas_ties", "sphere_inside", "nonfinite_samples")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


HIT_DTYPE = np.dtype([("t", np.float32), ("inst", np.uint32), ("prim", np.uint32), ("b1", np.float32), ("b2", np.float32)])


class BounceTrace(C.Structure):
    _fields_ = [("hit", C.c_uint32), ("t", C.c_float), ("inst", C.c_uint32), ("prim", C.c_uint32), ("b1", C.c_float),
                ("b2", C.c_float), ("pos", C.c_float * 3), ("normal", C.c_float * 3), ("radiance_after_nee", C.c_float * 3),
                ("f", C.c_float * 3), ("wi", C.c_float * 3), ("pr", C.c_float), ("pr_is_mass", C.c_uint32),
                ("beta_after", C.c_float * 3)]


class PathTrace(C.Structure):
    _fields_ = [("ray_o", C.c_float * 3), ("ray_d", C.c_float * 3), ("n_bounces", C.c_uint32),
                ("bounce", BounceTrace * TRACE_MAX_BOUNCES), ("radiance", C.c_float * 3), ("panics", C.c_uint32)]


def build():
    """Compile the oracle with its Makefile (g++ -O2 -ffp-contract=off)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        # PBRS_ORACLE_LIB: the sanitizer build (tools/cpu_asan.sh: make -C oracle asan)
        path = os.environ.get("PBRS_ORACLE_LIB") or os.path.join(_HERE, "libpbrs_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.oracle_scene_build.restype = C.c_void_p
        L.oracle_scene_build.argtypes = [C.c_void_p]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_tlas_height.restype = C.c_uint32
        L.oracle_tlas_height.argtypes = [C.c_void_p]
        L.oracle_render_tile.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_texture_value.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_env_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_render_tile_integrator.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_trace_sample.argtypes = [C.c_void_p] + [C.c_uint32] * 6 + [C.c_uint64, C.c_void_p]
        L.oracle_intersect_rays.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 7
        L.oracle_camera_rays.argtypes = [C.c_void_p] + [C.c_uint32] * 7 + [C.c_uint64, C.c_void_p, C.c_void_p]
        L.oracle_numeric_eval.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_rng_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_temperature_to_color.argtypes = [C.c_float, C.c_void_p]
        L.oracle_spd_to_color.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_selftest.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
        L.oracle_selftest_count.restype = C.c_uint32
        L.oracle_selftest_name.restype = C.c_char_p
        L.oracle_selftest_name.argtypes = [C.c_uint32]
        _LIB = L
    return _LIB


NUMERIC_FNS = {"sin": 0, "cos": 1, "tan": 2, "atan": 3, "atan2": 4, "acos": 5, "exp": 6, "ln": 7, "hypot": 8, "div": 9,
               "sqrt": 10, "asin": 11, "powi": 12, "fract": 13, "floor": 14}


def numeric_eval(fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float32)
        yp = y.ctypes.data
    rc = lib().oracle_numeric_eval(NUMERIC_FNS[fn], x.size, x.ctypes.data, yp, out.ctypes.data)
    assert rc == 0
    return out


def temperature_to_color(kelvin):
    """radiometry/src/spectrum.rs:38-55 -> (rgb, panics)."""
    out = np.empty(3, dtype=np.float32)
    panics = lib().oracle_temperature_to_color(float(kelvin), out.ctypes.data)
    return out, panics


def spd_to_color(lambdas_nm, values):
    """sampled_spectrum_to_color (radiometry/src/spectrum.rs:57-70) over (lambda, value) samples -> (rgb, panics)."""
    lam = np.ascontiguousarray(lambdas_nm, dtype=np.float32)
    val = np.ascontiguousarray(values, dtype=np.float32)
    out = np.empty(3, dtype=np.float32)
    panics = lib().oracle_spd_to_color(len(lam), lam.ctypes.data, val.ctypes.data, out.ctypes.data)
    return out, panics


def rng_stream(seed, pixel, sample, n):
    out = np.empty(n, dtype=np.float32)
    lib().oracle_rng_stream(seed, pixel, sample, n, out.ctypes.data)
    return out


class OracleScene:
    def __init__(self, scene_builder):
        self._sb = scene_builder
        self._spec = scene_builder.build()
        self.width = self._spec.camera.width
        self.height = self._spec.camera.height
        self._h = lib().oracle_scene_build(C.addressof(self._spec))

    def close(self):
        if self._h:
            lib().oracle_scene_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def tlas_height(self):
        return lib().oracle_tlas_height(self._h)

    def render(self, strata_x, strata_y, depth, seed, tile=None, nthreads=None, integrator="path"):
        """integrator: "path" (src/pathintegrator.rs), "direct" (direct_lighting_integrator, src/directlighting.rs:14-47) or
        "materials" / "normals" (material_visualizer :234-271, normal_visualizer :273-289; strata 1 x 1)."""
        x0, y0, w, h = tile or (0, 0, self.width, self.height)
        out = np.empty((h, w, 3), dtype=np.float32)
        st = Stats()
        nthreads = nthreads or os.cpu_count() or 1
        kind = {"path": 0, "direct": 1, "materials": 2, "normals": 3}[integrator]
        rc = lib().oracle_render_tile_integrator(self._h, x0, y0, w, h, strata_x, strata_y, depth, seed, nthreads, kind, out.ctypes.data,
                                                 C.addressof(st))
        assert rc == 0
        return out, st.as_dict()

    def env_eval(self, dirs):
        """Scene::eval_env_light for ray directions (n, 3)."""
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        out = np.empty_like(dirs)
        lib().oracle_env_eval(self._h, len(dirs), dirs.ctypes.data, out.ctypes.data)
        return out

    def trace_sample(self, row, col, sample, strata_x, strata_y, depth, seed):
        tr = PathTrace()
        lib().oracle_trace_sample(self._h, row, col, sample, strata_x, strata_y, depth, seed, C.addressof(tr))
        return tr

    def camera_rays(self, sample, strata_x, strata_y, seed, tile=None):
        x0, y0, w, h = tile or (0, 0, self.width, self.height)
        o = np.empty((h * w, 3), dtype=np.float32)
        d = np.empty((h * w, 3), dtype=np.float32)
        lib().oracle_camera_rays(self._h, x0, y0, w, h, sample, strata_x, strata_y, seed, o.ctypes.data, d.ctypes.data)
        return o, d

    def intersect(self, origins, dirs, tmax, closest=True, anyhit=True):
        origins = np.ascontiguousarray(origins, dtype=np.float32)
        dirs = np.ascontiguousarray(dirs, dtype=np.float32)
        tmax = np.ascontiguousarray(tmax, dtype=np.float32)
        n = len(tmax)
        hits = np.empty(n, dtype=HIT_DTYPE) if closest else None
        occ = np.empty(n, dtype=np.uint8) if anyhit else None
        st = Stats()
        ties = np.zeros(n, dtype=np.uint8)  # per ray: tlas/src/bvh.rs:94 saw equal t on both sides
        lib().oracle_intersect_rays(self._h, n, origins.ctypes.data, dirs.ctypes.data, tmax.ctypes.data,
                                    hits.ctypes.data if closest else None, occ.ctypes.data if anyhit else None, C.addressof(st),
                                    ties.ctypes.data)
        d = st.as_dict()
        d["tie_mask"] = ties.astype(bool)
        return hits, occ, d


def texture_value(texture_spec, uv, pos):
    """Texture::value(uv, p) of one pbrs_texture_spec for n (uv, p) pairs -> (n, 3) colours, panics reached."""
    uv = np.ascontiguousarray(uv, dtype=np.float32)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    out = np.empty((len(uv), 3), dtype=np.float32)
    panics = lib().oracle_texture_value(C.addressof(texture_spec), len(uv), uv.ctypes.data, pos.ctypes.data, out.ctypes.data)
    return out, panics
