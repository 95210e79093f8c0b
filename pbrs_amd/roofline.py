"""Algorithmic-bytes model of the wavefront stages (SURVEY.md §8(d) byte table).

achieved GB/s = algorithmic bytes moved by a stage / HIP-event time of that stage.  The counts come
from the instrumented kernel variant (pbrs_render_params.collect_counters), which is deterministic
and therefore equal to what the timed variant does.
"""
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)

B_CLOSEST_RAY = 52    # queue read (o 12, d 12, t_max 4, path id 4) + hit write (t, inst, prim, b1, b2 = 20)
B_SHADOW_RAY = 68     # queue read 32 + pending contribution 12 + radiance r/w 24
B_NODE = 32           # bbox 24 + links/range/axis 8
B_INSTANCE = 64       # inverse 3x4 (48) + kind/ids (16)
B_INSTANCE_HIT = 48   # forward 3x4 on an accepted hit
B_TRIANGLE = 48       # three positions (+ padding lanes that carry the id)
B_TRI_SHADING = 60    # 3 normals 36 + 3 uvs 24
B_SPHERE, B_CUBOID, B_QUAD, B_DISK = 16, 24, 36, 36
B_SHADE = 320         # path state r/w 72 + material 64 + light 64 + new ray 32 + 2 shadow items 88
B_SAMPLE = 36         # accumulate: read L 12 + r/w pixel sum 24


def extend_bytes(s):
    return (B_CLOSEST_RAY * s["closest_rays"] + B_NODE * (s["tlas_nodes"] + s["blas_nodes"]) + B_INSTANCE * s["instances"] +
            B_INSTANCE_HIT * s["instance_hits"] + B_TRIANGLE * s["triangles"] + B_TRI_SHADING * s["tri_shading"] +
            B_SPHERE * s["spheres"] + B_CUBOID * s["cuboids"] + B_QUAD * s["quads"] + B_DISK * s["disks"])


def shadow_bytes(s):
    return (B_SHADOW_RAY * s["shadow_rays"] + B_NODE * (s["shadow_tlas_nodes"] + s["shadow_blas_nodes"]) +
            B_INSTANCE * s["shadow_instances"] + B_TRIANGLE * s["shadow_triangles"] + B_SPHERE * s["shadow_prims"])


def shade_bytes(s):
    return B_SHADE * s["shade_events"]


def accumulate_bytes(s):
    return B_SAMPLE * s["samples"]


STAGES = {
    "extend": ("k_extend", extend_bytes, "ms_extend", "launches_extend"),
    "shadow": ("k_shadow", shadow_bytes, "ms_shadow", "launches_shadow"),
    "shade": ("k_shade", shade_bytes, "ms_shade", "launches_shade"),
}


def stage_report(counters, times):
    """counters: stats dict of an instrumented frame; times: stats dict of a timed frame (same frame)."""
    out = {}
    for stage, (kernel, fn, ms_key, launch_key) in STAGES.items():
        ms = float(times[ms_key])
        nbytes = float(fn(counters))
        launches = max(int(times[launch_key]), 1)
        out[stage] = {
            "kernel": kernel,
            "bytes_per_launch": nbytes / launches,
            "ms_per_launch": ms / launches,
            "launches": launches,
            "achieved_GBps": (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else 0.0,
        }
    return out


def dominant(report):
    return max(report.items(), key=lambda kv: kv[1]["ms_per_launch"] * kv[1]["launches"])
