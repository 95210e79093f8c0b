"""Byte model of the wavefront stages (SURVEY.md §8(d) byte table, DESIGN.md §5 Measurement).

Every stage's algorithmic bytes are split in two:

* queue_state bytes — ray / hit / shadow-ray records, path state and queue entries.  They are written by one kernel
  and read by the next, tens of GB apart: they MUST cross HBM.  Per unit: 52 B per closest-hit ray,
  68 B per shadow ray, 320 B per shade event, 36 B per accumulated sample.
* scene bytes — BVH nodes, instances, triangles, analytic shapes: 32 B per node visit, 64 (+48) B per instance, 48 B per
  triangle test, ...  The scene is read-only and cache-resident (C2 / C3: a few KB, in every L2; C4: 137 MB, in the
  Infinity Cache with its upper levels in L2), so these bytes are a WORK RATE, not HBM traffic: priced at record size
  they exceed the HBM peak on a cache-resident scene (round 1's 1.03).

The HBM roofline of a stage is therefore
    achieved = (queue_state bytes + scene bytes that missed the caches) / kernel time
where the scene misses come from the measured traffic of that kernel (rocprofv3 TCC counters, separate passes,
profiles/latest_traffic_<config>.json): misses = clamp(traffic - queue_state bytes, 0, scene bytes), and 0 for a scene
that fits one XCD's L2.  Without a traffic file the misses count as 0 and the line says so.  achieved <= traffic-derived
bytes <= what HBM can move, so frac <= 1 by construction; bench.py asserts it.

Counts come from the instrumented kernel variant (pbrs_render_params.collect_counters): traversal is deterministic, so
they equal the timed work (and the oracle's counts, tests/test_gpu_render.py).  The instrumented k_extend evaluates
the full feature set; what the lean variants skip (tri_shading fetches of shading-proved meshes) is scene work only.
"""
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)
L2_BYTES_PER_XCD = 4 << 20
N_SIMD = 256 * 4        # 256 CUs x 4 SIMDs (same guide)
CLOCK_HZ = 2.4e9        # peak engine clock (same guide); the sustained clock is lower, so issue fractions are lower bounds
CYCLES_PER_WAVE_VALU = 4  # a wave64 vector instruction occupies its 16-lane SIMD for 4 cycles

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


def extend_queue_bytes(s):
    return B_CLOSEST_RAY * s["closest_rays"]


def extend_scene_bytes(s):
    return (B_NODE * (s["tlas_nodes"] + s["blas_nodes"]) + B_INSTANCE * s["instances"] + B_INSTANCE_HIT * s["instance_hits"] +
            B_TRIANGLE * s["triangles"] + B_TRI_SHADING * s["tri_shading"] +
            B_SPHERE * s["spheres"] + B_CUBOID * s["cuboids"] + B_QUAD * s["quads"] + B_DISK * s["disks"])


def shadow_queue_bytes(s):
    return B_SHADOW_RAY * s["shadow_rays"]


def shadow_scene_bytes(s):
    return (B_NODE * (s["shadow_tlas_nodes"] + s["shadow_blas_nodes"]) + B_INSTANCE * s["shadow_instances"] +
            B_TRIANGLE * s["shadow_triangles"] + B_SPHERE * s["shadow_prims"])


def shade_queue_bytes(s):
    return B_SHADE * s["shade_events"]


def zero(_s):
    return 0


def accumulate_bytes(s):
    return B_SAMPLE * s["samples"]


# stage: (kernel, queue/state bytes, scene bytes, time key, launch-count key)
STAGES = {
    "extend": ("k_extend", extend_queue_bytes, extend_scene_bytes, "ms_extend", "launches_extend"),
    "shadow": ("k_shadow", shadow_queue_bytes, shadow_scene_bytes, "ms_shadow", "launches_shadow"),
    "shade": ("k_shade", shade_queue_bytes, zero, "ms_shade", "launches_shade"),
}


def kernel_traffic(traffic_doc, kernel, stage_launches_per_frame=None):
    """Measured HBM bytes per stage launch of the uninstrumented instantiation(s) of `kernel` in a traffic document
    (tools/traffic_from_pmc.py), e.g. "k_extend<false, 4u>", "k_shade<0u, false, 5u>"; None when absent.  A stage launch may be
    several kernel launches (k_shade's variants over their class ranges): the bytes of all of them, over the frames the
    document covers, divided by the stage's launches in those frames."""
    if not traffic_doc:
        return None
    rows = [v for k, v in traffic_doc.get("kernels", {}).items()
            if k == kernel or (k.startswith(kernel + "<") and not k.startswith(kernel + "<true"))]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None
    total = sum(v["hbm_total"] * v["launches"] for v in rows)
    frames = traffic_doc.get("geometry", {}).get("frames")
    if stage_launches_per_frame and frames:
        return total / (frames * stage_launches_per_frame)
    return total / n


def kernel_valu(traffic_doc, kernel, stage_launches_per_frame=None):
    """(wave-level vector instructions per stage launch, mean active lanes) of the uninstrumented instantiation(s) of
    `kernel` in a traffic document that carries the SQ pass (tools/traffic_from_pmc.py), or None."""
    if not traffic_doc:
        return None
    rows = [v for k, v in traffic_doc.get("kernels", {}).items()
            if (k == kernel or (k.startswith(kernel + "<") and not k.startswith(kernel + "<true"))) and "valu_insts" in v]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None
    total = sum(v["valu_insts"] * v["launches"] for v in rows)
    lanes = sum(v["valu_insts"] * v["launches"] * v.get("valu_lanes_active", 0.0) for v in rows) / total if total else 0.0
    frames = traffic_doc.get("geometry", {}).get("frames")
    per_launch = total / (frames * stage_launches_per_frame) if stage_launches_per_frame and frames else total / n
    return per_launch, lanes


def stage_report(counters, times, scene_nbytes=0, traffic_doc=None):
    """counters: stats dict of an instrumented frame; times: per-frame stage times and launch counts of the timed frames;
    scene_nbytes: size of the resident scene; traffic_doc: parsed profiles/latest_traffic_<config>.json or None."""
    out = {}
    for stage, (kernel, qfn, sfn, ms_key, launch_key) in STAGES.items():
        ms = float(times[ms_key])
        launches = max(int(times[launch_key]), 1)
        q = float(qfn(counters)) / launches
        sc = float(sfn(counters)) / launches
        traffic = kernel_traffic(traffic_doc, kernel, launches)
        if scene_nbytes <= L2_BYTES_PER_XCD or sc == 0.0:
            miss, miss_src = 0.0, "scene resident in every XCD's L2" if sc else "no scene reads"
        elif traffic is None:
            miss, miss_src = 0.0, "unmeasured (no traffic file for this workload): counted as 0"
        else:
            miss, miss_src = min(max(traffic - q, 0.0), sc), "measured traffic - queue/state bytes, capped at the scene bytes"
        sec = ms / launches * 1e-3
        out[stage] = {
            "kernel": kernel,
            "launches": launches,
            "ms_per_launch": ms / launches,
            "queue_state_bytes_per_launch": q,
            "scene_bytes_per_launch": sc,
            "scene_miss_bytes_per_launch": miss,
            "scene_miss_source": miss_src,
            "traffic_bytes_per_launch": traffic,
            "achieved_GBps": ((q + miss) / sec / 1e9) if sec > 0 else 0.0,
            "cache_work_rate_GBps": ((q + sc) / sec / 1e9) if sec > 0 else 0.0,  # not an HBM figure: may exceed the HBM peak
        }
        out[stage]["frac"] = out[stage]["achieved_GBps"] / HBM_PEAK_GBS
        valu = kernel_valu(traffic_doc, kernel, launches)
        if valu and sec > 0:
            # share of the chip's vector issue slots the kernel fills (instruction counts measured offline, time live): what
            # binds the kernels that HBM does not — a lower bound, the sustained clock being below CLOCK_HZ
            out[stage]["valu_insts_per_launch"] = valu[0]
            out[stage]["valu_lanes_active"] = valu[1]
            out[stage]["valu_issue_frac"] = valu[0] * CYCLES_PER_WAVE_VALU / (N_SIMD * sec * CLOCK_HZ)
    return out


def dominant(report):
    return max(report.items(), key=lambda kv: kv[1]["ms_per_launch"] * kv[1]["launches"])


def traversal(report):
    """extend + shadow together (the north star's "during BVH traversal")."""
    e, s = report["extend"], report["shadow"]
    sec = (e["ms_per_launch"] * e["launches"] + s["ms_per_launch"] * s["launches"]) * 1e-3
    hbm = sum((r["queue_state_bytes_per_launch"] + r["scene_miss_bytes_per_launch"]) * r["launches"] for r in (e, s))
    work = sum((r["queue_state_bytes_per_launch"] + r["scene_bytes_per_launch"]) * r["launches"] for r in (e, s))
    ach = hbm / sec / 1e9 if sec > 0 else 0.0
    return {"kernels": "k_extend + k_shadow", "achieved": ach, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ach / HBM_PEAK_GBS,
            "cache_work_rate_GBps": work / sec / 1e9 if sec > 0 else 0.0,
            "note": "achieved = queue/state bytes + measured scene misses; cache_work_rate prices every node / triangle visit at record size "
                    "(SURVEY.md §8d) and is served by L2 / Infinity Cache, so it is not comparable with the HBM peak"}
