"""Byte model of the wavefront stages (SURVEY.md §8(d) byte table, DESIGN.md §5 Measurement).

Every stage's algorithmic bytes are split in two:

* queue_state bytes — ray / hit / shadow-ray records, path state and queue entries.  They are written by one kernel
  and read by the next, tens of GB apart: they MUST cross HBM.  The per-unit figures are those of the record layout the
  kernels actually move (device/kernels.h, PathState; PBRS_STATE_BYTES_PER_PATH is the resident footprint of one path):
    closest-hit ray  48 B   32 B ray read (origin + slot, direction + RNG word: two 16-byte vectors) + 16 B hit record written
                            (+ 1 B class byte in scenes with several shading classes)
    shadow ray       44 B   32 B ray read + the lone ray's outcome: 16 B read, 12 B written when unoccluded (an occlusion byte
                            otherwise); priced at the unoccluded case of the common lone ray
    shade event     192 B   64 B read (three path vectors + hit) + 32 B radiance read-modify-write + 48 B next path record +
                            48 B for the one shadow ray most vertices cast (two vectors + its outcome vector)
    accumulated sample 28 B 16 B radiance vector read + 12 B of the pixel sum, amortised over the pass
  SURVEY.md §8(d) priced the same events at 52 / 68 / 320 / 36 B for the column layout it sketched (round 1's); the dense
  16-byte records of round 2 move fewer bytes, and the model follows the records so that `achieved` is not flattered.
* scene bytes — BVH nodes, instances, triangles, analytic shapes: 32 B per node visit, 64 (+48) B per instance, 48 B per
  triangle test, ...  The scene is read-only and cache-resident (C2 / C3: a few KB, in every L2; C4: 137 MB, in the
  Infinity Cache with its upper levels in L2), so these bytes are a WORK RATE, not HBM traffic: priced at record size
  they exceed the HBM peak on a cache-resident scene (round 1's 1.03).

The HBM roofline of a stage is
    achieved = (queue_state bytes + scene bytes that missed the caches) / kernel time
where the scene misses come from the measured traffic of that kernel (rocprofv3 TCC counters, separate passes,
profiles/latest_traffic_<config>.json): misses = clamp(traffic - queue_state bytes, 0, scene bytes), and 0 for a scene
that fits one XCD's L2.  Without a traffic file the misses count as 0 and the line says so.  The queue/state bytes are a
MODEL (units x record sizes), not a measurement: `frac_measured` = measured traffic / time / peak is reported beside the
model fraction wherever a traffic file applies, and a model fraction above 1 is flagged (`roofline_inconsistent`), not
asserted away.  The traffic counters (FETCH_SIZE / WRITE_SIZE / TCC_EA0_*) count L2 -> fabric requests INCLUDING
Infinity-Cache hits (guide §HBM): for a scene that fits the 256 MiB Infinity Cache they are an upper bound of true HBM
traffic, hence the `l2_fabric_*` names.

Counts come from the instrumented kernel variant (pbrs_render_params.collect_counters): traversal is deterministic, so
they equal the timed work (and the oracle's counts, tests/test_gpu_render.py).  The instrumented k_extend evaluates
the full feature set; what the lean variants skip (tri_shading fetches of shading-proved meshes) is scene work only.
"""
import hashlib
import os

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters)
L2_BYTES_PER_XCD = 4 << 20
N_SIMD = 256 * 4        # 256 CUs x 4 SIMDs (same guide)
CLOCK_HZ = 2.4e9        # peak engine clock (same guide); the sustained clock is lower, so issue fractions are lower bounds
# Issue cost of a wave64 vector instruction (tools/microbench/issue_rates.hip, eight waves per SIMD, cycles at the clock
# attribute): 2.7 for the 32-bit-encoded f32 / integer forms (v_mul_f32, v_sub_f32, v_min_f32 ...), 3.7-4.4 for the 64-bit
# encodings (v_fma_f32, v_max3_f32, v_pk_*, DPP), every f64 operation and the f32 <-> f64 conversions, 8.6 for the
# transcendentals.  The fraction below prices every instruction at 4: an UPPER bound of the share of issue slots a kernel fills
# (rounds 1 and 2 took it for the exact figure); with the traversal kernels' mix it is about 0.8 of that.
CYCLES_PER_WAVE_VALU = 4
# ... what the 32-bit-encoded f32 / integer instructions — most of a traversal step — were measured to take (tools/microbench/issue_rates.hip)
CYCLES_PER_WAVE_VALU_MIN = 2.7
# The calibrated price (round 4): a kernel's dynamic instruction count (SQ_INSTS_VALU) at the mix of its STATIC code — the share of its
# vector instructions in a 32-bit encoding (tools/isa_stats.py --json: `_e32` forms without a literal) at 2.7 cycles, the rest (64-bit
# encodings, DPP, literals, f64, conversions: 3.8-4.4 measured) at 4.0.  The static mix stands in for the dynamic one (the hot loops are
# most of both); the result is the `valu_issue_frac` the bound is chosen with, the all-4 and all-2.7 figures stay beside it as brackets.
CYCLES_PER_WAVE_VALU_64BIT = 4.0
# scalar instructions of a wave issue one at a time, beside the vector work of OTHER waves of the SIMD: 4.2-4.3 cycles each (same microbenchmark)
CYCLES_PER_WAVE_SALU = 4.25
N_CU = 256
L1_ACCESSES_PER_CU_CYCLE = 1.0  # a CU's L1 takes one access (one lane of a load whose lanes name different lines) per cycle,
                                # whatever the load's width: tools/microbench/gather_rates.hip, 64.6 cycles per 64-lane load

B_CLOSEST_RAY = 48    # two 16-byte ray vectors read + the 16-byte hit record written (PathState::q[.][0..1], ::hit)
B_SHADOW_RAY = 44     # two 16-byte ray vectors read (::sr[0..1]) + a lone ray's outcome: 16 B read (::sr[2]), 12 B into L
B_NODE = 32           # bbox 24 + links/range/axis 8
B_INSTANCE = 64       # inverse 3x4 (48) + kind/ids (16)
B_INSTANCE_HIT = 48   # forward 3x4 on an accepted hit
B_TRIANGLE = 48       # three positions (+ padding lanes that carry the id)
B_TRI_SHADING = 60    # 3 normals 36 + 3 uvs 24
B_SPHERE, B_CUBOID, B_QUAD, B_DISK = 16, 24, 36, 36
B_SHADE = 192         # q[.][0..2] + hit read 64, L read-modify-write 32, next path record 48, one shadow ray (sr[0..2]) 48
B_SAMPLE = 28         # accumulate: L vector read 16 + the pixel sum's 12 (read and written once per pass)
SURVEY_8D = {"closest_ray": 52, "shadow_ray": 68, "shade_event": 320, "sample": 36}  # the column layout SURVEY.md §8(d) priced


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
    total = sum(v.get("l2_fabric_total", v.get("hbm_total", 0.0)) * v["launches"] for v in rows)
    frames = traffic_doc.get("geometry", {}).get("frames")
    if stage_launches_per_frame and frames:
        return total / (frames * stage_launches_per_frame)
    return total / n


def kernel_field(traffic_doc, kernel, field, stage_launches_per_frame=None, per_launch=True):
    """A per-launch counter (`per_launch`: summed over the launches of a stage launch, as kernel_traffic) or a share (mean
    over the kernel's instantiations weighted by launches) of the uninstrumented instantiation(s) of `kernel`; None when absent."""
    if not traffic_doc:
        return None
    rows = [v for k, v in traffic_doc.get("kernels", {}).items()
            if (k == kernel or (k.startswith(kernel + "<") and not k.startswith(kernel + "<true"))) and field in v]
    n = sum(v["launches"] for v in rows)
    if not n:
        return None
    total = sum(v[field] * v["launches"] for v in rows)
    if not per_launch:
        return total / n
    frames = traffic_doc.get("geometry", {}).get("frames")
    return total / (frames * stage_launches_per_frame) if stage_launches_per_frame and frames else total / n


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


def static_mix(isa_doc, traffic_doc, kernel):
    """Share of `kernel`'s vector instructions in a 32-bit encoding: static counts (tools/isa_stats.py --json) of the instantiations the
    traffic document saw running, weighted by their dynamic instruction counts; None without an ISA document for the same sources."""
    if not isa_doc or not traffic_doc or isa_doc.get("source_hash") != traffic_doc.get("source_hash"):
        return None
    num = den = 0.0
    for k, v in traffic_doc.get("kernels", {}).items():
        if not (k == kernel or (k.startswith(kernel + "<") and not k.startswith(kernel + "<true"))) or "valu_insts" not in v:
            continue
        st = isa_doc.get("kernels", {}).get(k)
        if not st or not st.get("valu"):
            continue
        w = v["valu_insts"] * v["launches"]
        num += w * st["valu32"] / st["valu"]
        den += w
    return num / den if den else None


def stage_report(counters, times, scene_nbytes=0, traffic_doc=None, isa_doc=None, serial_times=None):
    """counters: stats dict of an instrumented frame; times: per-frame stage times and launch counts of the timed frames;
    scene_nbytes: size of the resident scene; traffic_doc: parsed profiles/latest_traffic_<config>.json or None; isa_doc: parsed
    profiles/latest_isa_mix.json (static encoding mix per kernel) or None; serial_times: stage times of one frame rendered with the
    pass overlap off (exclusive times: in the timed frames a stage's event brackets include the time its kernels share the chip with
    the other stream's) — reported beside the live figures as *_serial, and what `dominant` ranks the stages by."""
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
        # The issue / L1 / measured-traffic shares below divide offline per-launch counters by a per-launch time: the exclusive time of a
        # frame rendered with the pass overlap off where the caller has one (in the timed frames a late-bounce launch shares the chip with
        # the next pass's kernels, and its bracket is the longer for it), else the live one.  `frac` stays on the live time.
        xsec = (float(serial_times[ms_key]) / launches * 1e-3) if (serial_times and serial_times.get(ms_key)) else sec
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
        if serial_times and serial_times.get(ms_key):
            ssec = float(serial_times[ms_key]) / launches * 1e-3
            out[stage]["ms_per_launch_serial"] = float(serial_times[ms_key]) / launches
            out[stage]["frac_serial"] = (q + miss) / ssec / 1e9 / HBM_PEAK_GBS
        # what the counters saw (L2 -> fabric requests, Infinity-Cache hits included), beside the model
        out[stage]["frac_measured"] = (traffic / xsec / 1e9 / HBM_PEAK_GBS) if (traffic is not None and xsec > 0) else None
        valu = kernel_valu(traffic_doc, kernel, launches)
        sec = xsec
        if valu and sec > 0:
            # share of the chip's vector issue slots the kernel fills (instruction counts measured offline, time live): what
            # binds the kernels that HBM does not — a lower bound, the sustained clock being below CLOCK_HZ
            out[stage]["valu_insts_per_launch"] = valu[0]
            out[stage]["valu_lanes_active"] = valu[1]
            out[stage]["valu_issue_upper_price"] = valu[0] * CYCLES_PER_WAVE_VALU / (N_SIMD * sec * CLOCK_HZ)  # every instruction at 4 cycles: may pass 1
            # the same count at the cheapest measured issue cost: the true share lies between the two
            out[stage]["valu_issue_frac_min"] = valu[0] * CYCLES_PER_WAVE_VALU_MIN / (N_SIMD * sec * CLOCK_HZ)
            # ... and at the kernel's static encoding mix: the calibrated figure, the one the bound is chosen with
            mix = static_mix(isa_doc, traffic_doc, kernel)
            out[stage]["valu_32bit_encoding_share"] = mix
            # (no ISA document for these sources: the midpoint of the two prices, and the line says the mix is unknown)
            cyc = (mix * CYCLES_PER_WAVE_VALU_MIN + (1.0 - mix) * CYCLES_PER_WAVE_VALU_64BIT) if mix is not None else 0.5 * (CYCLES_PER_WAVE_VALU_MIN + CYCLES_PER_WAVE_VALU_64BIT)
            out[stage]["valu_issue_frac"] = min(valu[0] * cyc / (N_SIMD * sec * CLOCK_HZ), 1.0)
        salu = kernel_field(traffic_doc, kernel, "salu_insts", launches)
        if salu is not None and sec > 0:
            out[stage]["salu_insts_per_launch"] = salu
            out[stage]["salu_issue_frac"] = min(salu * CYCLES_PER_WAVE_SALU / (N_SIMD * sec * CLOCK_HZ), 1.0)
        l1 = kernel_field(traffic_doc, kernel, "l1_accesses", launches)
        if l1 is not None and sec > 0:
            # share of the L1s' access slots: a load whose 64 lanes name 64 lines is 64 accesses whatever its width, so a walk's
            # node, triangle and record fetches count per lane — the other resource the traversal kernels fill
            out[stage]["l1_accesses_per_launch"] = l1
            out[stage]["l1_access_frac"] = l1 / (N_CU * L1_ACCESSES_PER_CU_CYCLE * sec * CLOCK_HZ)
        for share in ("ta_busy_share", "td_busy_share"):
            v = kernel_field(traffic_doc, kernel, share, per_launch=False)
            if v is not None:
                out[stage][share] = v
    return out


def dominant(report):
    """The stage with the most time per frame: by exclusive (serial) time where the report has it, else by the live brackets."""
    return max(report.items(), key=lambda kv: kv[1].get("ms_per_launch_serial", kv[1]["ms_per_launch"]) * kv[1]["launches"])


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


def bound_shares(stage_row):
    """The shares (each <= 1) of the resources a kernel can be bound by: HBM (measured traffic where there is any, else the model),
    the vector issue slots at the kernel's static encoding mix, the scalar issue slots, the L1s' access slots and the busy share of
    the texture data path."""
    hbm = stage_row.get("frac_measured")
    return {"hbm": min(hbm if hbm is not None else stage_row["frac"], 1.0), "valu_issue": stage_row.get("valu_issue_frac") or 0.0,
            "salu_issue": stage_row.get("salu_issue_frac") or 0.0, "l1_access": min(stage_row.get("l1_access_frac") or 0.0, 1.0),
            "td_busy": min(stage_row.get("td_busy_share") or 0.0, 1.0)}


def bound_of(stage_row):
    """The resource the kernel fills the largest share of (bound_shares); "hbm" when no counters apply."""
    return max(bound_shares(stage_row).items(), key=lambda kv: kv[1])[0]


def counters_apply(traffic_doc, current=None):
    """Whether a counter file speaks about the kernels of the sources in the tree: measured on these very sources
    (`source_hash`), or on sources whose kernels NAMED IN THE FILE compile to the same instructions and resource
    directives — an entry of `same_isa_as_measured` (written on the builder side from tools/isa_same_kernels.py's
    comparison of the two device assemblies, its log under profiles/) that names the tree's hash and lists the kernels
    that do differ.  Returns (ok, how)."""
    current = current or source_hash()
    if traffic_doc.get("source_hash") == current:
        return True, "measured on these sources"
    for e in traffic_doc.get("same_isa_as_measured", []):
        if e.get("source_hash") == current and not (set(e.get("differing_kernels", [])) & set(traffic_doc.get("kernels", {}))):
            return True, (f"measured on sources {traffic_doc.get('source_hash')}; every kernel of the file compiles to the same "
                          f"instructions from {current} ({e.get('log')})")
    return False, None


def source_hash(root=None):
    """sha256 over the sources the kernels are built from (pbrs_amd/csrc/**, include/*.h): stamps a traffic file
    (tools/profile_frame.py -> tools/traffic_from_pmc.py) so that bench.py can tell whether the offline counters it
    quotes were measured on the kernels it is timing."""
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = []
    for base in (os.path.join(root, "pbrs_amd", "csrc"), os.path.join(root, "include")):
        for d, _, names in os.walk(base):
            files += [os.path.join(d, n) for n in names if n.endswith((".h", ".hip", ".cpp")) or n == "Makefile"]
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(os.path.relpath(f, root).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
