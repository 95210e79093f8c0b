"""pbrs_amd/roofline.py: the split of a stage's bytes into queue/state bytes (must cross HBM) and scene bytes (cache work),
and the scene-miss estimate from measured traffic.  Pure host logic."""
from pbrs_amd import roofline

COUNTS = {"closest_rays": 1000, "tlas_nodes": 4000, "blas_nodes": 30000, "instances": 2000, "instance_hits": 900, "triangles": 8000,
          "tri_shading": 700, "spheres": 10, "cuboids": 0, "quads": 0, "disks": 0, "shadow_rays": 600, "shadow_tlas_nodes": 2000,
          "shadow_blas_nodes": 9000, "shadow_instances": 800, "shadow_triangles": 3000, "shadow_prims": 5, "shade_events": 900, "samples": 500}
TIMES = {"ms_extend": 2.0, "ms_shadow": 1.0, "ms_shade": 1.5, "launches_extend": 2, "launches_shadow": 2, "launches_shade": 2}


def test_queue_bytes_follow_the_record_layout_and_scene_bytes_the_survey_table():
    """Queue/state bytes: the 16-byte records the kernels move (device/kernels.h PathState) — 48 B per closest ray (two ray vectors
    read, the hit record written), 44 B per shadow ray, 192 B per shade event — not the column layout SURVEY.md §8(d) sketched
    (52 / 68 / 320 B), which would flatter `achieved`.  Scene bytes: §8(d)'s per-visit figures."""
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30)
    assert rep["extend"]["queue_state_bytes_per_launch"] == 48 * 1000 / 2
    assert rep["shadow"]["queue_state_bytes_per_launch"] == 44 * 600 / 2
    assert rep["shade"]["queue_state_bytes_per_launch"] == 192 * 900 / 2 and rep["shade"]["scene_bytes_per_launch"] == 0
    assert roofline.SURVEY_8D == {"closest_ray": 52, "shadow_ray": 68, "shade_event": 320, "sample": 36}
    assert rep["extend"]["frac_measured"] is None  # no traffic file: nothing measured to put beside the model
    assert rep["extend"]["scene_bytes_per_launch"] == (32 * 34000 + 64 * 2000 + 48 * 900 + 48 * 8000 + 60 * 700 + 16 * 10) / 2
    # no traffic file: misses count as zero and the report says so
    assert rep["extend"]["scene_miss_bytes_per_launch"] == 0 and "unmeasured" in rep["extend"]["scene_miss_source"]
    assert rep["extend"]["achieved_GBps"] < rep["extend"]["cache_work_rate_GBps"]


def test_scene_misses_come_from_measured_traffic_and_are_capped():
    doc = {"kernels": {"k_extend<false, 5u>": {"launches": 4, "hbm_total": 500_000.0}, "k_extend<true, 7u>": {"launches": 1, "hbm_total": 9e9},
                       "k_shadow<false, 5u>": {"launches": 4, "hbm_total": 9e12}, "k_shade<0u, false>": {"launches": 4, "hbm_total": 1.0}}}
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc)
    e, s = rep["extend"], rep["shadow"]
    assert e["traffic_bytes_per_launch"] == 500_000.0  # the instrumented variant is not the timed kernel
    assert e["scene_miss_bytes_per_launch"] == 500_000.0 - e["queue_state_bytes_per_launch"]
    assert s["scene_miss_bytes_per_launch"] == s["scene_bytes_per_launch"]  # traffic beyond the scene bytes is not scene misses
    assert rep["shade"]["scene_miss_bytes_per_launch"] == 0
    # a scene that fits one XCD's L2 cannot miss
    small = roofline.stage_report(COUNTS, TIMES, scene_nbytes=10_000, traffic_doc=doc)
    assert small["extend"]["scene_miss_bytes_per_launch"] == 0 and small["shadow"]["scene_miss_bytes_per_launch"] == 0


def test_a_stage_launch_made_of_several_kernel_launches():
    """k_shade's variants over their class ranges: two kernel launches per bounce, one stage launch."""
    doc = {"geometry": {"frames": 2}, "kernels": {"k_shade<0u, false, 5u>": {"launches": 4, "hbm_total": 300.0},
                                                    "k_shade<0u, false, 0u>": {"launches": 4, "hbm_total": 100.0}}}
    assert roofline.kernel_traffic(doc, "k_shade", stage_launches_per_frame=2) == (4 * 300.0 + 4 * 100.0) / (2 * 2)
    assert roofline.kernel_traffic(doc, "k_shade") == 200.0  # without the stage's launch count: the plain average


def test_fractions_are_fractions_of_the_hbm_peak():
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30)
    for r in rep.values():
        assert r["frac"] == r["achieved_GBps"] / roofline.HBM_PEAK_GBS
    t = roofline.traversal(rep)
    assert t["frac"] == t["achieved"] / roofline.HBM_PEAK_GBS and t["cache_work_rate_GBps"] >= t["achieved"]
    assert roofline.dominant(rep)[0] == "extend"


def test_vector_issue_share_from_the_sq_pass():
    """A kernel HBM does not bind: wave-level VALU instructions per launch (offline SQ pass) x 4 cycles over the chip's SIMD
    cycles in the live kernel time."""
    doc = {"geometry": {"frames": 1}, "kernels": {"k_extend<false, 13u>": {"launches": 2, "hbm_total": 1.0, "valu_insts": 1.2e6, "valu_lanes_active": 40.0},
                                                    "k_extend<true, 15u>": {"launches": 1, "hbm_total": 1.0, "valu_insts": 9e9, "valu_lanes_active": 1.0},
                                                    "k_shade<0u, false, 3u>": {"launches": 2, "hbm_total": 1.0}}}
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc)
    e = rep["extend"]
    assert e["valu_insts_per_launch"] == 1.2e6 and e["valu_lanes_active"] == 40.0
    assert e["valu_issue_upper_price"] == 1.2e6 * 4 / (1024 * 1.0e-3 * 2.4e9)  # 2.0 ms over 2 launches
    assert abs(e["valu_issue_frac_min"] - e["valu_issue_upper_price"] * 2.7 / 4) < 1e-12  # ... at the cheapest measured issue cost
    # no static mix for these sources: the midpoint of the two prices, and the share of 32-bit encodings is reported as unknown
    assert e["valu_32bit_encoding_share"] is None and abs(e["valu_issue_frac"] - 1.2e6 * 3.35 / (1024 * 1.0e-3 * 2.4e9)) < 1e-12
    assert "valu_issue_frac" not in rep["shade"] and "valu_issue_frac" not in rep["shadow"]  # no SQ figures: not reported


def test_calibrated_issue_share_from_the_static_encoding_mix_and_the_scalar_share():
    """Round 4: the dynamic instruction count at the kernel's static mix of encodings (tools/isa_stats.py --json), capped at 1, is the
    figure the bound is chosen with; scalar instructions are a share of their own; every share of bound_shares is <= 1."""
    doc = {"source_hash": "abc", "geometry": {"frames": 1},
           "kernels": {"k_extend<false, 13u>": {"launches": 2, "l2_fabric_total": 4.0e6, "valu_insts": 6.0e8, "valu_lanes_active": 37.0, "salu_insts": 2.0e8,
                                                "l1_accesses": 5.0e8, "td_busy_share": 0.96}}}
    isa = {"source_hash": "abc", "kernels": {"k_extend<false, 13u>": {"valu": 1000, "valu32": 600, "salu": 500}}}
    e = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc, isa_doc=isa)["extend"]
    slots = 1024 * 1.0e-3 * 2.4e9
    assert e["valu_32bit_encoding_share"] == 0.6
    assert abs(e["valu_issue_frac"] - 6.0e8 * (0.6 * 2.7 + 0.4 * 4.0) / slots) < 1e-12 and e["valu_issue_frac"] < 1.0 <= e["valu_issue_upper_price"] * 1.03
    assert abs(e["salu_issue_frac"] - 2.0e8 * 4.25 / slots) < 1e-12
    shares = roofline.bound_shares(e)
    assert set(shares) == {"hbm", "valu_issue", "salu_issue", "l1_access", "td_busy"} and all(0.0 <= v <= 1.0 for v in shares.values())
    assert roofline.bound_of(e) == "td_busy"
    # an ISA document of other sources does not apply
    other = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc, isa_doc=dict(isa, source_hash="xyz"))["extend"]
    assert other["valu_32bit_encoding_share"] is None


def test_measured_fraction_bound_and_the_new_key_names():
    """frac_measured = measured traffic / time / peak beside the model fraction; l2_fabric_* keys (the counters include
    Infinity-Cache hits) are read like the hbm_* keys of older files; the bound is valu_issue where issue exceeds the HBM share."""
    doc = {"geometry": {"frames": 1}, "kernels": {"k_extend<false, 13u>": {"launches": 2, "l2_fabric_total": 4.0e6, "valu_insts": 1.2e6, "valu_lanes_active": 40.0},
                                                    "k_shade<0u, false, 3u>": {"launches": 2, "l2_fabric_total": 8.0e6}}}
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc)
    e = rep["extend"]
    assert e["traffic_bytes_per_launch"] == 4.0e6
    assert e["frac_measured"] == 4.0e6 / 1.0e-3 / 1e9 / roofline.HBM_PEAK_GBS
    assert roofline.bound_of(e) == "valu_issue" and roofline.bound_of(rep["shade"]) == "hbm"


def test_source_hash_follows_the_kernel_sources(tmp_path):
    import os
    (tmp_path / "pbrs_amd" / "csrc" / "device").mkdir(parents=True)
    (tmp_path / "include").mkdir()
    (tmp_path / "pbrs_amd" / "csrc" / "device" / "k.h").write_text("a")
    (tmp_path / "include" / "x.h").write_text("b")
    h0 = roofline.source_hash(str(tmp_path))
    assert h0 == roofline.source_hash(str(tmp_path)) and len(h0) == 16
    (tmp_path / "pbrs_amd" / "csrc" / "device" / "k.h").write_text("a ")
    assert roofline.source_hash(str(tmp_path)) != h0
    assert roofline.source_hash() == roofline.source_hash(os.path.dirname(os.path.dirname(os.path.abspath(roofline.__file__))))


def test_l1_access_share_from_the_tcp_pass():
    """The traversal kernels' other resource: a load whose 64 lanes name 64 lines is 64 L1 accesses whatever its width, and a
    CU's L1 serves one per cycle (tools/microbench/gather_rates.hip).  Accesses per launch come from the offline TCP pass."""
    doc = {"geometry": {"frames": 1}, "kernels": {"k_extend<false, 13u>": {"launches": 2, "l2_fabric_total": 4.0e6, "valu_insts": 1.0e5, "valu_lanes_active": 40.0,
                                                                             "l1_accesses": 5.0e8, "td_busy_share": 0.8, "ta_busy_share": 0.6},
                                                    "k_extend<true, 15u>": {"launches": 1, "l2_fabric_total": 1.0, "l1_accesses": 9e12}}}
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc)
    e = rep["extend"]
    assert e["l1_accesses_per_launch"] == 5.0e8  # the instrumented variant is not the timed kernel
    assert e["l1_access_frac"] == 5.0e8 / (256 * 1.0e-3 * 2.4e9)
    assert e["td_busy_share"] == 0.8 and e["ta_busy_share"] == 0.6
    assert roofline.bound_of(e) == "l1_access"
    assert "l1_access_frac" not in rep["shadow"]


def test_stages_are_ranked_by_exclusive_time_when_a_serial_frame_is_given():
    """With the pass overlap a stage's live event brackets include the time its kernels share the chip with the other stream's: the
    dominant stage is chosen by the exclusive times of a frame rendered with the overlap off, reported beside the live figures."""
    live = dict(TIMES, ms_shade=TIMES["ms_extend"] * 3.0)  # late-stream brackets stretched by the next pass's kernels
    serial = {"ms_extend": TIMES["ms_extend"], "ms_shade": TIMES["ms_extend"] * 0.2, "ms_shadow": TIMES["ms_extend"] * 0.5}
    rep = roofline.stage_report(COUNTS, live, scene_nbytes=1 << 30, serial_times=serial)
    assert roofline.dominant(rep)[0] == "extend"
    assert roofline.dominant(roofline.stage_report(COUNTS, live, scene_nbytes=1 << 30))[0] == "shade"
    e = rep["extend"]
    assert e["ms_per_launch_serial"] == serial["ms_extend"] / e["launches"] and abs(e["frac_serial"] - e["frac"]) < 1e-12
    assert rep["shade"]["frac_serial"] > rep["shade"]["frac"]


def test_offline_counter_shares_divide_by_the_exclusive_time():
    """Counters per launch come from offline passes; the time they are divided by is the exclusive per-launch time of the serial frame
    where there is one (a late-bounce launch's live bracket includes the time it shares the chip with the next pass's kernels)."""
    doc = {"geometry": {"frames": 1}, "kernels": {"k_shade<0u, false, 101u>": {"launches": 2, "l2_fabric_total": 8.0e6, "valu_insts": 2.0e8, "valu_lanes_active": 60.0,
                                                                                   "l1_accesses": 1.0e8, "salu_insts": 1.0e7}}}
    live = dict(TIMES, ms_shade=TIMES["ms_shade"] * 2.5)
    serial = {"ms_extend": TIMES["ms_extend"], "ms_shade": TIMES["ms_shade"], "ms_shadow": TIMES["ms_shadow"]}
    a = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30, traffic_doc=doc)["shade"]
    b = roofline.stage_report(COUNTS, live, scene_nbytes=1 << 30, traffic_doc=doc, serial_times=serial)["shade"]
    for key in ("valu_issue_frac", "salu_issue_frac", "l1_access_frac", "frac_measured"):
        assert abs(a[key] - b[key]) < 1e-12, key
    assert abs(b["frac"] * 2.5 - a["frac"]) < 1e-12 and abs(b["frac_serial"] - a["frac"]) < 1e-12


def test_counter_files_carry_over_only_to_sources_with_the_same_kernels():
    doc = {"source_hash": "abc", "kernels": {"k_extend<false, 13u>": {}, "k_shade<0u, false, 35u>": {}}}
    assert roofline.counters_apply(doc, "abc")[0] and not roofline.counters_apply(doc, "xyz")[0]
    doc["same_isa_as_measured"] = [{"source_hash": "xyz", "differing_kernels": ["k_shade<0u, false, 24u>"], "log": "profiles/x.log"}]
    ok, how = roofline.counters_apply(doc, "xyz")
    assert ok and "abc" in how and "profiles/x.log" in how and not roofline.counters_apply(doc, "other")[0]
    # a file that quotes a kernel whose instructions changed does not carry over
    doc["same_isa_as_measured"][0]["differing_kernels"].append("k_shade<0u, false, 35u>")
    assert not roofline.counters_apply(doc, "xyz")[0]


def test_the_committed_counter_files_apply_to_the_sources_in_the_tree():
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(roofline.__file__)))
    stale = []
    for p in glob.glob(os.path.join(root, "profiles", "latest_traffic_*.json")):
        with open(p) as f:
            doc = json.load(f)
        assert doc.get("source_hash") and isinstance(doc.get("same_isa_as_measured", []), list), p
        if not roofline.counters_apply(doc)[0]:
            stale.append(os.path.basename(p))
    if stale:  # a kernel source was edited since: bench.py leaves the counters out until tools/round_artifacts.sh has run again
        import pytest
        pytest.skip(f"counter files measured on other sources: {stale}")
