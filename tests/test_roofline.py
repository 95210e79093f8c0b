"""pbrs_amd/roofline.py: the split of a stage's bytes into queue/state bytes (must cross HBM) and scene bytes (cache work),
and the scene-miss estimate from measured traffic.  Pure host logic."""
from pbrs_amd import roofline

COUNTS = {"closest_rays": 1000, "tlas_nodes": 4000, "blas_nodes": 30000, "instances": 2000, "instance_hits": 900, "triangles": 8000,
          "tri_shading": 700, "spheres": 10, "cuboids": 0, "quads": 0, "disks": 0, "shadow_rays": 600, "shadow_tlas_nodes": 2000,
          "shadow_blas_nodes": 9000, "shadow_instances": 800, "shadow_triangles": 3000, "shadow_prims": 5, "shade_events": 900, "samples": 500}
TIMES = {"ms_extend": 2.0, "ms_shadow": 1.0, "ms_shade": 1.5, "launches_extend": 2, "launches_shadow": 2, "launches_shade": 2}


def test_queue_and_scene_bytes_follow_the_survey_table():
    rep = roofline.stage_report(COUNTS, TIMES, scene_nbytes=1 << 30)
    assert rep["extend"]["queue_state_bytes_per_launch"] == 52 * 1000 / 2
    assert rep["shadow"]["queue_state_bytes_per_launch"] == 68 * 600 / 2
    assert rep["shade"]["queue_state_bytes_per_launch"] == 320 * 900 / 2 and rep["shade"]["scene_bytes_per_launch"] == 0
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
    assert e["valu_issue_frac"] == 1.2e6 * 4 / (1024 * 1.0e-3 * 2.4e9)  # 2.0 ms over 2 launches
    assert "valu_issue_frac" not in rep["shade"] and "valu_issue_frac" not in rep["shadow"]  # no SQ figures: not reported
