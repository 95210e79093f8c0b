"""tools/traffic_from_pmc.py: per-kernel counter sums of the rocprofv3 passes -> bytes, instructions and L1 accesses per launch.
Pure host logic on synthetic summaries in the format tools/pmc_summary.py writes."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summary(path, rows):
    with open(path, "w") as f:
        for kernel, dispatches, counters in rows:
            f.write(f"{kernel} dispatches {dispatches}\n")
            for k, v in counters.items():
                f.write(f"   {k:<28s} {v:g}\n")


def test_traffic_instructions_and_l1_accesses_per_launch(tmp_path):
    geo = {"config": "c4", "pixels": 1000, "spp": 16, "passes": 1, "samples_per_pass": 16, "frames": 1, "source_hash": "abc", "git_head": "def"}
    (tmp_path / "geo.json").write_text(json.dumps(geo))
    k = "void k_extend<false, 13u>(DevScene, PathState)"
    _summary(tmp_path / "fetch.txt", [(k, 4, {"FETCH_SIZE": 4096.0})])                       # KiB, halved on gfx950
    _summary(tmp_path / "write.txt", [(k, 4, {"WRITE_SIZE": 1024.0})])
    _summary(tmp_path / "sizes.txt", [(k, 4, {"TCC_EA0_RDREQ_sum": 400.0, "TCC_EA0_RDREQ_32B_sum": 0.0, "TCC_EA0_RDREQ_64B_sum": 0.0, "TCC_EA0_RDREQ_128B_sum": 400.0})])
    _summary(tmp_path / "wr.txt", [(k, 4, {"TCC_EA0_WRREQ_sum": 40.0, "TCC_EA0_WRREQ_64B_sum": 40.0, "TCC_HIT_sum": 900.0, "TCC_MISS_sum": 100.0})])
    _summary(tmp_path / "sq.txt", [(k, 4, {"SQ_INSTS_VALU": 8.0e6, "SQ_THREAD_CYCLES_VALU": 3.2e8, "SQ_WAVE_CYCLES": 1.0e9, "SQ_WAIT_INST_ANY": 2.0e8})])
    _summary(tmp_path / "tcp.txt", [(k, 4, {"TCP_TOTAL_CACHE_ACCESSES_sum": 4.0e8, "TCP_TCC_READ_REQ_sum": 6.0e7})])
    _summary(tmp_path / "tatd.txt", [(k, 4, {"TA_TA_BUSY_sum": 256.0 * 600.0, "TD_TD_BUSY_sum": 256.0 * 900.0, "GRBM_GUI_ACTIVE": 8.0 * 1000.0})])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "traffic_from_pmc.py")] + [str(tmp_path / n) for n in
                         ("geo.json", "fetch.txt", "write.txt", "sizes.txt", "wr.txt", "sq.txt", "tcp.txt", "tatd.txt")], capture_output=True, text=True, check=True)
    doc = json.loads(out.stdout)
    assert doc["source_hash"] == "abc" and doc["git_head"] == "def"  # what bench.py compares with the sources it runs
    row = doc["kernels"]["k_extend<false, 13u>(DevScene, PathState)"]
    assert row["launches"] == 4
    assert row["l2_fabric_read"] == 4096.0 * 1024 / 4 * 2.0 and row["l2_fabric_write"] == 1024.0 * 1024 / 4
    assert row["l2_fabric_read_by_request_size"] == 128 * 400.0 / 4
    assert row["l2_hit_rate"] == 0.9
    assert row["valu_insts"] == 2.0e6 and row["valu_lanes_active"] == 40.0 and row["wave_wait_share"] == 0.2
    assert row["l1_accesses"] == 1.0e8 and row["l1_miss_requests"] == 1.5e7
    assert abs(row["ta_busy_share"] - 0.6) < 1e-12 and abs(row["td_busy_share"] - 0.9) < 1e-12
