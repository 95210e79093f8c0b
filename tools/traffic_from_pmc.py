"""Turns rocprofv3 TCC counter passes (tools/pmc_summary.py outputs) into memory-side (L2 -> fabric) traffic per launch per kernel.

    python tools/traffic_from_pmc.py GEOMETRY.json fetch.txt write.txt [rdreq_sizes.txt [wrreq.txt [sq_wave_cycles.txt [tcp.txt [ta_td.txt [sq_waves.txt]]]]]]

Units and gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: FETCH_SIZE / WRITE_SIZE are in
KiB and come from the L2's memory-side request counters (Infinity-Cache hits are counted, not excluded); on gfx950
FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact.  The
doubling is calibrated on k_accumulate (a pure coalesced stream of known size) and recorded.  Because the guide calls
other access widths uncalibrated, the optional third pass gives an independent figure for every kernel from the
request-size counters of gfx950 (TCC_EA0_RDREQ_{32B,64B,128B}): read bytes = 32 n32 + 64 n64 + 128 n128, with requests
of no recorded size counted at 64 B — reported as l2_fabric_read_by_request_size next to l2_fabric_read; bench.py uses the larger of
the two so that a roofline fraction is never flattered.  The optional sixth file (the SQ pass) adds, per kernel, the vector
instructions of a launch and the lanes active in them: what bench.py turns into the share of the chip's vector issue slots a
kernel uses — one bound of the kernels HBM does not bind.  The optional seventh / eighth files (TCP and TA / TD passes) add the
other one: the L1 accesses of a launch (one per lane and load instruction when the lanes name different lines: a CU's L1 serves
one per cycle, tools/microbench/gather_rates.hip) and the busy cycles of the texture address / data units.
"""
import json
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"^(\S.*) dispatches (\d+)", line)
        if m:
            cur = m.group(1).replace("void ", "")
            out[cur] = {"dispatches": int(m.group(2))}
        elif cur and line.strip():
            k, v = line.split()
            out[cur][k] = float(v)
    return out


geo = json.load(open(sys.argv[1]))
fetch, write = parse(sys.argv[2]), parse(sys.argv[3])
sizes = parse(sys.argv[4]) if len(sys.argv) > 4 else {}
wr = parse(sys.argv[5]) if len(sys.argv) > 5 else {}
sq = parse(sys.argv[6]) if len(sys.argv) > 6 else {}
tcp = parse(sys.argv[7]) if len(sys.argv) > 7 else {}
tatd = parse(sys.argv[8]) if len(sys.argv) > 8 else {}
sq2 = parse(sys.argv[9]) if len(sys.argv) > 9 else {}  # the second SQ pass: scalar, memory and LDS instruction counts
n_pixels = geo["pixels"]
n_slots = n_pixels * geo["samples_per_pass"]
res = {"unit": "bytes per launch", "geometry": geo, "source_hash": geo.get("source_hash"),
       **({"git_head": geo["git_head"]} if geo.get("git_head") else {}),
       "note": "l2_fabric_* = bytes of the L2's memory-side requests (FETCH_SIZE / WRITE_SIZE / TCC_EA0_*): they INCLUDE Infinity-Cache "
               "hits, so for a scene that fits the 256 MiB Infinity Cache they are an upper bound of true HBM traffic", "kernels": {}}
acc = fetch.get("k_accumulate")
if acc:
    known_read = float(geo.get("accumulate_read_bytes_per_slot", 12)) * n_slots + 12.0 * n_pixels
    res["fetch_size_scale_calibrated_on_k_accumulate"] = known_read / (acc["FETCH_SIZE"] * 1024 / acc["dispatches"])
for k in fetch:
    if k.startswith("__") or k not in write:
        continue
    d = fetch[k]["dispatches"]
    rd = fetch[k]["FETCH_SIZE"] * 1024 / d * 2.0   # guide: FETCH_SIZE = half the streamed bytes on gfx950
    wb = write[k]["WRITE_SIZE"] * 1024 / d
    row = {"launches": d, "l2_fabric_read": rd, "l2_fabric_write": wb}
    if k in sizes:
        s, ds = sizes[k], sizes[k]["dispatches"]
        n32, n64, n128 = s.get("TCC_EA0_RDREQ_32B_sum", 0.0), s.get("TCC_EA0_RDREQ_64B_sum", 0.0), s.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        rest = max(s.get("TCC_EA0_RDREQ_sum", 0.0) - n32 - n64 - n128, 0.0)
        row["l2_fabric_read_by_request_size"] = (32 * n32 + 64 * (n64 + rest) + 128 * n128) / ds
        row["read_requests"] = {"32B": n32 / ds, "64B": n64 / ds, "128B": n128 / ds, "unsized": rest / ds}
        rd = max(rd, row["l2_fabric_read_by_request_size"])
    if k in wr:
        s, ds = wr[k], wr[k]["dispatches"]
        row["write_requests"] = {"all": s.get("TCC_EA0_WRREQ_sum", 0.0) / ds, "64B": s.get("TCC_EA0_WRREQ_64B_sum", 0.0) / ds}
        hit, miss = s.get("TCC_HIT_sum"), s.get("TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            row["l2_hit_rate"] = hit / (hit + miss)
    if k in sq and sq[k].get("SQ_INSTS_VALU"):
        q, dq = sq[k], sq[k]["dispatches"]
        row["valu_insts"] = q["SQ_INSTS_VALU"] / dq  # wave-level vector instructions per launch
        row["valu_lanes_active"] = q.get("SQ_THREAD_CYCLES_VALU", 0.0) / q["SQ_INSTS_VALU"]  # of 64
        if q.get("SQ_WAVE_CYCLES"):
            row["wave_wait_share"] = q.get("SQ_WAIT_INST_ANY", 0.0) / q["SQ_WAVE_CYCLES"]  # of a wave's cycles spent waiting on an instruction's operands
    if k in sq2 and sq2[k].get("SQ_INSTS_SALU") is not None:
        q, dq = sq2[k], sq2[k]["dispatches"]
        row["salu_insts"] = q["SQ_INSTS_SALU"] / dq               # wave-level scalar instructions per launch
        row["vmem_rd_insts"] = q.get("SQ_INSTS_VMEM_RD", 0.0) / dq  # wave-level vector memory reads per launch
        row["lds_insts"] = q.get("SQ_INSTS_LDS", 0.0) / dq          # wave-level LDS instructions per launch
    if k in tcp and tcp[k].get("TCP_TOTAL_CACHE_ACCESSES_sum") is not None:
        q, dq = tcp[k], tcp[k]["dispatches"]
        row["l1_accesses"] = q["TCP_TOTAL_CACHE_ACCESSES_sum"] / dq           # per launch, summed over the CUs
        row["l1_miss_requests"] = q.get("TCP_TCC_READ_REQ_sum", 0.0) / dq       # read requests the L1s sent to the L2
    if k in tatd and tatd[k].get("GRBM_GUI_ACTIVE"):
        q = tatd[k]
        # GRBM_GUI_ACTIVE sums the 8 XCDs; TA / TD busy cycles sum the 256 CUs
        row["ta_busy_share"] = q.get("TA_TA_BUSY_sum", 0.0) / 256.0 / (q["GRBM_GUI_ACTIVE"] / 8.0)
        row["td_busy_share"] = q.get("TD_TD_BUSY_sum", 0.0) / 256.0 / (q["GRBM_GUI_ACTIVE"] / 8.0)
    row["l2_fabric_total"] = rd + wb
    res["kernels"][k] = row
print(json.dumps(res, indent=1))
