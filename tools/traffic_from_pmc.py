"""Turns the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, TCC slots) into per-launch HBM
traffic per kernel, with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE
are in KiB; FETCH_SIZE reports half of the bytes of a coalesced stream (calibrated here on k_accumulate, whose byte
count is known: it reads 12 B per path slot + 12 B per pixel and writes 12 B per pixel)."""
import json, re, sys

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

fetch, write = parse(sys.argv[1]), parse(sys.argv[2])
n_slots, n_pixels = int(sys.argv[3]), int(sys.argv[4])
acc = fetch["k_accumulate"]
known_read = 12.0 * n_slots + 12.0 * n_pixels
read_scale = known_read / (acc["FETCH_SIZE"] * 1024 / acc["dispatches"])
res = {"unit": "bytes per launch", "fetch_size_scale_calibrated_on_k_accumulate": read_scale, "kernels": {}}
for k in fetch:
    if k.startswith("__") or k not in write:
        continue
    d = fetch[k]["dispatches"]
    rd = fetch[k]["FETCH_SIZE"] * 1024 / d * 2.0   # guide: FETCH_SIZE = half the streamed bytes on gfx950
    wr = write[k]["WRITE_SIZE"] * 1024 / d
    res["kernels"][k] = {"launches": d, "hbm_read": rd, "hbm_write": wr, "hbm_total": rd + wr}
print(json.dumps(res, indent=1))
