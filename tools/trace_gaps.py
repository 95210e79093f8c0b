"""Developer tool: per-launch durations and the idle gaps between consecutive kernels of one frame, from a rocprofv3 --kernel-trace CSV
(where the frame's wall time goes that no kernel's duration shows: launch gaps, near-empty launches of late bounces).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/profile_frame.py --config c4 --no-warm ; python tools/trace_gaps.py DIR"""
import csv, glob, os, sys, collections
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(rows[i + 1][0] - rows[i][1], 0) for i in range(len(rows) - 1)]
print("launches %d, first start to last end %.2f ms, sum of durations %.2f ms, sum of gaps %.2f ms (mean %.1f us, max %.1f us)" % (
    len(rows), (t1 - t0) / 1e6, busy / 1e6, sum(gaps) / 1e6, sum(gaps) / max(len(gaps), 1) / 1e3, max(gaps) / 1e3))
by = collections.defaultdict(list)
for s, e, k in rows:
    by[k].append((e - s) / 1e3)
print("%-36s %6s %10s %10s %10s %10s" % ("kernel", "calls", "total ms", "max us", "median us", "min us"))
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print("%-36s %6d %10.2f %10.1f %10.1f %10.1f" % (k[:36], len(v), sum(v) / 1e3, v2[-1], v2[len(v2) // 2], v2[0]))
short = [(e - s) / 1e3 for s, e, k in rows if (e - s) < 200e3]
print("launches shorter than 200 us: %d, %.2f ms in all" % (len(short), sum(short) / 1e3))
# the sequence of one pass of the first frame: kernel, duration
print("first 80 launches (us):", " ".join("%s:%.0f" % (k.split("<")[0].replace("k_", ""), (e - s) / 1e3) for s, e, k in rows[:80]))
