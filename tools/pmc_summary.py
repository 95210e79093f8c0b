"""Sums rocprofv3 --pmc counter_collection.csv per kernel name (developer tool)."""
import csv, sys, collections, glob, os
d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row["Dispatch_Id"])
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(acc[k].items()):
        print("   %-28s %.4g" % (c, v))
