"""Join rocprofv3 kernel stats with the FETCH_SIZE / WRITE_SIZE passes of scripts/gpu_stream_copy.sh."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


dur = collections.defaultdict(list)
f = newest(root + "/stream_copy_stats/**/*kernel_trace.csv")
for r in csv.DictReader(open(f)):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = newest(root + f"/stream_copy_{c}/**/*counter_collection.csv")
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            ctr[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
print(f"{'kernel':86s} {'n':>3s} {'min_us':>8s} {'avg_us':>8s} {'read_GB(x2)':>11s} {'write_GB':>8s} {'TB/s(min)':>9s}")
for k, d in dur.items():
    rd = 2 * 1024 * sum(ctr[k]["FETCH_SIZE"]) / max(len(ctr[k]["FETCH_SIZE"]), 1) / 1e9
    wr = 1024 * sum(ctr[k]["WRITE_SIZE"]) / max(len(ctr[k]["WRITE_SIZE"]), 1) / 1e9
    mn, av = min(d) / 1e3, sum(d) / len(d) / 1e3
    print(f"{k[:86]:86s} {len(d):3d} {mn:8.1f} {av:8.1f} {rd:11.3f} {wr:8.3f} {(rd + wr) / mn * 1e3:9.2f}")
