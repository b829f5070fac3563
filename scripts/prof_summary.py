"""Summarise rocprofv3 output directories: kernel stats and (if present) PMC counters per kernel.
usage: prof_summary.py <dir> [<dir> ...]"""
import csv, glob, sys, os, collections
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print("==", f)
        for r in list(csv.DictReader(open(f)))[:12]:
            print(f"{r['Name'][:72]:72s} calls={int(r['Calls']):4d} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        if k.startswith("void at::"): continue
        s = " ".join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items()))
        extra = ""
        if "FETCH_SIZE" in c: extra += f" read_GB={2*sum(c['FETCH_SIZE'])/len(c['FETCH_SIZE'])*1024/1e9:.3f}"
        if "WRITE_SIZE" in c: extra += f" write_GB={sum(c['WRITE_SIZE'])/len(c['WRITE_SIZE'])*1024/1e9:.3f}"
        print(f"{k[:60]:60s} {s}{extra}")
