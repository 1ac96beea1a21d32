"""Condenses gpurun_out/profiles_<tag>/ into small committed summaries under profiles/.
Usage: python tools/summarize_profiles.py <tag>"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
src = "gpurun_out/profiles_%s" % tag
dst = "profiles"
os.makedirs(dst, exist_ok=True)
ours = ("vs::", "_GLOBAL__N_", "anonymous namespace")
summary = {}
for f in sorted(glob.glob(src + "/*_kernel_stats.csv")):
    name = os.path.basename(f).replace("_kernel_stats.csv", "")
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ours)]
    with open("%s/%s_%s_kernel_stats.csv" % (dst, tag, name), "w") as out:
        w = csv.writer(out)
        w.writerow(["Name", "Calls", "AverageNs", "MinNs", "MaxNs", "TotalDurationNs", "Percentage"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["TotalDurationNs"], r["Percentage"]])
    summary[name] = {r["Name"][:60]: float(r["AverageNs"]) for r in rows}
traffic = {}
for f in sorted(glob.glob(src + "/*_counters.csv")):
    name = os.path.basename(f).replace("_counters.csv", "")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in ours):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    traffic[name] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
json.dump({"kernel_avg_ns": summary, "pmc_avg_per_dispatch": traffic}, open("%s/%s_summary.json" % (dst, tag), "w"), indent=1)
for name, ks in traffic.items():
    print(name)
    for k, cs in ks.items():
        print("   ", k[:58].ljust(58), {c: round(v, 1) for c, v in cs.items()})
