"""Median per-launch value of every PMC counter rocprofv3 collected for k_env_step (summary of p_counter_collection.csv files)."""
import collections
import csv
import glob
import statistics
import sys

out = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if "k_env_step" in r["Kernel_Name"]:
            per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])     # summed over the counter's dimensions
    for (d, name), v in per_dispatch.items():
        out[name].append(v)
for name, v in sorted(out.items()):
    print(f"{name:24s} median {statistics.median(v):16.1f}  launches {len(v)}")
