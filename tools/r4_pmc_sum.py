"""sum rocprofv3 counter_collection csvs of k_env_step launches (average per launch, the first 5 launches skipped)"""
import csv, glob, os, sys, collections
root, sub = sys.argv[1], sys.argv[2]
d = os.path.join(root, sub)
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_env_step" in r.get("Kernel_Name", ""):
            per[r["Counter_Name"]][r.get("Dispatch_Id", "0")] += float(r["Counter_Value"])
for k, dd in sorted(per.items()):
    v = [dd[i] for i in sorted(dd, key=lambda x: int(x))]
    v = v[5:] if len(v) > 10 else v
    print(f"{sub} {k:40s} {sum(v) / max(len(v), 1):18.1f}   (launches {len(v)})")
