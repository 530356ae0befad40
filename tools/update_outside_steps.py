"""From a `rocprofv3 --kernel-trace --output-format csv` directory of tools/update_trace_probe.py: what one iteration's update phase runs OUTSIDE its
optimiser steps — the kernels between the rollout's last launch (k_gae) and the first optimiser step's first GEMM (permutation, minibatch gather), and
the time from the last k_adam_clip to the next iteration's first k_env_step."""
import csv, glob, sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gae = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_gae")]
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam_clip")]
a = gae[-1]
first_adam = next(i for i in adam if i > a)
t0 = int(rows[a]["Start_Timestamp"])
prev = t0
print("#   start  duration   gap kernel")
for r in rows[a:first_adam + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:6.1f}  {r['Kernel_Name'][:100]}")
    prev = max(prev, e)
last = [i for i in adam if i > a][:20][-1]
print(f"# update: k_gae start -> 20th k_adam_clip end {(int(rows[last]['End_Timestamp']) - t0) / 1e3:.1f} us; first optimiser step ends at {(int(rows[first_adam]['End_Timestamp']) - t0) / 1e3:.1f} us")
