"""Turn a `rocprofv3 --kernel-trace --output-format csv` directory of tools/rollout_trace_probe.py into a per-control-step timeline:
the kernels between the last two k_env_step launches, microseconds from the start of the first, with their queue."""
import csv, glob, sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_env_step")]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
queues = {}
print("#   start       end  duration queue kernel")
for r in rows[a:b + 1]:
    q = queues.setdefault(r["Queue_Id"], len(queues) + 1)
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f} q{q:2d}   {r['Kernel_Name'][:100]}")
