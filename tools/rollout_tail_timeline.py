"""From a `rocprofv3 --kernel-trace --output-format csv` directory of tools/rollout_trace_probe.py: the kernels that follow the LAST control step
of the last rollout (batched critic, bootstrap, GAE), microseconds from the start of that step's k_env_step; and the loop's per-step period."""
import csv, glob, sys

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_env_step")]
starts = [int(rows[i]["Start_Timestamp"]) for i in idx[-24:]]
per = [(b - a) / 1e3 for a, b in zip(starts[:-1], starts[1:])]
print(f"# last rollout: {len(per)} step periods, median {sorted(per)[len(per) // 2]:.1f} us, min {min(per):.1f}, max {max(per):.1f}")
a = idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
print("#   start       end  duration kernel")
for r in rows[a:]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}   {r['Kernel_Name'][:110]}")
