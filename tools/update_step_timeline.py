"""From a `rocprofv3 --kernel-trace --output-format csv` directory of tools/update_trace_probe.py: the kernels of ONE optimiser step (between two
consecutive k_adam_clip launches near the end of the trace), microseconds from the first, with durations and gaps; and the step's totals by family."""
import csv, glob, sys, re

d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_adam_clip")]
a, b = idx[-3] + 1, idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
fam = {}
print("#   start  duration   gap kernel")
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    m = re.search(r"MT\d+x\d+x\d+", n)
    short = n[:70] + (" " + m.group(0) if m else "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:6.1f}  {short}")
    key = "k_gemm2" if "k_gemm2" in n else "Cijk (library GEMM)" if n.startswith("Cijk") else "at::native (torch elementwise / reduce / copy)" if "at::" in n or "rocclr" in n else "pbhc other"
    fam[key] = fam.get(key, 0.0) + (e - s) / 1e3
    prev_end = max(prev_end, e)
print(f"# step: {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a + 1} kernels")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"#   {k:50s} {v:8.1f} us")
