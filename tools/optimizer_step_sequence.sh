# Kernel sequence (name, duration) of ONE optimiser step of the general-tracking teacher update, from a rocprofv3 kernel trace: bash tools/optimizer_step_sequence.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/stepseq
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/stepseq/t -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary ${PBHC_SEQ_ARGS:---workload v2_teacher29 --clips 256} > gpurun_out/stepseq/log.txt 2>&1
python3 - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/stepseq/t/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"MT\d+x\d+x\d+", n)
    if n.startswith("Cijk"): return "GEMM " + ("SB " if "_SB_" in n else "") + (m.group(0) if m else "")
    n = re.sub(r"at::native::|\(anonymous namespace\)::|void ", "", n)
    return n[:70]
# find the last k_ppo_loss (start of the last optimiser step's backward) and print the 260 kernels before it (that step's forward) and 200 after
idx = [i for i, r in enumerate(rows) if "k_ppo_loss" in r["Kernel_Name"]]
i0 = idx[-2]
out = open("gpurun_out/stepseq/seq.txt", "w")
for r in rows[i0 - 10:idx[-1] + 5]:
    out.write(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {short(r['Kernel_Name'])}\n")
PY
rm -rf gpurun_out/stepseq/t
wc -l gpurun_out/stepseq/seq.txt
