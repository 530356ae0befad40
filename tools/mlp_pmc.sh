# PMC counters of the whole-stack policy kernel (k_mlp_fwd, 4096 rows): bash tools/mlp_pmc.sh   (one --pmc pass, no other tracing domains)
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mpmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/mpmc/p1 -- python3 tools/mlp_probe.py 4096 > gpurun_out/mpmc/p1.log 2>&1
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/mpmc/p1/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"]
    if "k_mlp_fwd" in n:
        acc[(n[:20], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (n, g), d in acc.items():
    print(f"# {n} grid {g} (first group of launches: the actor's stack, second: the critic's)")
    for k, v in d.items():
        print(f"{k:28s} actor {sum(v[:200]) / max(len(v[:200]), 1):14.0f}   all launches {sum(v) / len(v):14.0f}  (n={len(v)})")
PY
find gpurun_out/mpmc -name "*.csv" -size +2M -delete
