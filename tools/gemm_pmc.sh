# PMC counters (SQ wait / busy, LDS conflicts, TA, L2 hits) of one fused-GEMM configuration: bash tools/gemm_pmc.sh N K shape variant
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gpmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/gpmc/p1 -- python3 tools/gemm_one.py $1 $2 $3 $4 > gpurun_out/gpmc/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/gpmc/p2 -- python3 tools/gemm_one.py $1 $2 $3 $4 > gpurun_out/gpmc/p2.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/gpmc/p3 -- python3 tools/gemm_one.py $1 $2 $3 $4 > gpurun_out/gpmc/p3.log 2>&1 || true
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    fs = glob.glob(f"gpurun_out/gpmc/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counters"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "k_gemm" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[5:]
        print(f"{p} {k:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
find gpurun_out/gpmc -name "*.csv" -size +4M -delete
