# SQ / instruction-cache counters of k_env_step (tools/kernel_probe.py 4096: v1 walk, 4096 envs, 120 launches): bash tools/env_step_pmc.sh [tag]
# One --pmc pass per group of <= 8 SQ counters, no tracing domain other than --kernel-trace; the probe itself is the program after `--`.
set -e
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-sq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
pass() {
  n=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$n -- python3 tools/kernel_probe.py 4096 > $OUT/$n.log 2>&1
  echo "pass $n done"
}
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH
pass p2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass p3 SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_IFETCH
pass p4 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ
pass p5 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VSKIPPED
python3 tools/env_step_pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
