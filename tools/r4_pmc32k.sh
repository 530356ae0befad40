# Memory-pipeline counters of k_env_step at 32768 envs (throughput regime): bash tools/r4_pmc32k.sh <tag> [N]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4_pmc}
N=${2:-32768}
mkdir -p $O
export PBHC_PROBE_STEPS=20
python3 tools/kernel_probe.py $N > $O/prebuild.txt 2>&1
: > $O/summary.txt
pass() {
  n=$1; shift
  echo "pass $n: $@"
  timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$n -- python3 tools/kernel_probe.py $N > $O/$n.log 2>&1
  echo "pass $n done rc=$?"
  python3 tools/r4_pmc_sum.py $O $n >> $O/summary.txt 2>&1
  rm -rf $O/$n
}
pass q1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass q2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS
pass q3 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass q4 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr
pass q5 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum
pass q6 TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum
pass q7 TCC_BUSY_sum TCC_EA0_WRREQ_STALL_sum
pass q8 TCC_REQ_sum TCC_HIT_sum
pass q9 GRBM_GUI_ACTIVE
cat $O/summary.txt
