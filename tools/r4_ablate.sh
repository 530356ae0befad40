# Timing ablations of k_env_step (specialised build, -DPBHC_ABL_*): launch time at 4096 / 32768 envs without one phase at a time.
# bash tools/r4_ablate.sh <tag>     -> gpurun_out/<tag>/ablation.txt
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r4_abl}
O=gpurun_out/$TAG
mkdir -p $O
: > $O/ablation.txt
for abl in NONE FK OBS RNG EBODY F PHASEC PHASED HIST JOINT WB; do
  defs="-DPBHC_ABL_NORESET"
  [ "$abl" != "NONE" ] && defs="$defs -DPBHC_ABL_$abl"
  for n in 4096 32768; do
    line=$(PBHC_SPEC_DEFINES="$defs" timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N")
    echo "$abl: $line" | tee -a $O/ablation.txt
  done
done
