export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r4_abl2}
mkdir -p $O
: > $O/ablation.txt
for abl in NONE OBSX4 "OBSX4 -DPBHC_ABL_WB" "OBS -DPBHC_ABL_WB" "OBS -DPBHC_ABL_WB -DPBHC_ABL_HIST"; do
  defs="-DPBHC_ABL_NORESET"
  [ "$abl" != "NONE" ] && defs="$defs -DPBHC_ABL_$abl"
  for n in 4096 32768; do
    line=$(PBHC_SPEC_DEFINES="$defs" timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N")
    echo "$abl: $line" | tee -a $O/ablation.txt
  done
done
