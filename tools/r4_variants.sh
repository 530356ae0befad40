# A/B of k_env_step build variants inside ONE call (one box, one clock): bash tools/r4_variants.sh <tag> "<defs1>" "<defs2>" ...
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
: > $O/variants.txt
for rep in 1 2; do
for defs in "$@"; do
  for n in 4096 32768; do
    line=$(PBHC_SPEC_DEFINES="$defs" timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N")
    echo "[$defs] $line" | tee -a $O/variants.txt
  done
done
done
