# A/B of env-var variants inside ONE call: bash tools/r4_variants_env.sh <tag> "VAR=val ..." "VAR=val ..."
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
: > $O/variants.txt
for rep in 1 2; do
for ev in "$@"; do
  for n in 4096 32768; do
    line=$(env $ev timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N")
    echo "[$ev] $line" | tee -a $O/variants.txt
  done
done
done
