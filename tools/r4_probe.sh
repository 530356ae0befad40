# Round-4 kernel iteration on the GPU box: bash tools/r4_probe.sh <tag> [quick|full|none]   (writes gpurun_out/<tag>/)
#   parity subset (or the full -m gpu suite), then phase stamps of k_env_step at 4096 envs and launch times at 256 / 4096 / 32768 envs
set -o pipefail
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r4_x}
MODE=${2:-quick}
O=gpurun_out/$TAG
mkdir -p $O
export PBHC_SPECIALISE_STRICT=1
rc=0
if [ "$MODE" = "quick" ]; then
  timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_v2.py tests/test_gpu_specialise.py -x -q -m gpu -k "env_step or general_tracking or trace or specialis or reinit or lazy or deploy or recorded" > $O/tests.log 2>&1; rc=$?
elif [ "$MODE" = "full" ]; then
  timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; rc=$?
fi
echo "tests rc=$rc" >> $O/tests.log
tail -4 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
PBHC_SPEC_DEFINES=-DPBHC_STAMPS timeout -k 10 200 python3 tools/kernel_probe.py 4096 > $O/stamps.txt 2>&1 || exit 1
for n in 256 4096 32768; do timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N"; done > $O/time_vs_envs.txt
cat $O/time_vs_envs.txt
grep -A14 "role A" $O/stamps.txt | head -16
grep -A6 "role B" $O/stamps.txt
