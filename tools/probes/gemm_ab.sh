# A/B inside one call: bench.py with pbhc_gemm.hip before / after a change.  Put the previous version next to this script first:
#   git show HEAD:pbhc_amd/csrc/pbhc_gemm.hip > tools/probes/_gemm_prev.hip.txt   (not committed)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], 'rollout', j['rollout_ms'], 'update', j['update_ms'])"; }
cp pbhc_amd/csrc/pbhc_gemm.hip /tmp/gemm_new.hip
run "new "
cp tools/probes/_gemm_prev.hip.txt pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "prev"
cp /tmp/gemm_new.hip pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "new "
cp tools/probes/_gemm_prev.hip.txt pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "prev"
cp /tmp/gemm_new.hip pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "new "
cp tools/probes/_gemm_prev.hip.txt pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "prev"
cp /tmp/gemm_new.hip pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "new "
