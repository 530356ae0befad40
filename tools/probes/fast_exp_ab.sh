# A/B inside one call: bench.py's update phase with the library expf / the hardware 2^x in the fused GEMMs' epilogues (rebuilds the library on the box)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], 'rollout', j['rollout_ms'], 'update', j['update_ms'])"; }
run "fastexp"
touch pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc DEFS=-DPBHC_GEMM_LIBM_EXP > /dev/null 2>&1
run "expf   "
touch pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "fastexp"
touch pbhc_amd/csrc/pbhc_gemm.hip && make -C pbhc_amd/csrc DEFS=-DPBHC_GEMM_LIBM_EXP > /dev/null 2>&1
run "expf   "
