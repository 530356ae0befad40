# A/B inside one call: the stack kernel with 8 / 16 waves per workgroup (rebuilds pbhc_mlp.hip on the box)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('$1', j['value'], 'rollout', j['rollout_ms'], 'update', j['update_ms'])"; }
for rep in 1 2; do
touch pbhc_amd/csrc/pbhc_mlp.hip && make -C pbhc_amd/csrc > /dev/null 2>&1
run "waves 8 "
touch pbhc_amd/csrc/pbhc_mlp.hip && make -C pbhc_amd/csrc DEFS=-DMLP_WAVES=16 > /dev/null 2>&1
run "waves 16"
done
