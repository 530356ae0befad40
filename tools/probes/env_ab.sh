# A/B of an environment switch at bench level inside one call: bash tools/probes/env_ab.sh "VAR=a" "VAR=b" [repetitions]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=${3:-3}
run() { env $1 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal 2>&1 | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('[$1]', j['value'], 'rollout', j['rollout_ms'], 'update', j['update_ms'])"; }
for i in $(seq 1 $R); do run "$1"; run "$2"; done
