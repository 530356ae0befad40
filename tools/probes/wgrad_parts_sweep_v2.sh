# The same sweep on the general-tracking teacher (BASELINE configs[2]): bash tools/probes/wgrad_parts_sweep_v2.sh
run() { PBHC_WGRAD_P="$1" python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --workload v2_teacher29 --clips 256 2>/dev/null | grep "^{" | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('[$1]', r['update_ms'])"; }
for c in "" "768x464:8,768x466:8" "768x464:2,768x466:2" "512x768:8" "256x512:16" "" "768x464:8,768x466:8"; do run "$c"; done
