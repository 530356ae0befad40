# Split-K factors of the library weight-gradient GEMMs with bench.py itself as the meter (PBHC_WGRAD_P="NxK:P,..."): bash tools/probes/wgrad_parts_sweep.sh
run() { PBHC_WGRAD_P="$1" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dp-rehearsal --no-secondary 2>/dev/null | grep "^{" | python3 -c "
import json,sys; r=json.loads(sys.stdin.read()); print('[$1]', r['update_ms'])"; }
for c in "" "512x768:8,768x630:8" "512x768:4,768x630:8" "512x768:8,768x630:16" "512x768:16,768x630:16" "768x630:16" "" "512x768:8,768x630:8" "512x768:4,768x630:8"; do run "$c"; done
