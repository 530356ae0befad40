# Round-4 profile collection on the GPU box: bash tools/collect_round4.sh   (writes gpurun_out/prof4; the summaries are copied to profiles/)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof4
mkdir -p $O
step() { echo "== $1"; }
step prebuild
python3 tools/kernel_probe.py 4096 > $O/probe_specialised.txt 2>&1
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --no-secondary > $O/bench_prebuild.log 2>&1
step "v1 kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal > $O/v1_bench.log 2>&1
cp $(find $O/v1 -name "*kernel_stats.csv" | head -1) $O/round4_bench_kernel_stats.csv
step "v2 kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --workload v2_teacher29 --clips 256 > $O/v2_bench.log 2>&1
cp $(find $O/v2 -name "*kernel_stats.csv" | head -1) $O/round4_bench_kernel_stats_v2_teacher29.csv
rm -rf $O/v1 $O/v2
step "phase stamps"
PBHC_SPEC_DEFINES=-DPBHC_STAMPS timeout -k 10 200 python3 tools/kernel_probe.py 4096 > $O/round4_k_env_step_phase_stamps.txt 2>&1
step "time vs envs"
for n in 256 1024 2048 4096 8192 16384 32768; do timeout -k 10 120 python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N"; done > $O/round4_k_env_step_time_vs_envs.txt
cat $O/round4_k_env_step_time_vs_envs.txt
step "SQ counters (4096 envs)"
timeout -k 10 600 bash tools/env_step_pmc.sh prof4/sq > $O/sq.log 2>&1
cp gpurun_out/prof4/sq/summary.txt $O/round4_k_env_step_sq_raw.txt 2>/dev/null
step "HBM traffic, v1"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/kernel_probe.py 4096 > $O/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/kernel_probe.py 4096 > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/round4_k_env_step_pmc.json
rm -rf $O/pmc_fetch $O/pmc_write
step "HBM traffic, v2"
python3 tools/kernel_probe.py 4096 v2_teacher29 256 > $O/probe_v2.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/kernel_probe.py 4096 v2_teacher29 256 > $O/pmc_fetch_v2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/kernel_probe.py 4096 v2_teacher29 256 > $O/pmc_write_v2.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/round4_k_env_step_pmc_v2_teacher29.json "tools/kernel_probe.py 4096 v2_teacher29 256 (29-DoF general tracking, 256-clip library, 4096 envs)"
rm -rf $O/pmc_fetch $O/pmc_write
step "memory pipeline at 32768 envs"
timeout -k 10 900 bash tools/r4_pmc32k.sh prof4/mem32k > $O/mem32k.log 2>&1
cp gpurun_out/prof4/mem32k/summary.txt $O/round4_k_env_step_memory_pipeline_raw.txt 2>/dev/null
step "update step timeline (v1)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/upd -- python3 tools/update_trace_probe.py > $O/upd.log 2>&1
python3 tools/update_step_timeline.py $O/upd > $O/round4_update_step_timeline.txt 2>&1
rm -rf $O/upd
find $O -name "*.csv" -size +2M -delete
find gpurun_out/prof4 -name "*kernel_trace.csv" -delete
du -sh $O
ls $O
