# Round-3 profile collection on the GPU box: bash tools/collect_round3.sh   (writes gpurun_out/prof3; copy the summaries to profiles/)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/prof3
mkdir -p $O
# the specialised kernels of the probed configs are built first, outside any profiler (cached in pbhc_amd/_spec for the rest of the call)
python3 tools/kernel_probe.py 4096 > $O/probe_specialised.txt 2>&1
PBHC_SPECIALISE=off python3 tools/kernel_probe.py 4096 > $O/probe_generic.txt 2>&1
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-dp-rehearsal > $O/bench_prebuild.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal > $O/v1_bench.log 2>&1
echo "v1 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/v2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --workload v2_teacher29 --clips 256 > $O/v2_bench.log 2>&1
echo "v2 stats done"
bash tools/env_step_pmc.sh prof3/sq_specialised
PBHC_SPECIALISE=off bash tools/env_step_pmc.sh prof3/sq_generic
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/kernel_probe.py 4096 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 tools/kernel_probe.py 4096 > $O/pmc_write.log 2>&1
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/round3_k_env_step_pmc.json
PBHC_SPEC_DEFINES=-DPBHC_STAMPS python3 tools/kernel_probe.py 4096 > $O/stamps_specialised.txt 2>&1
for n in 256 1024 2048 4096 8192 16384 32768; do python3 tools/kernel_probe.py $n 2>&1 | grep "k_env_step N"; done > $O/time_vs_envs.txt
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*.csv" -size +2M -delete
du -sh $O
