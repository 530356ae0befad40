set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/prof2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2/v1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/prof2/v1_bench.log 2>&1
echo "v1 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2/v2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --workload v2_teacher29 --clips 256 > gpurun_out/prof2/v2_bench.log 2>&1
echo "v2 stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof2/pmc_fetch -- python3 tools/kernel_probe.py 4096 > gpurun_out/prof2/pmc_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof2/pmc_write -- python3 tools/kernel_probe.py 4096 > gpurun_out/prof2/pmc_write.log 2>&1
echo "write done"
python3 tools/pmc_summary.py gpurun_out/prof2/pmc_fetch gpurun_out/prof2/pmc_write gpurun_out/prof2/round2_k_env_step_pmc.json
PBHC_LIB=pbhc_amd/libpbhc_hip_stamps.so python3 tools/kernel_probe.py 4096 > gpurun_out/prof2/stamps.txt 2>&1
for n in 256 1024 2048 4096 8192 16384 32768; do python3 tools/kernel_probe.py $n 2>&1 | grep k_env_step; done > gpurun_out/prof2/time_vs_envs.txt
find gpurun_out/prof2 -name "*kernel_stats.csv" | head
# keep only the small summaries (the merge-back limit is 64 MiB)
find gpurun_out/prof2 -name "*kernel_trace.csv" -size +8M -delete
find gpurun_out/prof2 -name "*counter_collection.csv" -size +8M -delete
du -sh gpurun_out/prof2
# fused fp32-MFMA Linear kernels against the library path, the K-loop ablation and the sustained MFMA peak of the box
python3 tools/gemm_probe.py 24576 > gpurun_out/prof2/gemm_probe_24576.txt 2>&1
python3 tools/gemm_probe.py 4096 > gpurun_out/prof2/gemm_probe_4096.txt 2>&1
python3 tools/gemm_ablation.py 0 > gpurun_out/prof2/gemm_ablation.txt 2>&1
python3 tools/mfma_peak_probe.py > gpurun_out/prof2/mfma_peak.txt 2>&1
# rollout: one control step kernel by kernel on both streams, the loop's period and the tail (batched critic, bootstrap, GAE); the whole-stack
# policy kernel against the layer chain
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2/rtl -- python3 tools/rollout_trace_probe.py > gpurun_out/prof2/rtl.log 2>&1
python3 tools/rollout_timeline.py gpurun_out/prof2/rtl > gpurun_out/prof2/rollout_step_timeline.txt
python3 tools/rollout_tail_timeline.py gpurun_out/prof2/rtl > gpurun_out/prof2/rollout_tail_timeline.txt
find gpurun_out/prof2/rtl -name "*.csv" -size +4M -delete
for n in 64 1024 4096; do python3 tools/mlp_probe.py $n 2>&1 | grep rows; done > gpurun_out/prof2/mlp_stack_probe.txt
python3 tools/rollout_host_time.py 2>&1 | grep "^rollout" > gpurun_out/prof2/rollout_host_time.txt
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof2/utl -- python3 tools/update_trace_probe.py > gpurun_out/prof2/utl.log 2>&1
python3 tools/update_step_timeline.py gpurun_out/prof2/utl > gpurun_out/prof2/update_step_timeline.txt
find gpurun_out/prof2/utl -name "*.csv" -size +4M -delete
# raw traces are scratch: only the summaries travel back (the merge-back limit is 64 MiB)
find gpurun_out/prof2 -name "*kernel_trace.csv" -delete
find gpurun_out/prof2 -name "*counter_collection.csv" -delete
du -sh gpurun_out/prof2
