# The trace-based part of tools/collect_round4.sh alone (kernel stats of both workloads, update / rollout timelines): bash tools/collect_round4_tail.sh
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof4b
mkdir -p $O
python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --no-secondary > $O/bench_prebuild.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-dp-rehearsal > $O/v1_bench.log 2>&1
cp $(find $O/v1 -name "*kernel_stats.csv" | head -1) $O/round4_bench_kernel_stats.csv
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v2 -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dp-rehearsal --workload v2_teacher29 --clips 256 > $O/v2_bench.log 2>&1
cp $(find $O/v2 -name "*kernel_stats.csv" | head -1) $O/round4_bench_kernel_stats_v2_teacher29.csv
rm -rf $O/v1 $O/v2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/upd -- python3 tools/update_trace_probe.py > $O/upd.log 2>&1
python3 tools/update_step_timeline.py $O/upd > $O/round4_update_step_timeline.txt 2>&1
rm -rf $O/upd
PBHC_ROLLOUT_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/rtl -- python3 tools/rollout_trace_probe.py 4096 v2_teacher29 256 > $O/rtl.log 2>&1
python3 tools/rollout_timeline.py $O/rtl > $O/rollout_step_timeline_v2.txt
python3 tools/rollout_tail_timeline.py $O/rtl > $O/rollout_tail_timeline_v2.txt 2>&1
rm -rf $O/rtl
bash tools/optimizer_step_sequence.sh > $O/stepseq.log 2>&1; cp gpurun_out/stepseq/seq.txt $O/optimizer_step_sequence_v2.txt
rm -rf gpurun_out/stepseq/t
timeout -k 10 600 python3 bench.py --steps 5 --warmup 2 > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log > $O/bench_default_line.json
find $O -name "*.csv" -size +2M -delete
ls -la $O
