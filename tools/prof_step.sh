set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$R/gpurun_out
rm -rf $O/prof_tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tmp -- python3 $R/bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline --step-launch graph > $O/tmp_bench_profiled.json 2> $O/tmp_bench_profiled.log
python3 $R/tools/step_breakdown.py $(ls $O/prof_tmp/*/*kernel_trace.csv | head -1) 70 > $O/tmp_step_breakdown.txt
rm -rf $O/prof_tmp
cat $O/tmp_step_breakdown.txt
