#!/bin/bash
# Everything the committed per-round profiles come from, in ONE gpurun call (run from the repo root on the GPU box):
#   ROUND=r06 tools/refresh_profiles.sh  -> gpurun_out/r05_*; copy the summaries into profiles/ afterwards
# Steps are joined so that a failed or timed-out GPU step stops the script.
set -e
RD=${ROUND:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
# under `rocprofv3 --pmc` the profiler initialises the GPU before python starts: the graph-safe runtime setting must be
# INHERITED, or the fail-closed check turns the captured step off (bench.py now refuses an explicit graph request then)
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
O=$R/gpurun_out
mkdir -p $O
# 1. the default bench line (sampler legs and CPU baseline included)
timeout -k 10 600 python3 $R/bench.py > $O/${RD}_bench_unprofiled.json 2> $O/${RD}_bench_default.err
# 2. kernel trace of the captured training step
rm -rf $O/prof_${RD}b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${RD}b -- python3 $R/bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline --step-launch graph > $O/${RD}_bench_profiled.json 2> $O/${RD}_bench_profiled.log
cp $(ls $O/prof_${RD}b/*/*kernel_stats.csv | head -1) $O/${RD}_bench_kernel_stats.csv
python3 $R/tools/step_breakdown.py $(ls $O/prof_${RD}b/*/*kernel_trace.csv | head -1) 60 > $O/${RD}_step_breakdown.txt
# 3. HBM counters, one pass per counter (graph replay: the step as it is timed)
MODE=${PMC_MODE:-graph}
rm -rf $O/pmc_${RD}_fetch $O/pmc_${RD}_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${RD}_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch $MODE > /dev/null 2> $O/${RD}_pmc_f.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${RD}_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch $MODE > /dev/null 2> $O/${RD}_pmc_w.log
if [ "$MODE" = graph ]; then    # the counters are labelled "graph replay": make sure that is what ran
  grep -q "hipGraph replay" $O/${RD}_pmc_f.log
  grep -q "hipGraph replay" $O/${RD}_pmc_w.log
fi
python3 $R/tools/pmc_hbm.py $(ls $O/pmc_${RD}_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_${RD}_write/*/*counter_collection.csv | head -1) $O/${RD}_pmc_hbm.json > $O/${RD}_pmc_summary.txt
# 4. sampler kernel traces
rm -rf $O/prof_${RD}s $O/prof_${RD}sf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${RD}s -- python3 $R/tools/sampler_profile.py 512 bf16 > $O/${RD}_sampler_bf16.log 2>&1
cp $(ls $O/prof_${RD}s/*/*kernel_stats.csv | head -1) $O/${RD}_sampler_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${RD}sf -- python3 $R/tools/sampler_profile.py 256 f32 > $O/${RD}_sampler_f32.log 2>&1
cp $(ls $O/prof_${RD}sf/*/*kernel_stats.csv | head -1) $O/${RD}_sampler_f32_kernel_stats.csv
rm -rf $O/prof_${RD}sx
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${RD}sx -- python3 $R/tools/sampler_profile.py 512 f32x3 > $O/${RD}_sampler_f32x3.log 2>&1
cp $(ls $O/prof_${RD}sx/*/*kernel_stats.csv | head -1) $O/${RD}_sampler_f32x3_kernel_stats.csv
# the raw traces are large: only the summaries above travel back
rm -rf $O/prof_${RD}b $O/prof_${RD}s $O/prof_${RD}sf $O/prof_${RD}sx $O/pmc_${RD}_fetch $O/pmc_${RD}_write
echo done
