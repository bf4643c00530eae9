#!/bin/bash
# Everything the committed round-3 profiles come from, in ONE gpurun call (run from the repo root on the GPU box):
#   tools/refresh_profiles.sh            -> gpurun_out/r03_*; copy the summaries into profiles/ afterwards
# Steps are joined so that a failed or timed-out GPU step stops the script.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
# 1. the default bench line (sampler legs and CPU baseline included)
timeout -k 10 500 python3 $R/bench.py > $O/r03_bench_unprofiled.json 2> $O/r03_bench_default.err
# 2. kernel trace of the captured training step
rm -rf $O/prof_r3b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3b -- python3 $R/bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline --step-launch graph > $O/r03_bench_profiled.json 2> $O/r03_bench_profiled.log
cp $(ls $O/prof_r3b/*/*kernel_stats.csv | head -1) $O/r03_bench_kernel_stats.csv
python3 $R/tools/step_breakdown.py $(ls $O/prof_r3b/*/*kernel_trace.csv | head -1) 60 > $O/r03_step_breakdown.txt
# 3. HBM counters, one pass per counter (graph replay: the step as it is timed)
MODE=${PMC_MODE:-graph}
rm -rf $O/pmc_r3_fetch $O/pmc_r3_write
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_r3_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch $MODE > /dev/null 2> $O/r03_pmc_f.log
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_r3_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --step-launch $MODE > /dev/null 2> $O/r03_pmc_w.log
python3 $R/tools/pmc_hbm.py $(ls $O/pmc_r3_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_r3_write/*/*counter_collection.csv | head -1) $O/r03_pmc_hbm.json > $O/r03_pmc_summary.txt
# 4. sampler kernel traces
rm -rf $O/prof_r3s $O/prof_r3sf
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3s -- python3 $R/tools/sampler_profile.py 512 bf16 > $O/r03_sampler_bf16.log 2>&1
cp $(ls $O/prof_r3s/*/*kernel_stats.csv | head -1) $O/r03_sampler_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r3sf -- python3 $R/tools/sampler_profile.py 256 f32 > $O/r03_sampler_f32.log 2>&1
cp $(ls $O/prof_r3sf/*/*kernel_stats.csv | head -1) $O/r03_sampler_f32_kernel_stats.csv
echo done
