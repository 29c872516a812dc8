#!/bin/bash
# One profiling pass on the GPU box (run through gpurun from the repo root):  profiles/collect.sh <tag>
# Leaves everything under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/ with the round's prefix afterwards:
#   <tag>_bench_line.json                the default bench.py line (all variants, CPU baseline)
#   <tag>_kernel_stats.csv               rocprofv3 --kernel-trace --stats of `bench.py --steps 20` (+ the line printed under the profiler)
#   <tag>_pmc_traffic.json               two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), eager launches, profiles/pmc_traffic.py
#   <tag>_sq_counters.json               one --pmc pass of the SQ counters, profiles/sq_counters.py
#   <tag>_kernel_stats_{hidden256,hid1024,dialoguernn,onlysp}.csv     a few steps of those variants
# The program itself follows `--` (never env / bash -c: the profiler's preloaded library has initialised the GPU already).
tag=${1:-r03_z}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -o ks -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-variants > $out/${tag}_bench_line_under_rocprof.json 2> $out/ks.err
cp $out/ks/ks_kernel_stats.csv $out/${tag}_kernel_stats.csv 2>/dev/null || find $out/ks -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o fetch -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-graph --no-roofline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o write -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-graph --no-roofline > /dev/null 2>&1
python profiles/pmc_traffic.py $out/fetch $out/write > $out/${tag}_pmc_traffic.json
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/sq -o sq -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-variants --no-graph --no-roofline > /dev/null 2>&1
python profiles/sq_counters.py $out/sq > $out/${tag}_sq_counters.json
for v in "hidden256 scratch/sps_steps.py 256 5" "hid1024 scratch/sps_steps.py 1024 3 32 256 8" "dialoguernn scratch/drnn_steps.py 3" "onlysp scratch/onlysp_steps.py"; do
  set -- $v; name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$name -o ks -- python "$@" > $out/ks_$name.log 2>&1
  find $out/ks_$name -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats_$name.csv \;
done
rm -rf $out/ks $out/fetch $out/write $out/sq $out/ks_hidden256 $out/ks_hid1024 $out/ks_dialoguernn $out/ks_onlysp
ls -la $out
