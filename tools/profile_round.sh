#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command and
# the two PMC passes for HBM traffic. Summaries land in gpurun_out/prof_$1/ (copied to profiles/ by hand).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o bench -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp /tmp/prof_kt/bench_kernel_stats.csv $OUT/kernel_stats.csv
python3 tools/trace_timeline.py /tmp/prof_kt/bench_kernel_trace.csv 60 > $OUT/timeline.txt
# config 2 (full-A PCG): kernel stats + the same two PMC passes, appended to the same CSVs
rm -rf /tmp/prof_kt2 /tmp/prof_f2 /tmp/prof_w2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt2 -o fa -- python3 bench.py --workload fullA --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_fullA_under_rocprof.json 2> /dev/null
cp /tmp/prof_kt2/fa_kernel_stats.csv $OUT/kernel_stats_fullA.csv
bash tools/pmc_pass.sh $TAG
grep "mi::" $OUT/kernel_stats.csv | cut -c1-160
