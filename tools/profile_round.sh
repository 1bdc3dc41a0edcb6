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
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o pmc -- python3 bench.py --steps 10 --warmup 2 --kernel-reps 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o pmc -- python3 bench.py --steps 10 --warmup 2 --kernel-reps 20 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
ls /tmp/prof_f /tmp/prof_w
python3 tools/hbm_traffic.py /tmp/prof_f/pmc_counter_collection.csv /tmp/prof_w/pmc_counter_collection.csv $OUT/hbm_traffic.json > $OUT/hbm_traffic.log 2>&1
tail -30 $OUT/hbm_traffic.log
grep "mi::" $OUT/kernel_stats.csv | cut -c1-160
