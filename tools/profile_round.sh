#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats of the default bench command and of config 2, the
# deflated / recycling loops (tools/profile_defl.sh) and the two PMC passes for HBM traffic (tools/pmc_pass.sh).
# Summaries land in gpurun_out/prof_$1/ (copied to profiles/ by hand).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o bench -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp /tmp/prof_kt/bench_kernel_stats.csv $OUT/kernel_stats.csv
grep "mi::" $OUT/kernel_stats.csv > $OUT/kernel_stats_mi_only.csv
python3 tools/trace_timeline.py /tmp/prof_kt/bench_kernel_trace.csv 60 > $OUT/timeline.txt
rm -rf /tmp/prof_kt2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt2 -o fa -- python3 bench.py --workload fullA --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_fullA_under_rocprof.json 2> /dev/null
cp /tmp/prof_kt2/fa_kernel_stats.csv $OUT/kernel_stats_fullA.csv
bash tools/profile_defl.sh $TAG > $OUT/profile_defl.log 2>&1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --many-subdomains > $OUT/bench_many_subdomains.json 2> /dev/null
bash tools/pmc_pass.sh $TAG
python3 tools/kstats.py $OUT/kernel_stats.csv > $OUT/kstats.txt; head -12 $OUT/kstats.txt
