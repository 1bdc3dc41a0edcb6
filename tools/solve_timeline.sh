#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of a short headline run; prints the timeline of one solve and the host turnaround
# between solves (tools/trace_timeline.py). $1 = output directory under gpurun_out/.
set -u
OUT=${1:-gpurun_out/timeline}
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python3 bench.py --steps 60 --warmup 10 --kernel-reps 0 --no-cpu-baseline --no-secondary > $OUT/bench.json 2> $OUT/bench.err
python3 tools/trace_timeline.py /tmp/prof_tl/tl_kernel_trace.csv 4 > $OUT/timeline.txt
tail -45 $OUT/timeline.txt
