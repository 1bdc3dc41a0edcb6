#!/bin/bash
# The two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes) for configs 3 and 2.
# bench.py's set-up goes through the library's own device kernels (mi_schur_setup_run, mi_nn_pinv: no rocSOLVER call at config 3),
# so only this library's kernels are dispatched under counter collection.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_f /tmp/prof_w /tmp/prof_f2 /tmp/prof_w2
for c in FETCH_SIZE:f WRITE_SIZE:w; do
  ctr=${c%%:*}; d=${c#*:}
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d /tmp/prof_$d -o pmc -- python3 bench.py --steps 10 --warmup 2 --kernel-reps 2 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/pmc_$d.err || echo "pass $ctr (config 3) failed rc=$?"
  timeout -k 10 120 rocprofv3 --pmc $ctr --output-format csv -d /tmp/prof_${d}2 -o pmc -- python3 bench.py --workload fullA --steps 1 --warmup 1 --kernel-reps 0 --no-cpu-baseline --no-secondary > /dev/null 2> $OUT/pmc_${d}2.err || echo "pass $ctr (config 2) failed rc=$?"
  [ -f /tmp/prof_${d}2/pmc_counter_collection.csv ] && [ -f /tmp/prof_$d/pmc_counter_collection.csv ] && tail -n +2 /tmp/prof_${d}2/pmc_counter_collection.csv >> /tmp/prof_$d/pmc_counter_collection.csv
done
python3 tools/hbm_traffic.py /tmp/prof_f/pmc_counter_collection.csv /tmp/prof_w/pmc_counter_collection.csv $OUT/hbm_traffic.json > $OUT/hbm_traffic.log 2>&1
tail -40 $OUT/hbm_traffic.log
tail -3 $OUT/pmc_f.err
