#!/bin/bash
# Runs ON THE GPU BOX: rocprofv3 kernel stats of the deflated / recycling loops at config 3 (tools/profile_defl.py),
# folded (default) and with the deflated fold switched off. Summaries -> gpurun_out/prof_$1/
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/profile_defl.py save /tmp/p.npz || exit 1
for mode in defpcg eigdefpcg eigpcg; do
  rm -rf /tmp/prof_$mode
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o d -- python3 tools/profile_defl.py $mode /tmp/p.npz > $OUT/$mode.out 2> $OUT/$mode.err
  cp /tmp/prof_$mode/d_kernel_stats.csv $OUT/kernel_stats_$mode.csv
  echo "== $mode"; python3 tools/kstats.py $OUT/kernel_stats_$mode.csv | head -12
done
rm -rf /tmp/prof_nofold
MI355_NO_FOLD_DEFL=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_nofold -o d -- python3 tools/profile_defl.py defpcg /tmp/p.npz > $OUT/defpcg_nofold.out 2> $OUT/defpcg_nofold.err
cp /tmp/prof_nofold/d_kernel_stats.csv $OUT/kernel_stats_defpcg_unfolded.csv
echo "== defpcg, deflated fold off"; python3 tools/kstats.py $OUT/kernel_stats_defpcg_unfolded.csv | head -12
