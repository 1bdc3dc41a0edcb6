#!/bin/bash
# Runs ON THE GPU BOX: A/B of the CSR row-block size (MI355_SPMV_TILE, a compile-time constant) with libraries prebuilt into
# build_variants/ (make OUT=... CXXFLAGS=...-DMI355_SPMV_TILE=N): matrix-free S-apply (A_II, 7 M non-zeros) and config 2.
OUT=${1:-gpurun_out/spmv_tile}
mkdir -p $OUT
cp julia-phd-krylov-spdes_amd/libmi355schur.so /tmp/lib_default.so
for v in default t2048 t4096; do
  if [ $v = default ]; then cp /tmp/lib_default.so julia-phd-krylov-spdes_amd/libmi355schur.so; else cp build_variants/lib_$v.so julia-phd-krylov-spdes_amd/libmi355schur.so; fi
  echo "== $v" >> $OUT/probe.log
  timeout -k 10 300 python tools/interior_probe.py 2>/dev/null | head -2 >> $OUT/probe.log
  timeout -k 10 200 python bench.py --workload fullA --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config 2:', d['value'], d['roofline'])" >> $OUT/probe.log
done
cp /tmp/lib_default.so julia-phd-krylov-spdes_amd/libmi355schur.so
cat $OUT/probe.log
