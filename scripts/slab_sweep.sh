#!/bin/bash
# Sweep of the row-block x column-slab kernels on BASELINE configs[2] (dense) and configs[1] (sparse):
# slab width and rows per wave.  Usage (GPU box): bash scripts/slab_sweep.sh <out-prefix>
out=${1:-gpurun_out/slab}
for cfg in "1,0,4" "2,0,4" "2,0,2" "2,400,4" "2,800,2" "2,1100,2"; do
  python bench.py --config c2 --steps 5 --warmup 1 --slab $cfg > ${out}_c2_${cfg//,/_}.json 2>> ${out}.err || exit 1
done
for cfg in "1,0,4" "2,0,4" "2,0,2" "2,800,2"; do
  python bench.py --config c1 --steps 5 --warmup 1 --no-cpu --slab $cfg > ${out}_c1_${cfg//,/_}.json 2>> ${out}.err || exit 1
done
