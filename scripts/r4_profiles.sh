#!/bin/bash
# Round-4 evidence: bash scripts/r4_profiles.sh r4_z  ->  gpurun_out/r4_z_*  (copied into profiles/ afterwards)
#   kernel stats of the default bench command (headline + extra_configs), PMC passes of configs[1], the bench lines
#   (default = what the driver runs, and --exact).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/${TAG}_kernel_stats.csv
echo "kernel stats done"; cut -c1-150 gpurun_out/${TAG}_kernel_stats.csv | head -12
bash scripts/pmc_run.sh $TAG --no-extra > /dev/null 2>&1
echo "pmc c1 done"
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --exact --steps 5 --no-extra > gpurun_out/${TAG}_bench_exact.json 2>> gpurun_out/${TAG}_bench.err
echo "bench lines done"
rm -rf gpurun_out/pmc_$TAG gpurun_out/prof_$TAG
for f in gpurun_out/${TAG}_bench*.json; do echo $f; cut -c1-300 $f; done
