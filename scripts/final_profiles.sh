#!/bin/bash
# Per-round evidence: kernel stats (rocprofv3 --kernel-trace --stats), PMC passes (scripts/pmc_run.sh), bench lines of every
# BASELINE config.   bash scripts/final_profiles.sh r2_k   ->   gpurun_out/r2_k_*
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/${TAG}_kernel_stats.csv
bash scripts/pmc_run.sh $TAG > /dev/null 2>&1
python3 bench.py > gpurun_out/${TAG}_bench_c1.json 2> gpurun_out/${TAG}_bench.err
python3 bench.py --exact --steps 5 > gpurun_out/${TAG}_bench_c1_exact.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --config c2 --steps 5 > gpurun_out/${TAG}_bench_c2.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --config c2 --steps 5 --exact > gpurun_out/${TAG}_bench_c2_exact.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --config c3 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_c3.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --config c3 --steps 3 --warmup 1 --exact > gpurun_out/${TAG}_bench_c3_exact.json 2>> gpurun_out/${TAG}_bench.err
python3 bench.py --config c4 --steps 3 --warmup 1 --no-cpu > gpurun_out/${TAG}_bench_c4.json 2>> gpurun_out/${TAG}_bench.err
bash scripts/pmc_run.sh ${TAG}_c3 --config c3 > /dev/null 2>&1
# only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf gpurun_out/pmc_$TAG gpurun_out/pmc_${TAG}_c3 gpurun_out/prof_$TAG
cut -c1-160 gpurun_out/${TAG}_kernel_stats.csv | head -8
for f in gpurun_out/${TAG}_bench_c*.json; do echo $f; cut -c1-330 $f; done
