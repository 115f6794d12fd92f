# configs[3] (triple product) evidence: one-off timings (default / exact) + rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
python3 scripts/run_c3c4.py c4 1.0 2>&1 | grep '^{' > gpurun_out/${TAG}_c4.jsonl
python3 scripts/run_c3c4.py c4 1.0 --exact 2>&1 | grep '^{' >> gpurun_out/${TAG}_c4.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 scripts/run_c3c4.py c4 1.0 > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/${TAG}_c4_kernel_stats.csv
cat gpurun_out/${TAG}_c4.jsonl; cut -c1-160 gpurun_out/${TAG}_c4_kernel_stats.csv
