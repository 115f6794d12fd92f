# evidence for the sparse / very wide regime (hash-set marker, hash numeric kernels): kernel stats of one bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu --rows 1000000 --cols 1000000 --density 0.00001 > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/${TAG}_sparse_1e6_kernel_stats.csv
grep '^{' gpurun_out/prof_$TAG.log > gpurun_out/${TAG}_sparse_1e6_bench.json
cut -c1-150 gpurun_out/${TAG}_sparse_1e6_kernel_stats.csv; cut -c1-300 gpurun_out/${TAG}_sparse_1e6_bench.json
rm -rf gpurun_out/prof_$TAG
