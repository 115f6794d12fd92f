# rocprofv3 kernel-trace stats of the default bench command -> gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu $BENCH_ARGS > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/prof_$TAG.kernel_stats.csv
cat gpurun_out/prof_$TAG.kernel_stats.csv | cut -c1-160
grep '^{' gpurun_out/prof_$TAG.log | cut -c1-300
