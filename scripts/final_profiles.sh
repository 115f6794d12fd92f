# final per-round evidence: kernel stats, PMC traffic passes, bench lines (default and --exact), c3/c4 one-offs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_$TAG.log 2>&1
f=$(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1)
(head -1 $f; grep smm:: $f) > gpurun_out/${TAG}_kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu > gpurun_out/pmc_$TAG.p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 bench.py --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1; grep '^{' gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
python3 bench.py --steps 3 --warmup 1 --exact > gpurun_out/${TAG}_bench_exact.log 2>&1; grep '^{' gpurun_out/${TAG}_bench_exact.log > gpurun_out/${TAG}_bench_exact.json
python3 scripts/run_c3c4.py c3 2>&1 | grep '^{' > gpurun_out/${TAG}_c3c4.jsonl
python3 scripts/run_c3c4.py c3 --exact 2>&1 | grep '^{' >> gpurun_out/${TAG}_c3c4.jsonl
python3 scripts/run_c3c4.py c4 1.0 2>&1 | grep '^{' >> gpurun_out/${TAG}_c3c4.jsonl
python3 scripts/run_c3c4.py c4 1.0 --exact 2>&1 | grep '^{' >> gpurun_out/${TAG}_c3c4.jsonl
cat gpurun_out/${TAG}_kernel_stats.csv | cut -c1-150; cut -c1-400 gpurun_out/${TAG}_bench.json; cat gpurun_out/${TAG}_c3c4.jsonl
