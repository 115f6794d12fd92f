# PMC passes over one bench step (separate passes: SQ has 8 slots, TCC 4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="bench.py --steps 1 --warmup 0 --no-cpu $BENCH_ARGS"
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_IFETCH" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 $ARGS > gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, os, collections
tag = os.environ.get("TAG", "x")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "smm" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {v:.4g}")
PY
