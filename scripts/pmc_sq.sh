# instruction-issue counters of one bench step (which pipe a kernel is bound by): two SQ passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-sq}
mkdir -p gpurun_out/pmc_$TAG
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu $BENCH_ARGS > gpurun_out/pmc_$TAG/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_$TAG/p$i.log; }
done
TAG=$TAG python3 - <<'PY' | tee gpurun_out/pmc_${TAG}_summary.txt
import csv, glob, os, collections
tag = os.environ.get("TAG", "x")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if "smm_symbolic" in k or "smm_numeric" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:36s} {v:.4g}")
PY
