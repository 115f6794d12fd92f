#!/bin/bash
# PMC passes (separate runs per counter set: SQ 8 slots, TCC 4; FETCH_SIZE / WRITE_SIZE alone) over one bench step.
#   bash scripts/pmc_run.sh TAG [bench args...]      ->  gpurun_out/pmc_TAG/, summary gpurun_out/TAG_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
mkdir -p gpurun_out/pmc_$TAG
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_$TAG/p$i -- python3 bench.py --steps 1 --warmup 0 --no-cpu "$@" > gpurun_out/pmc_$TAG/p$i.log 2>&1 || echo "pmc pass $i ($set) failed"
done
TAG=$TAG python3 - "$@" <<'PY' > gpurun_out/${TAG}_pmc.txt
import csv, glob, os, collections, sys
tag = os.environ["TAG"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "smm" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
print("# rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 1 --warmup 0 --no-cpu " + " ".join(sys.argv[1:]) + "   (scripts/pmc_run.sh; one pass per counter set)")
print("# per launch.  FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them (gfx950: x2 on FETCH_SIZE for the byte count, MI355X_MICROARCH.md)")
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:36s} {v / max(1, len(n[k][c])):.6g}")
PY
grep -A22 "dense_slab\|smm_numeric<" gpurun_out/${TAG}_pmc.txt | head -70
