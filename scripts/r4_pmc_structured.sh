#!/bin/bash
# round 4: LDS counters of the symbolic walk on an operand with dense runs (block diagonal family of structured_sweep.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_struct && mkdir -p gpurun_out/pmc_struct
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_struct/p1 -- python3 scripts/structured_sweep.py "${1:-block}" > gpurun_out/pmc_struct/p1.log 2>&1 || echo "pmc pass failed"
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("gpurun_out/pmc_struct/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "smm_symbolic" in k or "smm_numeric" in k or "plan_check" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:28s} {v / max(1, len(n[k][c])):.6g}")
PY
