"""Aggregate the rocprofv3 --pmc passes of scripts/final_profiles.sh (gpurun_out/pmc_<TAG>/p*/) into
profiles/<TAG>_final_pmc.txt and regenerate profiles/traffic_r1.json (HBM-side bytes of smm_numeric per
launch = 2 x FETCH_SIZE + WRITE_SIZE; the x2 is the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md,
calibrated on smm_symbolic whose index reads are known: 4 B x products).
usage: python scripts/pmc_summary.py TAG [products]"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
products = float(sys.argv[2]) if len(sys.argv) > 2 else 1.25008e10      # configs[1]: sum of nnz(B[a_ij,:])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(f"gpurun_out/pmc_{tag}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "smm" in k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
lines = ["# rocprofv3 --kernel-trace --pmc <set> -- python3 bench.py --steps 1 --warmup 0 --no-cpu   (scripts/final_profiles.sh; one pass per counter set)",
         "# per launch (one step).  FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them."]
per = {}
for k in sorted(agg):
    lines.append(k)
    for c, v in sorted(agg[k].items()):
        n = max(1, len(launches[k][c]))
        per[(k, c)] = v / n
        lines.append(f"   {c:36s} {v / n:.6g}")
open(f"profiles/{tag}_final_pmc.txt", "w").write("\n".join(lines) + "\n")
num = next(k for k in agg if "smm_numeric<0" in k)
symk = next(k for k in agg if "smm_symbolic" in k)
fetch, write = per[(num, "FETCH_SIZE")], per[(num, "WRITE_SIZE")]
calib = products * 4 / (per[(symk, "FETCH_SIZE")] * 1024)
traffic = (2.0 * fetch + write) * 1024
json.dump({"workload": "50000x50000 x 50000x50000 uniform random CSR d=0.01 -> CSR", "kernel": "smm_numeric", "mode": "default",
           "fetch_size_kib": fetch, "write_size_kib": write, "fetch_correction": 2.0,
           "calibration_on_smm_symbolic": calib, "traffic_bytes_per_launch": traffic,
           "source": f"profiles/{tag}_final_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; scripts/pmc_summary.py)"},
          open("profiles/traffic_r1.json", "w"), indent=1)
print(open("profiles/traffic_r1.json").read())
