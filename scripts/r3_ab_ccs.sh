# round 3: symbolic walk over the chunk-padded stream (SMM_SYM_CCS=1) against smm_symbolic (=0), configs[1], interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in 0 1; do
    SMM_SYM_CCS=$v timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu --no-extra > gpurun_out/r3_ccs_${v}_$rep.json 2> gpurun_out/r3_ccs_${v}_$rep.err || echo FAIL $v
    python3 - $v $rep <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r3_ccs_{sys.argv[1]}_{sys.argv[2]}.json"))
print("SMM_SYM_CCS=" + sys.argv[1], "ms/step", round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['roofline']['kernels_ms'].items()})
PY
  done
done
