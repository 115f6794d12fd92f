#!/bin/bash
# round 4: tile width / waves sweep of the persistent smm_numeric at configs[1]
cd $GRAFT_REPO_ROOT
for arg in "--lds-cols 16667" "--lds-cols 12500" "--lds-cols 10000" "--lds-cols 8334" "--lds-cols 8334 --waves 8" "--lds-cols 10000 --waves 8" "--lds-cols 16667 --waves 8"; do
  tag=$(echo "$arg" | tr -d ' -')
  timeout -k 10 200 python bench.py --config c1 $arg --steps 5 --warmup 1 --no-cpu --no-extra > gpurun_out/sweep_$tag.json 2> gpurun_out/sweep_$tag.err || echo FAIL $arg
  python3 - "$tag" "$arg" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/sweep_{sys.argv[1]}.json"))
    print(sys.argv[2], "ms/step", round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['roofline']['kernels_ms'].items()})
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
done
