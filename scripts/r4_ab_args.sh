#!/bin/bash
# round 4: interleaved A/B of bench arguments: bash scripts/r4_ab_args.sh <tag> "<common args>" "name=<extra args>" ...
cd $GRAFT_REPO_ROOT
tag=$1; args=$2; shift 2
for rep in 1 2 3; do
  for arm in "$@"; do
    name=${arm%%=*}; extra=${arm#*=}
    timeout -k 10 200 python bench.py $args $extra --steps 5 --warmup 1 --no-cpu --no-extra > gpurun_out/${tag}_${name}_${rep}.json 2> gpurun_out/${tag}_${name}_${rep}.err || echo FAIL $name
    python3 - "$tag" "$name" "$rep" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}_{sys.argv[3]}.json"))
    print(sys.argv[2], "ms/step", round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['roofline']['kernels_ms'].items()})
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
  done
done
