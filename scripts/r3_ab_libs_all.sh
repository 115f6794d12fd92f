#!/bin/bash
# round 3: A/B of library builds over all four BASELINE configs (default bench with extra_configs): name=lib.so ...
cd $GRAFT_REPO_ROOT
tag=$1; shift
for rep in 1 2; do
  for arm in "$@"; do
    name=${arm%%=*}; lib=${arm#*=}
    SMM_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/${tag}_${name}_${rep}.json 2> gpurun_out/${tag}_${name}_${rep}.err || echo FAIL $name
    python3 - "$tag" "$name" "$rep" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}_{sys.argv[3]}.json"))
    r = lambda k: {a: round(b, 2) for a, b in k.items()}
    print(sys.argv[2], "c1", round(d['ms_per_step'], 2), r(d['roofline']['kernels_ms']))
    for k, x in d["extra_configs"].items():
        print("    ", k, round(x["ms_per_step"], 2), r(x["roofline"]["kernels_ms"]))
except Exception as e:
    print(sys.argv[2], "no line", e)
PY
  done
done
