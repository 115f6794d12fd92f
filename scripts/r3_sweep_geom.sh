# round 3: tile geometry of the shared-tile walk on configs[1] (lds_cols x waves), 5 steps each
cd $GRAFT_REPO_ROOT
for cfg in "20000 16" "12600 16" "10000 16" "8400 16" "8400 8" "12600 8" "6300 8"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu --no-extra --lds-cols $1 --waves $2 > gpurun_out/r3_geom_$1_$2.json 2> gpurun_out/r3_geom_$1_$2.err || echo FAIL $cfg
  python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/r3_geom_{sys.argv[1]}_{sys.argv[2]}.json"))
    print(sys.argv[1], sys.argv[2], "ms/step", round(d['ms_per_step'], 2), "kernels", {k: round(v, 2) for k, v in d['roofline']['kernels_ms'].items()})
except Exception as e:
    print(sys.argv[1], sys.argv[2], "no line", e)
PY
done
