cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/h_$tag.log 2>&1 || { echo FAIL $tag; tail -3 gpurun_out/h_$tag.log; }
  python3 - "$tag" <<'PY'
import json, sys
for l in open(f"gpurun_out/h_{sys.argv[1]}.log"):
    if l.startswith('{'):
        d = json.loads(l); c = d['config']; print(sys.argv[1], "ms/step", round(d['ms_per_step'],3), "nnzC %.3g" % c['nnz_c_per_gpu'], "nnz/s %.3g" % d['value'])
PY
}
for cfg in "1000 1000 0.05" "20000 20000 0.001" "100000 100000 0.0001" "200000 200000 0.00005" "5000 5000 0.01" "10000 10000 0.005"; do
  set -- $cfg
  run "hash_$1" --rows $1 --cols $2 --density $3
  run "tile_$1" --rows $1 --cols $2 --density $3 --hash 0,0
done
