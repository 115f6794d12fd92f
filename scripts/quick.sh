cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/tq.log 2>&1; echo "tests EXIT $?" ; tail -3 gpurun_out/tq.log
run() { tag=$1; shift
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/q_$tag.log 2>&1 || { echo FAIL $tag; tail -5 gpurun_out/q_$tag.log; }
  python3 - "$tag" <<'PY'
import json, sys
for l in open(f"gpurun_out/q_{sys.argv[1]}.log"):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], "ms/step", round(d['ms_per_step'],1), "numeric", round(d['roofline']['kernel_ms'],1), "symbolic", round(d['roofline']['symbolic_kernel_ms'],1))
PY
}
run default
run exact --exact
for extra in "$@"; do run "x$(echo $extra | tr ' -' '__')" $extra; done
