# bench variants without the test-suite: each argument is one set of bench.py flags
cd $GRAFT_REPO_ROOT
i=0
for extra in "$@"; do i=$((i+1))
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu $extra > gpurun_out/bv_$i.log 2>&1 || { echo FAIL "$extra"; tail -3 gpurun_out/bv_$i.log; }
  python3 - "$i" "$extra" <<'PY'
import json, sys
for l in open(f"gpurun_out/bv_{sys.argv[1]}.log"):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[2] or "(default)", "| ms/step", round(d['ms_per_step'],1), "numeric", round(d['roofline']['kernel_ms'],1), "symbolic", round(d['roofline']['symbolic_kernel_ms'],1))
PY
done
