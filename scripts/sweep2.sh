# geometry sweep of the pipelined numeric kernel on BASELINE configs[1]
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t3.log 2>&1; echo "tests EXIT $?" ; tail -3 gpurun_out/t3.log
for cfg in "5000 1" "4000 1" "10000 2" "8000 2" "20000 4" "16000 4" "20000 8" "10000 4"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --lds-cols $1 --waves $2 > gpurun_out/sw2_$1_$2.log 2>&1 || echo FAIL $cfg
  python3 - "$1" "$2" <<'PY'
import json, sys
for l in open(f"gpurun_out/sw2_{sys.argv[1]}_{sys.argv[2]}.log"):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], sys.argv[2], "ms/step", round(d['ms_per_step'],1), "numeric", round(d['roofline']['kernel_ms'],1), "symbolic", round(d['roofline']['symbolic_kernel_ms'],1))
PY
done
