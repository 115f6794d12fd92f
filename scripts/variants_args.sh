# A/B of library builds on the bench with extra bench flags: variants_args.sh "<bench flags>" lib...
cd $GRAFT_REPO_ROOT
ARGS="$1"; shift
for v in "$@"; do
  export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/$v
  [ "$v" = "default" ] && unset SMM_LIB_PATH
  timeout -k 10 200 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu $ARGS > gpurun_out/va_$v.log 2>&1 || { echo FAIL $v; tail -3 gpurun_out/va_$v.log; }
  python3 - "$v" <<'PY'
import json, sys
for l in open(f"gpurun_out/va_{sys.argv[1]}.log"):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], "ms/step", round(d['ms_per_step'],1), "numeric", round(d['roofline']['kernel_ms'],1), "symbolic", round(d['roofline']['symbolic_kernel_ms'],1))
PY
done
