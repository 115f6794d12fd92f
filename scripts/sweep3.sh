cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1; echo "tests EXIT $?" ; tail -4 gpurun_out/t4.log
run() { # tag args...
  tag=$1; shift
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu "$@" > gpurun_out/sw3_$tag.log 2>&1 || echo FAIL $tag
  python3 - "$tag" <<'PY'
import json, sys
for l in open(f"gpurun_out/sw3_{sys.argv[1]}.log"):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], "ms/step", round(d['ms_per_step'],1), "numeric", round(d['roofline']['kernel_ms'],1), "symbolic", round(d['roofline']['symbolic_kernel_ms'],1))
PY
}
run sh_16384_16 --lds-cols 16384 --waves 16
run sh_16384_8 --lds-cols 16384 --waves 8
run sh_8192_8 --lds-cols 8192 --waves 8
run sh_8192_16 --lds-cols 8192 --waves 16
run sh_6400_8 --lds-cols 6400 --waves 8
run sh_20000_16 --lds-cols 20000 --waves 16
run ex_5000_1 --exact
export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/libsmm_hip_stamps.so
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu 2>&1 | grep SMM_STAMPS
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu --lds-cols 8192 --waves 8 2>&1 | grep SMM_STAMPS
