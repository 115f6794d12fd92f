# triple-product iteration loop on the GPU box: parity tests that touch the triple path, then configs[3] timing
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "triple or legacy_dense" > gpurun_out/tt.log 2>&1; echo "tests EXIT $?"; tail -3 gpurun_out/tt.log
timeout -k 10 300 python scripts/run_c3c4.py c4 > gpurun_out/c4.log 2>&1 || tail -5 gpurun_out/c4.log
grep '^{' gpurun_out/c4.log
for v in "$@"; do
  export $v; timeout -k 10 300 python scripts/run_c3c4.py c4 > gpurun_out/c4_$v.log 2>&1 || tail -5 gpurun_out/c4_$v.log
  echo $v; grep '^{' gpurun_out/c4_$v.log; unset ${v%%=*}
done
