cd $GRAFT_REPO_ROOT
export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/libsmm_hip_stamps.so
for cfg in "5000 1" "10000 2" "16000 4" "20000 8"; do
  set -- $cfg
  echo "== lds_cols $1 waves $2"
  timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu --lds-cols $1 --waves $2 2>&1 | grep -E "SMM_STAMPS|ms_per_step" | cut -c1-400
done
