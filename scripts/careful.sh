# after a risky change to smm_symbolic: the symbolic phase alone against scipy's structural product, under a short
# timeout, for each library build given (default = the in-tree one); nothing downstream of a wrong list runs
cd $GRAFT_REPO_ROOT
for v in "${@:-default}"; do
  ( [ "$v" != "default" ] && export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/$v
    echo "== $v"; timeout -k 10 90 python -u scripts/sym_check.py > gpurun_out/sym_$v.log 2>&1; echo "EXIT $?"; tail -18 gpurun_out/sym_$v.log )
done
