# per-kernel breakdown of one case for several library builds: breakdown_ab.sh "rows cols density" lib...
cd $GRAFT_REPO_ROOT
ARGS="$1"; shift
for v in "$@"; do
  ( [ "$v" != "default" ] && export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/$v
    echo "== $v"; timeout -k 10 200 python scripts/kernel_breakdown.py $ARGS 2>&1 | grep -v amdgpu.ids )
done
