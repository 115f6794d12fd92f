# timing of configs[3] with experimental builds of the library (sparse_matrix_mult_amd/lib/<name>) and env knobs
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=${v%%:*}; envs=${v#*:}; [ "$envs" = "$v" ] && envs=""
  ( [ "$lib" != "default" ] && export SMM_LIB_PATH=$GRAFT_REPO_ROOT/sparse_matrix_mult_amd/lib/$lib
    for e in $(echo $envs | tr ',' ' '); do export $e; done
    timeout -k 10 300 python scripts/run_c3c4.py c4 > gpurun_out/c4v.log 2>&1 || tail -5 gpurun_out/c4v.log
    echo "$v"; grep '^{' gpurun_out/c4v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms', round(d['ms'],1), 'stage1', round(d['stage1_ms'],1), 'stage2', round(d['stage2_ms'],1), 'checksum', d['checksum'])" )
done
