set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "8192 4" "8192 8" "16384 4" "16384 8" "4096 4" "4096 8"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --lds-cols $1 --waves $2 > gpurun_out/sw_o_$1_$2.log 2>&1 || echo FAIL $cfg
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --lds-cols $1 --waves $2 --unordered > gpurun_out/sw_u_$1_$2.log 2>&1 || echo FAIL u $cfg
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1a -- python3 bench.py --steps 3 --warmup 1 --no-cpu > gpurun_out/prof_r1a.log 2>&1
echo done
