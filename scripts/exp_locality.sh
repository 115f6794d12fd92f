cd $GRAFT_REPO_ROOT
for cfg in "50000 50000 0.01" "50000 20000 0.01" "50000 10000 0.02" "100000 25000 0.01"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu --rows $1 --cols $2 --density $3 > gpurun_out/loc_$1_$2.log 2>&1 || echo FAIL $cfg
  python3 - "$1" "$2" <<'PY'
import json, sys
for l in open(f"gpurun_out/loc_{sys.argv[1]}_{sys.argv[2]}.log"):
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        prod = c['nnz_a'] * (c['nnz_b'] / int(sys.argv[2]))
        print(sys.argv[1], sys.argv[2], "numeric ms", round(d['roofline']['kernel_ms'],2), "symbolic", round(d['roofline']['symbolic_kernel_ms'],2), "products %.3g" % prod, "gather GB/s %.0f" % (prod*12/d['roofline']['kernel_ms']/1e6), "B MB %.0f" % (c['nnz_b']*12/1e6), "nnzC %.3g" % c['nnz_c_per_gpu'])
PY
done
