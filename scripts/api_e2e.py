"""End-to-end timings through the drop-in Python API (scipy in, scipy/numpy out, PCIe and result
construction included).

    python scripts/api_e2e.py            # the orientation configs of BASELINE.md section 2
    python scripts/api_e2e.py big        # BASELINE configs[1] and [2]: 50 000 x 50 000, d = 0.01, -> CSR / dense

`big` builds the operands on the device (synthetic.py) and hands them to the API as scipy matrices on
the host; every call is timed cold (operand cache cleared) and warm (both operands cached)."""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import sparse_matrix_mult_amd as pkg  # noqa: E402
from sparse_matrix_mult_amd import sparse_matrix_multiply  # noqa: E402


def timed(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


if len(sys.argv) > 1 and sys.argv[1] == "big":
    import torch
    from sparse_matrix_mult_amd.synthetic import gen_csr_device
    dev = torch.device("cuda", 0)
    n, d = 50000, 0.01
    mats = []
    for seed in (1, 2):
        ip, ix, dv = (t.cpu().numpy() for t in gen_csr_device(torch, n, n, d, seed, dev))
        mats.append(sp.csr_matrix((dv, ix, ip), shape=(n, n)))
    torch.cuda.empty_cache()
    A, B = mats
    # round 3: what the full-content operand key costs on this host (both operands, all three arrays each)
    from sparse_matrix_mult_amd.matrix_ops import _operand_key, cache_stats
    _, dt_key = timed(lambda: (_operand_key(A), _operand_key(B)))
    _, dt_key2 = timed(lambda: (_operand_key(A), _operand_key(B)))
    print(json.dumps({"operand_keys_both_operands_s": [round(dt_key, 4), round(dt_key2, 4)], "bytes_hashed": int(2 * (A.data.nbytes + A.indices.nbytes + A.indptr.nbytes))}), flush=True)
    # round 3: same patterns, new values (in place): the values travel, the cached plan's numeric phase is replayed
    pkg.clear_cache()
    res = {}
    C, res["first"] = timed(lambda: sparse_matrix_multiply(A, B))
    del C
    for label in ("new_values_1", "new_values_2"):
        A.data *= 1.0001; B.data *= 0.9999
        C, res[label] = timed(lambda: sparse_matrix_multiply(A, B))
        del C
    C, res["unchanged"] = timed(lambda: sparse_matrix_multiply(A, B))
    del C
    print(json.dumps({"config": "configs[1] through sparse_matrix_multiply(): first call / new values on the same patterns / unchanged",
                      "seconds": {k: round(v, 3) for k, v in res.items()}, "cache_stats": dict(cache_stats)}), flush=True)
    pkg.clear_cache()
    for fmt in ("sparse", "dense"):
        res = {}
        for label in ("cold", "warm", "warm2"):
            if label == "cold":
                pkg.clear_cache()
            C, dt = timed(lambda: sparse_matrix_multiply(A, B, output_format=fmt))
            res[label] = round(dt, 3)
            if fmt == "sparse":
                info = {"nnzC": int(C.nnz), "indices_dtype": str(C.indices.dtype), "indptr_dtype": str(C.indptr.dtype),
                        "result_GB": round((C.data.nbytes + C.indices.nbytes + C.indptr.nbytes) / 1e9, 2)}
            else:
                info = {"result_GB": round(C.nbytes / 1e9, 2)}
            del C
        print(json.dumps({"config": f"{n}x{n} d={d} -> {fmt} through sparse_matrix_multiply()", "seconds": res, **info}), flush=True)
    sys.exit(0)

for n, d in ((1000, 0.05), (5000, 0.01), (10000, 0.005)):
    A = sp.random(n, n, density=d, format="csr", random_state=np.random.default_rng(1))
    B = sp.random(n, n, density=d, format="csr", random_state=np.random.default_rng(2))
    res = {}
    for name, kw in (("sparse", dict(output_format="sparse")), ("sparse_sym", dict(output_format="sparse", symmetric=True)),
                     ("dense", dict(output_format="dense"))):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); C = sparse_matrix_multiply(A, B, **kw); best = min(best, time.perf_counter() - t0)
        res[name] = round(best * 1e3, 2)
    t0 = time.perf_counter(); S = A @ B; res["scipy"] = round((time.perf_counter() - t0) * 1e3, 2)
    print(json.dumps({"n": n, "d": d, "nnzC": int(S.nnz), "ms": res}))
