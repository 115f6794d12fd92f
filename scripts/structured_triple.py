"""Round 4: H Q H^T at BASELINE configs[3]'s shape (H 20 000 x 80 000, d = 0.02, uniform random) with covariance-like
Q instead of the uniform random one: block diagonal (200 blocks of 400 x 400, d = 0.5, symmetric), a band (half-width
200), and an exponential-decay band stored to half-width 50.  Times stage 1 / stage 2, checks C x = H (Q (H^T x)).

    python scripts/structured_triple.py      # on the GPU box; one JSON line per Q
"""
import json
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, ".")
import torch  # noqa: E402
from sparse_matrix_mult_amd import engine  # noqa: E402
from sparse_matrix_mult_amd.synthetic import gen_csr_device, gen_symmetric_csr_device  # noqa: E402


def band(n, half, rng, decay=None):
    offs = np.arange(0, half + 1)
    diags = [rng.random(n - o) * (np.exp(-o / decay) if decay else 1.0) for o in offs]
    U = sp.diags(diags, offs, shape=(n, n), format="csr")
    return (U + sp.triu(U, 1).T).tocsr()


def blocks(nb, bs, d, rng):
    out = []
    for _ in range(nb):
        S = sp.random(bs, bs, density=d / 2, format="csr", random_state=rng)
        out.append(S + S.T)
    return sp.block_diag(out, format="csr")


def main():
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    m, n = 20000, 80000
    ip, ix, dv = gen_csr_device(torch, m, n, 0.02, 1, dev)
    ctx = engine.Context(0)
    ctx.timing(True)
    h = ctx.csr_from_torch(m, n, ip, ix, dv)
    H = sp.csr_matrix((dv.cpu().numpy(), ix.cpu().numpy(), ip.cpu().numpy()), shape=(m, n))
    out = torch.empty((m, m), dtype=torch.float64, device=dev)
    qs = [("uniform random symmetric d=0.005 (configs[3])", None),
          ("block diagonal 200 x (400 x 400, d=0.5)", lambda: blocks(200, 400, 0.5, rng)),
          ("band, half-width 200", lambda: band(n, 200, rng)),
          ("exponential-decay band, half-width 50", lambda: band(n, 50, rng, decay=10.0))]
    for name, make in qs:
        if make is None:
            qp, qi, qv = gen_symmetric_csr_device(torch, n, 0.005, 2, dev)
            q = ctx.csr_from_torch(n, n, qp, qi, qv)
            Q = sp.csr_matrix((qv.cpu().numpy(), qi.cpu().numpy(), qp.cpu().numpy()), shape=(n, n))
        else:
            Q = make(); Q.sort_indices()
            q = ctx.csr_from_scipy(Q)
        best, kern = 1e30, {}
        for rep in range(3):
            ctx.timing_reset(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.triple_into(h, q, out.data_ptr())
            ctx.synchronize(); torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if dt < best:
                best = dt
                kern = {k: round(ctx.kernel_time(k)[0], 2) for k in ("smm_numeric_dense", "smm_triple_stage2", "smm_ell_fill", "smm_pack_fill")
                        if ctx.kernel_time(k)[1]}
        # upper triangle of H Q H^T: compare u^T C v on the triangle through a full symmetric completion
        x = rng.standard_normal(m)
        want = H @ (Q @ (H.T @ x))
        C = torch.triu(out)
        C = C + torch.triu(out, 1).T
        got = (C @ torch.from_numpy(x).to(dev)).cpu().numpy()
        scale = abs(H) @ (abs(Q) @ (abs(H).T @ np.abs(x)))
        err = float(np.max(np.abs(got - want) / scale))
        del C
        print(json.dumps({"Q": name, "nnzQ": int(Q.nnz), "ms": round(best * 1e3, 2), "kernels_ms": kern, "checksum_rel_err": err}), flush=True)
        assert err < 1e-10
        q.close()


if __name__ == "__main__":
    main()
