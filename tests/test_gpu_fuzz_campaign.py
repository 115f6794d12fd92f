"""Randomized campaign on shapes the fixed tests do not enumerate: random row-length distributions
(Poisson, Pareto, a few very long rows), empty rows, column counts from 1 to 2e6 (every marker kind of
the symbolic phase: LDS hash sets, LDS bitmap, bitmap in global memory), symmetric on square cases;
both modes against the CPU oracle.  SMM_FUZZ_CASES (default 60) sets the number of cases,
SMM_FUZZ_SEED the stream: a long one-off run is `SMM_FUZZ_CASES=2000 pytest tests/test_gpu_fuzz_campaign.py`."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, assert_csr_equal

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("SMM_FUZZ_CASES", "60"))
SEED = int(os.environ.get("SMM_FUZZ_SEED", "0"))


def _rand_rows(r, m, n, mean, kind):
    if kind == 0:
        lens = r.poisson(mean, size=m)
    elif kind == 1:
        lens = (r.pareto(1.3, size=m) * mean * 0.4).astype(np.int64)
    else:
        lens = np.where(r.random(m) < 0.05, r.integers(0, min(n, int(50 * mean) + 1), size=m), r.poisson(mean * 0.3, size=m))
    lens = np.minimum(lens, n).astype(np.int64)
    lens[r.random(m) < 0.1] = 0
    indptr = np.zeros(m + 1, np.int64)
    indptr[1:] = np.cumsum(lens)
    if indptr[-1] > 3_000_000:
        return None
    cols = np.empty(int(indptr[-1]), np.int64)
    for i in np.flatnonzero(lens):
        k = int(lens[i])
        if k * 4 > n:
            c = r.choice(n, size=k, replace=False)
        else:
            c = np.unique(r.integers(0, n, size=k))
            while c.size < k:
                c = np.unique(np.concatenate([c, r.integers(0, n, size=k - c.size)]))
        cols[indptr[i]:indptr[i + 1]] = np.sort(c[:k])
    return sp.csr_matrix((r.uniform(-1, 1, size=cols.size), cols.astype(np.int32), indptr.astype(np.int32)), shape=(m, n))


@pytest.mark.parametrize("it", range(NCASES))
def test_campaign(ctx, oracle, it):
    r = np.random.default_rng(SEED * 100000 + it)
    m, k = int(r.integers(1, 3000)), int(r.integers(1, 3000))
    n = m if it % 4 == 0 else int(10 ** r.uniform(0, 6.3))
    ma, mb = 10 ** r.uniform(0, 1.8), 10 ** r.uniform(0, 2.2)
    A, B = _rand_rows(r, m, k, ma, int(r.integers(0, 3))), _rand_rows(r, k, n, mb, int(r.integers(0, 3)))
    if A is None or B is None or np.diff(B.indptr)[A.indices].sum() > 3e7:
        pytest.skip("case larger than the campaign's budget")
    a, b = ctx.csr_from_scipy(A), ctx.csr_from_scipy(B)
    try:
        for sym in ((False, True) if m == n else (False,)):
            want = oracle.sparse(arrays(A), arrays(B), n, symmetric=sym)
            assert_csr_equal(ctx.spgemm_host(a, b, symmetric=sym, exact=True), want, values="bits")
            # default mode: same pattern and order, values to the rounding of a different summation order.  The
            # entries are signed here, so sums cancel and 1e-10 of the RESULT is not a meaningful bar; the error
            # is held to 1e-12 of the sum of the products' magnitudes (the oracle on |A|, |B|: same pattern).
            gp, gi, gv = ctx.spgemm_host(a, b, symmetric=sym)
            assert np.array_equal(np.asarray(gp, np.int64), np.asarray(want[0], np.int64)) and np.array_equal(gi, want[1])
            mag = oracle.sparse(arrays(abs(A)), arrays(abs(B)), n, symmetric=sym)[2]
            assert np.all(np.abs(gv - want[2]) <= 1e-12 * mag), f"max {np.max(np.abs(gv - want[2]) / np.maximum(mag, 1e-300)):.3e} of the magnitude sum"
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("it", range(max(NCASES // 3, 1)))
def test_campaign_triple(ctx, oracle, it):
    """H Q H^T on random shapes: n across the 16-row blocks and 1024-row k-groups of stage 2, K across its
    1024-column chunks; SMM_EXACT must reproduce the CPU loop bit for bit, upper triangle and full matrix."""
    r = np.random.default_rng(SEED * 100000 + 50000 + it)
    n, K = int(r.integers(1, 1400)), int(r.integers(1, 5000))
    H = _rand_rows(r, n, K, 10 ** r.uniform(0, 1.5), int(r.integers(0, 3)))
    S = sp.random(K, K, density=min(1.0, 10 ** r.uniform(0, 1.3) / K), format="csr", random_state=r)
    Q = (S + S.T).tocsr()
    Q.sort_indices()
    if H is None or n * n * (H.nnz / max(n, 1)) > 6e8:
        pytest.skip("case larger than the campaign's budget")
    h, q = ctx.csr_from_scipy(H), ctx.csr_from_scipy(Q)
    try:
        for full in (0, 1):
            want = oracle.triple(arrays(H), arrays(Q), K, full=full)
            got = ctx.triple_host(h, q, full=bool(full), exact=True)
            assert np.array_equal(got.view(np.int64), want.view(np.int64))
    finally:
        h.close(); q.close()
