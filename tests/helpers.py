"""Shared input generators and comparison helpers of the test-suite."""
import numpy as np
import scipy.sparse as sp


def rand_csr(m, n, density, seed, dtype=np.float64):
    """BASELINE's generator: uniform[0,1) values, sorted unique indices."""
    return sp.random(m, n, density=density, format="csr", random_state=np.random.default_rng(seed), dtype=dtype)


def wide_csr(m, n, per_row, seed):
    """~per_row distinct random columns per row for very wide matrices (scipy.sparse.random permutes all
    m*n positions, which is hopeless at 1e6 columns)."""
    rng = np.random.default_rng(seed)
    cols = np.sort(rng.integers(0, n, size=(m, per_row)), axis=1)
    keep = np.ones_like(cols, dtype=bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    indptr = np.zeros(m + 1, np.int32)
    indptr[1:] = np.cumsum(keep.sum(axis=1))
    return sp.csr_matrix((rng.standard_normal(int(indptr[-1])), cols[keep].astype(np.int32), indptr), shape=(m, n))


def arrays(m):
    return (np.ascontiguousarray(m.indptr, dtype=np.int32), np.ascontiguousarray(m.indices, dtype=np.int32),
            np.ascontiguousarray(m.data, dtype=np.float64))


def signed(m, seed):
    """Same pattern, values in [-1,1): exercises cancellation and signed zeros."""
    r = np.random.default_rng(seed)
    out = m.copy()
    out.data = r.uniform(-1.0, 1.0, size=out.nnz)
    return out


def shuffle_rows(m, seed):
    """Same matrix with the entries of every row in random order (non-canonical CSR)."""
    r = np.random.default_rng(seed)
    m = m.tocsr().copy()
    for i in range(m.shape[0]):
        s, e = m.indptr[i], m.indptr[i + 1]
        p = r.permutation(e - s)
        m.indices[s:e] = m.indices[s:e][p]
        m.data[s:e] = m.data[s:e][p]
    m.has_sorted_indices = False
    return m


def assert_csr_equal(got, want, values="bits", rtol=1e-10):
    """indptr and indices bit-exact (reference first-touch order); values bit-exact or within
    the north star's tolerance (1e-10 relative)."""
    gp, gi, gv = got
    wp, wi, wv = want
    assert np.array_equal(np.asarray(gp, dtype=np.int64), np.asarray(wp, dtype=np.int64)), "indptr differs"
    assert np.array_equal(gi, wi), "indices differ (first-touch order)"
    if values == "bits":
        assert np.array_equal(gv.view(np.int64), wv.view(np.int64)), \
            f"values differ bitwise (max rel {rel_err(gv, wv):.3e})"
    else:
        assert rel_err(gv, wv) <= rtol, f"values differ: max rel {rel_err(gv, wv):.3e}"


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    if a.size == 0:
        return 0.0
    d = np.abs(a - b)
    s = np.maximum(np.abs(b), np.finfo(np.float64).tiny)
    ok = d == 0
    return float(np.max(np.where(ok, 0.0, d / s)))
