"""Round 4: the ring kernel of triple-product stage 2 (csrc/smm_ring.hpp) -- the tile of T as a ring of column pieces,
the 16 waves of a workgroup synchronised by progress words in LDS instead of barriers, the order of every lane's
entries scheduled once per H by a lock-step simulation of the k-group.  It is an ALTERNATIVE to the default chunk
kernel (slower on MI355X: DESIGN.md), selected with smm_ctx_tune_stage2 / SMM_S2_RING=1, and must give the reference's
result (src/sparse_sparse_dense.cpp:201-216) exactly as the default does: bit for bit under SMM_EXACT."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import arrays, rand_csr, rel_err, signed

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
RTOL = 1e-10


@pytest.fixture
def ring(ctx):
    ctx.tune_stage2(True)
    yield ctx
    ctx.tune_stage2(False)


def _q(k, d, seed):
    S = sp.random(k, k, density=d / 2, format="csr", random_state=np.random.default_rng(seed))
    return (S + S.T).tocsr()


@pytest.mark.parametrize("n,k,dh,dq", [(1, 1, 1.0, 1.0), (60, 90, 0.1, 0.1), (500, 500, 0.3, 0.3), (300, 9000, 0.02, 0.004),
                                       (257, 5000, 0.05, 0.01), (1500, 2500, 0.02, 0.004), (2100, 700, 0.01, 0.01)])
@pytest.mark.parametrize("full", [0, 1])
@pytest.mark.parametrize("exact", [False, True])
def test_ring_kernel_matches_oracle(ring, oracle, n, k, dh, dq, full, exact):
    H = rand_csr(n, k, dh, 3)
    Q = _q(k, dq, 4)
    if exact:
        H, Q = signed(H, 5), signed(Q, 6)
    want = oracle.triple(arrays(H), arrays(Q), k, full)
    h, q = ring.csr_from_scipy(H), ring.csr_from_scipy(Q)
    try:
        ring.timing(True); ring.timing_reset()
        got = ring.triple_host(h, q, full=bool(full), exact=exact)
        if H.nnz and Q.nnz:                                    # (a zero operand returns zeros without any kernel)
            assert ring.kernel_time("smm_ring_build")[1] >= 2      # the schedule was built (count + fill passes): the ring path ran
        ring.timing(False)
        # new values on the same pattern: the scheduled streams are re-filled in place
        H2 = signed(H, 7)
        h.update_values(H2.data)
        got2 = ring.triple_host(h, q, full=bool(full), exact=exact)
        want2 = oracle.triple(arrays(H2), arrays(Q), k, full)
    finally:
        h.close(); q.close()
    for g, w in ((got, want), (got2, want2)):
        if exact:
            assert np.array_equal(g.view(np.int64), w.view(np.int64)), f"max rel {rel_err(g, w):.3e}"
        else:
            mag = oracle.triple(arrays(abs(H) if g is got else abs(H2)), arrays(abs(Q)), k, full)
            assert np.all(np.abs(g - w) <= 1e-12 * np.maximum(mag, 1e-300))


def test_ring_kernel_row_ranges_and_many_k_groups(ring, oracle):
    H = rand_csr(2600, 1800, 0.01, 11)                    # three k-groups of 1024 rows, four column pieces
    Q = _q(1800, 0.01, 12)
    want = oracle.triple(arrays(H), arrays(Q), 1800, 0)
    h, q = ring.csr_from_scipy(H), ring.csr_from_scipy(Q)
    try:
        whole = ring.triple_host(h, q, exact=True)
        top = ring.triple_host(h, q, row_begin=0, row_end=1030, exact=True)
        bot = ring.triple_host(h, q, row_begin=1030, row_end=2600, exact=True)
    finally:
        h.close(); q.close()
    assert np.array_equal(whole, want) and np.array_equal(np.vstack([top, bot]), want)
    assert not whole[np.tril_indices(2600, -1)].any()     # the reference's calloc'd lower triangle


def test_ring_kernel_at_baseline_shape_scaled(ring, oracle):
    """BASELINE configs[3] at 1/4 scale in both dimensions (H 5000 x 20000, d = 0.02: 400 entries per row over 40
    pieces, five k-groups): sampled rows against the oracle, and U x = H (Q (H^T x)) on the whole upper triangle."""
    H = rand_csr(5000, 20000, 0.02, 21)
    Q = _q(20000, 0.005, 22)
    h, q = ring.csr_from_scipy(H), ring.csr_from_scipy(Q)
    try:
        got = ring.triple_host(h, q, exact=True)
    finally:
        h.close(); q.close()
    a, b = arrays(H), arrays(Q)
    for r in (0, 17, 1023, 1024, 2500, 4999):
        want = oracle.triple(a, b, 20000, 0, r, r + 1)[r]
        assert np.array_equal(got[r], want)
    x = np.random.default_rng(23).uniform(0.5, 1.5, 5000)
    full = got + got.T - np.diag(np.diag(got))
    assert np.allclose(full @ x, H @ (Q @ (H.T @ x)), rtol=1e-9, atol=0)
