"""Synthetic operands of the BASELINE shapes, generated ON THE DEVICE (torch is plumbing here:
random numbers, sort, cumsum).  Used by bench.py, the full-size GPU tests and scripts/ -- never by
sparse_matrix_multiply().

The distribution is scipy.sparse.random's (uniform[0,1) float64 values, sorted unique
column indices): every cell is kept with probability `density`, so row lengths are binomial --
scipy's are hypergeometric with the same mean, indistinguishable at these sizes.  The random
STREAM differs from scipy's, which is why parity at these sizes is always checked on the very
arrays generated here (tests/test_gpu_baseline_configs.py)."""

__all__ = ["gen_csr_device", "gen_symmetric_csr_device"]


def gen_csr_device(torch, rows, cols, density, seed, device):
    """(indptr int32, indices int32, data float64) CUDA tensors of a rows x cols uniform random CSR."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    chunk = max(1, min(rows, (64 << 20) // max(cols, 1)))
    idx_parts, cnt_parts = [], []
    for r0 in range(0, rows, chunk):
        r1 = min(rows, r0 + chunk)
        mask = torch.rand((r1 - r0, cols), generator=g, device=device) < density
        nz = mask.nonzero(as_tuple=False)                 # row-major -> sorted inside rows
        idx_parts.append(nz[:, 1].to(torch.int32))
        cnt_parts.append(mask.sum(dim=1))
        del mask, nz
    indices = torch.cat(idx_parts)
    counts = torch.cat(cnt_parts)
    indptr = torch.zeros(rows + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    assert int(indptr[-1]) < 2 ** 31
    data = torch.rand(indices.numel(), generator=g, device=device, dtype=torch.float64)
    return indptr.to(torch.int32), indices, data


def gen_symmetric_csr_device(torch, n, density, seed, device):
    """Q = S + S^T with S = random(n, n, density / 2): symmetric, density ~ `density`, sorted rows
    (BASELINE configs[3]: "B 80000 x 80000 symmetric d=0.005"; SURVEY 8d builds it the same way)."""
    ip, ix, dv = gen_csr_device(torch, n, n, density / 2.0, seed, device)
    rows = torch.repeat_interleave(torch.arange(n, device=device, dtype=torch.int64), (ip[1:] - ip[:-1]).to(torch.int64))
    cols = ix.to(torch.int64)
    key = torch.cat([rows * n + cols, cols * n + rows])
    val = torch.cat([dv, dv])
    key, order = torch.sort(key)
    val = val[order]
    ukey, inverse = torch.unique_consecutive(key, return_inverse=True)
    uval = torch.zeros(ukey.numel(), dtype=torch.float64, device=device)
    uval.index_add_(0, inverse, val)                      # S[i,j] + S[j,i] where both exist (and 2 S[i,i])
    urow = ukey // n
    counts = torch.bincount(urow, minlength=n)
    indptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    indptr[1:] = torch.cumsum(counts, 0)
    assert int(indptr[-1]) < 2 ** 31
    return indptr.to(torch.int32), (ukey % n).to(torch.int32), uval
