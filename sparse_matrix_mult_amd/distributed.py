"""Row-sharded multi-GPU driver: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has exactly one parallel strategy -- contiguous row blocks handed to OpenMP
threads (limits(), src/workdivision.cpp:16-89; driver src/sparse_sparse_sparse.cpp:228-249)
and a serial stitch that concatenates the blocks in order (:269-291).  This module is the
same idea across GPUs:

  * balanced_row_shards(): contiguous blocks balanced by WORK (products per row) instead of
    by row count as limits() does;
  * every rank multiplies its block of A with the replicated B (engine.Context);
  * the exchange step: an all-gather of the shards' row counts gives the global row pointer
    (global_indptr); allgather_csr() additionally reassembles indices/values on every rank
    with ONE variable-length all-gather.  RCCL has no all-gatherv, and xGMI is point-to-
    point (7 links per GPU): the v-gather is issued as one batch of isend/irecv pairs, so
    that every link carries exactly one peer's shard concurrently instead of a ring that
    would serialise on one link.

Because blocks are contiguous and each block is bit-identical to the same rows of the
single-device result, concatenation in rank order IS the single-device CSR (tested on gloo,
world_size 2, in tests/test_distributed_cpu.py).
"""
import numpy as np

__all__ = ["balanced_row_shards", "row_work", "triple_row_shards", "global_indptr", "allgatherv", "allgather_csr",
           "allgather_rows", "spgemm_row_sharded", "dense_row_sharded", "triple_row_sharded"]


def balanced_row_shards(work, n_shards):
    """Contiguous [begin,end) row ranges whose summed `work` is as even as a greedy prefix
    split allows.  Every shard is non-empty while rows last (like limits(), which clamps the
    number of blocks to the number of rows, workdivision.cpp:26-29)."""
    work = np.asarray(work, dtype=np.float64)
    rows = len(work)
    n = max(1, min(int(n_shards), rows)) if rows else 1
    if rows == 0:
        return [(0, 0)]
    csum = np.concatenate([[0.0], np.cumsum(work + 1e-9)])       # +eps keeps it strictly increasing
    total = csum[-1]
    bounds = [0]
    for s in range(1, n):
        cut = int(np.searchsorted(csum, total * s / n, side="left"))
        cut = max(cut, bounds[-1] + 1)                           # non-empty
        cut = min(cut, rows - (n - s))                           # leave rows for the rest
        bounds.append(cut)
    bounds.append(rows)
    return [(bounds[i], bounds[i + 1]) for i in range(n)]


def row_work(a_indptr, a_indices, b_row_len):
    """products per row of A: sum of nnz(B[r,:]) over the row's entries, from the structure alone
    (prefix-sum difference: empty rows anywhere, trailing ones included, give 0)."""
    per_entry = np.asarray(b_row_len, dtype=np.int64)[np.asarray(a_indices)]
    cs = np.concatenate([[0], np.cumsum(per_entry, dtype=np.int64)])
    ptr = np.asarray(a_indptr, dtype=np.int64)
    return cs[ptr[1:]] - cs[ptr[:-1]]


def _world(dist, group=None):
    return dist.get_world_size(group), dist.get_rank(group)


def _stage(t, dist, group):
    """gloo moves host memory only: stage device tensors through the host there (rehearsals and
    the CPU tests); RCCL takes device tensors as they are."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        return t.cpu(), t.device
    return t, None


def allgatherv(local, dist, group=None):
    """Variable-length all-gather of a 1-D tensor: returns (concatenation in rank order,
    per-rank lengths).  Sizes are exchanged first (one tiny all_gather), then every rank
    sends its shard to every peer and receives theirs in ONE batch of point-to-point ops."""
    import torch
    world, rank = _world(dist, group)
    local, back = _stage(local.contiguous(), dist, group)
    out, sizes = _allgatherv(local, dist, group, world, rank, torch)
    return (out.to(back) if back is not None else out), sizes


def _allgatherv(local, dist, group, world, rank, torch):
    n_local = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(s.item()) for s in sizes]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out = torch.empty(int(offs[-1]), dtype=local.dtype, device=local.device)
    out[offs[rank]:offs[rank + 1]] = local
    if world > 1:
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            if sizes[rank] > 0:
                ops.append(dist.P2POp(dist.isend, local, _global_rank(dist, group, peer), group))
            if sizes[peer] > 0:
                ops.append(dist.P2POp(dist.irecv, out[offs[peer]:offs[peer + 1]], _global_rank(dist, group, peer), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
    return out, sizes


def _global_rank(dist, group, group_rank):
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def global_indptr(local_indptr, dist, group=None, equal_rows=False):
    """local_indptr: int64, rows_local+1, starting at 0.  Returns the global row pointer
    (sum(rows)+1 entries) of the row-concatenated result on every rank.  equal_rows=True (every
    rank holds the same number of rows, as in bench.py) uses one plain all_gather."""
    import torch
    counts = (local_indptr[1:] - local_indptr[:-1]).contiguous()
    if equal_rows:
        world, _ = _world(dist, group)
        send, back = _stage(counts, dist, group)
        all_counts = torch.empty(world * send.numel(), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(all_counts, send, group=group)
        if back is not None:
            all_counts = all_counts.to(back)
    else:
        all_counts, _ = allgatherv(counts, dist, group)
    out = torch.zeros(all_counts.numel() + 1, dtype=torch.int64, device=local_indptr.device)
    torch.cumsum(all_counts, 0, out=out[1:])
    return out


def allgather_csr(indptr, indices, data, dist, group=None):
    """Reassemble the whole CSR on every rank: (global indptr, indices, data)."""
    g_indptr = global_indptr(indptr, dist, group)
    g_indices, _ = allgatherv(indices, dist, group)
    g_data, _ = allgatherv(data, dist, group)
    return g_indptr, g_indices, g_data


def spgemm_row_sharded(ctx, matrix_a, matrix_b, dist, symmetric=False, gather=True, group=None, exact=False):
    """C = A @ B with A's rows sharded over the ranks of `group`; A and B are scipy CSR
    matrices every rank holds (the replicated-input case of SURVEY 8e).  Returns torch
    tensors on this rank's GPU: the whole CSR when gather=True, else this rank's row block
    plus the global row pointer: (g_indptr, (row_begin, row_end), indices_local, data_local)."""
    world, rank = _world(dist, group)
    b = ctx.csr_from_scipy(matrix_b)
    try:
        # work per row from the structure alone: sum of nnz(B[r,:]) over A's entries
        b_len = np.diff(matrix_b.indptr).astype(np.int64)
        work = row_work(matrix_a.indptr, matrix_a.indices, b_len)
        shards = balanced_row_shards(work, world)
        r0, r1 = shards[rank] if rank < len(shards) else (matrix_a.shape[0], matrix_a.shape[0])
        a = ctx.csr_from_scipy(matrix_a[r0:r1])
        try:
            indptr, indices, data = ctx.spgemm_torch(a, b, symmetric=symmetric, row_offset=r0, exact=exact)
            ctx.synchronize()        # the product runs on the context's stream, the collectives on torch's
        finally:
            a.close()
    finally:
        b.close()
    if gather:
        return allgather_csr(indptr, indices, data, dist, group)
    return global_indptr(indptr, dist, group), (r0, r1), indices, data


def triple_row_shards(n, n_shards, full=False):
    """Row blocks of the n x n triple product balanced by the work of stage 2: row i computes the
    cells k >= i, i.e. n - i of them (all n with compute_full_matrix) -- the reference hands rows
    out with schedule(dynamic, 64) for the same reason (src/sparse_sparse_dense.cpp:183)."""
    i = np.arange(n, dtype=np.float64)
    return balanced_row_shards(np.full(n, float(n)) if full else (n - i), n_shards)


def allgather_rows(local, rows_per_rank, dist, group=None):
    """All-gather of a row-sharded dense matrix: `local` is this rank's (rows_r x n) block.  Blocks of equal
    height travel in ONE plain all_gather_into_tensor (RCCL's native all-gather, SURVEY 8e "Dense output");
    unequal blocks -- the triple product's shards are balanced by sum(n - i), so the last one is ~5x as tall
    as the first -- are flattened and gathered by the batched variable-length all-gather above straight into
    their place in the result: no padding to the tallest block (2.8x the result at 8 ranks) and no copy to drop
    it again.  Returns the (sum(rows) x n) matrix on every rank."""
    import torch
    world, rank = _world(dist, group)
    rows_per_rank = [int(r) for r in rows_per_rank]
    if len(rows_per_rank) != world or local.shape[0] != rows_per_rank[rank]:
        raise ValueError(f"allgather_rows: rank {rank} holds {local.shape[0]} rows, rows_per_rank says {rows_per_rank}")
    n = int(local.shape[1])
    total = sum(rows_per_rank)
    send, back = _stage(local.contiguous(), dist, group)
    if len(set(rows_per_rank)) == 1:
        out = torch.empty((total, n), dtype=send.dtype, device=send.device)
        if total * n > 0:
            dist.all_gather_into_tensor(out, send, group=group)
    else:
        flat, sizes = _allgatherv(send.reshape(-1), dist, group, world, rank, torch)
        if sizes != [r * n for r in rows_per_rank]:           # (a real check: `python -O` strips asserts)
            raise RuntimeError(f"allgather_rows: the ranks sent {sizes} elements, expected {[r * n for r in rows_per_rank]}")
        out = flat.reshape(total, n)
    return out.to(back) if back is not None else out


def _shard_rows(shards, world):
    return [(shards[r][1] - shards[r][0]) if r < len(shards) else 0 for r in range(world)]


def dense_row_sharded(ctx, matrix_a, matrix_b, dist, symmetric=False, gather=True, group=None, exact=False):
    """C = A @ B as a dense matrix, A's rows sharded over the ranks (dense_nosym / dense_sym run
    their rows under `omp parallel for`, src/sparse_sparse_dense.cpp:38,106).  Returns the whole
    m x n torch tensor on every rank (gather=True), else ((row_begin, row_end), local block)."""
    import torch
    world, rank = _world(dist, group)
    m, n = matrix_a.shape[0], matrix_b.shape[1]
    work = row_work(matrix_a.indptr, matrix_a.indices, np.diff(matrix_b.indptr))
    shards = balanced_row_shards(work, world)
    r0, r1 = shards[rank] if rank < len(shards) else (m, m)
    dev = torch.device("cuda", ctx.device)
    local = torch.empty((r1 - r0, n), dtype=torch.float64, device=dev)
    if r1 > r0 and n > 0:
        b = ctx.csr_from_scipy(matrix_b)
        try:
            a = ctx.csr_from_scipy(matrix_a[r0:r1])
            try:
                ctx.dense_into(a, b, local.data_ptr(), symmetric=symmetric, row_offset=r0, exact=exact)
                ctx.synchronize()
            finally:
                a.close()
        finally:
            b.close()
    if not gather:
        return (r0, r1), local
    return allgather_rows(local, _shard_rows(shards, world), dist, group)


def triple_row_sharded(ctx, matrix_h, matrix_q, dist, gather=True, group=None, exact=False):
    """Upper triangle of H Q H^T (compute_full_matrix=0, the BASELINE configuration) with the rows
    of the result sharded over the ranks, blocks balanced by sum(n - i); H and Q are replicated
    (stage 2 reads all of H).  Returns the n x n tensor on every rank, else ((r0, r1), block)."""
    import torch
    world, rank = _world(dist, group)
    n = matrix_h.shape[0]
    shards = triple_row_shards(n, world)
    r0, r1 = shards[rank] if rank < len(shards) else (n, n)
    dev = torch.device("cuda", ctx.device)
    local = torch.empty((r1 - r0, n), dtype=torch.float64, device=dev)
    if r1 > r0:
        h, q = ctx.csr_from_scipy(matrix_h), ctx.csr_from_scipy(matrix_q)
        try:
            ctx.triple_into(h, q, local.data_ptr(), row_begin=r0, row_end=r1, exact=exact)
            ctx.synchronize()
        finally:
            h.close(); q.close()
    if not gather:
        return (r0, r1), local
    return allgather_rows(local, _shard_rows(shards, world), dist, group)
