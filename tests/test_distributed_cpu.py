"""The N>1 path on CPU: two gloo ranks shard the rows, each produces its block (here with the
CPU oracle standing in for the GPU engine -- the collective logic is what is under test), the
variable-length all-gather reassembles the CSR, and the result must be the single-device CSR."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import arrays, rand_csr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, symmetric, empty_rows, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import allgather_csr, balanced_row_shards, global_indptr, row_work
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, B = rand_csr(203, 150, 0.06, 1), rand_csr(150, 203, 0.06, 2)
        if empty_rows:              # leading, interior and trailing empty rows of A (reduceat raised there)
            A = A.tolil()
            for r in (0, 1, 57, 58, 59, 201, 202):
                A[r, :] = 0
            A = A.tocsr()
            A.eliminate_zeros()
        a, b = arrays(A), arrays(B)
        work = row_work(a[0], a[1], np.diff(b[0]))
        assert work.shape == (203,) and (work[np.diff(a[0]) == 0] == 0).all()
        shards = balanced_row_shards(work, world)
        r0, r1 = shards[rank]
        cnt, idx, val = oracle.sparse_rows(a, b, 203, r0, r1, symmetric=symmetric)
        indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
        g_ptr, g_idx, g_val = allgather_csr(indptr, torch.from_numpy(idx), torch.from_numpy(val), dist)
        g_only = global_indptr(indptr, dist)
        want = oracle.sparse(a, b, 203, symmetric=symmetric)
        # equal-row form used by bench.py: one plain all_gather of the counts
        loc = torch.from_numpy(np.concatenate([[0], np.cumsum(np.arange(1, 8) * (rank + 1))]).astype(np.int64))
        eq = global_indptr(loc, dist, equal_rows=True).numpy()
        eq_want = np.concatenate([[0], np.cumsum(np.concatenate([np.arange(1, 8) * (r + 1) for r in range(world)]))])
        assert np.array_equal(eq, eq_want)
        ok = (np.array_equal(g_ptr.numpy(), want[0]) and np.array_equal(g_idx.numpy(), want[1])
              and np.array_equal(g_val.numpy(), want[2]) and np.array_equal(g_only.numpy(), want[0]))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,symmetric,empty_rows", [(2, False, False), (2, True, True), (3, False, True), (3, True, False)])
def test_row_sharded_allgatherv_reassembles_single_device_csr(world, symmetric, empty_rows):
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, symmetric, empty_rows, ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


def _dense_worker(rank, world, port, ret):
    """dense_row_sharded / triple_row_sharded without a GPU: the oracle computes each rank's row
    block, allgather_rows (padded equal tiles, one all_gather_into_tensor) reassembles it."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import (allgather_rows, balanced_row_shards, row_work,
                                                    triple_row_shards, _shard_rows)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, B = rand_csr(61, 40, 0.1, 3), rand_csr(40, 61, 0.1, 4)
        a, b = arrays(A), arrays(B)
        ok = True
        for symmetric in (False, True):
            shards = balanced_row_shards(row_work(a[0], a[1], np.diff(b[0])), world)
            r0, r1 = shards[rank]
            blk = oracle.dense(a, b, 61, symmetric=symmetric, row_begin=r0, row_end=r1)
            full = allgather_rows(torch.from_numpy(blk), _shard_rows(shards, world), dist).numpy()
            ok &= np.array_equal(full, oracle.dense(a, b, 61, symmetric=symmetric))
        S = sp.random(40, 40, density=0.1, format="csr", random_state=np.random.default_rng(5))
        q = arrays((S + S.T).tocsr())
        shards = triple_row_shards(61, world)
        r0, r1 = shards[rank]
        blk = oracle.triple(a, q, 40, 0, r0, r1)[r0:r1]
        full = allgather_rows(torch.from_numpy(np.ascontiguousarray(blk)), _shard_rows(shards, world), dist).numpy()
        ok &= np.array_equal(full, oracle.triple(a, q, 40, 0))
        # more ranks than rows: the trailing ranks hold empty blocks
        tiny = triple_row_shards(1, world)
        rows = _shard_rows(tiny, world)
        mine = torch.full((rows[rank], 3), float(rank))
        ok &= allgather_rows(mine, rows, dist).shape == (1, 3)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_dense_and_triple_row_blocks_allgather(world):
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dense_worker, args=(world, port, ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}


def _few_rows_worker(rank, world, port, ret):
    """Fewer rows than ranks: the trailing ranks own no row at all and send / receive nothing in the
    variable-length all-gather (sizes 0), yet every rank ends with the whole CSR and the whole dense matrix."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import (allgather_csr, allgather_rows, allgatherv, balanced_row_shards,
                                                    row_work, _shard_rows)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, B = rand_csr(2, 30, 0.4, 7), rand_csr(30, 25, 0.3, 8)
        a, b = arrays(A), arrays(B)
        shards = balanced_row_shards(row_work(a[0], a[1], np.diff(b[0])), world)
        assert len(shards) == 2                                      # clamped to the rows, like limits()
        r0, r1 = shards[rank] if rank < len(shards) else (2, 2)
        cnt, idx, val = oracle.sparse_rows(a, b, 25, r0, r1)
        indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
        g_ptr, g_idx, g_val = allgather_csr(indptr, torch.from_numpy(idx), torch.from_numpy(val), dist)
        want = oracle.sparse(a, b, 25)
        ok = np.array_equal(g_ptr.numpy(), want[0]) and np.array_equal(g_idx.numpy(), want[1]) and np.array_equal(g_val.numpy(), want[2])
        flat, sizes = allgatherv(torch.from_numpy(val), dist)
        ok &= sizes[-1] == 0 and sum(sizes) == len(want[2]) and np.array_equal(flat.numpy(), want[2])
        blk = oracle.dense(a, b, 25, row_begin=r0, row_end=r1)
        full = allgather_rows(torch.from_numpy(blk), _shard_rows(shards, world), dist).numpy()
        ok &= np.array_equal(full, oracle.dense(a, b, 25))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_more_ranks_than_rows_leaves_empty_shards():
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_few_rows_worker, args=(3, port, ret), nprocs=3, join=True)
        assert dict(ret) == {0: True, 1: True, 2: True}


def _world8_worker(rank, world, port, ret):
    """Eight ranks, the shapes the 8-GPU node will see first (round 4): (a) a CSR product sharded by work with empty rows
    at both ends, (b) the triple product's row blocks -- balanced by sum(n - i), so the last block is several times as
    tall as the first -- through the unequal-height path of allgather_rows, (c) 7 rows on 8 ranks: the last rank owns
    nothing, sends nothing, and still ends with the whole CSR and the whole dense matrix."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scipy.sparse as sp
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import (allgather_csr, allgather_rows, balanced_row_shards, global_indptr, row_work,
                                                    triple_row_shards, _shard_rows)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        # (a) CSR, work-balanced contiguous blocks
        A, B = rand_csr(203, 150, 0.06, 11), rand_csr(150, 203, 0.06, 12)
        A = A.tolil()
        for r in (0, 1, 100, 201, 202):
            A[r, :] = 0
        A = A.tocsr(); A.eliminate_zeros()
        a, b = arrays(A), arrays(B)
        shards = balanced_row_shards(row_work(a[0], a[1], np.diff(b[0])), world)
        ok &= len(shards) == world and shards[0][0] == 0 and shards[-1][1] == 203
        ok &= all(shards[i][1] == shards[i + 1][0] for i in range(world - 1))           # contiguous, in rank order
        r0, r1 = shards[rank]
        for symmetric in (False, True):
            cnt, idx, val = oracle.sparse_rows(a, b, 203, r0, r1, symmetric=symmetric)
            indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
            g_ptr, g_idx, g_val = allgather_csr(indptr, torch.from_numpy(idx), torch.from_numpy(val), dist)
            want = oracle.sparse(a, b, 203, symmetric=symmetric)
            ok &= np.array_equal(g_ptr.numpy(), want[0]) and np.array_equal(g_idx.numpy(), want[1]) and np.array_equal(g_val.numpy(), want[2])
            ok &= np.array_equal(global_indptr(indptr, dist).numpy(), want[0])
        # (b) triple product: unequal block heights (last / first > 2.5 at 8 ranks)
        H = rand_csr(97, 60, 0.1, 13)
        S = sp.random(60, 60, density=0.1, format="csr", random_state=np.random.default_rng(14))
        h, q = arrays(H), arrays((S + S.T).tocsr())
        tsh = triple_row_shards(97, world)
        heights = _shard_rows(tsh, world)
        ok &= sum(heights) == 97 and heights[-1] > 2.5 * heights[0] and len(set(heights)) > 1
        t0, t1 = tsh[rank]
        blk = np.ascontiguousarray(oracle.triple(h, q, 60, 0, t0, t1)[t0:t1])
        ok &= np.array_equal(allgather_rows(torch.from_numpy(blk), heights, dist).numpy(), oracle.triple(h, q, 60, 0))
        # a wrong height list is an exception, not a silent misplacement
        if rank == 0:
            try:
                allgather_rows(torch.from_numpy(blk), [h_ + 1 for h_ in heights], dist)    # raises before any collective
                ok = False
            except ValueError:
                pass
        # (c) fewer rows than ranks: rank 7 owns nothing
        A7, B7 = rand_csr(7, 30, 0.4, 15), rand_csr(30, 25, 0.3, 16)
        a7, b7 = arrays(A7), arrays(B7)
        sh7 = balanced_row_shards(row_work(a7[0], a7[1], np.diff(b7[0])), world)
        ok &= len(sh7) == 7
        s0, s1 = sh7[rank] if rank < len(sh7) else (7, 7)
        cnt, idx, val = oracle.sparse_rows(a7, b7, 25, s0, s1)
        indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
        g_ptr, g_idx, g_val = allgather_csr(indptr, torch.from_numpy(idx), torch.from_numpy(val), dist)
        want = oracle.sparse(a7, b7, 25)
        ok &= np.array_equal(g_ptr.numpy(), want[0]) and np.array_equal(g_idx.numpy(), want[1]) and np.array_equal(g_val.numpy(), want[2])
        dblk = oracle.dense(a7, b7, 25, row_begin=s0, row_end=s1)
        ok &= np.array_equal(allgather_rows(torch.from_numpy(dblk), _shard_rows(sh7, world), dist).numpy(), oracle.dense(a7, b7, 25))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_world_8_shards_gathers_and_empty_ranks():
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_world8_worker, args=(8, port, ret), nprocs=8, join=True)
        assert dict(ret) == {r: True for r in range(8)}
