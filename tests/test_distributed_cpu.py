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


def _worker(rank, world, port, symmetric, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import allgather_csr, balanced_row_shards, global_indptr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A, B = rand_csr(203, 150, 0.06, 1), rand_csr(150, 203, 0.06, 2)
        a, b = arrays(A), arrays(B)
        work = np.add.reduceat(np.diff(b[0])[a[1]], a[0][:-1]) * (np.diff(a[0]) > 0)
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


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("symmetric", [False, True])
def test_row_sharded_allgatherv_reassembles_single_device_csr(world, symmetric):
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, symmetric, ret), nprocs=world, join=True)
        assert dict(ret) == {r: True for r in range(world)}
