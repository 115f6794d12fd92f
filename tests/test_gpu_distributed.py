"""Row-sharded product end to end on the GPU box: two ranks (both on GPU 0, gloo collectives --
RCCL refuses two ranks on one device) run sparse_matrix_mult_amd.distributed.spgemm_row_sharded
and must each end up with the single-device CSR, bit for bit (SMM_EXACT) / to 1e-10 (default)."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import arrays, rand_csr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import oracle
    from sparse_matrix_mult_amd.distributed import spgemm_row_sharded
    from sparse_matrix_mult_amd.engine import Context
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctx = Context(0)
    try:
        A, B = rand_csr(333, 280, 0.05, 1), rand_csr(280, 333, 0.05, 2)
        ok = True
        for symmetric in (False, True):
            want = oracle.sparse(arrays(A), arrays(B), 333, symmetric=symmetric)
            p, i, v = spgemm_row_sharded(ctx, A, B, dist, symmetric=symmetric, gather=True, exact=True)
            ok &= np.array_equal(p.cpu().numpy(), want[0]) and np.array_equal(i.cpu().numpy(), want[1])
            ok &= np.array_equal(v.cpu().numpy(), want[2])
            gp, (r0, r1), li, lv = spgemm_row_sharded(ctx, A, B, dist, symmetric=symmetric, gather=False)
            ok &= np.array_equal(gp.cpu().numpy(), want[0])
            ok &= np.array_equal(li.cpu().numpy(), want[1][want[0][r0]:want[0][r1]])
            ok &= np.allclose(lv.cpu().numpy(), want[2][want[0][r0]:want[0][r1]], rtol=1e-10, atol=0)
        ret[rank] = bool(ok)
    finally:
        ctx.close()
        dist.destroy_process_group()


def test_two_ranks_reassemble_the_single_device_csr():
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: True, 1: True}
