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
    from sparse_matrix_mult_amd.distributed import dense_row_sharded, spgemm_row_sharded, triple_row_sharded
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
            # dense_nosym / dense_sym row blocks, one all_gather_into_tensor of padded tiles
            D = dense_row_sharded(ctx, A, B, dist, symmetric=symmetric, exact=True)
            ok &= np.array_equal(D.cpu().numpy(), oracle.dense(arrays(A), arrays(B), 333, symmetric=symmetric))
            (d0, d1), blk = dense_row_sharded(ctx, A, B, dist, symmetric=symmetric, gather=False, exact=True)
            ok &= tuple(blk.shape) == (d1 - d0, 333)
        # triple product: row blocks balanced by sum(n - i)
        import scipy.sparse as sp
        H = rand_csr(190, 260, 0.06, 3)
        S = sp.random(260, 260, density=0.03, format="csr", random_state=np.random.default_rng(4))
        Q = (S + S.T).tocsr()
        T = triple_row_sharded(ctx, H, Q, dist, exact=True)
        ok &= np.array_equal(T.cpu().numpy(), oracle.triple(arrays(H), arrays(Q), 260, 0))
        ret[rank] = bool(ok)
    finally:
        ctx.close()
        dist.destroy_process_group()


def test_bench_launches_its_own_ranks_and_gathers(tmp_path):
    """`python bench.py --gpus 2` with no launcher starts its ranks itself (fresh child processes) and rank 0
    prints one JSON line with n_gpus = 2.  Rehearsal mode: both ranks on GPU 0 over gloo -- the plumbing of the
    N > 1 path (self-launch, shard, exchange step, --gather's all-gatherv), not a performance result."""
    import json
    import subprocess
    env = dict(os.environ, SMM_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    for extra in ([], ["--gather"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--rows", "3000", "--cols", "6000", "--density", "0.004", "--no-cpu"] + extra,
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["value"] > 0 and line["unit"] == "nnz/s"
        assert ("all-gatherv" in line["config"]["workload"]) == bool(extra)
        # round 3: who is in the job, and the collective part of a step on its own
        assert line["rccl"]["backend"] == "gloo" and line["rccl"]["world"] == 2 and len(line["rccl"]["devices"]) == 2
        assert line["rccl"]["distinct_devices"] == 1                     # the rehearsal puts both ranks on GPU 0
        assert line["gather_ms" if extra else "exchange_ms"] >= 0.0


def test_bench_rehearsal_of_the_config4_shape_verifies_against_one_gpu():
    """Round 4 (first-contact safety): the N > 1 bench line on a small instance of BASELINE configs[4]'s shape (2000 rows
    per rank, B 70 000 columns wide: beyond one slab of the symbolic walk, so the column-slab kernels of the real
    configs[4] run) -- weak scaling with the exchange step, and --gather with the all-gatherv -- carries the rccl /
    exchange_ms / gather_ms fields, and `--verify` shows that what the ranks exchanged / gathered, concatenated in rank
    order, IS the single-GPU CSR of the concatenated row blocks (src/sparse_sparse_sparse.cpp:235-241,269-291)."""
    import json
    import subprocess
    env = dict(os.environ, SMM_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    for extra in ([], ["--gather"], ["--gather", "--exact"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4", "--steps", "2", "--warmup", "1",
                              "--rows", "2000", "--cols", "70000", "--no-cpu", "--verify"] + extra,
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert line["n_gpus"] == 2 and line["rccl"]["world"] == 2 and line["rccl"]["backend"] == "gloo"
        assert line["gather_ms" if "--gather" in extra else "exchange_ms"] >= 0.0
        assert line["verify"]["global_indptr_equals_single_gpu"] is True
        if "--gather" in extra:
            assert line["verify"]["gathered_indices_equal_single_gpu"] is True
            assert line["verify"]["gathered_values_equal_single_gpu"] is True


def test_two_ranks_reassemble_the_single_device_csr():
    import torch.multiprocessing as mp
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
        assert dict(ret) == {0: True, 1: True}
