#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CSR x CSR -> CSR hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rows 50000] [--cols 50000] [--density 0.01]

Workload (BASELINE.json configs[1]): A (rows x cols) times B (cols x cols), both uniform
random CSR of the given density with uniform[0,1) float64 values -- the distribution of
scipy.sparse.random, generated on the device (inputs are resident in HBM before the timed
region; nothing from the host is in it).  One step = one whole product: symbolic phase
(row counts + first-touch column order), scan, numeric phase, result left in HBM.

Multi-GPU (weak scaling): rank r owns the contiguous row block [r*rows, (r+1)*rows) of a
global (N*rows x cols) A; B is replicated.  The exchange step is the all-gather of the
shards' row counts that turns local row pointers into the global CSR row pointer
(sparse_matrix_mult_amd/distributed.py); indices/values stay sharded (a full all-gatherv of
N x 30 GB does not fit one GPU at this size -- DESIGN.md "Multi-GPU").

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes of one product / average duration of the dominant kernel
                  (smm_numeric), measured with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference's algorithm) timed on a
                  bounded row sample of the same operands on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec


def gen_csr_device(torch, rows, cols, density, seed, device):
    """Uniform random CSR on the device: every cell is kept with probability `density`
    (row lengths are binomial, as scipy.sparse.random's are to within sampling noise),
    indices sorted inside rows, values uniform[0,1) float64."""
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


def cpu_baseline(torch, a, b, cols, gpu_result=None, target_s=12.0):
    """Time the oracle (kind 'port': our C restatement of src/sparsework.cpp) on the first R
    rows of A against all of B on this box's host cores (one thread per core, disjoint row
    ranges -- the reference's own row-block parallelism, sparse_sparse_sparse.cpp:228-249).  When the GPU result of the
    last step is passed in, the same rows are compared with it (full-size parity check:
    indptr / indices bit-exact, values within 1e-10 relative)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    ap, ai, av = (t.cpu().numpy() for t in a)
    bp, bi, bv = (t.cpu().numpy() for t in b)
    A_, B_ = (ap, ai, av), (bp, bi, bv)
    rows = len(ap) - 1
    # host threads actually used: one per core this process may run on (rows are independent;
    # ctypes releases the GIL, every call has its own marker array)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16, rows))          # the GPU box's CPU share for one GPU is 16
    probe = max(1, min(rows, 32))
    t0 = time.perf_counter()
    oracle.sparse_rows(A_, B_, cols, 0, probe)
    dt1 = max(time.perf_counter() - t0, 1e-6)
    # ~10-30 core-seconds of work, at most 20 000 rows (the sample's CSR is held on the host)
    sample = int(max(cores, min(rows, 20000, probe * target_s / dt1 * cores * 0.6)))
    bounds = [sample * i // cores for i in range(cores + 1)]
    with ThreadPoolExecutor(cores) as pool:
        t0 = time.perf_counter()
        parts = list(pool.map(lambda i: oracle.sparse_rows(A_, B_, cols, bounds[i], bounds[i + 1]), range(cores)))
        dt = time.perf_counter() - t0
    cnt = np.concatenate([p[0] for p in parts]); idx = np.concatenate([p[1] for p in parts])
    val = np.concatenate([p[2] for p in parts])
    out = {"value": float(cnt.sum() / dt), "unit": "nnz/s", "cores": cores, "kind": "port",
           "sample": f"first {sample} of {rows} rows of A x all of B on {cores} host threads, "
                     f"{int(cnt.sum())} output nnz in {dt:.2f} s (one thread alone: {probe / dt1 * cnt.sum() / sample:.3g} nnz/s)"}
    if gpu_result is not None:
        g_ptr, g_idx, g_val = gpu_result
        nn = int(cnt.sum())
        ok_ptr = np.array_equal(g_ptr[:sample + 1].cpu().numpy(), np.concatenate([[0], np.cumsum(cnt)]))
        ok_idx = np.array_equal(g_idx[:nn].cpu().numpy(), idx)
        gv = g_val[:nn].cpu().numpy()
        rel = float(np.max(np.abs(gv - val) / np.maximum(np.abs(val), 1e-300))) if nn else 0.0
        out["parity_on_sample"] = {"indptr_bit_exact": bool(ok_ptr), "indices_bit_exact": bool(ok_idx),
                                   "values_max_rel_err": rel, "values_bit_exact": bool(np.array_equal(gv, val))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--cols", type=int, default=50000)
    ap.add_argument("--density", type=float, default=0.01)
    ap.add_argument("--exact", action="store_true", help="SMM_EXACT: reference-order accumulation (bit-exact values)")
    ap.add_argument("--lds-cols", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--hash", type=str, default="", help="small,medium thresholds of the LDS-hash kernels (0,0 = off)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
        args.gpus = world
    # SMM_BENCH_REHEARSAL=1: every rank on GPU 0 with gloo collectives -- a way to rehearse the
    # N > 1 code path on a one-GPU box (RCCL refuses two ranks on one device).  Never a result.
    rehearsal = os.environ.get("SMM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from sparse_matrix_mult_amd.engine import Context
    from sparse_matrix_mult_amd import distributed as smm_dist

    stream = torch.cuda.current_stream().cuda_stream
    ctx = Context(local, stream)
    if args.lds_cols or args.waves:
        (ctx.tune if args.exact else ctx.tune_shared)(args.lds_cols, args.waves)

    if args.hash:
        ctx.tune_hash(*[int(x) for x in args.hash.split(",")])

    m, n, d = args.rows, args.cols, args.density
    a_t = gen_csr_device(torch, m, n, d, 1 + 1000 * rank, device)      # rank's row block of A
    b_t = gen_csr_device(torch, n, n, d, 2, device)                   # B, replicated
    A = ctx.csr_from_torch(m, n, *a_t)
    B = ctx.csr_from_torch(n, n, *b_t)
    nnz_a, nnz_b = A.nnz, B.nnz

    def step():
        plan = ctx.spgemm_plan(A, B, row_offset=rank * m, exact=args.exact)
        indptr = torch.empty(m + 1, dtype=torch.int64, device=device)
        indices = torch.empty(plan.nnz, dtype=torch.int32, device=device)
        data = torch.empty(plan.nnz, dtype=torch.float64, device=device)
        plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        plan.close()
        if world > 1:
            indptr = smm_dist.global_indptr(indptr, dist, equal_rows=True)   # the exchange step
        return indptr, indices, data

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
        del out
    ctx.timing(True)
    ctx.timing_reset()
    fence()
    t0 = time.perf_counter()
    for it in range(args.steps):
        out = step()
        nnz_c = int(out[1].numel())
        if it + 1 < args.steps or args.no_cpu or world > 1:
            del out                               # rank 0 keeps the last result for the parity check
    fence()
    elapsed = time.perf_counter() - t0
    num_ms, num_n = ctx.kernel_time("smm_numeric")
    sym_ms, sym_n = ctx.kernel_time("smm_symbolic")
    ctx.timing(False)

    t = torch.tensor([elapsed, float(nnz_c)], dtype=torch.float64, device="cpu" if rehearsal else device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_nnz = float(tmax[0]), float(t[1])
    else:
        total_nnz = float(nnz_c)

    if rank == 0:
        # HBM traffic of the dominant kernel, from the committed PMC passes of this same workload
        # (rocprofv3 cannot run inside the bench; profiles/traffic_r1.json says how it was taken)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r1.json")))
            if (m, n, d) == (50000, 50000, 0.01) and not args.exact and not (args.lds_cols or args.waves):
                traffic = tj["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        # compulsory one-touch bytes of one product
        alg_bytes = (4 * (m + 1) + 12 * nnz_a) + (4 * (n + 1) + 12 * nnz_b) + (8 * (m + 1) + 12 * nnz_c)
        num_avg_s = (num_ms / max(num_n, 1)) * 1e-3
        achieved = alg_bytes / num_avg_s / 1e9 if num_avg_s > 0 else 0.0
        line = {
            "metric": "output nnz/sec, CSR x CSR -> CSR SpGEMM (first-touch order, float64)",
            "value": total_nnz * args.steps / elapsed,
            "unit": "nnz/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{m}x{n} x {n}x{n} uniform random CSR d={d} -> CSR, per GPU "
                                   f"(BASELINE configs[1])",
                       "nnz_a": nnz_a, "nnz_b": nnz_b, "nnz_c_per_gpu": nnz_c,
                       "mode": "SMM_EXACT (values bit-identical to the CPU loop)" if args.exact
                               else "default (indices bit-exact, values to rounding)",
                       "parallelism": f"row-sharded x{world}" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "smm_numeric", "kernel_ms": num_ms / max(num_n, 1),
                         "algorithmic_bytes": alg_bytes,
                         # rate at which the kernel moves its measured HBM-side traffic (PMC bytes / live duration):
                         # how close the gather itself runs to the 8 TB/s peak, as opposed to `frac`, which prices
                         # only the compulsory bytes
                         "traffic_rate_GBs": (traffic / num_avg_s / 1e9) if (traffic and num_avg_s > 0) else None,
                         "symbolic_kernel_ms": sym_ms / max(sym_n, 1)},
        }
        if not args.no_cpu and world == 1:           # the CPU leg runs at N = 1 only
            line["cpu_baseline"] = cpu_baseline(torch, a_t, b_t, n, (out[0], out[1], out[2]))
        print(json.dumps(line), flush=True)

    A.close(); B.close(); ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
