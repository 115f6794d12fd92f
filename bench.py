#!/usr/bin/env python3
"""bench.py -- headline benchmark of the CSR x CSR hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4] [--gather] [--exact]

Default workload (BASELINE.json configs[1], `--config c1`): A (50 000 x 50 000) times B (50 000 x
50 000), both uniform random CSR of density 0.01 with uniform[0,1) float64 values -- the
distribution of scipy.sparse.random, generated on the device (inputs are resident in HBM before
the timed region; nothing from the host is in it).  One step = one whole product: symbolic phase
(row counts + first-touch column order), scan, numeric phase, result left in HBM.
The other BASELINE configs can be timed with the same harness (they are parity-test cases, not
the driver's bench line): c2 = the same product -> dense, c3 = H Q H^T with H 20 000 x 80 000
d=0.02 and Q 80 000^2 symmetric d=0.005, c4 = one rank's share of 200 000^2 d=0.005 per GPU.

Multi-GPU (weak scaling): rank r owns the contiguous row block [r*rows, (r+1)*rows) of a global
(N*rows x cols) A; B is replicated.  The exchange step is the all-gather of the shards' row
counts that turns local row pointers into the global CSR row pointer
(sparse_matrix_mult_amd/distributed.py); indices/values stay sharded (a full all-gatherv of
N x 30 GB does not fit one GPU at this size -- DESIGN.md "Multi-GPU").  `--gather` runs the full
variable-length all-gather of indices/values inside the step at a size where the whole C fits
every GPU (SURVEY 8e option ii: 200 000^2, d = 0.001, rows split over the ranks: strong scaling).
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run
as a child process, before this process touches the GPU).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes of one product / duration of the numeric phase's kernels,
                  measured with HIP events on the launch stream
  cpu_baseline -- the reference's own row kernel (src/sparsework.cpp compiled by oracle/Makefile into
                  oracle/_ref/, kind "reference") -- or, where that build is absent, the CPU oracle (a port of
                  the reference's algorithm, kind "port") -- timed on a bounded row sample of the same
                  operands on this box's host cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
NUMERIC_KERNELS = ("smm_numeric", "smm_dense_slab", "smm_emit")      # kernels of the numeric phase (whichever ran)


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
    # The reference's OWN row kernel (src/sparsework.cpp, unedited, built by oracle/Makefile into oracle/_ref/ with its
    # marker initialised to -1: oracle/marker_init.c) on the same row ranges and threads, when that build travelled
    # here: the reported baseline is then kind "reference" and the port's rate stays next to it.
    try:
        from oracle import ref_binding as rb
        if rb.m1_available():
            caps = [int(p[0].sum()) for p in parts]
            with ThreadPoolExecutor(cores) as pool:
                t0 = time.perf_counter()
                rparts = list(pool.map(lambda i: rb.sparsework_once(A_, B_, rows, len(bp) - 1, cols, bounds[i], bounds[i + 1], caps[i]),
                                       range(cores)))
                rdt = time.perf_counter() - t0
            same = all(np.array_equal(r[0], p[0]) and np.array_equal(r[1], p[1]) and np.array_equal(r[2], p[2])
                       for r, p in zip(rparts, parts))
            out.update({"value": float(cnt.sum() / rdt), "kind": "reference", "port_value": float(cnt.sum() / dt),
                        "reference_equals_port_bit_for_bit": bool(same),
                        "sample": out["sample"] + f"; the reference's own sparsework_nosym on the same ranges: {rdt:.2f} s"})
    except Exception as e:                       # the checker's build is optional on the box
        out["reference_kernel"] = f"not timed: {e!r}"
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


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start the N ranks as a fresh child process tree
    (torch.distributed.run) BEFORE this process makes any GPU call, pass its output through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def device_identity(torch, local):
    """PCI bus id of this rank's GPU (the first RCCL run must show N distinct devices)."""
    props = torch.cuda.get_device_properties(local)
    try:
        return f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}"
    except AttributeError:
        return str(getattr(props, "uuid", f"device{local}"))


def run_config(cfg, args, env, steps, warmup, cpu_leg):
    """Time `steps` passes of one BASELINE config (inputs generated on the device first, resident in HBM
    before the timed region) and return the fields of its JSON line (rank 0; None elsewhere).  Every
    buffer of the config is released before returning."""
    torch, ctx, dist, smm_dist = env["torch"], env["ctx"], env["dist"], env["smm_dist"]
    rank, world, device, rehearsal, tstream = env["rank"], env["world"], env["device"], env["rehearsal"], env["tstream"]
    from sparse_matrix_mult_amd.synthetic import gen_csr_device, gen_symmetric_csr_device

    if cfg == "c3":
        n, k = int(20000 * args.scale), int(80000 * args.scale)
        m, d = n, 0.02
        A = ctx.csr_from_torch(n, k, *gen_csr_device(torch, n, k, 0.02, 3, device))                 # H
        B = ctx.csr_from_torch(k, k, *gen_symmetric_csr_device(torch, k, 0.005, 4, device))         # Q
        # rows of the result are split over the ranks by sum(n - i) (strong scaling of one product)
        r0, r1 = smm_dist.triple_row_shards(n, world)[rank] if rank < n else (n, n)
        out_buf = torch.empty((r1 - r0, n), dtype=torch.float64, device=device)
        a_t = b_t = None
    elif cfg == "c1s":
        # BASELINE configs[1] on the north star's LITERAL operands (SURVEY 8d "Synthetic inputs"): generated on the host by
        # scipy (outside the timed region), uploaded once, resident in HBM before the first step
        import scipy.sparse as sp
        m = n = 50000
        d = 0.01
        mats = [sp.random(n, n, density=d, format="csr", random_state=np.random.default_rng(seed), dtype=np.float64) for seed in (1, 2)]
        A, B = (ctx.csr_from_scipy(x) for x in mats)
        del mats
        a_t = b_t = out_buf = None
    else:
        if args.gather and not (args.rows or args.cols or args.density):
            n, d = 200000, 0.001
            m = n // world
        elif cfg == "c4":
            n, d = args.cols or 200000, args.density or 0.005
            m = args.rows or n // 8                                 # one rank's share of the 8-way split
        else:
            n, d = args.cols or 50000, args.density or 0.01
            m = args.rows or 50000
        a_t = gen_csr_device(torch, m, n, d, 1 + 1000 * rank, device)      # rank's row block of A
        b_t = gen_csr_device(torch, n, n, d, 2, device)                    # B, replicated
        A = ctx.csr_from_torch(m, n, *a_t)
        B = ctx.csr_from_torch(n, n, *b_t)
        out_buf = torch.empty((m, n), dtype=torch.float64, device=device) if cfg == "c2" else None
    nnz_a, nnz_b = A.nnz, B.nnz
    gather_ev = []

    def step():
        if cfg == "c2":
            ctx.dense_into(A, B, out_buf.data_ptr(), row_offset=rank * m, exact=args.exact)
            return (out_buf,)
        if cfg == "c3":
            ctx.triple_into(A, B, out_buf.data_ptr(), row_begin=r0, row_end=r1, exact=args.exact)
            return (out_buf,)
        plan = ctx.spgemm_plan(A, B, row_offset=rank * m, exact=args.exact)
        indptr = torch.empty(m + 1, dtype=torch.int64, device=device)
        indices = torch.empty(plan.nnz, dtype=torch.int32, device=device)
        data = torch.empty(plan.nnz, dtype=torch.float64, device=device)
        plan.numeric_into(indptr.data_ptr(), indices.data_ptr(), data.data_ptr())
        plan.close()
        local_nnz = indices.numel()
        if world > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(tstream)
            if args.gather:                                          # the whole C on every rank
                indptr, indices, data = smm_dist.allgather_csr(indptr, indices, data, dist)
            else:                                                    # the exchange step: global row pointer
                indptr = smm_dist.global_indptr(indptr, dist, equal_rows=True)
            e1.record(tstream)
            gather_ev.append((e0, e1))
        return indptr, indices, data, local_nnz

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        out = step()
        del out
    gather_ev.clear()
    ctx.timing(True)
    ctx.timing_reset()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    units = 0
    keep_last = cpu_leg and world == 1 and cfg == "c1"       # rank 0 keeps the last result for the parity check
    fence()
    t0 = time.perf_counter()
    marks[0].record(tstream)
    for it in range(steps):
        out = step()
        units = out[3] if len(out) == 4 else out[0].numel()
        marks[it + 1].record(tstream)
        if it + 1 < steps or not keep_last:
            del out
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    ktimes = {k: ctx.kernel_time(k) for k in NUMERIC_KERNELS + ("smm_numeric_dense", "smm_numeric_hash", "smm_symbolic", "smm_symbolic_hash",
                                                                 "smm_runs", "smm_triple_stage2")}
    ctx.timing(False)
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_ev) / max(len(gather_ev), 1) if gather_ev else None

    # --verify (N > 1, CSR configs): one more, untimed step; rank 0 then multiplies the row-wise concatenation of every
    # rank's block of A with B on its own GPU and compares -- the exchanged global row pointer always, and with --gather
    # the gathered indices / values too: concatenation in rank order IS the single-device CSR
    # (the reference's stitch invariant, src/sparse_sparse_sparse.cpp:269-291).
    verify = None
    if args.verify and world > 1 and cfg in ("c1", "c4"):
        got = step()
        torch.cuda.synchronize()
        if rank == 0:
            blocks = [gen_csr_device(torch, m, n, d, 1 + 1000 * r, device) for r in range(world)]
            offs = [0]
            for blk in blocks:
                offs.append(offs[-1] + int(blk[1].numel()))
            ptr = torch.cat([blocks[0][0][:1].to(torch.int64)] + [blk[0][1:].to(torch.int64) + offs[r] for r, blk in enumerate(blocks)])
            whole = ctx.csr_from_torch(m * world, n, ptr.to(torch.int32), torch.cat([blk[1] for blk in blocks]),
                                       torch.cat([blk[2] for blk in blocks]))
            sp_, si_, sv_ = ctx.spgemm_torch(whole, B, exact=args.exact)
            ctx.synchronize()
            verify = {"global_indptr_equals_single_gpu": bool(torch.equal(got[0], sp_))}
            if args.gather:
                verify["gathered_indices_equal_single_gpu"] = bool(torch.equal(got[1], si_))
                verify["gathered_values_equal_single_gpu"] = bool(torch.equal(got[2], sv_) if args.exact else
                                                                  torch.allclose(got[2], sv_, rtol=1e-10, atol=0.0))
            whole.close()
            del blocks, ptr, sp_, si_, sv_
        del got
        fence()

    t = torch.tensor([elapsed, float(units)], dtype=torch.float64, device="cpu" if rehearsal else device)
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_units = float(tmax[0]), float(t[1])
    else:
        total_units = float(units)

    line = None
    if rank == 0:
        per_launch = lambda name: ktimes[name][0] / max(steps, 1)          # ms per step spent in that kernel
        line = {
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "ms_per_step_best": step_ms[0], "ms_per_step_median": step_ms[len(step_ms) // 2],
            "higher_is_better": True, "scaling": "strong" if (args.gather or cfg == "c3") else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
        }
        if gather_ms is not None:
            line["gather_ms" if args.gather else "exchange_ms"] = gather_ms      # the collective part of one step (rank 0)
        if verify is not None:
            line["verify"] = verify
        mode = "SMM_EXACT (values bit-identical to the CPU loop)" if args.exact else "default (indices bit-exact, values to rounding)"
        par = f"row-sharded x{world}" if world > 1 else "single GPU"
        if cfg in ("c1", "c1s", "c4"):
            nnz_c = units
            # compulsory one-touch bytes of one product
            alg_bytes = (4 * (m + 1) + 12 * nnz_a) + (4 * (n + 1) + 12 * nnz_b) + (8 * (m + 1) + 12 * nnz_c)
            num_ms = sum(per_launch(k) for k in NUMERIC_KERNELS)
            # HBM traffic of the numeric phase, from the committed PMC passes of this same workload
            # (rocprofv3 cannot run inside the bench; profiles/ says how it was taken)
            traffic, traffic_src = None, None
            ran = sorted(k for k in NUMERIC_KERNELS if per_launch(k) > 0)
            for name in ("traffic_r4.json", "traffic_r3.json", "traffic_r2.json", "traffic_r1.json"):
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", name)))
                    if (m, n, d) == (50000, 50000, 0.01) and not args.exact and not (args.lds_cols or args.waves or args.slab) \
                            and sorted(tj.get("kernels", ["smm_numeric"])) == ran:
                        traffic = tj["traffic_bytes_per_launch"]
                        traffic_src = f"profiles/{name} (static: PMC passes of this workload and these kernels, not measured in this run)"
                        break
                except (OSError, ValueError, KeyError):
                    pass
            achieved = alg_bytes / (num_ms * 1e-3) / 1e9 if num_ms > 0 else 0.0
            line.update({
                "metric": "output nnz/sec, CSR x CSR -> CSR SpGEMM (first-touch order, float64)",
                "value": total_units * steps / elapsed, "unit": "nnz/s",
                "config": {"workload": (f"{m}x{n} x {n}x{n} scipy.sparse.random(d={d}, random_state=default_rng(1|2)) -> CSR "
                                        f"(BASELINE configs[1], the north star's literal operands)") if cfg == "c1s" else
                                       f"{m}x{n} x {n}x{n} uniform random CSR d={d} -> CSR, per GPU "
                                       f"(BASELINE configs[{1 if cfg == 'c1' else 4}]{', all-gatherv inside the step' if args.gather else ''})",
                           "nnz_a": nnz_a, "nnz_b": nnz_b, "nnz_c_per_gpu": nnz_c, "mode": mode, "parallelism": par},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                             "kernel": "numeric phase: " + " + ".join(k for k in NUMERIC_KERNELS if per_launch(k) > 0),
                             "kernel_ms": num_ms,
                             "kernels_ms": {k: per_launch(k) for k in ktimes if per_launch(k) > 0},
                             "algorithmic_bytes": alg_bytes,
                             # rate at which the phase moves its measured HBM-side traffic (PMC bytes / live duration)
                             "traffic_rate_GBs": (traffic / (num_ms * 1e-3) / 1e9) if (traffic and num_ms > 0) else None,
                             "symbolic_kernel_ms": per_launch("smm_symbolic"),
                             # the same bytes over the WHOLE step (symbolic + runs + scans + numeric; median step): what one
                             # product achieves, next to `frac`, which prices the dominant kernel alone as the contract asks
                             "whole_step": {"achieved": alg_bytes / (step_ms[len(step_ms) // 2] * 1e-3) / 1e9, "unit": "GB/s",
                                            "frac": alg_bytes / (step_ms[len(step_ms) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "ms": step_ms[len(step_ms) // 2]}},
                "whole_step_frac": alg_bytes / (step_ms[len(step_ms) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            })
            if keep_last:                                                 # the CPU leg runs at N = 1 only
                line["cpu_baseline"] = cpu_baseline(torch, a_t, b_t, n, (out[0], out[1], out[2]))
                del out
        elif cfg == "c2":
            alg_bytes = (4 * (m + 1) + 12 * nnz_a) + (4 * (n + 1) + 12 * nnz_b) + 8 * m * n
            num_ms = per_launch("smm_numeric_dense") + per_launch("smm_dense_slab")
            achieved = alg_bytes / (num_ms * 1e-3) / 1e9 if num_ms > 0 else 0.0
            line.update({
                "metric": "output elements/sec, CSR x CSR -> dense (float64)",
                "value": total_units * steps / elapsed, "unit": "elements/s",
                "config": {"workload": f"{m}x{n} x {n}x{n} uniform random CSR d={d} -> dense, per GPU (BASELINE configs[2])",
                           "nnz_a": nnz_a, "nnz_b": nnz_b, "mode": mode, "parallelism": par},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel_ms": num_ms,
                             "kernels_ms": {k: per_launch(k) for k in ktimes if per_launch(k) > 0},
                             "algorithmic_bytes": alg_bytes},
            })
        else:
            nn, kk = A.rows, A.cols
            fma = nnz_a * (nnz_b / kk) + nn * (nn + 1) / 2 * (nnz_a / nn)       # stage 1 products + stage 2 gather-FMAs
            line.update({
                "metric": "multiply-adds/sec, H Q H^T upper triangle (float64)",
                "value": fma * steps / elapsed, "unit": "FMA/s",
                "config": {"workload": f"H {nn}x{kk} d=0.02, Q {kk}x{kk} symmetric d~0.005 -> dense upper triangle "
                                       f"(BASELINE configs[3])", "nnz_h": nnz_a, "nnz_q": nnz_b, "mode": mode, "parallelism": par},
                "roofline": {"bound": "vector-fp64 / LDS gather (no MFMA: an indexing path)", "achieved": 2 * fma * steps / elapsed / 1e12,
                             "peak": 78.6, "unit": "TFLOP/s", "frac": 2 * fma * steps / elapsed / 1e12 / 78.6, "traffic": None,
                             "kernels_ms": {k: per_launch(k) for k in ktimes if per_launch(k) > 0}},
            })
    out = None
    ctx.synchronize()            # (also reports what the kernels' bounds clamps recorded, if anything: SMM_ERR_INTERNAL)
    A.close(); B.close()
    del a_t, b_t, out_buf
    torch.cuda.empty_cache()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c1", choices=["c1", "c1s", "c2", "c3", "c4"],
                    help="BASELINE config: c1 50k^2 -> sparse (default, the bench line), c1s the same on the literal "
                         "scipy.sparse.random operands (host-generated, uploaded), c2 -> dense, c3 triple product, "
                         "c4 one rank's share of 200k^2 per GPU")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--density", type=float, default=0.0)
    ap.add_argument("--scale", type=float, default=1.0, help="c3 only: scale both dimensions of H")
    ap.add_argument("--gather", action="store_true",
                    help="sparse configs: all-gatherv of indices/values inside the step, rows split over the ranks "
                         "(default size 200000^2 d=0.001, SURVEY 8e option ii)")
    ap.add_argument("--exact", action="store_true", help="SMM_EXACT: reference-order accumulation (bit-exact values)")
    ap.add_argument("--lds-cols", type=int, default=0)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--hash", type=str, default="", help="small,medium thresholds of the LDS-hash kernels (0,0 = off)")
    ap.add_argument("--slab", type=str, default="", help="mode,ws,rows_per_wave of the row-block x column-slab kernels "
                                                         "(mode 0 auto / 1 off / 2 force; ws 0 = L2-sized)")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1, CSR configs: after the timed region rank 0 recomputes the product of the concatenated row blocks on "
                         "one GPU and compares it with what the ranks exchanged / gathered (small sizes: rehearsals, first contact)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="default run only: skip the short runs of the other BASELINE configs (extra_configs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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

    # The library launches on torch's current stream (0 = the null stream -> SMM_STREAM_DEFAULT): the
    # generator's kernels, the product's kernels and the events below are all ordered on it.
    tstream = torch.cuda.current_stream()
    ctx = Context(local, tstream.cuda_stream)
    if args.lds_cols or args.waves:
        (ctx.tune if args.exact else ctx.tune_shared)(args.lds_cols, args.waves)
    if args.hash:
        ctx.tune_hash(*[int(x) for x in args.hash.split(",")])
    if args.slab:
        ctx.tune_slab(*[int(x) for x in args.slab.split(",")])

    env = {"torch": torch, "ctx": ctx, "dist": dist, "smm_dist": smm_dist, "rank": rank, "world": world, "device": device,
           "rehearsal": rehearsal, "tstream": tstream}
    rccl = None
    if world > 1:
        # who is in the job: backend and the PCI bus id of every rank's GPU (N distinct devices expected)
        ids = [None] * world
        dist.all_gather_object(ids, device_identity(torch, local))
        rccl = {"backend": dist.get_backend(), "world": world, "devices": ids, "distinct_devices": len(set(ids))}

    line = run_config(args.config, args, env, args.steps, args.warmup, cpu_leg=not args.no_cpu)
    if rank == 0 and rccl is not None:
        line["rccl"] = rccl

    # The default invocation (the driver's `python bench.py --gpus 1`) also times the other BASELINE configs, three
    # steps each, in this same process: parity-test cases, reported next to the headline, never part of `value`.
    default_run = (world == 1 and args.config == "c1" and not args.no_extra and not args.exact and not args.gather and
                   not (args.rows or args.cols or args.density or args.lds_cols or args.waves or args.hash or args.slab))
    if default_run:
        extra = {}
        for name, cfg in (("c1_scipy", "c1s"), ("c2", "c2"), ("c3", "c3"), ("c4_share", "c4")):
            t0 = time.perf_counter()
            try:
                r = run_config(cfg, args, env, 3, 1, cpu_leg=False)
                extra[name] = {"ms_per_step": r["ms_per_step"], "ms_per_step_best": r["ms_per_step_best"], "metric": r["metric"],
                               "value": r["value"], "unit": r["unit"], "workload": r["config"]["workload"],
                               "roofline": {k: r["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "kernels_ms", "whole_step")},
                               "wall_s_incl_input_generation": time.perf_counter() - t0}
            except Exception as e:                           # the headline stands on its own
                extra[name] = {"error": repr(e)}
        line["extra_configs"] = extra
    if rank == 0:
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
