#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X: CSR SpMV achieved GB/s (+ CG iterations/s)
on the 5-point 2-D Poisson matrix, n = 3162^2 = 9,998,244 rows per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the synthetic matrix: y = A x (halo exchange
included when N > 1).  Inputs are resident in HBM before the timed region.  value =
algorithmic bytes of all ranks (12 nnz + 4 (n+1) + 8 m + 8 n each) / max-over-
ranks wall time.  N > 1 is weak scaling: every rank owns nx*ny rows of an nx x (N*ny) grid
(contiguous row blocks; one xy-line of halo to each neighbour over RCCL).
The same run then times K CG iterations (reported under "cg"), the dominant kernel with HIP
events on the launch stream ("roofline"), and -- rank 0, N = 1 only -- the CPU oracle on
the host cores on a bounded sample ("cpu_baseline").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def spmv_bytes(n, m, nnz):
    return 12 * nnz + 4 * (n + 1) + 8 * m + 8 * n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nx", type=int, default=3162)
    ap.add_argument("--ny", type=int, default=3162)
    ap.add_argument("--cg-steps", type=int, default=100)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"],
                    help="c2: 5-point 2-D Poisson, nx*ny rows PER GPU (weak scaling, the BASELINE metric); "
                         "c5: 7-point 3-D Laplacian m^3 (default 464^3) split in z-slabs over the GPUs (strong scaling)")
    ap.add_argument("--c5-edge", type=int, default=464, help="grid edge m of the c5 workload (m^3 rows)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the RCCL row-partition code path even with one rank (testing aid)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sigma_amd as sg
    from sigma_amd import problems as P

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SGM_BENCH_SAME_GPU"):      # testing aid: all ranks on device 0 (if RCCL allows it)
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    sg.init(local_rank)
    dev = torch.device("cuda", local_rank)
    # every kernel of the library is launched on THIS torch stream, so torch.cuda.Event
    # (HIP events) brackets exactly the launches it is recorded around
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    sg.use_torch_stream()
    sg.set_async(True)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        # torch.distributed is the CONTROL plane only (bootstrap of the RCCL id, barriers, the
        # max-over-ranks of the timings) and runs over gloo on the host, so that the data
        # plane -- the library's own RCCL communicator (halo send/recv + dot all-reduces on the
        # launch stream) -- is the only RCCL communicator of the process.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def sum_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- synthetic matrix ------------------------------------------------------------------
    nx, ny = args.nx, args.ny
    ptr = node = val = None
    if args.workload == "c2":
        # this rank's nx*ny rows of the nx x (world*ny) 5-point grid (weak scaling)
        n_loc = nx * ny
        n_glob = n_loc * world
        i0 = rank * n_loc
        starts = np.arange(world + 1, dtype=np.int64) * n_loc
        if not use_dist:
            ptr, node, val = P.poisson2d_csr(nx, ny)
            arrays = (torch.from_numpy(ptr).to(dev), torch.from_numpy(node).to(dev), torch.from_numpy(val).to(dev))
        else:
            arrays = local_rows_poisson2d(nx, ny, world, rank)
        workload = (f"5-point 2D Poisson CSR, {nx}x{ny} rows per GPU", "weak")
    else:
        # z-slabs of the m^3 7-point grid (strong scaling), generated on the device
        m = args.c5_edge
        zs = [(m * r) // world for r in range(world + 1)]
        starts = np.array([z * m * m for z in zs], dtype=np.int64)
        i0, n_loc, n_glob = int(starts[rank]), int(starts[rank + 1] - starts[rank]), m ** 3
        arrays = local_rows_laplace3d(m, int(zs[rank]), int(zs[rank + 1]), dev)
        workload = (f"7-point 3D Laplacian CSR {m}^3 split in z-slabs over {world} GPU(s)", "strong")
    nnz = int(arrays[2].numel() if hasattr(arrays[2], "numel") else len(arrays[2]))
    if not use_dist:
        if args.workload == "c5":   # global == local numbering on one GPU
            A = sg.csr_matrix(n_loc, n_loc, *arrays)
        else:
            A = sg.csr_matrix(n_loc, n_loc, *arrays)
        x_len = n_loc
    else:
        uid = [sg.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = sg.Comm(rank, world, uid[0])
        A = sg.dist_csr_matrix(comm, starts, *arrays)
        x_len = A.x_len
    del arrays
    x = torch.zeros(x_len, dtype=torch.float64, device=dev)
    i0 = rank * n_loc
    x[:n_loc] = torch.sin(0.001 * torch.arange(i0 + 1, i0 + n_loc + 1, dtype=torch.float64, device=dev))
    y = torch.zeros(n_loc, dtype=torch.float64, device=dev)
    bytes_rank = spmv_bytes(n_loc, n_loc, nnz)

    # ---- timed region: K SpMV steps --------------------------------------------------------
    for _ in range(args.warmup):
        A.matvec(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        A.matvec(x, y)
    barrier()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt)
    ms_per_step = 1e3 * dt / args.steps
    bytes_all = sum_over_ranks(bytes_rank)
    value = bytes_all * args.steps / dt / 1e9

    # ---- dominant kernel with HIP events on the launch stream -----------------------------
    def time_kernel(mat, reps=50):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            mat.matvec(x, y)
            b.record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) * 1e-3

    k_avg = time_kernel(A)
    main_kernel = A.kernel
    achieved = bytes_rank / k_avg / 1e9
    # the same matrix through the other CSR kernels (what matrices that do not qualify for the
    # default form get), for comparison; not part of `value`
    variants = {"sliced 4-bit codes, 2 rows per lane (default for rows <= 8 entries / <= 15 offsets)":
                {"kernel": main_kernel, "avg_launch_ms": 1e3 * k_avg, "GB/s_algorithmic": achieved,
                 "frac_of_hbm_peak": achieved / HBM_PEAK_GBS}}
    if not use_dist:
        for label, opts in (("offset_dict_u8_codes, LDS-staged row-owner kernel (stencil-like matrices with longer rows / more offsets)",
                             {"csr_sliced": 0}),
                            ("int32_columns, row-owner gather (general kernel, rows <= 32 entries)",
                             {"csr_offset_dict": 0, "csr_row_owner": 1}),
                            ("int32_columns, streaming gather (general kernel, any row length)",
                             {"csr_offset_dict": 0, "csr_row_owner": 0})):
            for k, v in opts.items():
                sg.set_option(k, v)
            for _ in range(5):
                A.matvec(x, y)
            kv = time_kernel(A)
            variants[label] = {"kernel": A.kernel, "avg_launch_ms": 1e3 * kv, "GB/s_algorithmic": bytes_rank / kv / 1e9,
                               "frac_of_hbm_peak": bytes_rank / kv / 1e9 / HBM_PEAK_GBS}
            sg.set_option("csr_offset_dict", 1)
            sg.set_option("csr_row_owner", 1)
            sg.set_option("csr_sliced", 1)
        # a handle created WITHOUT the dictionary: short rows take the sliced int32-column kernel
        # (what a matrix with arbitrary columns and rows <= 16 entries gets; 12 B per slot)
        sg.set_option("csr_offset_dict", 0)
        try:
            A32 = sg.csr_matrix(n_loc, n_loc, A.get("ptr", np.int32), A.get("node", np.int32), A.get("val", np.float64))
        finally:
            sg.set_option("csr_offset_dict", 1)
        for _ in range(5):
            A32.matvec(x, y)
        kv = time_kernel(A32)
        variants["int32_columns, sliced (general kernel, rows <= 16 entries of similar length)"] = {
            "kernel": A32.kernel, "avg_launch_ms": 1e3 * kv, "GB/s_algorithmic": bytes_rank / kv / 1e9,
            "frac_of_hbm_peak": bytes_rank / kv / 1e9 / HBM_PEAK_GBS}
        A32.destroy()

    # ---- CG iterations/s (device-resident loop, fixed iteration count) -------------------
    cg = None
    if args.cg_steps > 0:
        s = sg.cg(1e-300)
        s.set_max_iter(args.cg_steps)
        s.setup(A)
        bvec = torch.full((n_loc,), 1.0 / n_glob, dtype=torch.float64, device=dev)
        u = torch.zeros(n_loc, dtype=torch.float64, device=dev)
        s.solve(A, u, bvec, check=False)       # warm-up
        u.zero_()
        barrier()
        t0 = time.perf_counter()
        s.solve(A, u, bvec, check=False)
        barrier()
        dtc = time.perf_counter() - t0
        dtc = max_over_ranks(dtc)
        its = s.last_iterations
        cg_bytes = sum_over_ranks(bytes_rank + 72 * n_loc)     # SURVEY §8d fused floor B_csr + 72 n
        cg = {"iters_per_s": its / dtc, "iterations": its, "ms_per_iter": 1e3 * dtc / its,
              "bytes_per_iter": cg_bytes, "GB/s": cg_bytes * its / dtc / 1e9,
              "frac_of_hbm_peak": cg_bytes * its / dtc / 1e9 / (HBM_PEAK_GBS * world),
              "final_res2": s.res2}

    # ---- CPU baseline: the oracle (1 thread, like the reference) on a bounded sample -------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and ptr is not None:
        import oracle as orc
        # the SAME matrix as the GPU workload (800 MB per matvec: far beyond any host cache)
        Ao = orc.CsrMatrix(n_loc, n_loc, ptr, node, val)
        reps = 20
        sec = orc.time_csr_matvec(Ao, P.test_vector(n_loc), reps)
        cpu = {"value": spmv_bytes(Ao.n, Ao.n, Ao.nnz) / sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
               "sample": f"the full workload matrix (n={Ao.n}, nnz={Ao.nnz}), {reps} matvecs of "
                         f"oracle/sigma_oracle.c (csr_matvec_add restatement), {sec * 1e3:.1f} ms each; "
                         f"host has {os.cpu_count()} logical cores, the reference is single-threaded"}
        # SURVEY 8(d) "(ii) all cores": the same row loop under one OpenMP pragma (rows bit-identical)
        sec_omp, nthreads, y_omp = orc.time_csr_matvec_omp(Ao, P.test_vector(n_loc), reps)
        cpu["all_cores_openmp"] = {"GB/s": spmv_bytes(Ao.n, Ao.n, Ao.nnz) / sec_omp / 1e9, "threads": nthreads,
                                   "ms_per_matvec": 1e3 * sec_omp,
                                   "note": "arrays are numpy allocations first touched by one thread (one NUMA node); "
                                           "the thread count is the pod's OpenMP default, not a tuned placement",
                                   "rows_equal_single_thread": bool(np.array_equal(y_omp, Ao.matvec(P.test_vector(n_loc))))}
        # the REAL reference (compiled in place by oracle/build_ref.sh; the binary travels with
        # the snapshot, the sources do not), timed on a bounded sample of the same workload
        ref = reference_cpu_baseline()
        if ref:
            ref["port_on_full_workload"] = {"GB/s": cpu["value"], "sample": cpu["sample"],
                                            "all_cores_openmp": cpu["all_cores_openmp"]}
            cpu = ref

    # HBM bytes per launch from the PMC counters cannot be collected inside this process; they
    # come from the committed rocprofv3 --pmc passes over this same command (profiles/)
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, "profiles", "r01", "pmc_hbm_traffic.json")
    if os.path.exists(tf) and world == 1 and (nx, ny) == (3162, 3162) and args.workload == "c2":
        tj = json.load(open(tf))
        # only if the committed counters belong to the kernel that ran here
        if main_kernel.split("<")[0] in tj.get("dominant_kernel", ""):
            traffic, traffic_src = tj.get("hbm_traffic_bytes"), "profiles/r01/pmc_hbm_traffic.json"

    if rank == 0:
        out = {
            "metric": "SpMV GB/s (achieved HBM) + CG iters/sec on 5-pt Laplacian, N=1e7",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": workload[1], "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{workload[0]} (rank 0: n={n_loc}, nnz={nnz}), fp64 SpMV y=A*x",
                       "rows_per_gpu": n_loc, "nnz_per_gpu": int(nnz),
                       "parallelism": f"row-partition x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": main_kernel, "algorithmic_bytes_per_launch": bytes_rank,
                         "avg_launch_ms": 1e3 * k_avg, "traffic_source": traffic_src,
                         "note": "achieved = algorithmic bytes (12 nnz + 4(n+1) + 16 n, the reference's int32/fp64 "
                                 "arrays) / measured launch time; the kernel streams 4-bit column codes and no row pointers, "
                                 "so the HBM bytes it really moves (`traffic`) are below the algorithmic count"},
            "spmv_variants": variants, "cg": cg, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_dist:
        barrier()
        dist.destroy_process_group()


def reference_cpu_baseline(nx_mv=2000, reps=20, nx_cg=600):
    """cpu_baseline of kind "reference": oracle/_ref/sigma_ref_driver (our driver program linked
    against the reference's own modules) times A%matvec on the nx_mv^2 5-point Laplacian and one
    unpreconditioned CG solve on nx_cg^2 -- about 15 s of one host core."""
    import struct
    import subprocess
    import tempfile
    from sigma_amd import problems as P
    drv = os.path.join(ROOT, "oracle", "_ref", "sigma_ref_driver")
    if not os.path.exists(drv):
        return None

    def run(nx, solves, mode):
        n = nx * nx
        ei, ej, ev = P.poisson2d_edges(nx, nx)
        with tempfile.TemporaryDirectory() as td:
            inp = os.path.join(td, "in.bin")
            with open(inp, "wb") as f:
                f.write(struct.pack("<5i", n, n, len(ei), 1, len(solves)))
                f.write(np.asarray(ei, "<i4").tobytes())
                f.write(np.asarray(ej, "<i4").tobytes())
                f.write(np.asarray(ev, "<f8").tobytes())
                f.write(np.asarray(P.test_vector(n), "<f8").tobytes())
                f.write(np.full(n, 1.0 / n, "<f8").tobytes())
                for (sk, pk, tol) in solves:
                    f.write(struct.pack("<iid", sk, pk, tol))
            out = subprocess.run([drv, inp, os.path.join(td, "o"), mode], capture_output=True, text=True, timeout=300)
        return n, len(ei), out.stdout

    try:
        n, nnz, txt = run(nx_mv, [], f"time:{reps}")
        sec = float(txt.split("matvec_seconds_each=")[1].split()[0])
        n2, _, txt2 = run(nx_cg, [(1, 0, 1e-8)], "time:1")
        line = [ln for ln in txt2.splitlines() if ln.startswith("solve 1:")][0]
        its = int(line.split("iterations=")[1].split()[0])
        cg_sec = float(line.split("seconds=")[1].split()[0])
    except Exception as e:        # a baseline leg must never take the bench line down
        sys.stderr.write(f"[bench] reference baseline skipped: {e}\n")
        return None
    return {"value": spmv_bytes(n, n, nnz) / sec / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
            "sample": f"danshapero/sigma itself (amdflang -O2, oracle/build_ref.sh): csr A%matvec on the "
                      f"{nx_mv}^2 5-point Laplacian (n={n}, nnz={nnz}), {reps} calls, {sec * 1e3:.2f} ms each; "
                      f"cg%solve on {nx_cg}^2 to 1e-8: {its} iterations in {cg_sec:.2f} s",
            "cg_iters_per_s": its / cg_sec if cg_sec > 0 else None, "cg_n": n2}


def local_rows_laplace3d(m, z0, z1, dev):
    """Rows of the planes [z0, z1) of the m^3 7-point grid, built on the device: local 1-based
    ptr, GLOBAL 1-based node, val; entry order -z,-y,-x,C,+x,+y,+z like problems.laplace3d_csr."""
    import torch
    pl = m * m
    k = torch.arange(z0 * pl, z1 * pl, device=dev, dtype=torch.int64)
    i, j, l = k % m, (k // m) % m, k // pl
    one = torch.ones_like(k, dtype=torch.bool)
    offs = [(-pl, l > 0, -1.0), (-m, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 6.0), (1, i < m - 1, -1.0),
            (m, j < m - 1, -1.0), (pl, l < m - 1, -1.0)]
    mask = torch.stack([mk for _, mk, _ in offs], dim=1)
    ptr = torch.ones(k.numel() + 1, dtype=torch.int64, device=dev)
    ptr[1:] += torch.cumsum(mask.sum(dim=1), 0)
    cols = torch.stack([k + 1 + o for o, _, _ in offs], dim=1)[mask].to(torch.int32)
    vals = torch.tensor([v for _, _, v in offs], dtype=torch.float64, device=dev).expand(k.numel(), -1)[mask].contiguous()
    return ptr.to(torch.int32), cols, vals


def local_rows_poisson2d(nx, ny, world, rank):
    """Rows [rank*nx*ny, (rank+1)*nx*ny) of the nx x (world*ny) 5-point grid: local 1-based
    ptr, GLOBAL 1-based node, val -- same insertion order S,W,C,E,N as problems.poisson2d_csr."""
    n_loc = nx * ny
    k = np.arange(rank * n_loc, (rank + 1) * n_loc, dtype=np.int64)
    i, j = k % nx, k // nx
    NY = ny * world
    offs = [(-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, np.ones(n_loc, bool), 4.0), (1, i < nx - 1, -1.0),
            (nx, j < NY - 1, -1.0)]
    cols = np.stack([k + 1 + o for o, _, _ in offs], axis=1)
    mask = np.stack([m for _, m, _ in offs], axis=1)
    vals = np.broadcast_to(np.array([v for _, _, v in offs]), cols.shape)
    ptr = np.concatenate([[1], 1 + np.cumsum(mask.sum(axis=1))]).astype(np.int32)
    return ptr, cols[mask].astype(np.int32), vals[mask].astype(np.float64)


if __name__ == "__main__":
    main()
