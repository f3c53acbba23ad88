"""N > 1 path on CPU: two `gloo` processes run the row-partition protocol of
sgm_csr_create_dist / the distributed CG with the library's own host-only
halo planner (sgm_halo_plan_host, no GPU call) and the CPU oracle as the local arithmetic:

  * want-count all-gather + index-list swap  (what sgm_csr_create_dist does over RCCL)
  * per matvec: gather send list -> send/recv with the neighbour rank -> local matvec on
    [owned | halo]            (row sums must be BIT-identical to the serial matvec)
  * per dot: local partial + all_reduce(sum)   (solution within 1e-12 of the serial CG)
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        if case == "poisson2d":
            nx, ny = 24, 20
            n = nx * ny
            ptr, node, val = P.poisson2d_csr(nx, ny)
        else:
            n = 9 * 8 * 10
            ptr, node, val = P.laplace3d_csr(9, 8, 10)
        starts = (np.arange(world + 1) * n // world) // 2 * 2
        starts[-1] = n
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        n_own = r1 - r0
        k0, k1 = ptr[r0] - 1, ptr[r1] - 1
        lptr = (ptr[r0:r1 + 1] - k0).astype(np.int32)
        # --- index work: the library's host planner (bit-exact vs numpy in test_cabi_cpu)
        lnode, halo = sg.halo_plan_host(n_own, r0, node[k0:k1])
        A_loc = orc.CsrMatrix(n_own, n_own + len(halo), lptr, lnode, val[k0:k1])
        owner = np.searchsorted(starts, halo - 1, side="right") - 1
        # --- want counts all-gather, then neighbours swap index lists
        want = np.array([(owner == q_).sum() for q_ in range(world)], dtype=np.int64)
        allw = [torch.zeros(world, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allw, torch.from_numpy(want))
        allw = torch.stack(allw).numpy()                       # allw[q][me] = q wants this many of mine
        send_idx = {}
        reqs = []
        for q_ in range(world):
            if q_ == rank:
                continue
            if want[q_]:
                req = torch.from_numpy((halo[owner == q_] - 1 - starts[q_]).astype(np.int64))
                reqs.append(dist.isend(req, q_))
            if allw[q_][rank]:
                buf = torch.zeros(int(allw[q_][rank]), dtype=torch.int64)
                reqs.append(dist.irecv(buf, q_))
                send_idx[q_] = buf
        for r in reqs:
            r.wait()

        def exchange(xext):
            rq, bufs = [], {}
            for q_, idx in send_idx.items():
                rq.append(dist.isend(torch.from_numpy(xext[idx.numpy()].copy()), q_))
            for q_ in range(world):
                if q_ != rank and want[q_]:
                    bufs[q_] = torch.zeros(int(want[q_]), dtype=torch.float64)
                    rq.append(dist.irecv(bufs[q_], q_))
            for r in rq:
                r.wait()
            for q_, b in bufs.items():
                xext[n_own:][owner == q_] = b.numpy()

        def matvec(v_own):
            xext = np.zeros(n_own + len(halo))
            xext[:n_own] = v_own
            exchange(xext)
            return A_loc.matvec(xext)

        def gdot(a, b):
            t = torch.tensor([orc.lib().orc_dot(len(a), a.ctypes.data_as(__import__("ctypes").c_void_p),
                                                 b.ctypes.data_as(__import__("ctypes").c_void_p))],
                             dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())

        # --- matvec: bit-identical rows
        A = orc.CsrMatrix(n, n, ptr, node, val)
        x = P.test_vector(n)
        y_loc = matvec(x[r0:r1].copy())
        ok_mv = bool(np.array_equal(y_loc, A.matvec(x)[r0:r1]))
        # --- CG (cg_solvers.f90:128-146) with distributed dots
        b = np.full(n, 1.0 / n)
        bl = b[r0:r1].copy()
        xl = np.zeros(n_own)
        q_ = matvec(xl)
        r = bl - q_
        p = r.copy()
        res2 = gdot(r, r)
        its = 0
        tol = 1e-13
        while np.sqrt(res2) > tol:
            q_ = matvec(p)
            alpha = res2 / gdot(p, q_)
            xl = xl + alpha * p
            r = r - alpha * q_
            dpr = gdot(r, r)
            p = r + (dpr / res2) * p
            res2 = dpr
            its += 1
        xr, itr, _, _ = orc.cg(A, b, tol=tol)
        rel = float(np.abs(xl - xr[r0:r1]).max() / np.abs(xr).max())
        q.put((rank, ok_mv, its, itr, rel, len(halo)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["poisson2d", "laplace3d"])
def test_two_rank_gloo_partition_protocol(case):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_mv, its, itr, rel, nh in res:
        assert nh > 0
        assert ok_mv, f"rank {rank}: partitioned matvec rows differ from the serial matvec"
        assert abs(its - itr) <= 1
        assert rel <= 1e-12
