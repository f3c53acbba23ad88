"""N > 1 path on CPU: `gloo` processes run the row-partition protocol of sgm_csr_create_dist /
the distributed CG.  Every list RCCL would carry comes from the PRODUCT's own
host-only planners (the code sgm_csr_create_dist runs, exported for this purpose):

  sgm_halo_plan_host       sorted unique halo list + renumbering to [owned | halo]
  sgm_dist_plan_host       want counts per owner, request list in the owner's numbering
  sgm_dist_neighbors_host  neighbour table (peer, send/recv counts, halo offset) from the
                           all-gathered want matrix
  sgm_partition_links_host what sgm_csr_create_partitioned builds in one process: the send lists
                           received over gloo must equal these bit for bit

Only the transport (gloo instead of RCCL) and the local arithmetic (the CPU oracle instead of the
HIP kernels) differ from the GPU run:
  * want-count all-gather + request-list swap (what sgm_csr_create_dist does over RCCL)
  * per matvec: gather send list -> send/recv with the neighbour ranks -> local matvec on
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
        starts = sg.partition_rows_by_nnz(ptr, world, align=2)          # product partitioner
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        n_own = r1 - r0
        k0, k1 = ptr[r0] - 1, ptr[r1] - 1
        lptr = (ptr[r0:r1 + 1] - k0).astype(np.int32)
        # --- index work: the library's host planners
        lnode, halo = sg.halo_plan_host(n_own, r0, node[k0:k1])
        A_loc = orc.CsrMatrix(n_own, n_own + len(halo), lptr, lnode, val[k0:k1])
        want, want_off, req = sg.dist_plan_host(rank, world, starts, halo)
        # --- want counts all-gather (RCCL: ncclAllGather), then neighbours swap request lists
        allw = [torch.zeros(world, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(allw, torch.from_numpy(want.copy()))
        want_all = torch.stack(allw).numpy()
        nbrs = sg.dist_neighbors_host(rank, world, want_all)       # (peer, send_count, recv_count, recv_offset)
        send_idx = {}
        reqs = []
        for peer, sc, rc, ro in nbrs:                                # RCCL: one grouped ncclSend/ncclRecv
            if rc:
                reqs.append(dist.isend(torch.from_numpy(req[ro:ro + rc].copy()), peer))
            if sc:
                send_idx[peer] = torch.zeros(sc, dtype=torch.int32)
                reqs.append(dist.irecv(send_idx[peer], peer))
        for r in reqs:
            r.wait()
        # the lists this rank received must be the ones the in-process partition builds
        links = [l for l in sg.partition_links_host(starts, ptr, node) if l["sender"] == rank]
        ok_links = len(links) == len(send_idx)
        for l in links:
            got = send_idx.get(l["receiver"])
            ok_links = ok_links and got is not None and np.array_equal(got.numpy(), l["send_idx"])
            mine = [t for t in sg.partition_links_host(starts, ptr, node) if t["receiver"] == rank and t["sender"] == l["receiver"]]
            for t in mine:      # and my halo offsets are the receiver-side offsets of those links
                ok_links = ok_links and any(p_ == t["sender"] and ro_ == t["recv_offset"] and rc_ == len(t["send_idx"])
                                            for p_, _, rc_, ro_ in nbrs)

        def exchange(xext):
            rq, bufs = [], {}
            for peer, sc, rc, ro in nbrs:
                if sc:
                    rq.append(dist.isend(torch.from_numpy(xext[send_idx[peer].numpy()].copy()), peer))
                if rc:
                    bufs[peer] = (torch.zeros(rc, dtype=torch.float64), ro, rc)
                    rq.append(dist.irecv(bufs[peer][0], peer))
            for r in rq:
                r.wait()
            for peer, (b_, ro, rc) in bufs.items():
                xext[n_own + ro:n_own + ro + rc] = b_.numpy()

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
        q.put((rank, ok_mv, its, itr, rel, len(halo), bool(ok_links)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("poisson2d", 2), ("laplace3d", 2), ("laplace3d", 3)])
def test_gloo_partition_protocol_on_product_planners(case, world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_mv, its, itr, rel, nh, ok_links in res:
        assert nh > 0
        assert ok_links, f"rank {rank}: send lists over gloo differ from sgm_csr_create_partitioned's links"
        assert ok_mv, f"rank {rank}: partitioned matvec rows differ from the serial matvec"
        assert abs(its - itr) <= 1
        assert rel <= 1e-12


def test_planners_agree_without_any_transport():
    """dist_plan + neighbours for every rank, with the want matrix assembled locally, reproduce
    the in-process links (what the gloo test checks, for more ranks and an uneven partition)."""
    import sigma_amd as sg
    from sigma_amd import problems as P
    ptr, node, val = P.laplace3d_csr(7, 6, 11)
    n = 7 * 6 * 11
    for world in (2, 3, 5, 8):
        starts = sg.partition_rows_by_nnz(ptr, world, align=2)
        assert starts[0] == 0 and starts[-1] == n and np.all(np.diff(starts) > 0) and np.all(starts[:-1] % 2 == 0)
        plans = []
        for r in range(world):
            r0, r1 = int(starts[r]), int(starts[r + 1])
            _, halo = sg.halo_plan_host(r1 - r0, r0, node[ptr[r0] - 1: ptr[r1] - 1])
            plans.append((halo,) + sg.dist_plan_host(r, world, starts, halo))
        want_all = np.stack([p[1] for p in plans])
        links = sg.partition_links_host(starts, ptr, node)
        seen = 0
        for r in range(world):
            for peer, sc, rc, ro in sg.dist_neighbors_host(r, world, want_all):
                if rc:      # what r asks of peer == the link peer -> r
                    l = [t for t in links if t["sender"] == peer and t["receiver"] == r]
                    assert len(l) == 1 and l[0]["recv_offset"] == ro
                    assert np.array_equal(l[0]["send_idx"], plans[r][3][ro:ro + rc])
                    halo = plans[r][0]
                    assert np.array_equal(halo[ro:ro + rc] - 1 - starts[peer], l[0]["send_idx"])
                    seen += 1
                assert sc == want_all[peer][r]
        assert seen == len(links)


def test_partition_rows_by_nnz_balances_bytes():
    import sigma_amd as sg
    rs_ = np.random.RandomState(3)
    n = 20000
    cnt = rs_.randint(0, 40, size=n)
    cnt[:2000] = 200            # a dense head: equal-row blocks would be badly unbalanced
    ptr = np.concatenate([[1], 1 + np.cumsum(cnt)]).astype(np.int32)
    for parts in (2, 4, 8):
        starts = sg.partition_rows_by_nnz(ptr, parts, align=16)
        assert starts[0] == 0 and starts[-1] == n and np.all(starts[1:-1] % 16 == 0)
        w = np.array([12 * (ptr[starts[i + 1]] - ptr[starts[i]]) + 20 * (starts[i + 1] - starts[i]) for i in range(parts)], float)
        assert w.max() / w.mean() < 1.05


def test_rectangular_leaf_planners_without_any_transport():
    """An off-diagonal block of a composite (sgm_csr_create_dist_rect): rows partitioned by one list, x by another.
    The same planners, fed the COLUMN partition, give every rank its halo and the owners their request lists;
    the rows computed on [owned slice of x | halo] equal the serial block product bit for bit."""
    import scipy.sparse as sp
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P
    ptr, node, val = P.laplace3d_csr(9, 8, 10)
    n = 720
    Asp = sp.csr_matrix((val, node - 1, ptr - 1), shape=(n, n))
    m = 302
    B = Asp[:m, m:].tocsr()                      # 302 x 418: rows of block row 1, columns of block column 2
    B.sort_indices()
    Bo = orc.CsrMatrix(m, n - m, (B.indptr + 1).astype(np.int32), (B.indices + 1).astype(np.int32), B.data.copy())
    x = np.random.RandomState(2).standard_normal(n - m)
    y_ref = Bo.matvec(x)
    for world in (2, 3, 5):
        rs = sg.partition_rows_by_nnz((Asp[:m].indptr + 1).astype(np.int32), world, align=2)
        cs = sg.partition_rows_by_nnz((Asp[m:].indptr + 1).astype(np.int32), world, align=2)
        assert rs[-1] == m and cs[-1] == n - m and not np.array_equal(rs, cs)
        asked = {q: [] for q in range(world)}
        for r in range(world):
            L = B[int(rs[r]):int(rs[r + 1])]
            c0, nc = int(cs[r]), int(cs[r + 1] - cs[r])
            lnode, halo = sg.halo_plan_host(nc, c0, (L.indices + 1).astype(np.int32))
            assert np.all((halo - 1 < c0) | (halo - 1 >= c0 + nc)) and np.all(np.diff(halo) > 0)
            want, want_off, req = sg.dist_plan_host(r, world, cs, halo)
            assert want[r] == 0 and want.sum() == len(halo)
            for q in range(world):
                mine = req[want_off[q]:want_off[q] + want[q]]
                glob = halo[want_off[q]:want_off[q] + want[q]] - 1
                assert np.all((glob >= cs[q]) & (glob < cs[q + 1])), "a halo column was asked of the wrong owner"
                assert np.array_equal(mine, (glob - cs[q]).astype(np.int32))
                asked[q].append(len(mine))
            xext = np.concatenate([x[c0:c0 + nc], x[halo - 1]])
            A_loc = orc.CsrMatrix(L.shape[0], nc + len(halo), (L.indptr + 1).astype(np.int32), lnode, L.data.copy())
            assert np.array_equal(A_loc.matvec(xext), y_ref[int(rs[r]):int(rs[r + 1])]), (world, r)


def test_ell_degrees_recovered_from_the_padding_equal_the_graph_builds():
    """sgm_ell_degrees_host (what sgm_ell_create / sgm_ell_create_dist run: the product's interface has no degrees argument)
    against the degrees the oracle's restatement of ellpack_graph_build keeps (ellpack_graphs.f90:105-170): random graphs with
    ragged rows (padding = the last neighbour repeated), full rows, rows without any edge (node(:, i) = 0), a one-slot graph."""
    sys.path.insert(0, ROOT)
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P
    rs = np.random.RandomState(5)
    cases = [P.random_regular_ell(700, 12, 99, dmin=3), P.random_regular_ell(300, 8, 7, dmin=None), P.poisson2d_edges(17, 9)]
    n = 400
    i = np.repeat(np.arange(n), 3)
    j = np.clip(i + np.tile([-4, 0, 4], n), 0, n - 1)
    keep = i % 5 != 2                       # every fifth row has no edge at all
    key, first = np.unique(i[keep].astype(np.int64) * n + j[keep], return_index=True)
    cases.append((i[keep][np.sort(first)] + 1, j[keep][np.sort(first)] + 1, rs.standard_normal(first.size)))
    cases.append((np.arange(1, 51), np.arange(1, 51), np.ones(50)))
    for ei, ej, ev in cases:
        ne = int(max(np.max(ei), np.max(ej)))
        E = orc.EllMatrix.from_edges(ne, ne, ei, ej, ev)
        assert np.array_equal(sg.ell_degrees_host(E.node), E.degrees)
    assert sg.ell_degrees_host(np.zeros((0, 3), np.int32)).size == 0


@pytest.mark.parametrize("nparts", [1, 2, 3, 5])
def test_rows_of_a_left_permuted_matrix_cut_for_every_rank(nparts):
    """sgm_left_permute_rows_host (what A%left_permute(p) on a matrix distributed over ranks runs after the gather): for every
    rank's row range the rows of the oracle's permuted matrix (cs_matrices.f90:471-478), entries in their stored order, values
    with them; a p that is not a permutation is refused."""
    sys.path.insert(0, ROOT)
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P
    n = 1200
    A = orc.CsrMatrix.from_edges(n, n, *P.random_spd_edges(n, seed=5, skew=True))
    p = (np.random.RandomState(3).permutation(n) + 1).astype(np.int32)
    Ap = orc.permuted(A, p, None)
    starts = sg.partition_rows_by_nnz(A.ptr, nparts, align=2)
    for k in range(nparts):
        r0, r1 = int(starts[k]), int(starts[k + 1])
        lptr, lnode, lval = sg.left_permute_rows_host(p, A.ptr, A.node, A.val, r0, r1)
        k0, k1 = Ap.ptr[r0] - 1, Ap.ptr[r1] - 1
        assert np.array_equal(lptr, Ap.ptr[r0:r1 + 1] - k0)
        assert np.array_equal(lnode, Ap.node[k0:k1]) and np.array_equal(lval, Ap.val[k0:k1])
    bad = p.copy()
    bad[5] = bad[6]
    with pytest.raises(sg.SigmaError):
        sg.left_permute_rows_host(bad, A.ptr, A.node, A.val, 0, 10)
