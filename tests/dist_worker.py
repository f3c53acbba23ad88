"""One rank of the multi-process GPU tests (started by tests/test_gpu_multirank.py, one process
per rank).  Runs the PRODUCT's row-partitioned path -- sgm_comm_init, sgm_csr_create_dist, halo
exchange on the communication stream, all-reduced dots inside the device-resident Krylov loops --
and checks it against the CPU oracle.  With SGM_RCCL_LIB pointing at tests/mock_rccl the ranks
share ONE GPU (RCCL itself refuses that); with real RCCL every rank takes its own device.

    python tests/dist_worker.py RANK WORLD PORT CASE OUT.json
"""
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import faulthandler
    faulthandler.enable()          # a crash inside the library names the Python line that called it (the log of this rank)
    rank, world, port, case, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    res = {"rank": rank, "ok": False}
    try:
        run(rank, world, port, case, res)
        res["ok"] = True
    except Exception:
        res["error"] = traceback.format_exc()
    with open(out, "w") as f:
        json.dump(res, f)
    sys.exit(0 if res["ok"] else 1)


def run(rank, world, port, case, res):
    import numpy as np
    import torch
    import torch.distributed as dist
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P

    same_gpu = bool(os.environ.get("SGM_RCCL_LIB"))
    device = 0 if same_gpu else rank
    sg.init(device)
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        uid = [sg.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        comm = sg.Comm(rank, world, uid[0])
        res["group_ok"] = comm.group_ok          # sgm_comm_init's probe: pairs + all-reduce in one group on this transport

        reorder_only = case.endswith("+reorder")
        case = case.replace("+reorder", "")
        if case == "composite":
            run_composite(rank, world, comm, dev, res)
            comm.destroy()
            return
        if case.startswith("fuzz:"):
            _, seed0, count = case.split(":")
            run_fuzz(rank, world, comm, res, int(seed0), int(count))
            comm.destroy()
            return
        if case == "badrows":
            # rank 1 hands over rows whose ptr does not match nnz: EVERY rank must come back with an error (the verdict
            # travels in the all-gather) instead of rank 1 leaving and its peers waiting in the collective for ever
            n = 96 * 70
            ptr, node, val = P.poisson2d_csr(96, 70)
            starts = sg.partition_rows_by_nnz(ptr, world, align=2)
            r0, r1 = int(starts[rank]), int(starts[rank + 1])
            k0, k1 = ptr[r0] - 1, ptr[r1] - 1
            lptr = (ptr[r0:r1 + 1] - k0).astype(np.int32)
            if rank == 1:
                lptr[-1] += 3
            try:
                sg.dist_csr_matrix(comm, starts, lptr, np.ascontiguousarray(node[k0:k1]), np.ascontiguousarray(val[k0:k1]))
                raise AssertionError("sgm_csr_create_dist accepted rows whose ptr does not match nnz")
            except sg.SigmaError as e:
                msg = str(e)
                assert ("ptr(n+1)-1" in msg) if rank == 1 else ("rank 1 rejected its rows" in msg), msg
            res["n_halo"] = 1
            res["solves"] = {}
            comm.destroy()
            return
        if case == "poisson2d":
            n = 96 * 70
            ptr, node, val = P.poisson2d_csr(96, 70)
        elif case == "laplace3d":
            n = 24 * 20 * 30
            ptr, node, val = P.laplace3d_csr(24, 20, 30)
        elif case == "longrows":    # general SPD matrix, rows of ~150 entries inside a band: the long-row kernels on the
            import scipy.sparse as sp    # interior / boundary row ranges of every rank, halo from both neighbours
            n = 2600
            rs = np.random.RandomState(21)
            deg = rs.randint(60, 95, size=n)
            rows = np.repeat(np.arange(n), deg)
            cols = np.clip(rows + rs.randint(-250, 251, size=rows.size), 0, n - 1)
            B = sp.csr_matrix((rs.standard_normal(rows.size) * 0.01, (rows, cols)), shape=(n, n))
            B.sum_duplicates()
            S = (B + B.T).tocsr()
            S = (S + sp.diags(np.abs(S).sum(axis=1).A1 + 1.0)).tocsr()
            S.sort_indices()
            ptr, node, val = (S.indptr + 1).astype(np.int32), (S.indices + 1).astype(np.int32), S.data.copy()
        else:                       # nonsymmetric, irregular: no offset dictionary, halo from several ranks
            n = 1200
            Ar = orc.CsrMatrix.from_edges(n, n, *P.random_spd_edges(n, seed=5, skew=True))
            ptr, node, val = Ar.ptr, Ar.node, Ar.val
        A = orc.CsrMatrix(n, n, ptr, node, val)
        starts = sg.partition_rows_by_nnz(ptr, world, align=2)
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        n_own = r1 - r0
        k0, k1 = ptr[r0] - 1, ptr[r1] - 1
        lptr = (ptr[r0:r1 + 1] - k0).astype(np.int32)
        lnode_g = np.ascontiguousarray(node[k0:k1])
        lval = np.ascontiguousarray(val[k0:k1])

        H = sg.dist_csr_matrix(comm, starts, lptr, lnode_g, lval)
        _, halo = sg.halo_plan_host(n_own, r0, lnode_g)
        assert H.x_len == n_own + len(halo), (H.x_len, n_own, len(halo))
        res["n_halo"] = int(len(halo))

        # ---- the exchange plan the ranks agreed on over the transport == the in-process links
        links = sg.partition_links_host(starts, ptr, node)
        mine = sorted((l for l in links if l["sender"] == rank), key=lambda l: l["receiver"])
        to_me = {l["sender"]: l for l in links if l["receiver"] == rank}
        nbrs = H.halo_nbrs()
        sends = sorted((nb for nb in nbrs if nb["send_count"]), key=lambda nb: nb["peer"])
        assert [nb["peer"] for nb in sends] == [l["receiver"] for l in mine]
        for nb, l in zip(sends, mine):
            assert np.array_equal(nb["send_idx"], l["send_idx"]), "send list differs from the in-process partition"
        for nb in nbrs:
            if nb["recv_count"]:
                l = to_me[nb["peer"]]
                assert nb["recv_offset"] == l["recv_offset"] and nb["recv_count"] == len(l["send_idx"])
        assert sum(1 for nb in nbrs if nb["recv_count"]) == len(to_me)

        bl_box = [None]
        def reorder_checks(H, out, bl_holder):
            """re-orderings and permutations of the matrix distributed over the ranks (see below)"""
            # ---- re-orderings and permutations of the matrix distributed over the ranks (permutations.f90:22-205, cs_matrices.f90:
            #      471-490): every rank gets the reference's p / colours for the WHOLE graph, and after A%left_permute(p) /
            #      A%right_permute(p) every rank holds its row block of the reference's permuted matrix -- products bit for bit, then
            #      a solve on the permuted system
            pb = H.bfs_order()
            pbo = orc.bfs_order(A)
            if not np.array_equal(pb, pbo):
                Hs = sg.csr_matrix(n, n, ptr, node, val)
                single = bool(np.array_equal(Hs.bfs_order(), pbo))
                Hs.destroy()
                print(f"[diag] single-GPU bfs on the same matrix equals the oracle: {single}", flush=True)
                bad_i = np.nonzero(pb != pbo)[0]
                raise AssertionError(f"bfs_order over ranks differs from breadth_first_search: {bad_i.size} of {n} entries, first at {bad_i[:5]}: "
                                     f"{pb[bad_i[:5]]} vs {pbo[bad_i[:5]]}; ours is a permutation: {np.array_equal(np.sort(pb), np.arange(1, n + 1))}; single-GPU bfs ok: {single}")
            col, ncol_ = H.greedy_coloring()
            assert np.array_equal(col, orc.greedy_coloring(A)) and ncol_ == int(col.max())
            perm = None
            if case in ("poisson2d", "laplace3d"):
                perm, ptrs, nc_ = H.greedy_color_ordering()
                po, ptro, nco = orc.greedy_color_ordering(A)
                assert nc_ == nco and np.array_equal(perm, po) and np.array_equal(ptrs, ptro)
            else:
                perm = (np.random.RandomState(77).permutation(n) + 1).astype(np.int32)      # any permutation: rows cross every rank boundary
            Ap = orc.permuted(A, perm, perm)
            H.left_permute(perm)
            H.right_permute(perm)
            xq = np.random.RandomState(9).standard_normal(n)
            xeq = np.zeros(H.x_len)
            xeq[:n_own] = xq[r0:r1]
            yq = np.zeros(n_own)
            H.matvec(xeq, yq)
            assert np.array_equal(yq, Ap.matvec(xq)[r0:r1]), "rows of P A P^T over ranks differ from the reference's permuted matrix"
            tq = np.zeros(n_own)
            H.matvec_t(xq[r0:r1].copy(), tq)
            assert np.array_equal(tq, Ap.matvec_t(xq)[r0:r1]), "transposed product of the permuted distributed matrix"
            if case != "random":
                bq = np.empty(n); bq[perm - 1] = b
                bl_holder[0] = bq[r0:r1].copy()
                check("cg_on_permuted", sg.cg(1e-13), None, orc.cg(Ap, bq, tol=1e-13), lambda i: 1, 1e-10 if case == "longrows" else 1e-12)
            out["reorderings_over_ranks"] = {"iterations": 0, "bfs": True, "colours": int(ncol_), "permuted_products_bit_exact": True}

        if reorder_only:
            b = np.full(n, 1.0 / n)
            out = {}

            def check(name, solver, pc_mk, ref, it_tol, rel_tol):
                ur, itr = ref[0], ref[1]
                solver.setup(H)
                u = np.zeros(n_own)
                solver.solve(H, u, bl_box[0], None)
                rel = float(np.abs(u - ur[r0:r1]).max() / np.abs(ur).max())
                out[name] = {"iterations": int(solver.iterations), "oracle_iterations": int(itr), "rel": rel}
                assert abs(solver.iterations - itr) <= it_tol(itr) and rel <= rel_tol, (name, solver.iterations, itr, rel)
                solver.destroy()
            reorder_checks(H, out, bl_box)
            res["solves"] = out
            H.destroy()
            comm.destroy()
            return

        def probe(label):
            """SGM_WORKER_PROBE: after which section does the graph gathered over the ranks stop being the matrix's?"""
            if not os.environ.get("SGM_WORKER_PROBE"):
                return
            okp = bool(np.array_equal(H.bfs_order(), orc.bfs_order(A)))
            print(f"[probe] rank {rank} after {label}: bfs over ranks equals the oracle's: {okp}", flush=True)

        probe("creation")
        # ---- matvec: host vectors, then device tensors; rows bit-identical to the serial matvec
        x = P.test_vector(n) if case != "random" else np.random.RandomState(1).standard_normal(n)
        y_ref = A.matvec(x)[r0:r1]
        xe = np.zeros(H.x_len)
        xe[:n_own] = x[r0:r1]
        y = np.zeros(n_own)
        H.matvec(xe, y)
        assert np.array_equal(y, y_ref), "distributed matvec rows differ from the serial matvec"
        xd = torch.zeros(H.x_len, dtype=torch.float64, device=dev)
        xd[:n_own] = torch.from_numpy(x[r0:r1]).to(dev)
        yd = torch.full((n_own,), -3.0, dtype=torch.float64, device=dev)
        H.matvec(xd, yd)
        assert np.array_equal(yd.cpu().numpy(), y_ref)
        # a device vector is used in place: its halo region now holds the neighbours' x entries
        assert np.array_equal(xd[n_own:].cpu().numpy(), x[halo - 1]), "halo region does not hold the neighbours' x entries"
        y0 = np.random.RandomState(7).standard_normal(n)
        ya = y0[r0:r1].copy()
        H.matvec_add(xe, ya)
        assert np.array_equal(ya, A.matvec_add(x, y0.copy())[r0:r1])

        probe("matvec")
        # ---- transpose products: A^T built once as another distributed matrix (entries travel to the owner of
        #      their column); every y(i) sums the reference scatter's terms in the reference's order
        xt = np.random.RandomState(11).standard_normal(n)
        t_ref = A.matvec_t(xt)
        t = np.full(n_own, -2.0)
        H.matvec_t(xt[r0:r1].copy(), t)
        assert np.array_equal(t, t_ref[r0:r1]), "distributed matvec_t differs from csc_matvec_add"
        ta = y0[r0:r1].copy()
        H.matvec_t_add(xt[r0:r1].copy(), ta)
        assert np.array_equal(ta, A.matvec_t_add(xt, y0.copy())[r0:r1])
        td = torch.zeros(n_own, dtype=torch.float64, device=dev)
        H.matvec_t(torch.from_numpy(xt[r0:r1].copy()).to(dev), td)
        assert np.array_equal(td.cpu().numpy(), t_ref[r0:r1])

        probe("transpose")
        # ---- the same matrix created from DEVICE arrays (what bench.py --workload c5 does)
        H2 = sg.dist_csr_matrix(comm, starts, torch.from_numpy(lptr).to(dev), torch.from_numpy(lnode_g).to(dev),
                                torch.from_numpy(lval).to(dev))
        yd.fill_(-5.0)
        H2.matvec(xd, yd)
        assert np.array_equal(yd.cpu().numpy(), y_ref)
        H2.destroy()

        probe("device-array twin")
        # ---- ELLPACK rows (padding slots included) through the same machinery: sgm_ell_create_dist
        ne = 4000
        E = orc.EllMatrix.from_edges(ne, ne, *P.random_regular_ell(ne, 12, 99, dmin=7))
        es = sg.partition_rows_by_nnz(np.arange(1, 12 * ne + 2, 12, dtype=np.int32), world, align=2)
        e0, e1 = int(es[rank]), int(es[rank + 1])
        He = sg.dist_ellpack_matrix(comm, es, np.ascontiguousarray(E.node.reshape(ne, -1)[e0:e1]),
                                    np.ascontiguousarray(E.val.reshape(ne, -1)[e0:e1]))
        xe_full = np.random.RandomState(3).standard_normal(ne)
        xel = np.zeros(He.x_len)
        xel[:e1 - e0] = xe_full[e0:e1]
        ye = np.zeros(e1 - e0)
        He.matvec(xel, ye)
        assert np.array_equal(ye, E.matvec(xe_full)[e0:e1]), "distributed ELLPACK rows differ from ellpack_matvec_add"
        # ... and the transposed products (ellpack_matvec_t_add, ellpack_matrices.f90:670-693: the scatter runs over ALL max_d
        # slots, padding included -- they are stored entries here): every y(i) the reference's terms in the reference's order
        te_ref = E.matvec_t(xe_full)
        te = np.full(e1 - e0, 4.0)
        He.matvec_t(xe_full[e0:e1].copy(), te)
        assert np.array_equal(te, te_ref[e0:e1]), "distributed ELLPACK matvec_t differs from ellpack_matvec_t_add"
        ye0 = np.random.RandomState(5).standard_normal(ne)
        tea = ye0[e0:e1].copy()
        He.matvec_t_add(xe_full[e0:e1].copy(), tea)
        assert np.array_equal(tea, E.matvec_t_add(xe_full, ye0.copy())[e0:e1])
        # ... and ldu() on these rows: block-Jacobi ILDU(0) of every rank's diagonal block of the REAL entries (the reference's
        # pattern pass and fill go by the edge cursor, ellpack_graphs.f90:310-369: the first degrees(i) slots, never the padding,
        # which is stored here for the product's sake and would otherwise overwrite the last neighbour's value with 0.0)
        Ec = orc.ell_real_entries(E)
        rowsE = np.repeat(np.arange(ne), np.diff(Ec.ptr))
        blkE = np.searchsorted(es, np.arange(ne), side="right") - 1
        keepE = blkE[rowsE] == blkE[Ec.node - 1]
        cntE = np.bincount(rowsE[keepE], minlength=ne)
        EbD = orc.CsrMatrix(ne, ne, np.concatenate([[1], 1 + np.cumsum(cntE)]).astype(np.int32), Ec.node[keepE].copy(), Ec.val[keepE].copy())
        pcE = sg.ldu()
        pcE.setup(He)
        zE = np.zeros(e1 - e0)
        pcE.solve(He, zE, xe_full[e0:e1].copy())
        assert np.array_equal(zE, orc.Ildu(EbD).solve(xe_full)[e0:e1], equal_nan=True), "ILDU(0) of distributed ELLPACK rows saw the padding"
        pcE.destroy()
        He.destroy()

        probe("ellpack section")
        # ---- Krylov loops with all-reduced dots
        b = np.full(n, 1.0 / n) if case != "random" else P.test_vector(n)
        bl = b[r0:r1].copy()
        out = {}

        def check(name, solver, pc_mk, ref, it_tol, rel_tol):
            ur, itr = ref[0], ref[1]
            solver.setup(H)
            pc = pc_mk() if pc_mk else None
            if pc is not None:
                pc.setup(H)
            u = np.zeros(n_own)
            solver.solve(H, u, bl if bl_box[0] is None else bl_box[0], pc)      # (bl_box: the right-hand side of the permuted system, set by reorder_checks)
            rel = float(np.abs(u - ur[r0:r1]).max() / np.abs(ur).max())
            out[name] = {"iterations": int(solver.iterations), "oracle_iterations": int(itr), "rel": rel}
            assert abs(solver.iterations - itr) <= it_tol(itr), (name, solver.iterations, itr)
            assert rel <= rel_tol, (name, rel)
            solver.destroy()
            if pc is not None:
                pc.destroy()

        if case != "random":
            # (two iterates that both meet the ABSOLUTE 1e-13 differ by about tol / |u|: u is 1e-4-sized on the long-row matrix)
            rt = 1e-10 if case == "longrows" else 1e-12
            check("cg", sg.cg(1e-13), None, orc.cg(A, b, tol=1e-13), lambda i: 1, rt)
            check("cg_jacobi", sg.cg(1e-13), sg.jacobi, orc.cg(A, b, tol=1e-13, pc=orc.Jacobi(A)), lambda i: 1, rt)
            # block-Jacobi ILDU(0): the oracle factors the block-diagonal part of A
            rows = np.repeat(np.arange(n), np.diff(ptr))
            blk = np.searchsorted(starts, np.arange(n), side="right") - 1
            keep = blk[rows] == blk[node - 1]
            cnt = np.bincount(rows[keep], minlength=n)
            Ab = orc.CsrMatrix(n, n, np.concatenate([[1], 1 + np.cumsum(cnt)]).astype(np.int32), node[keep].copy(), val[keep].copy())
            check("cg_ildu_blockjacobi", sg.cg(1e-12), sg.ldu, orc.cg(A, b, tol=1e-12, pc=orc.Ildu(Ab)), lambda i: 1, max(rt, 1e-11))
            check("bicgstab", sg.bicgstab(1e-13), None, orc.bicgstab(A, b, tol=1e-13), lambda i: max(2, 0.1 * i), max(rt, 1e-11))
            probe("first solver checks")
            # ---- option dist_halo_fused: p's halo formed locally from the boundary rows of r (z) that travel beside the
            #      all-reduce of r.r (r.z) -- 1: one group with it, 2: a group of its own -- against 0, p exchanged by every
            #      product: the same bits, iterate for iterate
            fused = {}
            for mode in (0, 1, 2):
                for nm, pc_mk in (("cg", None), ("cg_jacobi", sg.jacobi), ("cg_ildu", sg.ldu)):
                    sv = sg.cg(1e-12)
                    sv.set_option("dist_halo_fused", mode)
                    sv.set_history(100000)
                    sv.setup(H)
                    pcm = pc_mk() if pc_mk else None
                    if pcm is not None:
                        pcm.setup(H)
                    u = np.full(n_own, 0.125)
                    sv.solve(H, u, bl, pcm)
                    fused[(nm, mode)] = (u, int(sv.iterations), np.array(sv.history))
                    sv.destroy()
                    if pcm is not None:
                        pcm.destroy()
            for (nm, mode), (u, it, hist) in fused.items():
                u0, it0, hist0 = fused[(nm, 0)]
                assert it == it0 and np.array_equal(u, u0) and np.array_equal(hist, hist0), ("dist_halo_fused", nm, mode, it, it0)
            out["halo_fused_modes"] = {"iterations": fused[("cg", 1)][1], "bit_identical_to_exchanging_p": True}
            probe("halo_fused modes")
            # ---- ldu(reorder="colour") over the ranks: every rank orders its own diagonal block (greedy_color_ordering of
            #      A_kk's graph, no communication), block-Jacobi ILDU(0) of the ordered blocks, the solve in the permuted
            #      order rank by rank.  Oracle: the same ordering block by block, A permuted by it, PCG with ILDU(0) of its
            #      block-diagonal part.  (Not for the long-row matrix: its blocks are not bipartite, nothing to gain.)
            if case in ("poisson2d", "laplace3d"):
                pglob = np.zeros(n, np.int32)
                for kq in range(world):
                    q0, q1 = int(starts[kq]), int(starts[kq + 1])
                    a0, a1 = Ab.ptr[q0] - 1, Ab.ptr[q1] - 1
                    Bq = orc.CsrMatrix(q1 - q0, q1 - q0, (Ab.ptr[q0:q1 + 1] - a0).astype(np.int32), (Ab.node[a0:a1] - q0).astype(np.int32), Ab.val[a0:a1].copy())
                    pglob[q0:q1] = orc.greedy_color_ordering(Bq)[0] + q0
                Apm = orc.permuted(A, pglob, pglob)
                rowsP = np.repeat(np.arange(n), np.diff(Apm.ptr))
                keepP = blk[rowsP] == blk[Apm.node - 1]
                cntP = np.bincount(rowsP[keepP], minlength=n)
                AbP = orc.CsrMatrix(n, n, np.concatenate([[1], 1 + np.cumsum(cntP)]).astype(np.int32), Apm.node[keepP].copy(), Apm.val[keepP].copy())
                bpm = np.empty(n); bpm[pglob - 1] = b
                urp, itrp = orc.cg(Apm, bpm, tol=1e-12, pc=orc.Ildu(AbP))[:2]
                for mode in (2, 0):
                    sv = sg.cg(1e-12)
                    sv.set_option("reorder_solve", mode)
                    sv.setup(H)
                    pcm = sg.ldu(reorder="colour")
                    pcm.setup(H)
                    u = np.zeros(n_own)
                    sv.solve(H, u, bl, pcm)
                    relc = float(np.abs(u - urp[pglob - 1][r0:r1]).max() / np.abs(urp).max())
                    assert abs(sv.iterations - itrp) <= 1 and relc <= 1e-10, ("cg_ildu_colour", mode, sv.iterations, itrp, relc)
                    out[f"cg_ildu_colour_mode{mode}"] = {"iterations": int(sv.iterations), "oracle_iterations": int(itrp), "rel": relc}
                    sv.destroy()
                    pcm.destroy()
        else:
            check("bicgstab_jacobi", sg.bicgstab(1e-12), sg.jacobi, orc.bicgstab(A, b, tol=1e-12, pc=orc.Jacobi(A)),
                  lambda i: max(2, 0.1 * i), 1e-10)
        probe("colour ildu")
        # ---- Lanczos over the ranks (SURVEY 8(f3) on a partitioned matrix): every rank its owned slice of q1 and Q, dots
        #      all-reduced, T the same everywhere; against the oracle's serial run
        if case in ("poisson2d", "laplace3d"):
            q1 = np.random.RandomState(12).random_sample(n) * 2 - 1
            Tl, Ql = sg.lanczos(H, 20, q1[r0:r1].copy())
            Tlo, Qlo = orc.lanczos(A, 20, q1)
            assert np.abs(Tl - Tlo).max() <= 1e-9, float(np.abs(Tl - Tlo).max())
            assert np.abs(Ql[:, :8] - Qlo[r0:r1, :8]).max() <= 1e-10
            out["lanczos"] = {"iterations": 20, "T_max_diff": float(np.abs(Tl - Tlo).max())}
            # generalized: B = a mass-like matrix on the same pattern and partition, B%solve = CG(1e-14) over the ranks
            rowsg = np.repeat(np.arange(1, n + 1), np.diff(ptr))
            bval = np.where(rowsg == node, 1.0 + (rowsg % 7) / 16.0, -1.0 / 16.0)
            HB = sg.dist_csr_matrix(comm, starts, lptr, lnode_g, np.ascontiguousarray(bval[k0:k1]))
            HB.set_solver(sg.cg(1e-14))
            Tg, Qg = sg.generalized_lanczos(H, HB, 12, q1[r0:r1].copy())
            Tgo, Qgo = orc.generalized_lanczos(A, orc.CsrMatrix(n, n, ptr, node, bval), 12, q1, 1e-14)
            assert np.abs(Tg - Tgo).max() <= 1e-9 and np.abs(Qg - Qgo[r0:r1]).max() <= 1e-9
            out["generalized_lanczos"] = {"iterations": 12, "T_max_diff": float(np.abs(Tg - Tgo).max())}
            HB.destroy()

        probe("lanczos")
        # ---- dot_order = 1: the running sum of every dot travels rank 0 -> 1 -> ... and each rank continues it over its
        #      own rows: the distributed iterates are BIT-IDENTICAL to the serial ones (the oracle's left-to-right dots)
        sg.set_option("dot_order", 1)
        try:
            def exact(name, solver, pc_mk, ref):
                ur, itr = ref[0], ref[1]
                solver.setup(H)
                pc = pc_mk() if pc_mk else None
                if pc is not None:
                    pc.setup(H)
                u = np.zeros(n_own)
                solver.solve(H, u, bl, pc)
                out[name] = {"iterations": int(solver.iterations), "oracle_iterations": int(itr),
                             "bit_identical": bool(np.array_equal(u, ur[r0:r1]))}
                assert solver.iterations == itr, (name, solver.iterations, itr)
                assert np.array_equal(u, ur[r0:r1]), (name, float(np.abs(u - ur[r0:r1]).max()))
                solver.destroy()
                if pc is not None:
                    pc.destroy()
            if case != "random":
                exact("seq_cg", sg.cg(1e-13), None, orc.cg(A, b, tol=1e-13))
                exact("seq_cg_jacobi", sg.cg(1e-13), sg.jacobi, orc.cg(A, b, tol=1e-13, pc=orc.Jacobi(A)))
                exact("seq_bicgstab", sg.bicgstab(1e-13), None, orc.bicgstab(A, b, tol=1e-13))
            else:
                exact("seq_bicgstab_jacobi", sg.bicgstab(1e-12), sg.jacobi, orc.bicgstab(A, b, tol=1e-12, pc=orc.Jacobi(A)))
        finally:
            sg.set_option("dot_order", 0)
        check("gmres30", sg.gmres(1e-12, 30), None, orc.gmres(A, b, tol=1e-12, restart=30), lambda i: 2, 1e-9 if case == "longrows" else 1e-10)
        reorder_checks(H, out, bl_box)
        res["solves"] = out
        H.destroy()
        comm.destroy()
    finally:
        dist.destroy_process_group()


def run_fuzz(rank, world, comm, res, seed0, count):
    """`count` seeded systems of tests/fuzz_solvers.py's generator, each cut into `world` row blocks at RANDOM even boundaries
    (blocks of a few rows, blocks without neighbours, every rank a neighbour of every other on the random graphs): products
    and transposed products of the owned rows bit-exact, CG in the reference's dot order with a random preconditioner the
    oracle's solve bit for bit, tree-order CG the same bits whether p is exchanged or its halo formed locally."""
    import numpy as np
    import oracle as orc
    import sigma_amd as sg
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz_solvers as F
    done = []
    for seed in range(seed0, seed0 + count):
        rs = np.random.RandomState(seed)
        kind, n, (ptr, node, val) = F.make(rs)
        if n < 4 * world or n > 120000:
            continue
        A = orc.CsrMatrix(n, n, ptr, node, val)
        cuts = np.sort(rs.choice(np.arange(1, n // 2), size=world - 1, replace=False)) * 2
        starts = np.concatenate([[0], cuts, [n]]).astype(np.int64)
        pck = ["none", "jacobi", "ildu", "ildu_colour"][int(rs.randint(0, 4))]
        r0, r1 = int(starts[rank]), int(starts[rank + 1])
        n_own = r1 - r0
        k0, k1 = ptr[r0] - 1, ptr[r1] - 1
        lptr = (ptr[r0:r1 + 1] - k0).astype(np.int32)
        H = sg.dist_csr_matrix(comm, starts, lptr, np.ascontiguousarray(node[k0:k1]), np.ascontiguousarray(val[k0:k1]))
        tag = (seed, kind, n, [int(v) for v in starts], pck)
        x, xt, b = rs.standard_normal(n), rs.standard_normal(n), rs.standard_normal(n)
        xe = np.zeros(H.x_len); xe[:n_own] = x[r0:r1]
        y = np.zeros(n_own); H.matvec(xe, y)
        assert np.array_equal(y, A.matvec(x)[r0:r1]), ("matvec", tag)
        t = np.zeros(n_own); H.matvec_t(xt[r0:r1].copy(), t)
        assert np.array_equal(t, A.matvec_t(xt)[r0:r1]), ("matvec_t", tag)
        # the oracle's system: the diagonal blocks for ILDU, the blockwise colour order for reorder="colour"
        Ao, bo, perm, opc, mk = A, b, None, None, None
        if pck == "jacobi":
            opc, mk = orc.Jacobi(A), sg.jacobi
        elif pck == "ildu":
            opc, mk = orc.Ildu(F.block_diagonal(A, starts)), sg.ldu
        elif pck == "ildu_colour":
            Ab = F.block_diagonal(A, starts)
            perm = np.zeros(n, np.int32)
            for k in range(world):
                q0, q1 = int(starts[k]), int(starts[k + 1])
                a0, a1 = Ab.ptr[q0] - 1, Ab.ptr[q1] - 1
                B = orc.CsrMatrix(q1 - q0, q1 - q0, (Ab.ptr[q0:q1 + 1] - a0).astype(np.int32), (Ab.node[a0:a1] - q0).astype(np.int32), Ab.val[a0:a1].copy())
                try:
                    perm[q0:q1] = orc.greedy_color_ordering(B)[0] + q0
                except ValueError:
                    perm = None
                    break
            if perm is None:                      # a block not connected from its first vertex: plain block-Jacobi ILDU instead
                pck, opc, mk = "ildu", orc.Ildu(Ab), sg.ldu
            else:
                Ao = orc.permuted(A, perm, perm)
                opc = orc.Ildu(F.block_diagonal(Ao, starts))
                bo = np.empty(n); bo[perm - 1] = b
                mk = lambda: sg.ldu(reorder="colour")
        ur, itr = orc.cg(Ao, bo, tol=1e-8, pc=opc, max_iter=300)[:2]
        if perm is not None:
            ur = ur[perm - 1]
        bl = b[r0:r1].copy()
        pc = mk() if mk else None
        if pc is not None:
            pc.setup(H)
        se = sg.cg(1e-8)
        se.set_option("dot_order", 1)
        se.set_max_iter(300)
        se.setup(H)
        u = np.zeros(n_own)
        se.solve(H, u, bl, pc, check=False)
        assert se.last_iterations == itr, ("dot_order=1 iterations", tag, se.last_iterations, itr)
        assert np.array_equal(u, ur[r0:r1]), ("dot_order=1 solution", tag, float(np.abs(u - ur[r0:r1]).max()))
        se.destroy()
        got = {}
        for mode in (0, 1):
            sv = sg.cg(1e-8)
            sv.set_option("dist_halo_fused", mode)
            sv.set_max_iter(300)
            sv.setup(H)
            u = np.zeros(n_own)
            sv.solve(H, u, bl, pc, check=False)
            got[mode] = (u, sv.last_iterations)
            sv.destroy()
        assert got[0][1] == got[1][1] and np.array_equal(got[0][0], got[1][0]), ("dist_halo_fused", tag, got[0][1], got[1][1])
        if pc is not None:
            pc.destroy()
        H.destroy()
        done.append(seed)
    res["n_halo"] = 1
    res["solves"] = {"fuzz": {"iterations": len(done)}}
    res["seeds"] = done


def run_composite(rank, world, comm, dev, res):
    """A 2 x 2 composite (sparse_matrix_composites.f90:41-162) whose four leaves are each distributed over the
    ranks -- block row i with a partition of its own, the off-diagonal leaves rectangular
    (sgm_csr_create_dist_rect) -- against the serial block loops of the oracle's leaves, bit for bit, and
    CG / Jacobi-PCG on it against the oracle's solves of the assembled matrix."""
    import numpy as np
    import scipy.sparse as sp
    import torch
    import oracle as orc
    import sigma_amd as sg
    from sigma_amd import problems as P

    nx, ny = 96, 70
    n = nx * ny
    ptr, node, val = P.poisson2d_csr(nx, ny)
    A = orc.CsrMatrix(n, n, ptr, node, val)
    Asp = sp.csr_matrix((val, node - 1, ptr - 1), shape=(n, n))
    Asp.sort_indices()
    m = 3362                                             # block boundary (not a grid line)
    cuts = [0, m, n]
    parts = []                                           # row partition of block row i (even boundaries)
    for i in range(2):
        sub = Asp[cuts[i]:cuts[i + 1]]
        parts.append(sg.partition_rows_by_nnz((sub.indptr + 1).astype(np.int32), world, align=2))
    leaves_o, leaves_d = [[None, None], [None, None]], [[None, None], [None, None]]
    for i in range(2):
        for j in range(2):
            B = Asp[cuts[i]:cuts[i + 1], cuts[j]:cuts[j + 1]].tocsr()
            B.sort_indices()
            nr, nc = B.shape
            leaves_o[i][j] = orc.CsrMatrix(nr, nc, (B.indptr + 1).astype(np.int32), (B.indices + 1).astype(np.int32), B.data.copy())
            r0, r1 = int(parts[i][rank]), int(parts[i][rank + 1])
            L = B[r0:r1]
            leaves_d[i][j] = sg.dist_csr_matrix(comm, parts[i], (L.indptr + 1).astype(np.int32), (L.indices + 1).astype(np.int32),
                                                L.data.copy(), col_starts=parts[j])
    S = sg.sparse_matrix(np.array([1, m + 1, n + 1], np.int32), np.array([1, m + 1, n + 1], np.int32))
    for i in range(2):
        for j in range(2):
            S.set_submatrix(i + 1, j + 1, leaves_d[i][j])

    def local(v):
        return np.concatenate([v[cuts[i] + int(parts[i][rank]):cuts[i] + int(parts[i][rank + 1])] for i in range(2)])

    rs = np.random.RandomState(4)
    x = rs.standard_normal(n)
    xs = [x[:m], x[m:]]
    # composite_matvec_add: row blocks outer, column blocks inner, y(i1:i2) += leaf * x(j1:j2)
    y0 = rs.standard_normal(n)
    yref = np.zeros(n)
    yaref = y0.copy()
    for i in range(2):
        for j in range(2):
            yref[cuts[i]:cuts[i + 1]] = leaves_o[i][j].matvec_add(xs[j], yref[cuts[i]:cuts[i + 1]].copy())
            yaref[cuts[i]:cuts[i + 1]] = leaves_o[i][j].matvec_add(xs[j], yaref[cuts[i]:cuts[i + 1]].copy())
    xl = local(x)
    y = np.full(len(xl), -7.0)
    S.matvec(xl, y)
    assert np.array_equal(y, local(yref)), "composite over distributed leaves: matvec differs from the serial block loops"
    ya = local(y0)
    S.matvec_add(xl, ya)
    assert np.array_equal(ya, local(yaref))
    yd = torch.zeros(len(xl), dtype=torch.float64, device=dev)
    S.matvec(torch.from_numpy(xl).to(dev), yd)
    assert np.array_equal(yd.cpu().numpy(), local(yref))
    # composite_matvec_t_add: column blocks outer, y(j1:j2) += leaf^T * x(i1:i2)
    tref = np.zeros(n)
    for j in range(2):
        for i in range(2):
            tref[cuts[j]:cuts[j + 1]] = leaves_o[i][j].matvec_t_add(xs[i], tref[cuts[j]:cuts[j + 1]].copy())
    t = np.full(len(xl), 3.0)
    S.matvec_t(xl, t)
    assert np.array_equal(t, local(tref)), "composite over distributed leaves: matvec_t differs from the serial block loops"
    # Krylov loops: all-reduced dots, Jacobi from the diagonal leaves
    b = np.full(n, 1.0 / n)
    out = {}
    for name, mk_pc, opc in (("cg", None, None), ("cg_jacobi", sg.jacobi, orc.Jacobi(A))):
        ur, itr, _, _ = orc.cg(A, b, tol=1e-13, pc=opc)
        s = sg.cg(1e-13)
        s.setup(S)
        pc = mk_pc() if mk_pc else None
        if pc is not None:
            pc.setup(S)
        u = np.zeros(len(xl))
        s.solve(S, u, local(b), pc)
        rel = float(np.abs(u - local(ur)).max() / np.abs(ur).max())
        out[name] = {"iterations": int(s.iterations), "oracle_iterations": int(itr), "rel": rel}
        assert abs(s.iterations - itr) <= 2 and rel <= 1e-11, (name, s.iterations, itr, rel)
    # Lanczos on the composite (src/eigensolver.f90:27-90: only A%matvec and dot products -- the block loops above and
    # all-reduces): T the same on every rank and the oracle's serial run's, Q this rank's slices of the block vectors
    q1 = np.random.RandomState(12).random_sample(n) * 2 - 1
    Tl, Ql = sg.lanczos(S, 16, local(q1))
    Tlo, Qlo = orc.lanczos(A, 16, q1)
    assert np.abs(Tl - Tlo).max() <= 1e-9, float(np.abs(Tl - Tlo).max())
    assert np.abs(Ql[:, :8] - np.stack([local(Qlo[:, c]) for c in range(8)], axis=1)).max() <= 1e-10
    out["lanczos_composite"] = {"iterations": 16, "T_max_diff": float(np.abs(Tl - Tlo).max())}
    # generalized Lanczos, A and B composites over the same distributed leaves' partitions, B%solve = CG(1e-14) over the
    # ranks (the reference's own test runs it on a composite: eigensolver_test_generalized_lanczos.f90:150)
    rowsg = np.repeat(np.arange(1, n + 1), np.diff(ptr))
    bval = np.where(rowsg == node, 1.0 + (rowsg % 7) / 16.0, -1.0 / 16.0)
    Bsp = sp.csr_matrix((bval, node - 1, ptr - 1), shape=(n, n))
    Bsp.sort_indices()
    SB = sg.sparse_matrix(np.array([1, m + 1, n + 1], np.int32), np.array([1, m + 1, n + 1], np.int32))
    leaves_b = []
    for i in range(2):
        for j in range(2):
            Bb = Bsp[cuts[i]:cuts[i + 1], cuts[j]:cuts[j + 1]].tocsr()
            Bb.sort_indices()
            r0, r1 = int(parts[i][rank]), int(parts[i][rank + 1])
            Lb = Bb[r0:r1]
            leaves_b.append(sg.dist_csr_matrix(comm, parts[i], (Lb.indptr + 1).astype(np.int32), (Lb.indices + 1).astype(np.int32),
                                               Lb.data.copy(), col_starts=parts[j]))
            SB.set_submatrix(i + 1, j + 1, leaves_b[-1])
    SB.set_solver(sg.cg(1e-14))
    Tg, Qg = sg.generalized_lanczos(S, SB, 10, local(q1))
    Tgo, Qgo = orc.generalized_lanczos(A, orc.CsrMatrix(n, n, ptr, node, bval), 10, q1, 1e-14)
    assert np.abs(Tg - Tgo).max() <= 1e-9 and np.abs(Qg - np.stack([local(Qgo[:, c]) for c in range(10)], axis=1)).max() <= 1e-9
    out["generalized_lanczos_composite"] = {"iterations": 10, "T_max_diff": float(np.abs(Tg - Tgo).max())}
    res["solves"] = out
    res["n_local"] = int(len(xl))
    res["n_halo"] = int(sum(leaves_d[i][j].x_len - leaves_d[i][j].nc_local for i in range(2) for j in range(2)))


if __name__ == "__main__":
    main()
