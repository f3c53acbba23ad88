#!/usr/bin/env python3
"""Graph-operation fuzzer (not collected by pytest; tests/test_gpu_reorder.py runs a few seeds): breadth_first_search,
greedy_coloring and greedy_color_ordering (permutations.f90:22-205) of seeded random matrix graphs -- grids with holes, random
bipartite and general symmetric graphs, trees, directed graphs, graphs not connected from vertex 1 -- on the device against the
oracle's sequential passes, bit for bit, through all three colouring passes (matrix option coloring_pass), and the symmetric
permutation by the ordering against the oracle's permuted matrix.

    python tests/fuzz_graphs.py [seconds] [first_seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp

import oracle as orc
import sigma_amd as sg


def make(rs):
    kind = ["grid_holes", "bipartite", "general", "tree", "directed", "islands", "hubs"][int(rs.randint(0, 7))]
    n = int(10 ** rs.uniform(0.7, 5.2))
    if kind == "grid_holes":
        nx = max(2, int(np.sqrt(n))); ny = max(1, n // nx); n = nx * ny
        idx = np.arange(n).reshape(ny, nx)
        a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]); b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
        keep = rs.rand(a.size) >= 10 ** rs.uniform(-3, -0.7)
        a, b = a[keep], b[keep]
    elif kind == "bipartite":
        h = max(1, n // 2); k = int(n * rs.uniform(1, 4))
        a, b = rs.randint(0, h, size=k), h + rs.randint(0, max(1, n - h), size=k)
    elif kind == "general":
        k = int(n * rs.uniform(0.8, 4))
        a, b = rs.randint(0, n, size=k), rs.randint(0, n, size=k)
    elif kind == "tree":
        a = np.arange(1, n); b = (a * rs.rand(a.size)).astype(np.int64)
    elif kind == "directed":
        k = int(n * rs.uniform(1, 3))
        a, b = rs.randint(0, n, size=k), rs.randint(0, n, size=k)
    elif kind == "islands":
        m = max(2, n // int(rs.randint(2, 6)))
        a = np.arange(0, n - 1); b = a + 1
        keep = (a % m) != (m - 1)
        a, b = a[keep], b[keep]
    else:                                       # a few vertices adjacent to very many (long rows)
        k = int(n * 1.5)
        a, b = rs.randint(0, n, size=k), rs.randint(0, n, size=k)
        hubs = rs.randint(0, n, size=min(5, n))
        a = np.concatenate([a, np.repeat(hubs, max(1, n // 3))]); b = np.concatenate([b, rs.randint(0, n, size=hubs.size * max(1, n // 3))])
    if kind in ("general", "hubs", "directed") and n > 1 and rs.rand() < 0.7:      # + a random spanning tree: connected from vertex 1
        t = np.arange(1, n)
        a = np.concatenate([a, t]); b = np.concatenate([b, (t * rs.rand(t.size)).astype(np.int64)])
        if kind == "directed":                                                      # (reachable along the edges' direction)
            a, b = np.concatenate([a, (t * rs.rand(t.size)).astype(np.int64)]), np.concatenate([b, t])
    if kind == "bipartite" and n > 3 and rs.rand() < 0.7:
        h = max(1, n // 2)
        s2 = np.arange(h, n); s1 = np.arange(1, h)
        a = np.concatenate([a, (s2 - h) % h * 0 + (rs.rand(s2.size) * h).astype(np.int64), s1])
        b = np.concatenate([b, s2, h + (rs.rand(s1.size) * (n - h)).astype(np.int64)])
    keep = (a != b) & (a < n) & (b < n)
    a, b = a[keep], b[keep]
    if kind == "directed":
        i, j = a, b
    else:
        i, j = np.concatenate([a, b]), np.concatenate([b, a])
    i = np.concatenate([i, np.arange(n)]); j = np.concatenate([j, np.arange(n)])
    S = sp.coo_matrix((np.ones(i.size), (i, j)), shape=(n, n)).tocsr()
    S.sum_duplicates()
    if rs.rand() < 0.5:
        S.sort_indices()
    S.data = rs.standard_normal(S.data.size)
    return kind, n, (S.indptr + 1).astype(np.int32), (S.indices + 1).astype(np.int32), S.data.copy()


def one(seed, verbose=True):
    rs = np.random.RandomState(seed)
    kind, n, ptr, node, val = make(rs)
    A = orc.CsrMatrix(n, n, ptr, node, val)
    H = sg.csr_matrix(n, n, ptr, node, val)
    bad = []
    if not np.array_equal(H.bfs_order(), orc.bfs_order(A)):
        bad.append("bfs_order")
    c_ref = orc.greedy_coloring(A)
    try:
        o_ref = orc.greedy_color_ordering(A)
    except ValueError:
        o_ref = None
    passes = []
    for mode in (0, 1, 2):
        H.set_option("coloring_pass", mode)
        c, nc = H.greedy_coloring()
        if not np.array_equal(c, c_ref) or nc != int(c_ref.max(initial=0)):
            bad.append(f"greedy_coloring pass {mode}")
        try:
            p, ptrs, ncol = H.greedy_color_ordering()
            if o_ref is None:
                bad.append(f"ordering of a graph not connected from vertex 1 accepted (pass {mode})")
            elif not (np.array_equal(p, o_ref[0]) and np.array_equal(ptrs, o_ref[1]) and ncol == o_ref[2]):
                bad.append(f"greedy_color_ordering pass {mode}")
        except sg.SigmaError:
            if o_ref is not None:
                bad.append(f"greedy_color_ordering refused (pass {mode})")
    H.set_option("coloring_pass", 0)
    if o_ref is not None and n <= 200000:
        p = o_ref[0]
        H.left_permute(p); H.right_permute(p)
        Ap = orc.permuted(A, p, p)
        x = rs.standard_normal(n)
        y = np.zeros(n); H.matvec(x, y)
        if not np.array_equal(y, Ap.matvec(x)):
            bad.append("product after the symmetric permutation")
    H.destroy()
    if verbose or bad:
        print(f"seed {seed}: {kind} n={n} nnz={val.size} colours={int(c_ref.max(initial=0))} ordering={'yes' if o_ref is not None else 'refused'}"
              + (f"  MISMATCH: {bad}" if bad else ""), flush=True)
    return bad


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 700000
    sg.init(0)
    t0 = time.time()
    failures, count = [], 0
    while time.time() - t0 < seconds:
        if one(seed):
            failures.append(seed)
        seed += 1
        count += 1
    print(f"{count} graphs, failing seeds: {failures}")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
