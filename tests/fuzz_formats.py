#!/usr/bin/env python3
"""Format-selection fuzzer (not collected by pytest; tests/test_gpu_parity.py runs a few seeds of it): seeded CSR matrices of
MID size -- where the layouts chosen at create differ from the small cases of the randomised tests: slices with a window of x in
LDS, the column-blocked two-phase form, SELL, the lean layouts -- with random structure and random per-matrix options.  Every
product (matvec, matvec_add, both transposes, after set_values, after a symmetric permutation) must equal the oracle's rows
bit for bit (cs_matrices.f90:600-622).

    python tests/fuzz_formats.py [seconds] [first_seed]          (every third seed is an ELLPACK matrix: one_ell)

Prints one line per matrix and the failing seed if any; exit code 1 on a mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle as orc
import sigma_amd as sg


def make(rs):
    """-> (kind, n, m, ptr, node, val), 1-based arrays."""
    kind = ["band", "band_outliers", "scattered", "powerlaw", "stencil_perturbed", "rect", "blocks"][int(rs.randint(0, 7))]
    n = int(10 ** rs.uniform(3.3, 6.3))
    cap = 12_000_000                            # stored entries at most (the oracle's product and the host arrays stay quick)

    def rows_of(deg):                           # deg -> (n, deg, rows) with at most `cap` entries
        nonlocal n
        tot = int(deg.sum())
        if tot > cap:
            n = max(1, int(n * cap / tot))
            deg = deg[:n]
        return deg, np.repeat(np.arange(n), deg)

    offs = None
    if kind == "band":
        bw = int(10 ** rs.uniform(0, 4.3))
        d = int(rs.randint(1, min(300, 2 * bw + 1) + 1))
        deg = np.full(n, d) if rs.rand() < 0.5 else rs.randint(max(1, d // 2), d + 1, size=n)
        deg, rows = rows_of(deg)
        m = n
        cols = rows + rs.randint(-bw, bw + 1, size=rows.size)
    elif kind == "band_outliers":
        bw = int(10 ** rs.uniform(1, 3.5))
        deg, rows = rows_of(np.full(n, int(rs.randint(4, 60))))
        m = n
        cols = rows + rs.randint(-bw, bw + 1, size=rows.size)
        far = rs.rand(rows.size) < 10 ** rs.uniform(-5, -2)           # a few entries anywhere
        cols = np.where(far, rs.randint(0, n, size=rows.size), cols)
    elif kind == "scattered":
        d0, d1 = sorted(int(v) for v in rs.randint(2, 33, size=2))
        deg, rows = rows_of(rs.randint(d0, d1 + 1, size=n))
        m = n
        cols = rs.randint(0, m, size=rows.size)
    elif kind == "powerlaw":
        deg, rows = rows_of(np.minimum((rs.pareto(1.3, size=n) * 3).astype(np.int64), 5000))
        m = n
        cols = rs.randint(0, m, size=rows.size)
    elif kind == "stencil_perturbed":
        nx = max(2, int(np.sqrt(n)))
        n = m = nx * nx
        offs = np.array([0, -1, 1, -nx, nx] + ([-nx - 1, nx + 1] if rs.rand() < 0.5 else []))
        deg = np.full(n, offs.size)
        rows = np.repeat(np.arange(n), offs.size)
        cols = rows + np.tile(offs, n)
        bad = rs.rand(rows.size) < 10 ** rs.uniform(-6, -2)
        cols = np.where(bad, rs.randint(0, n, size=rows.size), cols)
    elif kind == "rect":
        d = int(rs.randint(1, 24))
        deg, rows = rows_of(rs.randint(0, d + 1, size=n))
        m = int(n * 10 ** rs.uniform(-1, 1)) + 1
        cols = (rows * (m / n)).astype(np.int64) + rs.randint(-50, 51, size=rows.size)
    else:                                       # dense-ish diagonal blocks
        bs = int(rs.randint(4, 200))
        deg, rows = rows_of(np.full(n, int(rs.randint(1, bs + 1))))
        m = n
        cols = (rows // bs) * bs + rs.randint(0, bs, size=rows.size)
    keep = (cols >= 0) & (cols < m)
    if not keep.all():                          # out-of-range neighbours are dropped: the rows at the edges are shorter
        deg = np.bincount(rows[keep], minlength=n)
        rows, cols = rows[keep], cols[keep]
    if rs.rand() < 0.5:                          # ascending columns inside a row (the reference keeps insertion order: both occur)
        cols = cols[np.lexsort((cols, rows))]
    ptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    return kind, n, m, ptr, (cols + 1).astype(np.int32), rs.standard_normal(rows.size)


OPTIONS = (("csr_sliced", (1, 1, 1, 0)), ("csr_sell", (1, 1, 2, 0)), ("csr_xwindow", (1, 1, 0)), ("csr_lean", (1, 1, 0)),
           ("csr_offset_dict", (1, 1, 1, 0)), ("csr_row_owner", (1, 1, 0)), ("csr_row_lines", (1, 1, 0)), ("ell_colblock", (1, 1, 2, 0)),
           ("slice_sched", (0, 0, 1)))
DEFAULTS = {"csr_sliced": 1, "csr_sell": 1, "csr_xwindow": 1, "csr_lean": 1, "csr_offset_dict": 1, "csr_row_owner": 1, "csr_row_lines": 1,
            "ell_colblock": 1, "slice_sched": 0}


def one(seed, verbose=True):
    rs = np.random.RandomState(seed)
    kind, n, m, ptr, node, val = make(rs)
    opts = {k: int(v[rs.randint(0, len(v))]) for k, v in OPTIONS}
    A = orc.CsrMatrix(n, m, ptr, node, val)
    x, y0 = rs.standard_normal(m), rs.standard_normal(n)
    xt = rs.standard_normal(n)
    try:
        for k, v in opts.items():
            sg.set_option(k, v)
        H = sg.csr_matrix(n, m, ptr, node, val)
        kern = H.kernel
        bad = []
        y = np.zeros(n); H.matvec(x, y)
        if not np.array_equal(y, A.matvec(x)): bad.append("matvec")
        ya = y0.copy(); H.matvec_add(x, ya)
        if not np.array_equal(ya, A.matvec_add(x, y0.copy())): bad.append("matvec_add")
        t = np.zeros(m); H.matvec_t(xt, t)
        if not np.array_equal(t, A.matvec_t(xt)): bad.append("matvec_t")
        v2 = rs.standard_normal(val.size)
        H.set_values(v2)
        A2 = orc.CsrMatrix(n, m, ptr, node, v2)
        y = np.zeros(n); H.matvec(x, y)
        if not np.array_equal(y, A2.matvec(x)): bad.append("matvec after set_values")
        t = np.zeros(m); H.matvec_t(xt, t)
        if not np.array_equal(t, A2.matvec_t(xt)): bad.append("matvec_t after set_values")
        if n == m and n <= 400_000 and rs.rand() < 0.5:
            p = (rs.permutation(n) + 1).astype(np.int32)
            H.left_permute(p)
            H.right_permute(p)
            Ap = orc.permuted(A2, p, p)
            y = np.zeros(n); H.matvec(x, y)
            if not np.array_equal(y, Ap.matvec(x)): bad.append("matvec after permute")
            kern += " -> " + H.kernel
        H.destroy()
    finally:
        for k, v in DEFAULTS.items():
            sg.set_option(k, v)
    changed = {k: v for k, v in opts.items() if v != DEFAULTS[k]}
    if verbose or bad:
        print(f"seed {seed}: {kind} n={n} m={m} nnz={val.size} {changed} {kern}" + (f"  MISMATCH: {bad}" if bad else ""), flush=True)
    return bad


def one_ell(seed, verbose=True):
    """The ELLPACK twin (ellpack_matrices.f90:640-693): rows of 1..max_d neighbours padded the reference's way."""
    rs = np.random.RandomState(seed)
    kind = ["stencil", "band", "scattered", "ragged"][int(rs.randint(0, 4))]
    n = int(10 ** rs.uniform(2.5, 6.4))
    md = int(rs.choice([1, 2, 3, 5, 7, 8, 9, 12, 16, 20, 32, 40]))
    n = min(n, 12_000_000 // md)
    m = n if rs.rand() < 0.8 else max(1, int(n * rs.uniform(0.3, 2.0)))
    deg = np.full(n, md) if kind in ("stencil", "scattered") else rs.randint(1 if kind == "band" else 0, md + 1, size=n)
    if deg.max(initial=0) == 0:
        deg[0] = 1
    rows = np.repeat(np.arange(n), deg)
    if kind == "stencil":
        offs = np.sort(rs.choice(np.arange(-3 * md, 3 * md + 1), size=md, replace=False)) * int(rs.choice([1, 1, 37]))
        cols = rows + np.tile(offs, n)
        cols = np.clip(cols, 0, m - 1)                     # (clipped at the edges: repeated neighbours there, kept by set_value's rule)
    elif kind == "band":
        bw = int(10 ** rs.uniform(0.5, 3.5))
        cols = np.clip((rows * (m / n)).astype(np.int64) + rs.randint(-bw, bw + 1, size=rows.size), 0, m - 1)
    else:
        cols = rs.randint(0, m, size=rows.size)
    ei, ej, ev = (rows + 1).astype(np.int32), (cols + 1).astype(np.int32), rs.standard_normal(rows.size)
    E = orc.EllMatrix.from_edges(n, m, ei, ej, ev)
    opts = {"ell_offset_dict": int(rs.choice([1, 1, 0])), "csr_sliced": int(rs.choice([1, 1, 0])), "ell_colblock": int(rs.choice([1, 1, 2, 0])),
            "ell_colblock_cols": int(rs.choice([16384, 16384, 4096, 2048])), "ell_colblock_rows": int(rs.choice([0, 0, 256, 512]))}
    defaults = {"ell_offset_dict": 1, "csr_sliced": 1, "ell_colblock": 1, "ell_colblock_cols": 16384, "ell_colblock_rows": 0}
    x, y0, xt = rs.standard_normal(m), rs.standard_normal(n), rs.standard_normal(n)
    bad = []
    try:
        for k, v in opts.items():
            sg.set_option(k, v)
        H = sg.ellpack_matrix(n, m, E.node, E.val)
        kern = H.kernel
        y = np.zeros(n); H.matvec(x, y)
        if not np.array_equal(y, E.matvec(x)): bad.append("matvec")
        ya = y0.copy(); H.matvec_add(x, ya)
        if not np.array_equal(ya, E.matvec_add(x, y0.copy())): bad.append("matvec_add")
        t = np.zeros(m); H.matvec_t(xt, t)
        if not np.array_equal(t, E.matvec_t(xt)): bad.append("matvec_t")
        v2 = np.where(E.val != 0.0, rs.standard_normal(E.val.shape), E.val)         # new values on the same pattern
        H.set_values(v2)
        E2 = orc.EllMatrix(n, m, E.max_d, E.node, v2, E.degrees)
        y = np.zeros(n); H.matvec(x, y)
        if not np.array_equal(y, E2.matvec(x)): bad.append("matvec after set_values")
        t = np.zeros(m); H.matvec_t(xt, t)
        if not np.array_equal(t, E2.matvec_t(xt)): bad.append("matvec_t after set_values")
        H.destroy()
    finally:
        for k, v in defaults.items():
            sg.set_option(k, v)
    changed = {k: v for k, v in opts.items() if v != defaults[k]}
    if verbose or bad:
        print(f"seed {seed}: ell {kind} n={n} m={m} max_d={E.max_d} {changed} {kern}" + (f"  MISMATCH: {bad}" if bad else ""), flush=True)
    return bad


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    sg.init(0)
    t0 = time.time()
    failures = []
    count = 0
    while time.time() - t0 < seconds:
        if (one_ell if seed % 3 == 2 else one)(seed):
            failures.append(seed)
        seed += 1
        count += 1
    print(f"{count} matrices, failing seeds: {failures}")
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
