#!/usr/bin/env python3
"""One-off stress (not part of the suite): seeded random square matrices of many shapes through the device-side ILDU(0)
setup and every apply path -- factors and applies against the oracle's statement-for-statement restatement, bit for bit;
with and without a colour ordering, row-space sweeps on / off, pipelines on / off, a second setup with new values."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import sigma_amd as sg
from oracle import oracle as orc
sg.init(0)


def gen(rs):
    kind = rs.choice(["grid2", "grid3", "band", "random", "dups", "holes"])
    if kind == "grid2":
        nx, ny = int(rs.randint(2, 180)), int(rs.randint(2, 180)); n = nx * ny
        r = np.arange(n); i, j = [], []
        for d, ok in ((1, (r % nx) < nx - 1), (nx, r + nx < n)):
            i += [r[ok], r[ok] + d]; j += [r[ok] + d, r[ok]]
        i, j = np.concatenate(i), np.concatenate(j)
    elif kind == "grid3":
        nx, ny, nz = int(rs.randint(2, 70)), int(rs.randint(2, 40)), int(rs.randint(2, 30)); n = nx * ny * nz
        r = np.arange(n); i, j = [], []
        for d, ok in ((1, (r % nx) < nx - 1), (nx, (r // nx) % ny < ny - 1), (nx * ny, r + nx * ny < n)):
            i += [r[ok], r[ok] + d]; j += [r[ok] + d, r[ok]]
        i, j = np.concatenate(i), np.concatenate(j)
    else:
        n = int(rs.choice([1, 2, 17, 300, 2049, 9000, 40000]))
        deg = int(rs.choice([1, 2, 3, 5, 9, 14]))
        i = np.repeat(np.arange(n), deg)
        if kind == "band":
            j = np.clip(i + rs.randint(-12, 13, size=i.size), 0, n - 1)
        else:
            j = rs.randint(0, n, size=i.size)
    off = i != j
    i, j = i[off], j[off]
    if kind != "dups" and i.size:
        key = np.unique(i.astype(np.int64) * n + j)
        i, j = (key // n), (key % n)
    rows = np.arange(n)
    if kind == "holes" and n > 5:
        rows = rows[rs.rand(n) > 0.02]                       # a few missing diagonals: inf / nan factors, the same ones
    v = -rs.uniform(0.05, 1.0, size=i.size)
    ri = np.concatenate([i, rows]).astype(np.int64); rj = np.concatenate([j, rows]).astype(np.int64)
    rv = np.concatenate([v, np.full(rows.size, 4.0 + 2.0 * (i.size / max(n, 1)))])
    o = np.lexsort((rs.rand(ri.size), ri))
    ri, rj, rv = ri[o], rj[o], rv[o]
    ptr = (np.concatenate([[0], np.cumsum(np.bincount(ri, minlength=n))]) + 1).astype(np.int32)
    return kind, n, ptr, (rj + 1).astype(np.int32), rv


def check(tag, A, H, bad):
    if os.environ.get("STRESS_TRACE"):
        print("  check", tag, flush=True)
    ref = orc.Ildu(A)
    pc = sg.ldu(); pc.setup(H)
    for nm, dt, want in (("Lptr", np.int32, ref.Lptr), ("Lnode", np.int32, ref.Lnode), ("Uptr", np.int32, ref.Uptr), ("Unode", np.int32, ref.Unode),
                         ("Lval", np.float64, ref.Lval), ("Uval", np.float64, ref.Uval), ("D", np.float64, ref.D)):
        got = pc.get(nm, dt)
        if not np.array_equal(got, want[:got.size], equal_nan=True):
            bad.append((tag, nm)); return
    b = np.random.RandomState(A.n).standard_normal(A.n)
    want = ref.solve(b)
    for rows_on, strips_on in ((1, 1), (0, 1), (1, 0), (2, 0), (0, 0)):
        pc.set_option("ildu_rows", rows_on); pc.set_option("ildu_strips", strips_on)
        if os.environ.get("STRESS_TRACE"):
            print("    apply", rows_on, strips_on, flush=True); sg.synchronize()
        z = np.zeros(A.n); pc.solve(H, z, b)
        if not np.array_equal(z, want, equal_nan=True):
            bad.append((tag, "apply", rows_on, strips_on))
    pc.set_option("ildu_rows", 1); pc.set_option("ildu_strips", 1)
    v2 = A.val * (1.0 + 0.1 * np.cos(np.arange(A.val.size)))
    H.set_values(v2); pc.setup(H)
    A2 = orc.CsrMatrix(A.n, A.n, A.ptr, A.node, v2)
    ref2 = orc.Ildu(A2)
    z = np.zeros(A.n); pc.solve(H, z, b)
    if not np.array_equal(z, ref2.solve(b), equal_nan=True):
        bad.append((tag, "apply after new values"))
    H.set_values(A.val)
    pc.destroy()


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    bad, seen = [], {}
    for t in range(trials):
        rs = np.random.RandomState(5000 + t)
        kind, n, ptr, node, val = gen(rs)
        A = orc.CsrMatrix(n, n, ptr, node, val)
        H = sg.csr_matrix(n, n, ptr, node, val)
        seen[kind] = seen.get(kind, 0) + 1
        if os.environ.get("STRESS_TRACE"):
            print("trial", t, kind, n, len(node), flush=True)
        check((t, kind, n, "natural"), A, H, bad)
        if n > 1 and kind != "holes":
            try:
                p, ptrs, nc = H.greedy_color_ordering()
            except sg.SigmaError:                             # (the ordering walks the graph from vertex 1: a disconnected one is refused)
                seen["not connected"] = seen.get("not connected", 0) + 1
                H.destroy()
                continue
            try:                                              # the device ordering (union-find / level sweep / host pass) is the reference's
                pref, _, ncref = orc.greedy_color_ordering(A)
                if not np.array_equal(p, pref) or int(nc) != int(ncref):
                    bad.append(((t, kind, n), "greedy_color_ordering"))
            except ValueError:
                bad.append(((t, kind, n), "greedy_color_ordering: oracle refuses, library orders"))
            H.left_permute(p); H.right_permute(p)
            B = orc.permuted(A, p, p)
            check((t, kind, n, "colour", int(nc)), B, H, bad)
        H.destroy()
    print(json.dumps({"trials": trials, "mismatches": len(bad), "first": [str(b) for b in bad[:8]], "kinds": seen}))


if __name__ == "__main__":
    main()
