#!/usr/bin/env python3
"""One-off stress (not part of the suite): many seeded random matrices of every flavour through every CSR kernel
combination, each product against the first combination's bits and a host evaluation of sampled rows in stored order."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import sigma_amd as sg
sg.init(0)
# (offset dictionary, sliced forms, row owner, row lines, SELL-128-512: 2 = whenever its padding allows)
COMBOS = ((1, 1, 1, 1, 1), (1, 1, 1, 1, 2), (1, 0, 1, 1, 0), (0, 1, 1, 1, 2), (0, 1, 1, 1, 0), (0, 0, 1, 1, 0), (0, 0, 0, 1, 0), (0, 0, 0, 0, 0))


def opts(c):
    for name, v in zip(("csr_offset_dict", "csr_sliced", "csr_row_owner", "csr_row_lines", "csr_sell"), c):
        sg.set_option(name, v)


def host_rows(ptr, node, val, x, rows):
    out = {}
    for r in rows:
        z = 0.0
        for k in range(ptr[r] - 1, ptr[r + 1] - 1):
            z = z + val[k] * x[node[k] - 1]
        out[r] = 0.0 + z
    return out


def gen(rs):
    kind = rs.choice(["banded", "many", "short", "ragged", "long_uniform", "long_mixed"])
    n = int(rs.choice([1, 7, 64, 255, 257, 511, 513, 1000, 2049, 5000, 20011]))
    m = int(rs.choice([n, n, n + 13]))
    if kind == "banded":
        w = int(rs.choice([1, 3, 5, 7, 8]))
        offs = np.sort(rs.choice(np.arange(-9, 10), size=min(w + 2, 15), replace=False))
        deg = np.where(rs.rand(n) < 0.9, w, rs.randint(0, w + 1, size=n))
        rows = np.repeat(np.arange(n), deg)
        cols = np.clip(rows + offs[rs.randint(0, len(offs), size=rows.size)], 0, m - 1)
    elif kind == "many":
        deg = rs.randint(0, 30, size=n)
        rows = np.repeat(np.arange(n), deg)
        cols = np.clip(rows + rs.randint(-60, 61, size=rows.size), 0, m - 1)
    elif kind == "short":
        w = int(rs.choice([3, 6, 8, 11, 16, 20, 27, 32]))
        deg = np.where(rs.rand(n) < 0.9, w, rs.randint(0, w + 1, size=n))
        rows = np.repeat(np.arange(n), deg)
        cols = rs.randint(0, m, size=rows.size)
    elif kind == "ragged":
        deg = rs.randint(0, 6, size=n)
        deg[rs.randint(0, n, size=max(1, n // 50))] = rs.randint(40, 6000)
        rows = np.repeat(np.arange(n), deg)
        cols = rs.randint(0, m, size=rows.size)
    elif kind == "long_uniform":
        lo = int(rs.choice([17, 33, 65, 100, 300])); hi = int(lo * rs.uniform(1.0, 2.0))
        deg = rs.randint(lo, hi + 1, size=n)
        rows = np.repeat(np.arange(n), deg)
        cols = np.clip(rows + rs.randint(-3 * hi, 3 * hi + 1, size=rows.size), 0, m - 1)
    else:
        deg = rs.randint(10, 40, size=n)
        deg[rs.randint(0, n, size=max(1, n // 100))] = rs.randint(200, 3000)
        rows = np.repeat(np.arange(n), deg)
        cols = np.clip(rows + rs.randint(-4000, 4001, size=rows.size), 0, m - 1)
    ptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(np.int32)
    return kind, n, m, ptr, (cols + 1).astype(np.int32), rs.standard_normal(rows.size)


def main():
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seen = {}
    bad = 0
    for t in range(trials):
        rs = np.random.RandomState(1000 + t)
        kind, n, m, ptr, node, val = gen(rs)
        x, y0 = rs.standard_normal(m), rs.standard_normal(n)
        ref = None
        for c in COMBOS:
            opts(c)
            H = sg.csr_matrix(n, m, ptr, node, val)
            seen[H.kernel.split("<")[0] + ("/CW4" if "CW=4" in H.kernel else "")] = seen.get(H.kernel.split("<")[0] + ("/CW4" if "CW=4" in H.kernel else ""), 0) + 1
            y = np.zeros(n); H.matvec(x, y)
            ya = y0.copy(); H.matvec_add(x, ya)
            tt = np.zeros(m); H.matvec_t(y0, tt)
            if ref is None:
                ref = (y.copy(), ya.copy(), tt.copy())
                rows = rs.randint(0, n, size=min(n, 40))
                hr = host_rows(ptr, node, val, x, rows)
                for r, z in hr.items():
                    if not (y[r] == z or (np.isnan(z) and np.isnan(y[r]))):
                        bad += 1; print("ROW MISMATCH", t, kind, n, c, r)
            else:
                if not (np.array_equal(y, ref[0]) and np.array_equal(ya, ref[1]) and np.array_equal(tt, ref[2])):
                    bad += 1; print("KERNEL MISMATCH", t, kind, n, m, c, H.kernel)
            H.destroy()
    opts((1, 1, 1, 1, 1))
    print(json.dumps({"trials": trials, "mismatches": bad, "kernels_seen": seen}))


if __name__ == "__main__":
    main()
