"""Deterministic synthetic inputs for the SpMV / Krylov hot path.

Every generator returns the matrix the way a SiGMA user builds it: an EDGE LIST
in insertion order (1-based ``ei, ej``) plus the value passed to
``A%set_value(ei, ej, ev)`` -- the call sequence of
``test/solver_test_diffusion_1d.f90:55-78`` and ``test/solver_test_jacobi.f90:73-128``.
Turning an edge list into the reference's ``ptr/node/val`` (CSR) or
``node(max_d,n)/val(max_d,n)`` (ELLPACK) arrays is the job of the reference's
``cs_graph_build`` / ``ellpack_graph_build`` (restated in ``oracle/``); the
``*_csr`` helpers below build the same arrays directly and vectorised for the
large benchmark sizes, and the tests check them bit-for-bit against the oracle's
graph build at small sizes.

All indices are 1-based int32 exactly as the Fortran host holds them.
"""
from __future__ import annotations

import numpy as np

I4 = np.int32
F8 = np.float64


# --------------------------------------------------------------------------- #
# edge lists (insertion order)
# --------------------------------------------------------------------------- #
def tridiag_edges(n, diag, upper, lower):
    """Insertion order of test/solver_test_diffusion_1d.f90:58-63:
    for i=1..n-1: (i,i),(i,i+1),(i+1,i); then (n,n)."""
    i = np.arange(1, n, dtype=np.int64)
    ei = np.empty(3 * (n - 1) + 1, dtype=np.int64)
    ej = np.empty_like(ei)
    ev = np.empty(ei.shape, dtype=F8)
    ei[0:-1:3], ej[0:-1:3], ev[0:-1:3] = i, i, diag
    ei[1:-1:3], ej[1:-1:3], ev[1:-1:3] = i, i + 1, upper
    ei[2:-1:3], ej[2:-1:3], ev[2:-1:3] = i + 1, i, lower
    ei[-1], ej[-1], ev[-1] = n, n, diag
    return ei.astype(I4), ej.astype(I4), ev


def diffusion_1d(n):
    """tridiag(-1,2,-1), f = 2 dx^2, analytic v(i)=i dx (1 - i dx)
    (test/solver_test_diffusion_1d.f90:55-94)."""
    dx = 1.0 / (n + 1)
    ei, ej, ev = tridiag_edges(n, 2.0, -1.0, -1.0)
    f = np.full(n, 2.0 * dx ** 2, dtype=F8)
    i = np.arange(1, n + 1, dtype=F8)
    v = i * dx * (1.0 - i * dx)
    return (ei, ej, ev), f, v


def advection_diffusion_1d(n, c=0.5):
    """tridiag(-1-c dx/2, 2, -1+c dx/2) (test/solver_test_advection_diffusion_1d.f90:55-101)."""
    dx = 1.0 / (n + 1)
    ei, ej, ev = tridiag_edges(n, 2.0, -1.0 + c * dx / 2, -1.0 - c * dx / 2)
    f = np.full(n, 2.0 * dx ** 2, dtype=F8)
    x = np.arange(1, n + 1, dtype=F8) * dx
    v = 2.0 * (x - (np.exp(c * x) - 1) / (np.exp(c) - 1)) / c
    return (ei, ej, ev), f, v


def poisson2d_edges(nx, ny):
    """5-point Laplacian; row k=(j-1)*nx+i; per-row insertion order S,W,C,E,N
    with values -1,-1,4,-1,-1, boundary neighbours dropped."""
    ptr, node, val = poisson2d_csr(nx, ny)
    n = nx * ny
    ei = np.repeat(np.arange(1, n + 1, dtype=I4), np.diff(ptr))
    return ei, node, val


def laplace3d_edges(nx, ny, nz):
    """7-point Laplacian; insertion order -z,-y,-x,C,+x,+y,+z; values -1..6..-1."""
    ptr, node, val = laplace3d_csr(nx, ny, nz)
    n = nx * ny * nz
    ei = np.repeat(np.arange(1, n + 1, dtype=I4), np.diff(ptr))
    return ei, node, val


def _stencil_csr(offsets_valid, centre_val, n):
    """offsets_valid: list of (offset, valid_mask, value) in insertion order."""
    k = len(offsets_valid)
    cols = np.empty((n, k), dtype=np.int64)
    vals = np.empty((n, k), dtype=F8)
    mask = np.empty((n, k), dtype=bool)
    row = np.arange(1, n + 1, dtype=np.int64)
    for s, (off, valid, v) in enumerate(offsets_valid):
        cols[:, s] = row + off
        vals[:, s] = v
        mask[:, s] = valid
    cnt = mask.sum(axis=1)
    ptr = np.empty(n + 1, dtype=np.int64)
    ptr[0] = 1
    np.cumsum(cnt, out=ptr[1:])
    ptr[1:] += 1
    return ptr.astype(I4), cols[mask].astype(I4), vals[mask]


def poisson2d_csr(nx, ny):
    """Direct (vectorised) CSR arrays of poisson2d_edges; 1-based ptr/node."""
    n = nx * ny
    k = np.arange(n, dtype=np.int64)
    i, j = k % nx, k // nx
    return _stencil_csr([(-nx, j > 0, -1.0), (-1, i > 0, -1.0),
                         (0, np.ones(n, bool), 4.0),
                         (+1, i < nx - 1, -1.0), (+nx, j < ny - 1, -1.0)], 4.0, n)


def laplace3d_csr(nx, ny, nz):
    n = nx * ny * nz
    k = np.arange(n, dtype=np.int64)
    i, j, l = k % nx, (k // nx) % ny, k // (nx * ny)
    return _stencil_csr([(-nx * ny, l > 0, -1.0), (-nx, j > 0, -1.0), (-1, i > 0, -1.0),
                         (0, np.ones(n, bool), 6.0),
                         (+1, i < nx - 1, -1.0), (+nx, j < ny - 1, -1.0),
                         (+nx * ny, l < nz - 1, -1.0)], 6.0, n)


def tridiag_csr(n, diag, upper, lower):
    """CSR arrays that cs_graph_build produces from tridiag_edges: the stored
    order inside row i is the insertion order: row 1: (1,1),(1,2);
    row i: (i,i-1),(i,i),(i,i+1) ... because edge (i+1,i) is inserted while
    the loop is at i, i.e. BEFORE (i+1,i+1)."""
    cnt = np.full(n, 3, dtype=np.int64)
    cnt[0] = 2
    cnt[-1] = 2
    if n == 1:
        cnt[0] = 1
    ptr = np.empty(n + 1, dtype=np.int64)
    ptr[0] = 1
    ptr[1:] = 1 + np.cumsum(cnt)
    nnz = int(ptr[-1] - 1)
    node = np.empty(nnz, dtype=np.int64)
    val = np.empty(nnz, dtype=F8)
    r = np.arange(1, n + 1, dtype=np.int64)
    p0 = ptr[:-1] - 1
    # row 1: (1,1),(1,2)
    node[p0[0]] = 1
    val[p0[0]] = diag
    if n > 1:
        node[p0[0] + 1] = 2
        val[p0[0] + 1] = upper
        # rows 2..n: lower, diag, [upper]
        node[p0[1:]] = r[1:] - 1
        val[p0[1:]] = lower
        node[p0[1:] + 1] = r[1:]
        val[p0[1:] + 1] = diag
        node[p0[1:-1] + 2] = r[1:-1] + 1
        val[p0[1:-1] + 2] = upper
    return ptr.astype(I4), node.astype(I4), val


# --------------------------------------------------------------------------- #
# random problems (deterministic: fixed-seed generators, no time seeding)
# --------------------------------------------------------------------------- #
_LCG_A = np.uint64(6364136223846793005)
_LCG_C = np.uint64(1442695040888963407)


def _lcg_draws(seeds, ndraw):
    """ndraw successive 64-bit LCG states for every seed (vectorised, wrapping)."""
    out = np.empty((seeds.shape[0], ndraw), dtype=np.uint64)
    s = seeds.astype(np.uint64).copy()
    with np.errstate(over="ignore"):
        for t in range(ndraw):
            s = s * _LCG_A + _LCG_C
            out[:, t] = s
    return out


def random_regular_ell(n, d=32, seed=12345, dmin=None):
    """Random digraph in ELLPACK shape: row i holds (i,i) and
    then d-1 distinct off-diagonal columns drawn from a per-row 64-bit LCG
    (state0 = seed + i*0x9E3779B97F4A7C15, s <- s*6364136223846793005 +
    1442695040888963407, column = 1 + (s >> 16) mod n; a draw equal to the row
    or to an earlier pick is skipped).  With ``dmin`` the row degree is
    dmin + (i mod (d-dmin+1)) instead of d (exercises ELLPACK padding).
    value(slot k, row i) = 1/(k + (i mod 7)), 1-based k, i.

    Returns the edge list in insertion order (row by row)."""
    assert n > 2 * d
    rows = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        seeds = np.uint64(seed) + rows * np.uint64(0x9E3779B97F4A7C15)
    ndraw = d + 16
    draws = _lcg_draws(seeds, ndraw)
    cols = (1 + (draws >> np.uint64(16)) % np.uint64(n)).astype(np.int64)
    cand = np.concatenate([rows.astype(np.int64)[:, None], cols], axis=1)   # slot 0 = diagonal
    # keep first occurrences in order
    order = np.argsort(cand, axis=1, kind="stable")
    srt = np.take_along_axis(cand, order, axis=1)
    dup_sorted = np.zeros_like(srt, dtype=bool)
    dup_sorted[:, 1:] = srt[:, 1:] == srt[:, :-1]
    dup = np.empty_like(dup_sorted)
    np.put_along_axis(dup, order, dup_sorted, axis=1)
    keep = ~dup
    rank = np.cumsum(keep, axis=1)
    if dmin is None:
        deg = np.full(n, d, dtype=np.int64)
    else:
        deg = dmin + (np.arange(1, n + 1, dtype=np.int64) % (d - dmin + 1))
    keep &= rank <= deg[:, None]
    assert (keep.sum(axis=1) == deg).all(), "not enough distinct draws"
    ej = cand[keep]
    ei = np.repeat(np.arange(1, n + 1, dtype=np.int64), deg)
    slot = (rank[keep]).astype(np.int64)           # 1-based slot inside the row
    ev = 1.0 / (slot + (ei % 7)).astype(F8)
    return ei.astype(I4), ej.astype(I4), ev


# --------------------------------------------------------------------------- #
# the same stencil matrices generated with torch on a device (plumbing for the
# full-size configs: no 1 GB host arrays, no PCIe upload); entry order = the
# numpy generators' (tests compare them at small sizes)
# --------------------------------------------------------------------------- #
def stencil_csr_torch(k, offs_masks_vals, device):
    """Rows k (0-based global row numbers, int64 tensor) of a stencil matrix: offs_masks_vals = [(column offset, mask of
    the rows that hold the entry, value)] in stored order.  Returns local 1-based ptr (int32), GLOBAL 1-based node
    (int32), val (float64)."""
    import torch
    n = k.numel()
    cols = torch.stack([k + 1 + o for o, _, _ in offs_masks_vals], dim=1)
    mask = torch.stack([m for _, m, _ in offs_masks_vals], dim=1)
    vals = torch.tensor([v for _, _, v in offs_masks_vals], dtype=torch.float64, device=device).expand(n, -1)
    ptr = torch.ones(n + 1, dtype=torch.int64, device=device)
    ptr[1:] += torch.cumsum(mask.sum(dim=1), 0)
    return ptr.to(torch.int32), cols[mask].to(torch.int32), vals[mask].contiguous()


def tridiag_csr_torch(n, diag, upper, lower, device):
    """tridiag_csr on a device: the reference's insertion order leaves every row as (i,i-1),(i,i),(i,i+1)."""
    import torch
    k = torch.arange(n, device=device, dtype=torch.int64)
    one = torch.ones(n, dtype=torch.bool, device=device)
    return stencil_csr_torch(k, [(-1, k > 0, lower), (0, one, diag), (1, k < n - 1, upper)], device)


def laplace3d_rows_torch(nx, ny, nz, device, z0=0, z1=None):
    """Rows of the planes [z0, z1) of laplace3d_csr(nx, ny, nz) (entry order -z,-y,-x,C,+x,+y,+z)."""
    import torch
    z1 = nz if z1 is None else z1
    pl = nx * ny
    k = torch.arange(z0 * pl, z1 * pl, device=device, dtype=torch.int64)
    i, j, l = k % nx, (k // nx) % ny, k // pl
    one = torch.ones_like(k, dtype=torch.bool)
    return stencil_csr_torch(k, [(-pl, l > 0, -1.0), (-nx, j > 0, -1.0), (-1, i > 0, -1.0), (0, one, 6.0), (1, i < nx - 1, -1.0),
                                 (nx, j < ny - 1, -1.0), (pl, l < nz - 1, -1.0)], device)


def random_regular_ell_torch(n, d=32, seed=12345, device=None, chunk=1 << 20):
    """random_regular_ell(n, d, seed) (no padding variant) evaluated with torch on `device`, row chunk by
    row chunk: returns (node, val) as (n, d) int32 / float64 tensors -- the ELLPACK arrays the edge list of
    the numpy generator assembles to (every row holds exactly d slots in insertion order), entry for entry
    (tests/test_oracle_golden.py compares the two at a small size).  int64 arithmetic wraps like the uint64
    LCG; the logical shift is an arithmetic one with the sign bits masked off."""
    import torch
    assert n > 2 * d
    dev = device if device is not None else torch.device("cpu")
    A = 6364136223846793005
    Cc = 1442695040888963407
    G = 0x9E3779B97F4A7C15 - (1 << 64)          # the golden-ratio increment as a two's-complement int64
    ndraw = d + 16
    node = torch.empty((n, d), dtype=torch.int32, device=dev)
    val = torch.empty((n, d), dtype=torch.float64, device=dev)
    slot = torch.arange(1, d + 1, dtype=torch.int64, device=dev)
    for r0 in range(0, n, chunk):
        r1 = min(n, r0 + chunk)
        rows = torch.arange(r0 + 1, r1 + 1, dtype=torch.int64, device=dev)
        s = seed + rows * G
        cand = torch.empty((r1 - r0, ndraw + 1), dtype=torch.int64, device=dev)
        cand[:, 0] = rows
        for t in range(ndraw):
            s = s * A + Cc
            cand[:, t + 1] = 1 + ((s >> 16) & 0x0000FFFFFFFFFFFF) % n
        srt, order = torch.sort(cand, dim=1, stable=True)
        dup_sorted = torch.zeros_like(srt, dtype=torch.bool)
        dup_sorted[:, 1:] = srt[:, 1:] == srt[:, :-1]
        dup = torch.empty_like(dup_sorted)
        dup.scatter_(1, order, dup_sorted)
        keep = ~dup
        rank = torch.cumsum(keep, dim=1)
        keep &= rank <= d
        if not bool((keep.sum(dim=1) == d).all()):
            raise AssertionError("not enough distinct draws")
        node[r0:r1] = cand[keep].view(r1 - r0, d).to(torch.int32)
        val[r0:r1] = 1.0 / (slot[None, :] + (rows % 7)[:, None]).to(torch.float64)
    return node, val


def random_spd_edges(n, seed=1, p=None, skew=False):
    """The matrix family of test/solver_test_jacobi.f90:62-128 (random graph
    Laplacian + I, optionally with the skew perturbation of :240-257), made
    reproducible with numpy's MT19937 RandomState(seed) instead of the
    reference's time-seeded RNG.  Edge insertion order as in the test:
    for i: (i,i); for j>i with z<p: (i,j),(j,i)."""
    rs = np.random.RandomState(seed)
    if p is None:
        p = np.log2(n) / n
    ei, ej = [], []
    und = []
    for i in range(1, n + 1):
        ei.append(i)
        ej.append(i)
        z = rs.random_sample(n - i)
        for j in np.nonzero(z < p)[0] + i + 1:
            ei += [i, int(j)]
            ej += [int(j), i]
            und.append((i, int(j)))
    w = rs.random_sample(len(und))
    A = {}
    diag = np.ones(n + 1)
    for (i, j), z in zip(und, w):
        A[(i, j)] = -z
        A[(j, i)] = -z
        diag[i] += z
        diag[j] += z
    if skew:
        s = (2 * rs.random_sample(len(und)) - 1) / 16
        for (i, j), z in zip(und, s):
            A[(i, j)] += z
            A[(j, i)] -= z
    ev = [diag[i] if i == j else A[(i, j)] for i, j in zip(ei, ej)]
    return np.array(ei, I4), np.array(ej, I4), np.array(ev, F8)


def test_vector(m):
    """x(i) = sin(0.001 i), 1-based i."""
    return np.sin(0.001 * np.arange(1, m + 1, dtype=F8))
