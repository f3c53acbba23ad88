"""ctypes face of oracle/sigma_oracle.c (liborc.so) -- TEST INFRASTRUCTURE ONLY.

Mirrors the reference objects loosely: CsrMatrix / EllMatrix built from an edge list
the way the reference's ll_graph -> cs_graph/ellpack_graph copy does it, plus
cg / bicgstab / gmres with optional jacobi / ildu preconditioners.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

I4 = np.int32
F8 = np.float64


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    src = os.path.join(_HERE, "sigma_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_cs_graph_build.restype = C.c_int64
        _LIB.orc_max_degree.restype = C.c_int32
        _LIB.orc_cg.restype = C.c_int64
        _LIB.orc_bicgstab.restype = C.c_int64
        _LIB.orc_gmres.restype = C.c_int64
        _LIB.orc_dot.restype = C.c_double
        _LIB.orc_time_csr_matvec.restype = C.c_double
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _SolveArgs(C.Structure):
    _fields_ = [("fmt", C.c_int32), ("n", C.c_int32), ("max_d", C.c_int32),
                ("ptr", C.c_void_p), ("node", C.c_void_p), ("val", C.c_void_p),
                ("pc_kind", C.c_int32), ("idiag", C.c_void_p),
                ("Lptr", C.c_void_p), ("Lnode", C.c_void_p), ("Lval", C.c_void_p),
                ("Uptr", C.c_void_p), ("Unode", C.c_void_p), ("Uval", C.c_void_p),
                ("D", C.c_void_p),
                ("tol", C.c_double), ("max_iter", C.c_int64),
                ("history", C.c_void_p), ("hist_cap", C.c_int64)]


class CsrMatrix:
    """cs_matrix restatement: 1-based ptr/node, val (cs_matrices.f90:32-38)."""
    fmt = 1

    def __init__(self, n, m, ptr, node, val):
        self.n, self.m = int(n), int(m)
        self.ptr = np.ascontiguousarray(ptr, I4)
        self.node = np.ascontiguousarray(node, I4)
        self.val = np.ascontiguousarray(val, F8)
        self.max_d = int(np.diff(self.ptr).max()) if n else 0
        self.degrees = None

    @classmethod
    def from_edges(cls, n, m, ei, ej, ev):
        ei = np.ascontiguousarray(ei, I4)
        ej = np.ascontiguousarray(ej, I4)
        ev = np.ascontiguousarray(ev, F8)
        ptr = np.zeros(n + 1, I4)
        node = np.zeros(max(len(ei), 1), I4)
        md = C.c_int32(0)
        ne = lib().orc_cs_graph_build(C.c_int32(n), C.c_int64(len(ei)), _p(ei), _p(ej),
                                      _p(ptr), _p(node), C.byref(md))
        node = node[:ne].copy()
        val = np.zeros(ne, F8)
        lib().orc_csr_set_values(C.c_int32(n), _p(ptr), _p(node), _p(val),
                                 C.c_int64(len(ei)), _p(ei), _p(ej), _p(ev))
        return cls(n, m, ptr, node, val)

    @property
    def nnz(self):
        return int(self.ptr[-1] - 1)

    def matvec_add(self, x, y):
        x = np.ascontiguousarray(x, F8)
        lib().orc_csr_matvec_add(C.c_int32(self.n), _p(self.ptr), _p(self.node),
                                 _p(self.val), _p(x), _p(y))
        return y

    def matvec(self, x):
        y = np.full(self.n, -7.0)
        x = np.ascontiguousarray(x, F8)
        lib().orc_matvec(C.c_int32(1), C.c_int32(self.n), C.c_int32(0), _p(self.ptr),
                         _p(self.node), _p(self.val), _p(x), _p(y))
        return y

    def matvec_t_add(self, x, y):
        x = np.ascontiguousarray(x, F8)
        lib().orc_csr_matvec_t_add(C.c_int32(self.n), _p(self.ptr), _p(self.node), _p(self.val),
                                   _p(x), _p(y))
        return y

    def matvec_t(self, x):       # linear_operator_interface.f90:199-208: y = 0 ; matvec_t_add
        return self.matvec_t_add(x, np.zeros(self.m, F8))


class EllMatrix:
    """ellpack_matrix restatement: node/val as Fortran (max_d, n) column-major, kept
    here as C arrays of shape (n, max_d) (same memory)."""
    fmt = 2

    def __init__(self, n, m, max_d, node, val, degrees):
        self.n, self.m, self.max_d = int(n), int(m), int(max_d)
        self.node = np.ascontiguousarray(node, I4).reshape(n, max_d)
        self.val = np.ascontiguousarray(val, F8).reshape(n, max_d)
        self.degrees = np.ascontiguousarray(degrees, I4)
        self.ptr = None

    @classmethod
    def from_edges(cls, n, m, ei, ej, ev):
        ei = np.ascontiguousarray(ei, I4)
        ej = np.ascontiguousarray(ej, I4)
        ev = np.ascontiguousarray(ev, F8)
        md = lib().orc_max_degree(C.c_int32(n), C.c_int64(len(ei)), _p(ei), _p(ej))
        node = np.zeros((n, md), I4)
        deg = np.zeros(n, I4)
        lib().orc_ellpack_graph_build(C.c_int32(n), C.c_int64(len(ei)), _p(ei), _p(ej),
                                      C.c_int32(md), _p(node), _p(deg))
        val = np.zeros((n, md), F8)
        lib().orc_ell_set_values(C.c_int32(n), C.c_int32(md), _p(node), _p(deg), _p(val),
                                 C.c_int64(len(ei)), _p(ei), _p(ej), _p(ev))
        return cls(n, m, md, node, val, deg)

    def matvec_add(self, x, y):
        x = np.ascontiguousarray(x, F8)
        lib().orc_ell_matvec_add(C.c_int32(self.n), C.c_int32(self.max_d), _p(self.node),
                                 _p(self.val), _p(x), _p(y))
        return y

    def matvec(self, x):
        y = np.full(self.n, -7.0)
        x = np.ascontiguousarray(x, F8)
        lib().orc_matvec(C.c_int32(2), C.c_int32(self.n), C.c_int32(self.max_d), None,
                         _p(self.node), _p(self.val), _p(x), _p(y))
        return y

    def matvec_t_add(self, x, y):
        x = np.ascontiguousarray(x, F8)
        lib().orc_ell_matvec_t_add(C.c_int32(self.n), C.c_int32(self.max_d), _p(self.node),
                                   _p(self.val), _p(x), _p(y))
        return y

    def matvec_t(self, x):
        return self.matvec_t_add(x, np.zeros(self.m, F8))


class Jacobi:
    kind = 1

    def __init__(self, A):
        self.n = A.n
        self.idiag = np.zeros(A.n, F8)
        lib().orc_jacobi_setup(C.c_int32(A.fmt), C.c_int32(A.n), C.c_int32(A.max_d),
                               _p(A.ptr), _p(A.node), _p(A.val), _p(A.degrees),
                               _p(self.idiag))

    def solve(self, b):
        x = np.zeros(self.n, F8)
        b = np.ascontiguousarray(b, F8)
        lib().orc_jacobi_solve(C.c_int32(self.n), _p(self.idiag), _p(x), _p(b))
        return x


def ell_real_entries(E):
    """What the reference's edge cursor hands out for an ellpack_matrix (ellpack_graphs.f90:310-369, get_entries
    ellpack_matrices.f90): row after row the first degrees(i) slots of node(:, i) / val(:, i) -- the same stream a CSR matrix
    with exactly these rows yields (cs_graphs.f90 cursor), which is all incomplete_ldu_sparsity_pattern (ldu_solvers.f90:397-440)
    and the fill loop of sparse_static_pattern_ldu_factorization (:306-321) see of A.  Returned as that CsrMatrix."""
    deg = np.asarray(E.degrees, np.int64)
    ptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(I4)
    keep = np.arange(E.max_d)[None, :] < deg[:, None]
    return CsrMatrix(E.n, E.m, ptr, np.ascontiguousarray(E.node[keep], I4), np.ascontiguousarray(E.val[keep], F8))


class Ildu:
    """sparse_ldu_solver, ILDU(0) (ldu_solvers.f90:95-176) of a CSR matrix, or of an ELLPACK matrix through its edge stream."""
    kind = 2

    def __init__(self, A):
        if A.fmt == 2:
            A = ell_real_entries(A)
        n = self.n = A.n
        self.Lptr = np.zeros(n + 1, I4)
        self.Uptr = np.zeros(n + 1, I4)
        nl, nu = C.c_int64(0), C.c_int64(0)
        lib().orc_ildu_pattern(C.c_int32(n), _p(A.ptr), _p(A.node), _p(self.Lptr), None,
                               _p(self.Uptr), None, C.byref(nl), C.byref(nu))
        self.Lnode = np.zeros(max(nl.value, 1), I4)
        self.Unode = np.zeros(max(nu.value, 1), I4)
        lib().orc_ildu_pattern(C.c_int32(n), _p(A.ptr), _p(A.node), _p(self.Lptr),
                               _p(self.Lnode), _p(self.Uptr), _p(self.Unode),
                               C.byref(nl), C.byref(nu))
        self.Lnode = self.Lnode[:nl.value].copy()
        self.Unode = self.Unode[:nu.value].copy()
        self.Lval = np.zeros(max(nl.value, 1), F8)
        self.Uval = np.zeros(max(nu.value, 1), F8)
        self.D = np.zeros(n, F8)
        lib().orc_ildu_factor(C.c_int32(n), _p(A.ptr), _p(A.node), _p(A.val),
                              _p(self.Lptr), _p(self.Lnode), _p(self.Lval),
                              _p(self.Uptr), _p(self.Unode), _p(self.Uval), _p(self.D))
        self.Lval = self.Lval[:nl.value].copy() if nl.value else self.Lval
        self.Uval = self.Uval[:nu.value].copy() if nu.value else self.Uval

    def solve(self, b):
        x = np.zeros(self.n, F8)
        b = np.ascontiguousarray(b, F8)
        lib().orc_ldu_solve(C.c_int32(self.n), _p(self.Lptr), _p(self.Lnode), _p(self.Lval),
                            _p(self.D), _p(self.Uptr), _p(self.Unode), _p(self.Uval),
                            _p(x), _p(b))
        return x


def _args(A, pc, tol, max_iter, history):
    a = _SolveArgs()
    a.fmt, a.n, a.max_d = A.fmt, A.n, A.max_d
    a.ptr, a.node, a.val = _p(A.ptr), _p(A.node), _p(A.val)
    a.pc_kind = 0 if pc is None else pc.kind
    if pc is not None and pc.kind == 1:
        a.idiag = _p(pc.idiag)
    if pc is not None and pc.kind == 2:
        a.Lptr, a.Lnode, a.Lval = _p(pc.Lptr), _p(pc.Lnode), _p(pc.Lval)
        a.Uptr, a.Unode, a.Uval = _p(pc.Uptr), _p(pc.Unode), _p(pc.Uval)
        a.D = _p(pc.D)
    a.tol, a.max_iter = float(tol), int(max_iter)
    a.history = _p(history)
    a.hist_cap = 0 if history is None else len(history)
    return a


def _solve(fn, A, b, x0, pc, tol, max_iter, history, extra=()):
    x = np.zeros(A.n, F8) if x0 is None else np.array(x0, F8)
    b = np.ascontiguousarray(b, F8)
    hist = None if not history else np.zeros(history, F8)
    a = _args(A, pc, tol, max_iter, hist)
    res = C.c_double(0.0)
    its = fn(C.byref(a), *extra, _p(x), _p(b), C.byref(res))
    if hist is not None:
        hist = hist[:min(its, len(hist))]
    return x, int(its), res.value, hist


def cg(A, b, x0=None, pc=None, tol=1e-16, max_iter=0, history=0):
    """Returns (x, iterations, res2, history-of-res2)."""
    return _solve(lib().orc_cg, A, b, x0, pc, tol, max_iter, history)


def bicgstab(A, b, x0=None, pc=None, tol=1e-16, max_iter=0, history=0):
    return _solve(lib().orc_bicgstab, A, b, x0, pc, tol, max_iter, history)


def gmres(A, b, x0=None, pc=None, tol=1e-16, max_iter=0, restart=30, history=0, orth="mgs"):
    """Returns (x, iterations, |residual|, history-of-res^2).  No reference counterpart.
    orth: "mgs" (modified Gram-Schmidt) or "cgs2" (classical Gram-Schmidt twice)."""
    lib().orc_set_gmres_orth(C.c_int(1 if orth == "cgs2" else 0))
    return _solve(lib().orc_gmres, A, b, x0, pc, tol, max_iter, history,
                  extra=(C.c_int32(restart),))


def lanczos(A, nsteps, q1):
    """lanczos(A, T, Q) (eigensolver.f90:27-90) with the start vector given: returns
    (T[3, nsteps] as the reference indexes it, Q[n, nsteps])."""
    T = np.zeros((nsteps, 3), F8)
    Q = np.zeros((nsteps, A.n), F8)
    q1 = np.ascontiguousarray(q1, F8)
    lib().orc_lanczos(C.c_int32(A.fmt), C.c_int32(A.n), C.c_int32(A.max_d), _p(A.ptr), _p(A.node),
                      _p(A.val), C.c_int32(nsteps), _p(q1), _p(T), _p(Q))
    return T.T.copy(), Q.T.copy()


def generalized_lanczos(A, B, nsteps, q1, tol=1e-14):
    """generalized_lanczos(A, B, T, Q) (eigensolver.f90:95-155), B%solve = CG(tol), start vector
    given: returns (T[3, nsteps], Q[n, nsteps]).  CSR operands."""
    T = np.zeros((nsteps, 3), F8)
    Q = np.zeros((nsteps, A.n), F8)
    q1 = np.ascontiguousarray(q1, F8)
    lib().orc_generalized_lanczos(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(A.val), _p(B.ptr), _p(B.node), _p(B.val),
                                  C.c_double(tol), C.c_int32(nsteps), _p(q1), _p(T), _p(Q))
    return T.T.copy(), Q.T.copy()


def bfs_order(A):
    """breadth_first_search(p, g) (permutations.f90:22-78) on the matrix graph: p(i) = visiting number."""
    p = np.zeros(A.n, I4)
    lib().orc_bfs_order(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(p))
    return p


def greedy_coloring(A):
    """greedy_coloring(colors, g) (permutations.f90:83-157)."""
    c = np.zeros(A.n, I4)
    lib().orc_greedy_coloring(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(c))
    return c


def greedy_color_ordering(A):
    """greedy_color_ordering(p, ptrs, num_colors, g) (permutations.f90:162-205):
    returns (p, ptrs[:num_colors+1], num_colors)."""
    p = np.zeros(A.n, I4)
    ptrs = np.zeros(A.n + 2, I4)
    lib().orc_greedy_color_ordering.restype = C.c_int32
    nc = lib().orc_greedy_color_ordering(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(p), _p(ptrs))
    if nc < 0:
        raise ValueError("greedy_color_ordering: the graph is not connected from vertex 1")
    return p, ptrs[:nc + 1].copy(), int(nc)


def permuted(A, p_left=None, p_right=None):
    """A%left_permute(p_left) then A%right_permute(p_right) (cs_matrices.f90:471-490) as a new CsrMatrix."""
    ptr, node, val = A.ptr.copy(), A.node.copy(), A.val.copy()
    if p_left is not None:
        p_left = np.ascontiguousarray(p_left, I4)
        ptr2, node2, val2 = np.zeros_like(ptr), np.zeros_like(node), np.zeros_like(val)
        lib().orc_csr_left_permute(C.c_int32(A.n), _p(ptr), _p(node), _p(val), _p(p_left), _p(ptr2), _p(node2), _p(val2))
        ptr, node, val = ptr2, node2, val2
    if p_right is not None:
        p_right = np.ascontiguousarray(p_right, I4)
        lib().orc_csr_right_permute(C.c_int64(len(node)), _p(node), _p(p_right))
    return CsrMatrix(A.n, A.m, ptr, node, val)


def ell_graph_as_csr(E):
    """The neighbour lists of an ELLPACK graph (the first degrees(i) slots of row i) as a CsrMatrix,
    for the graph routines above (they only look at ptr/node)."""
    deg = E.degrees.astype(np.int64)
    ptr = np.concatenate([[1], 1 + np.cumsum(deg)]).astype(I4)
    mask = np.arange(E.max_d)[None, :] < deg[:, None]
    node = np.ascontiguousarray(E.node.reshape(E.n, E.max_d)[mask], I4)
    return CsrMatrix(E.n, E.m, ptr, node, np.ones(len(node), F8))


def ell_permuted(E, p_left=None, p_right=None):
    """ellpack left/right permute (ellpack_matrices.f90:601-632): returns (node, val, degrees) as (n, max_d) arrays."""
    node = np.ascontiguousarray(E.node.reshape(E.n, E.max_d), I4).copy()
    val = np.ascontiguousarray(E.val.reshape(E.n, E.max_d), F8).copy()
    deg = np.ascontiguousarray(E.degrees, I4).copy()
    if p_left is not None:
        p_left = np.ascontiguousarray(p_left, I4)
        n2, v2, d2 = np.zeros_like(node), np.zeros_like(val), np.zeros_like(deg)
        lib().orc_ell_left_permute(C.c_int32(E.n), C.c_int32(E.max_d), _p(node), _p(val), _p(deg), _p(p_left), _p(n2), _p(v2), _p(d2))
        node, val, deg = n2, v2, d2
    if p_right is not None:
        p_right = np.ascontiguousarray(p_right, I4)
        lib().orc_ell_right_permute(C.c_int32(E.n), C.c_int32(E.max_d), _p(node), _p(p_right))
    return node, val, deg


def set_dot_mode(mode):
    """0: left-to-right dot products; 1: four interleaved partial sums (a vectorising
    compiler's dot_product).  Both are valid restatements of the Fortran intrinsic."""
    lib().orc_set_dot_mode(C.c_int(mode))


def time_csr_matvec(A, x, reps):
    y = np.zeros(A.n, F8)
    x = np.ascontiguousarray(x, F8)
    return lib().orc_time_csr_matvec(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(A.val),
                                     _p(x), _p(y), C.c_int32(reps))


def time_csr_matvec_omp(A, x, reps):
    """All host cores (OpenMP over rows; bit-identical rows): returns (seconds per matvec, threads)."""
    y = np.zeros(A.n, F8)
    x = np.ascontiguousarray(x, F8)
    nt = C.c_int32(0)
    lib().orc_time_csr_matvec_omp.restype = C.c_double
    sec = lib().orc_time_csr_matvec_omp(C.c_int32(A.n), _p(A.ptr), _p(A.node), _p(A.val), _p(x), _p(y),
                                        C.c_int32(reps), C.byref(nt))
    return sec, int(nt.value), y
