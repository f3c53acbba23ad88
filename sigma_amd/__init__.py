"""sigma_amd -- MI355X-native SpMV + Krylov path for SiGMA (danshapero/sigma).

This package is the thin host side above the C ABI of ``libsigma_hip.so``
(include/sigma_hip.h): it mirrors the reference's operator / solver interface for the
hot path -- same names, argument meaning and error behaviour --

    A = csr_matrix(n, m, ptr, node, val)       cs_matrices.f90:32-151  (1-based arrays)
    A = ellpack_matrix(n, m, node, val)        ellpack_matrices.f90:28-105
    A.matvec(x, y); A.matvec_add(x, y)         linear_operator_interface.f90:185-194
    solver = cg(tolerance)                     cg_solvers.f90:36-47
    solver = bicgstab(tolerance)               bicgstab_solvers.f90:37-48
    pc = jacobi(); pc = ldu(incomplete, level) jacobi_solvers.f90:23-31, ldu_solvers.f90:73-86
    solver.setup(A); pc.setup(A)
    solver.solve(A, x, b[, pc])                generic solve: linear_solve / linear_solve_pc
    solver.iterations, solver.tolerance, solver.nn, solver.initialized
    solver.destroy(); A.destroy()

All arithmetic happens in hand-written HIP kernels for gfx950; there is NO CPU fallback:
importing works anywhere (so the build and symbol checks run on a CPU box), but any
compute call without a GPU raises SigmaError.  Vectors may be numpy arrays (copied over
PCIe inside the call) or CUDA/HIP torch tensors (used in place in HBM).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsigma_hip.so")

SGM_HOST, SGM_DEVICE = 0, 1
FMT_CSR, FMT_ELL = 1, 2

_lib = None


class SigmaError(RuntimeError):
    """Nonzero status from libsigma_hip.so (the reference would print and exit(1))."""

    def __init__(self, code, msg):
        super().__init__(f"[sigma_hip status {code}] {msg}")
        self.code = code


def build(force=False):
    """Compile libsigma_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    import subprocess
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def _share_torch_hip_runtime():
    """One HIP/HSA runtime per process.  A torch wheel carries its own libamdhip64.so.7 and a
    libhsa-runtime64.so its other libraries ask for by a name the system copy does not answer to,
    so `import torch` AFTER this library has pulled in /opt/rocm's runtime starts a second HSA
    runtime and torch then finds no GPU.  When torch is installed but not yet imported, load its
    libamdhip64 first; the dynamic linker then hands the same copy to libsigma_hip.so (same
    soname), exactly as when torch was imported first.  SGM_HIP_RUNTIME=system skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("SGM_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded C-ABI library.  Fails loudly when the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SigmaError(-1, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        _share_torch_hip_runtime()
        L = C.CDLL(LIB_PATH)
        L.sgm_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def _ck(rc):
    if rc != 0:
        raise SigmaError(rc, lib().sgm_last_error().decode(errors="replace"))


def init(device=0):
    _ck(lib().sgm_init(C.c_int(device)))


_adopted_stream = None      # hipStream_t the library launches on when it is not its own


def set_stream(stream_ptr):
    global _adopted_stream
    _ck(lib().sgm_set_stream(C.c_void_p(stream_ptr)))
    _adopted_stream = stream_ptr or None


def set_async(on):
    _ck(lib().sgm_set_async(C.c_int(1 if on else 0)))


def synchronize():
    _ck(lib().sgm_synchronize())


def set_option(name, value):
    """sgm_set_option: the DEFAULT that matrices / solvers / preconditioners created later start with (options are per
    handle: include/sigma_hip.h lists them; A.set_option / solver.set_option / pc.set_option change one handle)."""
    _ck(lib().sgm_set_option(name.encode(), C.c_int(int(value))))


def use_torch_stream():
    """Launch on torch's current HIP stream so torch.cuda.Event brackets our kernels."""
    import torch
    set_stream(torch.cuda.current_stream().cuda_stream)


# ------------------------------------------------------------------------------------ #
def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "is_cuda")


def _arg(a, dtype, writable=False):
    """(pointer, where, keepalive) for a numpy array or a device torch tensor."""
    if _is_torch(a):
        import torch
        want = {np.float64: torch.float64, np.int32: torch.int32, np.int64: torch.int64}[dtype]
        if a.dtype != want or not a.is_contiguous():
            raise TypeError(f"torch tensor must be contiguous {want}")
        if not a.is_cuda:
            return _arg(a.numpy(), dtype, writable)
        # the library launches on its own stream unless it adopted torch's (use_torch_stream): a
        # tensor torch is still producing on another stream must be finished before we read it
        cur = torch.cuda.current_stream(a.device)
        if cur.cuda_stream != _adopted_stream:
            cur.synchronize()
        return C.c_void_p(a.data_ptr()), SGM_DEVICE, a
    arr = np.asarray(a)
    if arr.dtype != dtype or not arr.flags.c_contiguous or (writable and not arr.flags.writeable):
        if writable:
            raise TypeError(f"output array must be a contiguous writable {np.dtype(dtype)} numpy array")
        arr = np.ascontiguousarray(arr, dtype)
    return C.c_void_p(arr.ctypes.data), SGM_HOST, arr


def _len(a):
    return int(a.numel()) if _is_torch(a) else int(np.asarray(a).size)


def _need(a, n, what):
    """The C ABI takes raw pointers: a short vector would be read or written out of bounds."""
    if _len(a) < n:
        raise SigmaError(2, f"{what}: vector has {_len(a)} entries, the operation needs {n}")


def _same_where(*ws):
    if len(set(ws)) != 1:
        raise TypeError("all vectors of one call must live in the same place (host or device)")
    return ws[0]


class _Matrix:
    """linear_operator face (nrow, ncol, matvec, matvec_add, destroy)."""

    def __init__(self):
        self._h = C.c_void_p()
        self.nrow = self.ncol = 0
        self.solver = None       # linear_operator%solver / %pc (linear_operator_interface.f90:29)
        self.pc = None

    # -- linear_operator_interface.f90:185-194 ------------------------------------------
    def _lens(self):
        """(entries matvec reads from x, entries it writes to y): ncol / nrow, or owned+halo /
        owned rows for a matrix distributed over processes."""
        return self.x_len, getattr(self, "n_local", self.nrow)

    def matvec(self, x, y):
        nx, ny = self._lens()
        _need(x, nx, "matvec x"); _need(y, ny, "matvec y")
        px, wx, _k1 = _arg(x, np.float64)
        py, wy, _k2 = _arg(y, np.float64, writable=True)
        _ck(lib().sgm_mat_matvec(self._h, px, py, C.c_int(_same_where(wx, wy))))
        return y

    def matvec_add(self, x, y):
        nx, ny = self._lens()
        _need(x, nx, "matvec_add x"); _need(y, ny, "matvec_add y")
        px, wx, _k1 = _arg(x, np.float64)
        py, wy, _k2 = _arg(y, np.float64, writable=True)
        _ck(lib().sgm_mat_matvec_add(self._h, px, py, C.c_int(_same_where(wx, wy))))
        return y

    # -- linear_operator_interface.f90:199-208 ------------------------------------------
    def matvec_t(self, x, y):
        _need(x, getattr(self, "n_local", self.nrow), "matvec_t x"); _need(y, getattr(self, "nc_local", self.ncol), "matvec_t y")
        px, wx, _k1 = _arg(x, np.float64)
        py, wy, _k2 = _arg(y, np.float64, writable=True)
        _ck(lib().sgm_mat_matvec_t(self._h, px, py, C.c_int(_same_where(wx, wy))))
        return y

    def matvec_t_add(self, x, y):
        _need(x, getattr(self, "n_local", self.nrow), "matvec_t_add x"); _need(y, getattr(self, "nc_local", self.ncol), "matvec_t_add y")
        px, wx, _k1 = _arg(x, np.float64)
        py, wy, _k2 = _arg(y, np.float64, writable=True)
        _ck(lib().sgm_mat_matvec_t_add(self._h, px, py, C.c_int(_same_where(wx, wy))))
        return y

    def halo_nbrs(self, part=0):
        """Exchange plan of local part `part` (parity checks): list of dicts with peer, send_count,
        recv_count, recv_offset and the send list (0-based local indices)."""
        n = C.c_int32(0)
        _ck(lib().sgm_mat_halo_nbr(self._h, C.c_int32(part), C.c_int32(-1), C.byref(n), None, None, None, None, None, C.c_int32(0)))
        out = []
        for k in range(n.value):
            pe, sc, rc, ro = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _ck(lib().sgm_mat_halo_nbr(self._h, C.c_int32(part), C.c_int32(k), None, C.byref(pe), C.byref(sc), C.byref(rc),
                                       C.byref(ro), None, C.c_int32(0)))
            idx = np.zeros(sc.value, np.int32)
            if sc.value:
                _ck(lib().sgm_mat_halo_nbr(self._h, C.c_int32(part), C.c_int32(k), None, None, None, None, None,
                                           C.c_void_p(idx.ctypes.data), C.c_int32(sc.value)))
            out.append({"peer": pe.value, "send_count": sc.value, "recv_count": rc.value, "recv_offset": ro.value,
                        "send_idx": idx})
        return out

    @property
    def x_len(self):
        n = C.c_int64(0)
        _ck(lib().sgm_mat_info(self._h, None, None, None, None, C.byref(n)))
        return n.value

    def footprint(self):
        """(resident_bytes, matvec_bytes): what the handle keeps in HBM, and what one y = A x moves by
        construction with the kernel the current options select (sgm_mat_footprint)."""
        if hasattr(self, "_build"):
            self._build()
        r, m = C.c_int64(0), C.c_int64(0)
        _ck(lib().sgm_mat_footprint(self._h, C.byref(r), C.byref(m)))
        return r.value, m.value

    # -- src/graph/permutations.f90 + cs_matrices.f90:471-490 -----------------------------
    def bfs_order(self):
        """breadth_first_search(p, g) on the matrix graph: p(i) = visiting number (1-based), -1 unreached."""
        p = np.zeros(self.nrow, np.int32)
        _ck(lib().sgm_graph_bfs_order(self._h, p.ctypes.data_as(C.c_void_p)))
        return p

    def greedy_coloring(self):
        """greedy_coloring(colors, g): returns (colors, num_colors)."""
        c = np.zeros(self.nrow, np.int32)
        nc = C.c_int32(0)
        _ck(lib().sgm_graph_greedy_coloring(self._h, c.ctypes.data_as(C.c_void_p), C.byref(nc)))
        return c, nc.value

    def greedy_color_ordering(self):
        """greedy_color_ordering(p, ptrs, num_colors, g): returns (p, ptrs[:num_colors+1], num_colors)."""
        p = np.zeros(self.nrow, np.int32)
        ptrs = np.zeros(self.nrow + 2, np.int32)
        nc = C.c_int32(0)
        _ck(lib().sgm_graph_greedy_color_order(self._h, p.ctypes.data_as(C.c_void_p), ptrs.ctypes.data_as(C.c_void_p),
                                               C.c_int32(len(ptrs)), C.byref(nc)))
        return p, ptrs[:nc.value + 1].copy(), nc.value

    def left_permute(self, p):
        pp, w, _k = _arg(p, np.int32)
        _ck(lib().sgm_mat_left_permute(self._h, pp, C.c_int(w)))

    def right_permute(self, p):
        pp, w, _k = _arg(p, np.int32)
        _ck(lib().sgm_mat_right_permute(self._h, pp, C.c_int(w)))

    def set_option(self, name, value):
        """sgm_mat_set_option: this matrix's own kernel-selection option (same bits whichever)."""
        if hasattr(self, "_build"):
            self._build()
        _ck(lib().sgm_mat_set_option(self._h, name.encode(), C.c_int(int(value))))
        return self

    @property
    def kernel(self):
        """Name of the SpMV kernel variant this matrix runs with under its options."""
        buf = C.create_string_buffer(64)
        _ck(lib().sgm_mat_kernel(self._h, buf, C.c_int(64)))
        return buf.value.decode()

    # -- linear_operator_interface.f90:213-280: A%solve facade ---------------------------
    def set_solver(self, solver):
        self.solver = solver
        solver.setup(self)

    def set_preconditioner(self, pc):
        self.pc = pc
        pc.setup(self)

    def solve(self, x, b):
        if self.solver is None:
            raise SigmaError(1, "A%solve: no solver set (linear_operator_interface.f90:213-233)")
        return self.solver.solve(self, x, b, self.pc)

    def get(self, name, dtype):
        """Read a leaf matrix back in the reference's layout (sgm_mat_get)."""
        need = C.c_size_t(0)
        _ck(lib().sgm_mat_get(self._h, name.encode(), None, C.c_size_t(0), C.byref(need)))
        out = np.zeros(need.value // np.dtype(dtype).itemsize, dtype)
        _ck(lib().sgm_mat_get(self._h, name.encode(), C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes), None))
        return out

    @classmethod
    def from_edges(cls, nrow, ncol, ei, ej, ev):
        """Assemble on the device from an edge list in insertion order (1-based), like
        g%add_edge ... convert_graph_type ... A%set_value (sgm_csr/ell_from_edges)."""
        self = cls.__new__(cls)
        _Matrix.__init__(self)
        pi, w1, _k1 = _arg(ei, np.int32)
        pj, w2, _k2 = _arg(ej, np.int32)
        pv, w3, _k3 = _arg(ev, np.float64)
        ne = ei.numel() if _is_torch(ei) else len(ei)
        self.nrow, self.ncol = int(nrow), int(ncol)
        fn = lib().sgm_csr_from_edges if cls is csr_matrix else lib().sgm_ell_from_edges
        _ck(fn(C.byref(self._h), C.c_int32(nrow), C.c_int32(ncol), C.c_int64(ne), pi, pj, pv,
               C.c_int(_same_where(w1, w2, w3))))
        return self

    # -- sparse_matrix_to_file (sparse_matrix_interfaces.f90:601-653): text dump -----------
    def to_file(self, filename, trans=False):
        """`nrow ncol nnz`, then one `i j value` line per stored entry in stored order (rows in
        order, a row's entries as they lie in memory); `trans` swaps i/j and the dimensions like the
        reference.  Values are written with 17 significant digits (they read back bit for bit)."""
        if not isinstance(self, csr_matrix):
            raise SigmaError(7, "to_file: CSR matrices only")
        ptr, node, val = self.get("ptr", np.int32), self.get("node", np.int32), self.get("val", np.float64)
        rows = np.repeat(np.arange(1, self.nrow + 1, dtype=np.int64), np.diff(ptr))
        a, b = (node, rows) if trans else (rows, node)
        dims = (self.ncol, self.nrow) if trans else (self.nrow, self.ncol)
        with open(filename, "w") as f:
            f.write(f" {dims[0]} {dims[1]} {len(val)}\n")
            f.writelines(f" {int(i)} {int(j)} {float(v)!r}\n" for i, j, v in zip(a, b, val))

    @classmethod
    def from_file(cls, filename):
        """Read a file written by to_file (or by the reference's A%to_file): the entries are
        inserted in file order, so a CSR matrix written and read back has identical arrays."""
        with open(filename) as f:
            nrow, ncol, nnz = (int(t) for t in f.readline().split())
            data = np.loadtxt(f, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, 3))
        if data.shape[0] != nnz:
            raise SigmaError(2, f"from_file: header says {nnz} entries, file holds {data.shape[0]}")
        return cls.from_edges(nrow, ncol, data[:, 0].astype(np.int32), data[:, 1].astype(np.int32),
                              np.ascontiguousarray(data[:, 2]))

    def destroy(self):
        if self._h:
            _ck(lib().sgm_mat_destroy(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class csr_matrix(_Matrix):
    """csr_matrix (cs_matrices.f90:112-151): g%ptr(n+1), g%node(nnz), val(nnz), 1-based."""

    def __init__(self, nrow, ncol, ptr, node, val):
        super().__init__()
        pp, w1, _k1 = _arg(ptr, np.int32)
        pn, w2, _k2 = _arg(node, np.int32)
        pv, w3, _k3 = _arg(val, np.float64)
        self.nrow, self.ncol = int(nrow), int(ncol)
        self.nnz = int(len(val))
        _ck(lib().sgm_csr_create(C.byref(self._h), C.c_int32(nrow), C.c_int32(ncol), C.c_int64(self.nnz),
                                 pp, pn, pv, C.c_int(_same_where(w1, w2, w3))))

    def set_values(self, val):
        pv, w, _k = _arg(val, np.float64)
        _ck(lib().sgm_csr_set_values(self._h, pv, C.c_int(w)))


class partitioned_csr_matrix(_Matrix):
    """The multi-GPU row partition with all P row blocks on one GPU (test vehicle)."""

    def __init__(self, nrow, ncol, ptr, node, val, row_starts):
        super().__init__()
        ptr = np.ascontiguousarray(ptr, np.int32)
        node = np.ascontiguousarray(node, np.int32)
        val = np.ascontiguousarray(val, np.float64)
        rs = np.ascontiguousarray(row_starts, np.int64)
        self.nrow, self.ncol, self.nnz = int(nrow), int(ncol), int(len(val))
        _ck(lib().sgm_csr_create_partitioned(C.byref(self._h), C.c_int32(len(rs) - 1), C.c_void_p(rs.ctypes.data),
                                             C.c_int32(nrow), C.c_int32(ncol), C.c_int64(self.nnz),
                                             C.c_void_p(ptr.ctypes.data), C.c_void_p(node.ctypes.data),
                                             C.c_void_p(val.ctypes.data)))


    @classmethod
    def from_parts(cls, row_starts, parts):
        """sgm_csr_create_partitioned_parts: the same in-process partition handed over part by part -- parts[k] =
        (ptr_local, node_global, val) of rows row_starts[k] .. row_starts[k+1]-1, numpy arrays or device tensors (all
        of one kind) -- for matrices too large to assemble whole on the host."""
        self = cls.__new__(cls)
        _Matrix.__init__(self)
        rs = np.ascontiguousarray(row_starts, np.int64)
        P = len(rs) - 1
        assert len(parts) == P
        keep, wheres = [], []
        pp, pn, pv = (C.c_void_p * P)(), (C.c_void_p * P)(), (C.c_void_p * P)()
        nnz = (C.c_int64 * P)()
        for k, (a, b, c) in enumerate(parts):
            p1, w1, k1 = _arg(a, np.int32)
            p2, w2, k2 = _arg(b, np.int32)
            p3, w3, k3 = _arg(c, np.float64)
            keep += [k1, k2, k3]
            wheres += [w1, w2, w3]
            pp[k], pn[k], pv[k] = p1, p2, p3
            nnz[k] = int(len(c))
        self.nrow = self.ncol = int(rs[-1])
        self.nnz = int(sum(nnz))
        _ck(lib().sgm_csr_create_partitioned_parts(C.byref(self._h), C.c_int32(P), C.c_void_p(rs.ctypes.data), nnz, pp, pn, pv,
                                                   C.c_int(_same_where(*wheres))))
        return self


class dist_csr_matrix(_Matrix):
    """This rank's row block of a matrix partitioned over processes (RCCL).  col_starts: the partition of x when it
    is not the rows' (sgm_csr_create_dist_rect: an off-diagonal block of a composite)."""

    def __init__(self, comm, row_starts, ptr_local, node_global, val, col_starts=None):
        super().__init__()
        rs = np.ascontiguousarray(row_starts, np.int64)
        cs = rs if col_starts is None else np.ascontiguousarray(col_starts, np.int64)
        pp, w1, _k1 = _arg(ptr_local, np.int32)
        pn, w2, _k2 = _arg(node_global, np.int32)
        pv, w3, _k3 = _arg(val, np.float64)
        self.nrow, self.ncol = int(rs[-1]), int(cs[-1])
        self.n_local = int(rs[comm.rank + 1] - rs[comm.rank])
        self.nc_local = int(cs[comm.rank + 1] - cs[comm.rank])
        self.nnz = int(len(val))
        _ck(lib().sgm_csr_create_dist_rect(C.byref(self._h), comm._h, C.c_void_p(rs.ctypes.data), C.c_void_p(cs.ctypes.data),
                                           C.c_int64(self.nnz), pp, pn, pv, C.c_int(_same_where(w1, w2, w3))))


class dist_ellpack_matrix(_Matrix):
    """This rank's rows of an ELLPACK matrix partitioned over processes (sgm_ell_create_dist): node / val
    as C arrays of shape (n_local, max_d) = the reference's (max_d, n_local), GLOBAL 1-based columns."""

    def __init__(self, comm, row_starts, node_global, val):
        super().__init__()
        rs = np.ascontiguousarray(row_starts, np.int64)
        max_d = int(node_global.shape[1])
        pn, w1, _k1 = _arg(node_global, np.int32)
        pv, w2, _k2 = _arg(val, np.float64)
        self.nrow = self.ncol = int(rs[-1])
        self.n_local = self.nc_local = int(rs[comm.rank + 1] - rs[comm.rank])
        self.max_d = max_d
        _ck(lib().sgm_ell_create_dist(C.byref(self._h), comm._h, C.c_void_p(rs.ctypes.data), C.c_int32(max_d), pn, pv,
                                      C.c_int(_same_where(w1, w2))))


class sparse_matrix(_Matrix):
    """The composite "matrix of matrices" (sparse_matrix_composites.f90:41-162): row_ptr /
    col_ptr are the 1-based block offsets, set_submatrix(it, jt, B) places a leaf (1-based
    block indices); build() freezes the layout.  matvec_add loops over the blocks like
    composite_matvec_add (:1076-1099)."""

    def __init__(self, row_ptr, col_ptr):
        super().__init__()
        self.row_ptr = np.ascontiguousarray(row_ptr, np.int32)
        self.col_ptr = np.ascontiguousarray(col_ptr, np.int32)
        self.num_row_mats, self.num_col_mats = len(self.row_ptr) - 1, len(self.col_ptr) - 1
        self.nrow, self.ncol = int(self.row_ptr[-1] - 1), int(self.col_ptr[-1] - 1)
        self.sub_mats = [[None] * self.num_col_mats for _ in range(self.num_row_mats)]

    def set_submatrix(self, it, jt, B):
        self.sub_mats[it - 1][jt - 1] = B
        if self._h:
            self.destroy()

    def _build(self):
        if self._h:
            return
        arr = (C.c_void_p * (self.num_row_mats * self.num_col_mats))()
        for i, row in enumerate(self.sub_mats):
            for j, B in enumerate(row):
                arr[i * self.num_col_mats + j] = B._h if B is not None else None
        _ck(lib().sgm_composite_create(C.byref(self._h), C.c_int32(self.num_row_mats), C.c_int32(self.num_col_mats),
                                       C.c_void_p(self.row_ptr.ctypes.data), C.c_void_p(self.col_ptr.ctypes.data), arr))
        leaves = [B for row in self.sub_mats for B in row if B is not None]
        if leaves and hasattr(leaves[0], "n_local"):
            # over distributed leaves the vectors are this rank's slices of the block vectors, concatenated
            self.n_local = sum(next(B.n_local for B in row if B is not None) for row in self.sub_mats)
            self.nc_local = sum(next(self.sub_mats[i][j].nc_local for i in range(self.num_row_mats) if self.sub_mats[i][j] is not None)
                                for j in range(self.num_col_mats))

    def matvec(self, x, y):
        self._build()
        return super().matvec(x, y)

    def matvec_add(self, x, y):
        self._build()
        return super().matvec_add(x, y)

    def matvec_t(self, x, y):
        self._build()
        return super().matvec_t(x, y)

    def matvec_t_add(self, x, y):
        self._build()
        return super().matvec_t_add(x, y)


class ellpack_matrix(_Matrix):
    """ellpack_matrix (ellpack_matrices.f90:28-105): node(max_d,n), val(max_d,n) in Fortran
    order, i.e. a C array of shape (n, max_d); padding = last neighbour / 0.0."""

    def __init__(self, nrow, ncol, node, val):
        super().__init__()
        if _is_torch(node):
            max_d = int(node.shape[1])
        else:
            node = np.ascontiguousarray(node, np.int32)
            val = np.ascontiguousarray(val, np.float64)
            max_d = int(node.shape[1]) if node.ndim == 2 else int(node.size // max(nrow, 1))
        pn, w1, _k1 = _arg(node, np.int32)
        pv, w2, _k2 = _arg(val, np.float64)
        self.nrow, self.ncol, self.max_d = int(nrow), int(ncol), max_d
        _ck(lib().sgm_ell_create(C.byref(self._h), C.c_int32(nrow), C.c_int32(ncol), C.c_int32(max_d), pn, pv,
                                 C.c_int(_same_where(w1, w2))))

    def set_values(self, val):
        pv, w, _k = _arg(val, np.float64)
        _ck(lib().sgm_ell_set_values(self._h, pv, C.c_int(w)))


# ------------------------------------------------------------------------------------ #
class _Preconditioner:
    """linear_solver used as a preconditioner (setup / solve / destroy)."""
    _kind = 0

    def __init__(self):
        self._h = C.c_void_p()
        self.nn = 0
        self.initialized = False
        self._opts = {}

    def setup(self, A):
        if hasattr(A, "_build"):
            A._build()
        if not self._h:
            # the factory (jacobi() / ldu()) has seen no matrix: options set before the first setup decide how it builds
            _ck(lib().sgm_pc_create(C.byref(self._h), C.c_int32(self._kind)))
            for k, v in self._opts.items():
                _ck(lib().sgm_pc_set_option(self._h, k.encode(), C.c_int(int(v))))
        _ck(lib().sgm_pc_setup(self._h, A._h))
        self.nn = getattr(A, "n_local", A.nrow)
        self.initialized = True

    def info(self, part=0):
        """sgm_pc_info: {"levels": (L, U), "path": 0..4, "colours": n, "est_us": per apply, "name": e.g. "strip pipeline,
        6323 levels"} -- whether an apply will be a dependency chain (natural-order ILDU of a grid) or a few bandwidth-bound
        sweeps (ldu(reorder="colour"))."""
        o = (C.c_int32 * 4)()
        us = C.c_double()
        nm = C.create_string_buffer(160)
        _ck(lib().sgm_pc_info(self._h, C.c_int32(part), o, C.byref(us), nm, C.c_int(160)))
        return {"levels": (int(o[0]), int(o[1])), "path": int(o[2]), "colours": int(o[3]), "est_us": float(us.value), "name": nm.value.decode()}

    def set_option(self, name, value):
        """sgm_pc_set_option: this preconditioner's own "ildu_strips" / "ildu_rows" / "pipeline_spin_limit"."""
        self._opts[name] = int(value)
        if self._h:
            _ck(lib().sgm_pc_set_option(self._h, name.encode(), C.c_int(int(value))))
        return self

    def solve(self, A, x, b):
        """pc%solve(A, x, b): x = M^-1 b."""
        _need(b, self.nn, "pc%solve b"); _need(x, self.nn, "pc%solve x")
        pb, wb, _k1 = _arg(b, np.float64)
        px, wx, _k2 = _arg(x, np.float64, writable=True)
        _ck(lib().sgm_pc_apply(self._h, pb, px, C.c_int(_same_where(wb, wx))))
        return x

    def get(self, name, dtype):
        need = C.c_size_t(0)
        _ck(lib().sgm_pc_get(self._h, name.encode(), None, C.c_size_t(0), C.byref(need)))
        out = np.zeros(need.value // np.dtype(dtype).itemsize, dtype)
        _ck(lib().sgm_pc_get(self._h, name.encode(), C.c_void_p(out.ctypes.data), C.c_size_t(out.nbytes), None))
        return out

    def destroy(self):
        if self._h:
            _ck(lib().sgm_pc_destroy(self._h))
            self._h = C.c_void_p()
        self.initialized = False

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class jacobi_solver(_Preconditioner):
    _kind = 1       # SGM_PC_JACOBI

    @property
    def idiag(self):
        return self.get("idiag", np.float64)


class sparse_ldu_solver(_Preconditioner):
    _kind = 2       # SGM_PC_ILDU0


def jacobi():
    """jacobi() factory (jacobi_solvers.f90:23-31)."""
    return jacobi_solver()


def ldu(incomplete=True, level=0, reorder=None):
    """ldu(incomplete, level) factory (ldu_solvers.f90:73-86).  Like the reference
    (ldu_set_params :143-151) the arguments are accepted and forced to ILDU(0).
    reorder="colour" (an extension, off by default): ILDU(0) of the colour-ordered matrix P A P^T -- P = the reference's
    greedy_color_ordering of A's graph -- applied as z = P^T M^-1 P r; A, b and x stay as the caller holds them."""
    pc = sparse_ldu_solver()
    if reorder in ("colour", "color"):
        pc.set_option("ildu_reorder", 1)
    elif reorder not in (None, "natural"):
        raise SigmaError(1, f"ldu: reorder is None or 'colour', not {reorder!r}")
    return pc


class _Solver:
    """linear_solver (linear_operator_interface.f90:61-73)."""

    def __init__(self, tolerance):
        self._h = C.c_void_p()
        self.tolerance = 1.0e-16 if tolerance is None else float(tolerance)   # cg_solvers.f90:106
        self.nn = 0
        self.initialized = False
        self._max_iter = 0
        self._hist = 0
        self._opts = {}

    def _create(self):
        raise NotImplementedError

    def set_option(self, name, value):
        """sgm_solver_set_option: this solver's own "cg_small" / "bicgstab_small" / "krylov_graph" / "dot_order" /
        "gmres_cgs2" (read at the next solve)."""
        self._opts[name] = int(value)
        if self._h:
            _ck(lib().sgm_solver_set_option(self._h, name.encode(), C.c_int(int(value))))
        return self

    def setup(self, A):
        if hasattr(A, "_build"):
            A._build()
        if not self._h:
            self._create()
            if self._max_iter:
                _ck(lib().sgm_solver_set_max_iter(self._h, C.c_int64(self._max_iter)))
            if self._hist:
                _ck(lib().sgm_solver_set_history(self._h, C.c_int64(self._hist)))
            for k, v in self._opts.items():
                _ck(lib().sgm_solver_set_option(self._h, k.encode(), C.c_int(int(v))))
        _ck(lib().sgm_solver_setup(self._h, A._h))
        self.nn = A.nrow
        self.initialized = True

    def set_params(self, tolerance=None):
        """solver%set_params(tolerance) (cg_solvers.f90:95-111): without an argument the tolerance goes back to 1e-16."""
        self.tolerance = 1.0e-16 if tolerance is None else float(tolerance)
        return self

    def set_max_iter(self, max_iter):
        """Extension: the reference has no iteration cap.  <= 0 = unbounded."""
        self._max_iter = int(max_iter)
        if self._h:
            _ck(lib().sgm_solver_set_max_iter(self._h, C.c_int64(self._max_iter)))

    def set_history(self, capacity):
        self._hist = int(capacity)
        if self._h:
            _ck(lib().sgm_solver_set_history(self._h, C.c_int64(self._hist)))

    def solve(self, A, x, b, pc=None, check=True):
        """solver%solve(A, x, b[, pc]): x holds the initial guess on entry, the solution on
        exit.  With set_max_iter, hitting the cap raises unless check=False."""
        nloc = getattr(A, "n_local", A.nrow)
        _need(x, nloc, "solve x"); _need(b, nloc, "solve b")
        px, wx, _k1 = _arg(x, np.float64, writable=True)
        pb, wb, _k2 = _arg(b, np.float64)
        # solver.tolerance is a live public field, read at every solve like cg_solvers.f90:133
        _ck(lib().sgm_solver_set_tolerance(self._h, C.c_double(float(self.tolerance))))
        rc = lib().sgm_solver_solve(self._h, A._h, px, pb, pc._h if pc is not None else None,
                                    C.c_int(_same_where(wx, wb)))
        if rc == 5 and not check:
            return x
        _ck(rc)
        return x

    def _info(self):
        it, last = C.c_int64(0), C.c_int64(0)
        r, cv = C.c_double(0.0), C.c_int32(0)
        _ck(lib().sgm_solver_info(self._h, C.byref(it), C.byref(r), C.byref(cv), C.byref(last)))
        return it.value, r.value, cv.value, last.value

    @property
    def iterations(self):
        return self._info()[0] if self._h else 0

    @property
    def last_iterations(self):
        return self._info()[3]

    @property
    def res2(self):
        return self._info()[1]

    @property
    def converged(self):
        return bool(self._info()[2])

    @property
    def history(self):
        cnt = C.c_int64(0)
        _ck(lib().sgm_solver_get_history(self._h, None, C.c_int64(0), C.byref(cnt)))
        out = np.zeros(cnt.value, np.float64)
        if cnt.value:
            _ck(lib().sgm_solver_get_history(self._h, C.c_void_p(out.ctypes.data), C.c_int64(cnt.value), None))
        return out

    def destroy(self):
        if self._h:
            _ck(lib().sgm_solver_destroy(self._h))
            self._h = C.c_void_p()
        self.initialized = False

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class cg_solver(_Solver):
    def _create(self):
        _ck(lib().sgm_cg_create(C.byref(self._h), C.c_double(self.tolerance)))


class bicgstab_solver(_Solver):
    def _create(self):
        _ck(lib().sgm_bicgstab_create(C.byref(self._h), C.c_double(self.tolerance)))


class gmres_solver(_Solver):
    def __init__(self, tolerance, restart):
        super().__init__(tolerance)
        self.restart = int(restart)

    def _create(self):
        _ck(lib().sgm_gmres_create(C.byref(self._h), C.c_double(self.tolerance), C.c_int32(self.restart)))


def cg(tolerance=None):
    """cg(tolerance) factory (cg_solvers.f90:36-47); default tolerance 1e-16 (:106)."""
    return cg_solver(tolerance)


def bicgstab(tolerance=None):
    """bicgstab(tolerance) factory (bicgstab_solvers.f90:37-48)."""
    return bicgstab_solver(tolerance)


def gmres(tolerance=None, restart=30):
    """GMRES(restart).  Not in the reference; conventions follow cg."""
    return gmres_solver(tolerance, restart)


# ------------------------------------------------------------------------------------ #
class Comm:
    """RCCL communicator for the row partition (one process per GPU)."""

    def __init__(self, rank, nranks, unique_id, halo_unique_id=None):
        self._h = C.c_void_p()
        self.rank, self.nranks = int(rank), int(nranks)
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        _ck(lib().sgm_comm_init(C.byref(self._h), C.c_int(rank), C.c_int(nranks), buf))
        if halo_unique_id is not None:        # a second communicator for the halo send / recv pairs
            buf2 = (C.c_char * 128).from_buffer_copy(bytes(halo_unique_id))
            _ck(lib().sgm_comm_attach_halo_comm(self._h, buf2))

    @staticmethod
    def unique_id():
        buf = (C.c_char * 128)()
        _ck(lib().sgm_comm_unique_id(buf))
        return bytes(buf)

    def group_selftest(self):
        """sgm_comm_group_selftest: one group of send / recv (to itself) + all-reduce on this communicator's transport;
        returns (delivered, allreduced, microseconds) -- delivered must be 42 + rank, allreduced nranks."""
        out = (C.c_double * 3)()
        _ck(lib().sgm_comm_group_selftest(self._h, out))
        return float(out[0]), float(out[1]), float(out[2])

    @property
    def group_ok(self):
        """sgm_comm_group_ok: did this transport take the probe group of sgm_comm_init (pairs + all-reduce in one group)?"""
        return bool(lib().sgm_comm_group_ok(self._h))

    def destroy(self):
        if self._h:
            _ck(lib().sgm_comm_destroy(self._h))
            self._h = C.c_void_p()


DIST_PHASES = ("halo_post_to_done", "interior_rows", "halo_wait_exposed", "boundary_rows", "dot_reduce_kernels", "allreduce")


def dist_profile(on):
    """Switch the phase timers of the row-partitioned path on / off (and clear them)."""
    _ck(lib().sgm_dist_profile(C.c_int(1 if on else 0)))


def dist_profile_read():
    """{phase: {"ms": total, "count": spans}} since the last read (synchronises the library's streams)."""
    ms = (C.c_double * 6)()
    cnt = (C.c_int64 * 6)()
    _ck(lib().sgm_dist_profile_read(ms, cnt))
    return {nm: {"ms": float(ms[k]), "count": int(cnt[k])} for k, nm in enumerate(DIST_PHASES)}


def lanczos(A, nsteps, q1, want_Q=True):
    """lanczos(A, T, Q) (src/eigensolver.f90:27-90) on the device with the start vector q1:
    returns (T[3, nsteps], Q[n, nsteps] or None).  The eigenvalues of the tridiagonal
    (diag T[1], off-diagonal T[2][:-1]) are what eigensolve :160-208 gets from dstev.
    A may be row-partitioned: an in-process partition takes and returns global vectors, a rank of a matrix distributed
    over processes its owned slice of q1 and of Q (T is the same on every rank)."""
    if hasattr(A, "_build"):
        A._build()
    pq, w, _k = _arg(q1, np.float64)
    nloc = getattr(A, "n_local", A.nrow)          # a rank of a distributed matrix works on its owned slice
    _need(q1, nloc, "lanczos q1")
    T = np.zeros((nsteps, 3), np.float64)
    Q = np.zeros((nsteps, nloc), np.float64) if want_Q else None
    if w != SGM_HOST:
        raise TypeError("lanczos: pass the start vector as a numpy array")
    _ck(lib().sgm_lanczos(A._h, C.c_int32(nsteps), pq, C.c_void_p(T.ctypes.data),
                          C.c_void_p(Q.ctypes.data) if want_Q else None, C.c_int(SGM_HOST)))
    return T.T.copy(), (Q.T.copy() if want_Q else None)


def generalized_lanczos(A, B, nsteps, q1, want_Q=True):
    """generalized_lanczos(A, B, T, Q) (src/eigensolver.f90:95-155) on the device: Lanczos for
    A x = lambda B x with the start vector q1.  Like the reference it uses the solver (and
    preconditioner) set on B with B.set_solver / B.set_preconditioner for the per-step solve
    `B%solve(w, v)`.  Returns (T[3, nsteps], Q[n, nsteps] or None)."""
    for M in (A, B):
        if hasattr(M, "_build"):
            M._build()
    if B.solver is None:
        raise SigmaError(1, "generalized_lanczos: B has no solver set (eigensolver.f90:101-103 assumes B%set_solver)")
    pq, w, _k = _arg(q1, np.float64)
    if w != SGM_HOST:
        raise TypeError("generalized_lanczos: pass the start vector as a numpy array")
    nloc = getattr(A, "n_local", A.nrow)
    _need(q1, nloc, "generalized_lanczos q1")
    T = np.zeros((nsteps, 3), np.float64)
    Q = np.zeros((nsteps, nloc), np.float64) if want_Q else None
    _ck(lib().sgm_solver_set_tolerance(B.solver._h, C.c_double(float(B.solver.tolerance))))
    _ck(lib().sgm_generalized_lanczos(A._h, B._h, B.solver._h, B.pc._h if B.pc is not None else None, C.c_int32(nsteps), pq,
                                      C.c_void_p(T.ctypes.data), C.c_void_p(Q.ctypes.data) if want_Q else None,
                                      C.c_int(SGM_HOST)))
    return T.T.copy(), (Q.T.copy() if want_Q else None)


def _ritz(T, Q, normalise_sign):
    """The LAPACK tail of eigensolve (src/eigensolver.f90:174-186): dstev('V') on the tridiagonal, V = Q Z.
    Host side (scipy's LAPACK), like the reference's."""
    from scipy.linalg import eigh_tridiagonal
    lam, Z = eigh_tridiagonal(T[1], T[2][:-1])
    V = Q @ Z
    if normalise_sign:                                   # V(:,i) = V(1,i) / |V(1,i)| * V(:,i)   (:182-184)
        V = V * (V[0] / np.abs(V[0]))
    return lam, V


def eigensolve(A, n, q1):
    """eigensolve(A, lambda, V) (src/eigensolver.f90:160-188): n Lanczos steps on the device, the
    tridiagonal eigenproblem on the host.  Returns (lambda[n] ascending, V[nrow, n])."""
    T, Q = lanczos(A, n, q1)
    return _ritz(T, Q, True)


def generalized_eigensolve(A, B, n, q1):
    """generalized_eigensolve(A, B, lambda, V) (src/eigensolver.f90:193-208); B needs a solver set."""
    T, Q = generalized_lanczos(A, B, n, q1)
    return _ritz(T, Q, False)


def halo_plan_host(n_own, col_begin, node_global):
    """Host-only index work of the row partition (no GPU needed): returns
    (node_local 1-based, halo_cols sorted unique global 1-based)."""
    node = np.ascontiguousarray(node_global, np.int32)
    out = np.zeros(len(node), np.int32)
    halo = np.zeros(max(len(node), 1), np.int32)
    nh = C.c_int32(0)
    _ck(lib().sgm_halo_plan_host(C.c_int32(n_own), C.c_int64(col_begin), C.c_int64(len(node)),
                                 C.c_void_p(node.ctypes.data), C.c_void_p(out.ctypes.data),
                                 C.c_void_p(halo.ctypes.data), C.byref(nh)))
    return out, halo[:nh.value].copy()


def dist_plan_host(rank, nranks, row_starts, halo_cols):
    """sgm_dist_plan_host (no GPU needed): (want[nranks], want_off[nranks+1], req[len(halo)])."""
    rs = np.ascontiguousarray(row_starts, np.int64)
    halo = np.ascontiguousarray(halo_cols, np.int32)
    want, off, req = np.zeros(nranks, np.int32), np.zeros(nranks + 1, np.int32), np.zeros(max(len(halo), 1), np.int32)
    _ck(lib().sgm_dist_plan_host(C.c_int32(rank), C.c_int32(nranks), C.c_void_p(rs.ctypes.data), C.c_int32(len(halo)),
                                 C.c_void_p(halo.ctypes.data), C.c_void_p(want.ctypes.data), C.c_void_p(off.ctypes.data),
                                 C.c_void_p(req.ctypes.data)))
    return want, off, req[:len(halo)].copy()


def dist_neighbors_host(rank, nranks, want_all):
    """sgm_dist_neighbors_host: list of (peer, send_count, recv_count, recv_offset) from the
    all-gathered want matrix (row q = rank q's want)."""
    wa = np.ascontiguousarray(want_all, np.int32).reshape(nranks, nranks)
    arr = [np.zeros(max(nranks, 1), np.int32) for _ in range(4)]
    n = C.c_int32(0)
    _ck(lib().sgm_dist_neighbors_host(C.c_int32(rank), C.c_int32(nranks), C.c_void_p(wa.ctypes.data),
                                      *[C.c_void_p(a.ctypes.data) for a in arr], C.byref(n)))
    return [tuple(int(a[i]) for a in arr) for i in range(n.value)]


def partition_links_host(row_starts, ptr, node):
    """sgm_partition_links_host: the (sender, receiver, recv_offset, send list) links that
    sgm_csr_create_partitioned builds for these row blocks (host-only)."""
    rs = np.ascontiguousarray(row_starts, np.int64)
    ptr = np.ascontiguousarray(ptr, np.int32)
    node = np.ascontiguousarray(node, np.int32)
    nparts = len(rs) - 1
    nl, need = C.c_int32(0), C.c_int64(0)
    _ck(lib().sgm_partition_links_host(C.c_int32(nparts), C.c_void_p(rs.ctypes.data), C.c_void_p(ptr.ctypes.data),
                                       C.c_void_p(node.ctypes.data), C.byref(nl), None, None, None, None, None,
                                       C.c_int64(0), C.byref(need)))
    a = [np.zeros(max(nl.value, 1), np.int32) for _ in range(4)]
    idx = np.zeros(max(need.value, 1), np.int32)
    _ck(lib().sgm_partition_links_host(C.c_int32(nparts), C.c_void_p(rs.ctypes.data), C.c_void_p(ptr.ctypes.data),
                                       C.c_void_p(node.ctypes.data), C.byref(nl), *[C.c_void_p(v.ctypes.data) for v in a],
                                       C.c_void_p(idx.ctypes.data), C.c_int64(len(idx)), None))
    out, off = [], 0
    for i in range(nl.value):
        cnt = int(a[3][i])
        out.append({"sender": int(a[0][i]), "receiver": int(a[1][i]), "recv_offset": int(a[2][i]),
                    "send_idx": idx[off:off + cnt].copy()})
        off += cnt
    return out


def partition_rows_by_nnz(ptr, nparts, align=512):
    """sgm_partition_rows_by_nnz: contiguous row blocks balanced by 12 nnz + 20 rows, boundaries at
    multiples of `align` rows; returns row_starts (nparts+1, 0-based, int64)."""
    ptr = np.ascontiguousarray(ptr, np.int32)
    rs = np.zeros(nparts + 1, np.int64)
    _ck(lib().sgm_partition_rows_by_nnz(C.c_int32(len(ptr) - 1), C.c_void_p(ptr.ctypes.data), C.c_int32(nparts),
                                        C.c_int32(align), C.c_void_p(rs.ctypes.data)))
    return rs


def ell_degrees_host(node):
    """sgm_ell_degrees_host: degrees(n) of an ELLPACK graph (node as (n, max_d) = the reference's (max_d, n), 1-based) recovered
    from the padding the reference keeps -- what sgm_ell_create does, since the product's interface carries no degrees."""
    node = np.ascontiguousarray(node, np.int32)
    n, md = node.shape
    deg = np.zeros(max(n, 1), np.int32)
    _ck(lib().sgm_ell_degrees_host(C.c_int32(n), C.c_int32(md), C.c_void_p(node.ctypes.data), C.c_void_p(deg.ctypes.data)))
    return deg[:n]


def left_permute_rows_host(p, ptr, node, val, r0, r1):
    """sgm_left_permute_rows_host: rows [r0, r1) (0-based) of A%left_permute(p) cut out of the whole matrix (1-based ptr / node):
    what a rank keeps of a permuted matrix distributed over ranks.  Returns (lptr, lnode, lval)."""
    p = np.ascontiguousarray(p, np.int32)
    ptr = np.ascontiguousarray(ptr, np.int32)
    node = np.ascontiguousarray(node, np.int32)
    val = np.ascontiguousarray(val, np.float64)
    n = len(ptr) - 1
    lptr = np.zeros(r1 - r0 + 1, np.int32)
    need = C.c_int64(0)
    args = (C.c_int32(n), C.c_void_p(p.ctypes.data), C.c_void_p(ptr.ctypes.data), C.c_void_p(node.ctypes.data),
            C.c_void_p(val.ctypes.data), C.c_int64(r0), C.c_int64(r1), C.c_void_p(lptr.ctypes.data))
    _ck(lib().sgm_left_permute_rows_host(*args, None, None, C.c_int64(0), C.byref(need)))
    lnode = np.zeros(max(need.value, 1), np.int32)
    lval = np.zeros(max(need.value, 1), np.float64)
    _ck(lib().sgm_left_permute_rows_host(*args, C.c_void_p(lnode.ctypes.data), C.c_void_p(lval.ctypes.data), C.c_int64(need.value), None))
    return lptr, lnode[:need.value], lval[:need.value]


def slice_sched_host(n_slices, period_rows, grid, band_slices=64):
    """sgm_slice_sched_host: the order in which `grid` workgroups take the 512-row slices of a sliced
    matrix whose rows carry a far offset of `period_rows`; returns the (iters, grid) int32 table
    (slice or -1)."""
    it = C.c_int32(0)
    _ck(lib().sgm_slice_sched_host(C.c_int64(n_slices), C.c_int64(period_rows), C.c_int32(grid), C.c_int32(band_slices),
                                   None, C.c_int64(0), C.byref(it)))
    tab = np.empty(it.value * grid, np.int32)
    _ck(lib().sgm_slice_sched_host(C.c_int64(n_slices), C.c_int64(period_rows), C.c_int32(grid), C.c_int32(band_slices),
                                   C.c_void_p(tab.ctypes.data), C.c_int64(tab.size), C.byref(it)))
    return tab.reshape(it.value, grid)


def heartbeat():
    """sgm_heartbeat: where the thread driving the library is right now; callable from another thread (no HIP call).
    {"phase": words, "phase_code", "beats", "iteration", "halo_posts", "allreduce_posts", "solves"}."""
    out = (C.c_int64 * 6)()
    name = C.create_string_buffer(160)
    lib().sgm_heartbeat(out, name, C.c_int(160))
    return {"phase": name.value.decode(), "phase_code": int(out[0]), "beats": int(out[1]), "iteration": int(out[2]),
            "halo_posts": int(out[3]), "allreduce_posts": int(out[4]), "solves": int(out[5])}


def dot(a, b):
    pa, wa, _k1 = _arg(a, np.float64)
    pb, wb, _k2 = _arg(b, np.float64)
    n = a.numel() if _is_torch(a) else len(a)
    r = C.c_double(0.0)
    _ck(lib().sgm_dot(C.c_int64(n), pa, pb, C.byref(r), C.c_int(_same_where(wa, wb))))
    return r.value


def axpy(alpha, x, y):
    px, wx, _k1 = _arg(x, np.float64)
    py, wy, _k2 = _arg(y, np.float64, writable=True)
    n = x.numel() if _is_torch(x) else len(x)
    _ck(lib().sgm_axpy(C.c_int64(n), C.c_double(alpha), px, py, C.c_int(_same_where(wx, wy))))
    return y
