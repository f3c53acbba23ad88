// Matrix handles of the C ABI (include/sigma_hip.h): create / set_values / products and transpose products / composite /
// getters, and the staging of caller vectors.
#include "sgm_spmv_select.hpp"

namespace sgm {

// Stage a caller vector on the device if it lives on the host (or is not 16-B aligned).
struct Staged {
    double *dev = nullptr;
    bool owned = false;
    ~Staged() { if (owned) dfree(dev); }
};
int stage_in(Staged &s, const double *v, int64_t n, int where, bool copy)
{
    if (where == SGM_DEVICE && (reinterpret_cast<uintptr_t>(v) & 15) == 0) {
        s.dev = const_cast<double *>(v);
        return SGM_OK;
    }
    SGM_TRY(dalloc(&s.dev, (size_t)n));
    s.owned = true;
    if (copy)
        SGM_HIP(hipMemcpyAsync(s.dev, v, (size_t)n * 8,
                               where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               g_rt.stream));
    return SGM_OK;
}
int stage_out(const Staged &s, double *v, int64_t n, int where)
{
    if (!s.owned) return SGM_OK;
    SGM_HIP(hipMemcpyAsync(v, s.dev, (size_t)n * 8,
                           where == SGM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                           g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

__global__ void k_gather_perm(double *__restrict__ dst, const double *__restrict__ src,
                              const int32_t *__restrict__ perm, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[perm[i]];
}

// Transpose products (linear_operator_interface.f90:199-208 -> csc_matvec_add
// cs_matrices.f90:627-647 / ellpack_matvec_t_add ellpack_matrices.f90:670-693).  The reference
// scatters y(node(k)) += val(k)*x(j) for j = 1..n, k in stored order; a scatter needs atomics
// on a GPU and would lose the summation order.  Instead A^T is built once (device radix
// sort, stable in (j, k)), so y(i) is a ROW SUM over the same terms in the same order and the
// ordinary SpMV kernels apply (for matvec_t_add the sum is chained onto y(i), bit for bit
// like the scatter).  ELLPACK padding slots are kept (they add val=0 * x(j) like the reference).
// keys (= column of the entry) and source indices of all entries in (row j, slot k) order
__global__ void k_tr_keys_csr(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                              int32_t *__restrict__ key, int32_t *__restrict__ src, int32_t *__restrict__ rowid,
                              int32_t *__restrict__ count)
{
    const int32_t j = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    for (int32_t k = rowptr[j] + lane; k < rowptr[j + 1]; k += 64) {
        key[k] = col[k];
        src[k] = k;
        rowid[k] = j + 1;                        // 1-based row of A = column index in A^T
        atomicAdd(&count[col[k]], 1);
    }
}
__global__ void k_tr_keys_ell(int32_t n, int32_t max_d, const int32_t *__restrict__ ecol, int32_t *__restrict__ key,
                              int32_t *__restrict__ src, int32_t *__restrict__ rowid, int32_t *__restrict__ count)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // t = j*max_d + k
    if (t >= (int64_t)n * max_d) return;
    const int32_t j = (int32_t)(t / max_d), k = (int32_t)(t % max_d);
    const int64_t s = (int64_t)k * n + j;                                   // slot-major device layout
    const int32_t c = ecol[s];
    key[t] = c;
    src[t] = (int32_t)s;
    rowid[t] = j + 1;
    atomicAdd(&count[c], 1);
}
__global__ void k_tr_gather_rows(int64_t nnz, int64_t stride_t, int32_t max_d, int32_t n_src, const int32_t *__restrict__ src_sorted,
                                 const int32_t *__restrict__ rowid, int32_t *__restrict__ tnode)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        const int32_t s = src_sorted[i];
        // CSR: rowid is indexed by the entry; ELLPACK: by t = j*max_d + k with s = k*n + j
        tnode[i] = max_d ? rowid[(int64_t)(s % n_src) * max_d + s / n_src] : rowid[s];
    }
    (void)stride_t;
}
__global__ void k_inc1(int64_t n, int32_t *a)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] += 1;
}

static int ensure_transpose(sgm_mat A)
{
    if (A->distributed()) return fail(SGM_ERR_UNSUPPORTED, "matvec_t: not available on a row-partitioned matrix");
    Part &p = A->parts[0];
    const bool ell = A->fmt == SGM_FMT_ELL;
    const int64_t nnz = ell ? (int64_t)p.n * p.max_d : p.nnz;
    if (!A->T) {
        // A^T on the device: a STABLE radix sort of the entries by column (hipCUB) keeps them in
        // (row j, slot k) order inside every column, which is the order the reference's scatter adds
        // them in; the column histogram's prefix sum is A^T's row pointer.
        hipStream_t st = g_rt.stream;
        const int32_t nt = A->ncol;                        // rows of A^T
        const size_t m = (size_t)std::max<int64_t>(nnz, 1);
        int32_t *key = nullptr, *src = nullptr, *rowid = nullptr, *key2 = nullptr, *src2 = nullptr, *tptr = nullptr, *tnode = nullptr;
        double *zeros = nullptr;
        void *tmp = nullptr;
        size_t tb_sort = 0, tb_scan = 0;
        int rc = dalloc(&key, m);
        if (rc == SGM_OK) rc = dalloc(&src, m);
        if (rc == SGM_OK) rc = dalloc(&rowid, m);
        if (rc == SGM_OK) rc = dalloc(&key2, m);
        if (rc == SGM_OK) rc = dalloc(&src2, m);
        if (rc == SGM_OK) rc = dalloc(&tptr, (size_t)nt + 2);
        if (rc == SGM_OK) rc = dalloc(&tnode, m);
        if (rc == SGM_OK) rc = dalloc(&zeros, m);
        if (rc == SGM_OK) {
            int end_bit = 1;
            while (end_bit < 31 && (1ll << end_bit) <= (int64_t)nt) ++end_bit;
            (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb_sort, key, key2, src, src2, (int)std::min<int64_t>(nnz, INT32_MAX), 0, end_bit, st);
            (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb_scan, tptr, tptr, nt + 1, st);
            if (hipMalloc(&tmp, std::max<size_t>(std::max(tb_sort, tb_scan), 16)) != hipSuccess) rc = fail(SGM_ERR_HIP, "matvec_t: sort workspace");
            if (rc == SGM_OK) {
                (void)hipMemsetAsync(tptr, 0, ((size_t)nt + 2) * 4, st);
                (void)hipMemsetAsync(zeros, 0, m * 8, st);
                if (nnz && !ell && csr_need_arrays(p) != SGM_OK) rc = SGM_ERR_ALLOC;
                if (nnz && rc == SGM_OK) {
                    if (!ell)
                        hipLaunchKernelGGL(k_tr_keys_csr, dim3((unsigned)(((int64_t)p.n * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                                           p.n, (const int32_t *)p.rowptr, (const int32_t *)p.col, key, src, rowid, tptr);
                    else
                        hipLaunchKernelGGL(k_tr_keys_ell, dim3((unsigned)((nnz + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, p.n, p.max_d,
                                           (const int32_t *)p.ecol, key, src, rowid, tptr);
                    (void)hipcub::DeviceRadixSort::SortPairs(tmp, tb_sort, key, key2, src, src2, (int)nnz, 0, end_bit, st);
                    hipLaunchKernelGGL(k_tr_gather_rows, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, nnz, (int64_t)0, ell ? p.max_d : 0, p.n,
                                       (const int32_t *)src2, (const int32_t *)rowid, tnode);
                }
                (void)hipcub::DeviceScan::ExclusiveSum(tmp, tb_scan, tptr, tptr, nt + 1, st);
                hipLaunchKernelGGL(k_inc1, dim3(vec_grid(nt + 1)), dim3(kBlock), 0, st, (int64_t)nt + 1, tptr);     // 1-based, like the reference
                if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) rc = fail(SGM_ERR_HIP, "matvec_t: transpose build failed");
            }
        }
        sgm_mat T = nullptr;
        if (rc == SGM_OK) {
            T = new sgm_mat_s;
            T->fmt = SGM_FMT_CSR;
            T->nrow = A->ncol;
            T->ncol = A->nrow;
            T->nnz = nnz;
            T->parts.resize(1);
            T->parts[0].opt = p.opt;                       // A^T runs with A's options
            rc = build_csr_part(T->parts[0], T->nrow, T->ncol, 0, nnz, tptr, tnode, zeros, SGM_DEVICE, false);
        }
        if (tmp) (void)hipFree(tmp);
        dfree(key); dfree(src); dfree(rowid); dfree(key2); dfree(tptr); dfree(tnode); dfree(zeros);
        if (!ell) csr_release_arrays(p);
        if (rc != SGM_OK) { dfree(src2); if (T) sgm_mat_destroy(T); return rc; }
        A->tperm = src2;                                   // entry of A behind every entry of A^T
        A->T = T;
        A->t_stale = true;
    }
    if (A->t_stale && nnz) {
        Part &tp = A->T->parts[0];
        if (!ell) SGM_TRY(csr_need_arrays(p));             // A's values in CSR order (a lean part rebuilds them from its slices)
        SGM_TRY(lean_val_buffer(tp));
        hipLaunchKernelGGL(k_gather_perm, dim3(vec_grid(nnz)), dim3(kBlock), 0, g_rt.stream, tp.val,
                           (const double *)(ell ? p.eval : p.val), (const int32_t *)A->tperm, nnz);
        SGM_HIP(hipGetLastError());
        SGM_TRY(pack_sliced(tp));
        if (tp.cb_P) SGM_TRY(refresh_ell_colblock_values(tp));      // (a transpose with scattered columns has the column-blocked form)
        csr_release_arrays(tp);
        if (!ell) csr_release_arrays(p);
    }
    A->t_stale = false;
    return SGM_OK;
}

static int matvec_t_impl(sgm_mat A, const double *x, double *y, int where, bool add)
{
    SGM_TRY(require_init());
    if (!A || !x || !y) return fail(SGM_ERR_BAD_ARG, "matvec_t: null argument");
    if (A->fmt == SGM_FMT_COMPOSITE) {
        // composite_matvec_t_add (sparse_matrix_composites.f90:1104-1127): column blocks outer
        const int64_t nr = A->parts[0].n, nc = A->parts[0].ncol_own;        // (local lengths over distributed leaves)
        Staged sx, sy;
        SGM_TRY(stage_in(sx, x, nr, where, true));
        SGM_TRY(stage_in(sy, y, nc, where, add));
        if (!add) SGM_HIP(hipMemsetAsync(sy.dev, 0, (size_t)nc * 8, g_rt.stream));
        const int nrb = (int)A->blk_row_ptr.size() - 1, ncb = (int)A->blk_col_ptr.size() - 1;
        for (int jt = 0; jt < ncb; ++jt)
            for (int it = 0; it < nrb; ++it) {
                sgm_mat C = A->blocks[(size_t)it * ncb + jt];
                if (!C) continue;
                if (C->comm) {          // A^T of the leaf is a distributed matrix of its own (sgm_dist.hip)
                    SGM_TRY(matvec_t_dist(C, sx.dev + A->blk_row_ptr[it], sy.dev + A->blk_col_ptr[jt], SGM_DEVICE, true));
                    continue;
                }
                SGM_TRY(ensure_transpose(C));
                const double *xs[1] = {sx.dev + A->blk_row_ptr[it]};
                double *ys[1] = {sy.dev + A->blk_col_ptr[jt]};
                SGM_TRY(spmv_parts(C->T, xs, ys, true, nullptr, nullptr, nullptr, 0x7fffffff, true));
            }
        SGM_TRY(stage_out(sy, y, nc, where));
        return finish();
    }
    if (A->comm) return matvec_t_dist(A, x, y, where, add);
    SGM_TRY(ensure_transpose(A));
    Staged sx, sy;
    SGM_TRY(stage_in(sx, x, A->nrow, where, true));
    SGM_TRY(stage_in(sy, y, A->ncol, where, add));
    const double *xs[1] = {sx.dev};
    double *ys[1] = {sy.dev};
    SGM_TRY(spmv_parts(A->T, xs, ys, add, nullptr, nullptr, nullptr, 0x7fffffff, /*chain=*/add));
    SGM_TRY(stage_out(sy, y, A->ncol, where));
    return finish();
}

static int matvec_impl(sgm_mat A, const double *x, double *y, int where, bool add)
{
    SGM_TRY(require_init());
    if (!A || !x || !y) return fail(SGM_ERR_BAD_ARG, "matvec: null argument");
    const size_t P = A->parts.size();
    if (P == 1) {
        Part &p = A->parts[0];
        Staged sx, sy;
        SGM_TRY(stage_in(sx, x, p.xlen(), where, true));
        SGM_TRY(stage_in(sy, y, p.n, where, add));
        const double *xs[1] = {sx.dev};
        double *ys[1] = {sy.dev};
        SGM_TRY(spmv_parts(A, xs, ys, add, nullptr, nullptr, nullptr));
        SGM_TRY(stage_out(sy, y, p.n, where));
        return finish();
    }
    // in-process row partition: x and y are plain global-length vectors
    Staged sx, sy;
    SGM_TRY(stage_in(sx, x, A->ncol, where, true));
    SGM_TRY(stage_in(sy, y, A->nrow, where, add));
    std::vector<const double *> xs(P);
    std::vector<double *> ys(P);
    for (size_t ip = 0; ip < P; ++ip) {
        Part &p = A->parts[ip];
        SGM_HIP(hipMemcpyAsync(p.xext, sx.dev + p.row_begin, (size_t)p.ncol_own * 8,
                               hipMemcpyDeviceToDevice, g_rt.stream));
        xs[ip] = p.xext;
        ys[ip] = sy.dev + p.row_begin;
    }
    SGM_TRY(spmv_parts(A, xs.data(), ys.data(), add, nullptr, nullptr, nullptr));
    SGM_TRY(stage_out(sy, y, A->nrow, where));
    return finish();
}

// y = A x on device vectors laid out like sgm_mat_matvec's: one part (also one rank of a distributed matrix: x holds
// [owned | halo room]) or an in-process partition (plain global vectors); stream-ordered, no synchronisation
int matvec_plain(sgm_mat A, const double *x, double *y)
{
    const bool was_async = g_rt.async;
    g_rt.async = true;
    const int rc = matvec_impl(A, x, y, SGM_DEVICE, false);
    g_rt.async = was_async;
    return rc;
}

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_csr_create(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t nnz, const int32_t *ptr,
                   const int32_t *node, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || nnz < 0 || !ptr || (nnz && (!node || !val)))
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_create: bad argument");
    if (nnz > INT32_MAX - 4) return fail(SGM_ERR_UNSUPPORTED, "sgm_csr_create: nnz exceeds int32 ptr");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_CSR;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = nnz;
    A->parts.resize(1);
    int rc = build_csr_part(A->parts[0], nrow, ncol, 0, nnz, ptr, node, val, where, true);
    if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
    *out = A;
    return SGM_OK;
}

int sgm_csr_set_values(sgm_mat A, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!A || A->fmt != SGM_FMT_CSR || !val) return fail(SGM_ERR_BAD_ARG, "sgm_csr_set_values: bad argument");
    A->t_stale = true;
    A->version += 1;
    int64_t off = 0;
    for (auto &p : A->parts) {
        SGM_TRY(lean_val_buffer(p));
        SGM_HIP(hipMemcpyAsync(p.val, val + off, (size_t)p.nnz * 8,
                               where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                               g_rt.stream));
        off += p.nnz;
        SGM_TRY(pack_sliced(p));
        if (p.cb_P) SGM_TRY(refresh_ell_colblock_values(p));
        csr_release_arrays(p);
    }
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

// degrees(i) of an ELLPACK row as the reference holds it (ellpack_graphs.f90:14-21): add_edge / graph_build set the whole rest
// of the row to the neighbour just added (`g%node(d+1:, i) = j`, :164,:394-397) and never store a neighbour twice, so the last
// slot holds the last real neighbour and its FIRST occurrence is slot degrees(i); an empty row keeps node(:, i) = 0.
__global__ void k_ell_degrees(int32_t n, int32_t max_d, const int32_t *__restrict__ node /* (max_d, n) as handed over, 1-based */,
                              int32_t *__restrict__ deg)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t *row = node + (int64_t)i * max_d;
    const int32_t last = row[max_d - 1];
    int32_t d = 0;
    if (last != 0)
        for (d = 1; d < max_d && row[d - 1] != last; ++d) {}
    deg[i] = d;
}

int sgm_ell_create(sgm_mat *out, int32_t nrow, int32_t ncol, int32_t max_d, const int32_t *node,
                   const double *val, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || max_d < 0 || (nrow && max_d && (!node || !val)))
        return fail(SGM_ERR_BAD_ARG, "sgm_ell_create: bad argument");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_ELL;
    A->nrow = nrow;
    A->ncol = ncol;
    A->nnz = (int64_t)nrow * max_d;
    A->parts.resize(1);
    Part &p = A->parts[0];
    p.n = nrow;
    p.ncol_own = ncol;
    p.max_d = max_d;
    const size_t total = (size_t)nrow * max_d;
    int rc = dalloc(&p.ecol, total);
    if (rc == SGM_OK) rc = dalloc(&p.eval, total);
    if (rc != SGM_OK) { sgm_mat_destroy(A); return rc; }
    *out = A;
    if (total == 0) return SGM_OK;
    // (from here on *out owns A: an error return leaves a handle the caller may destroy -- except for rejected
    // index arrays, where nothing usable exists)
    int32_t *tn = nullptr;
    unsigned long long *bad = nullptr, hbad = ~0ull;
    SGM_TRY(dalloc(&bad, 1));
    SGM_HIP(hipMemsetAsync(bad, 0xff, sizeof(unsigned long long), g_rt.stream));
    const int32_t *src = node;
    if (where == SGM_HOST) {
        SGM_TRY(dalloc(&tn, total));
        SGM_HIP(hipMemcpyAsync(tn, node, total * 4, hipMemcpyHostToDevice, g_rt.stream));
        src = tn;
    }
    hipLaunchKernelGGL(k_ell_transpose, dim3(vec_grid(total)), dim3(kBlock), 0, g_rt.stream, src, (const double *)nullptr, p.ecol,
                       p.eval, nrow, max_d, ncol, bad);
    // degrees(n): not an argument (the product never reads it, ellpack_matrices.f90:640-665), but what the reference's edge
    // cursor and get_value go by (ellpack_graphs.f90:310-369) -- recovered from the padding the reference keeps
    SGM_TRY(dalloc(&p.edeg, (size_t)nrow));
    hipLaunchKernelGGL(k_ell_degrees, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, g_rt.stream, nrow, max_d, src, p.edeg);
    SGM_HIP(hipMemcpyAsync(&hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, g_rt.stream));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    dfree(tn);
    dfree(bad);
    if (hbad != ~0ull) {
        int32_t c = 0;
        const int64_t e = (int64_t)hbad;        // entry (slot k, row i) of the (max_d, n) array: e = i * max_d + k
        SGM_HIP(hipMemcpy(&c, node + e, sizeof c, where == SGM_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost));
        *out = nullptr;
        sgm_mat_destroy(A);
        return fail(SGM_ERR_DIMS, "ellpack create: node(%lld,%lld) = %d is outside 0..%d", (long long)(e % max_d) + 1,
                    (long long)(e / max_d) + 1, c, ncol);
    }
    SGM_TRY(build_ell_offset_dict(p));
    SGM_TRY(build_ell_colblock(p));
    return sgm_ell_set_values(A, val, where);
}

int sgm_ell_set_values(sgm_mat A, const double *val, int where)
{
    SGM_TRY(require_init());
    if (!A || A->fmt != SGM_FMT_ELL || !val) return fail(SGM_ERR_BAD_ARG, "sgm_ell_set_values: bad argument");
    Part &p = A->parts[0];
    A->t_stale = true;
    A->version += 1;
    const size_t total = (size_t)p.n * p.max_d;
    if (!total) return SGM_OK;
    double *tv = nullptr;
    const double *src = val;
    if (where == SGM_HOST) {
        SGM_TRY(dalloc(&tv, total));
        SGM_HIP(hipMemcpyAsync(tv, val, total * 8, hipMemcpyHostToDevice, g_rt.stream));
        src = tv;
    }
    hipLaunchKernelGGL(k_ell_transpose, dim3(vec_grid(total)), dim3(kBlock), 0, g_rt.stream,
                       (const int32_t *)nullptr, src, p.ecol, p.eval, p.n, p.max_d);
    SGM_HIP(hipGetLastError());
    SGM_TRY(pack_sliced(p));
    SGM_TRY(refresh_ell_colblock_values(p));
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    dfree(tv);
    return SGM_OK;
}

/* sgm_mat_set_option: this matrix's own copy of a kernel-selection option (sgm_set_option only changes what matrices
 * created LATER start with).  Options that choose among forms the handle already holds (csr_sliced, csr_offset_dict,
 * csr_row_owner, csr_row_lines, csr_sell, ell_offset_dict, ell_colblock 0 / nonzero, slice_sched) act from the next
 * product on; the ones a form is BUILT with (ell_colblock 0 <-> built, ell_colblock_cols / _rows, csr_lean)
 * rebuild / release that form here.  Every choice gives the same bits.  On a composite: applied to every block. */
int sgm_mat_set_option(sgm_mat A, const char *name, int value)
{
    SGM_TRY(require_init());
    if (!A || !name) return fail(SGM_ERR_BAD_ARG, "sgm_mat_set_option: null argument");
    int v = 0;
    SGM_TRY(normalise_option(name, value, &v));
    MatOptions probe;
    if (!mat_option_field(probe, name)) return fail(SGM_ERR_BAD_ARG, "sgm_mat_set_option: '%s' is not a matrix option", name);
    if (A->fmt == SGM_FMT_COMPOSITE) {
        for (sgm_mat_s *B : A->blocks)
            if (B) SGM_TRY(sgm_mat_set_option(B, name, value));
        return SGM_OK;
    }
    const bool cb_shape = !strcmp(name, "ell_colblock_cols") || !strcmp(name, "ell_colblock_rows");
    for (Part &p : A->parts) {
        int *f = mat_option_field(p.opt, name);
        const int old = *f;
        *f = v;
        if (old == v) continue;
        if ((p.ecol || (!p.lean && p.rowptr && p.col && p.val && p.n_halo == 0 && !p.sval && !p.sl_val)) &&
            (cb_shape || (!strcmp(name, "ell_colblock") && ((old != 0) != (v != 0) || v == 2 || old == 2)))) {
            SGM_TRY(build_ell_colblock(p));           // (frees the old form first; decides again whether the matrix wants one)
            SGM_TRY(refresh_ell_colblock_values(p));
            SGM_HIP(hipStreamSynchronize(g_rt.stream));
        }
        if (!strcmp(name, "csr_lean") && !p.ecol) {
            if (v == 0) { SGM_TRY(csr_need_arrays(p)); p.lean = false; }
            else csr_go_lean(p);
        }
        if (!strcmp(name, "slice_sched")) free_slice_sched(p);
    }
    if (A->T) SGM_TRY(sgm_mat_set_option(A->T, name, value));
    return SGM_OK;
}

int sgm_mat_matvec(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_impl(A, x, y, where, false);
}

int sgm_mat_matvec_add(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_impl(A, x, y, where, true);
}

int sgm_composite_create(sgm_mat *out, int32_t nrb, int32_t ncb, const int32_t *row_ptr, const int32_t *col_ptr,
                         const sgm_mat *blocks)
{
    SGM_TRY(require_init());
    if (!out || nrb < 1 || ncb < 1 || !row_ptr || !col_ptr || !blocks)
        return fail(SGM_ERR_BAD_ARG, "sgm_composite_create: bad argument");
    sgm_mat A = new sgm_mat_s;
    A->fmt = SGM_FMT_COMPOSITE;
    for (int i = 0; i <= nrb; ++i) A->blk_row_ptr.push_back(row_ptr[i] - 1);
    for (int j = 0; j <= ncb; ++j) A->blk_col_ptr.push_back(col_ptr[j] - 1);
    A->nrow = A->blk_row_ptr[nrb];
    A->ncol = A->blk_col_ptr[ncb];
    A->blocks.assign(blocks, blocks + (size_t)nrb * ncb);
    sgm_comm comm = nullptr;
    bool any_local = false;
    for (int it = 0; it < nrb; ++it)
        for (int jt = 0; jt < ncb; ++jt) {
            sgm_mat C = A->blocks[(size_t)it * ncb + jt];
            if (!C) continue;
            if (C->parts.size() != 1 || C->fmt == SGM_FMT_COMPOSITE || C->nrow != A->blk_row_ptr[it + 1] - A->blk_row_ptr[it] ||
                C->ncol != A->blk_col_ptr[jt + 1] - A->blk_col_ptr[jt]) {
                delete A;
                return fail(SGM_ERR_DIMS, "sgm_composite_create: block (%d,%d) does not fit its slot", it + 1, jt + 1);
            }
            if (C->comm) { if (comm && comm != C->comm) { delete A; return fail(SGM_ERR_BAD_ARG, "sgm_composite_create: leaves on different communicators"); } comm = C->comm; }
            else any_local = true;
            A->nnz += C->nnz;
        }
    int64_t nloc_r = A->nrow, nloc_c = A->ncol;
    if (comm) {
        // Leaves distributed over processes: block row i must be partitioned the same way in all its leaves, block
        // column j likewise, and (so that the operator maps a vector layout onto itself) block row i like block
        // column i.  The block offsets become the LOCAL ones: this rank's slices of the block vectors, concatenated.
        const int me = comm->rank;
        auto bad = [&](const char *why) { delete A; return fail(SGM_ERR_UNSUPPORTED, "sgm_composite_create over distributed leaves: %s", why); };
        if (any_local) return bad("every leaf must be distributed (sgm_csr_create_dist / _rect / sgm_ell_create_dist)");
        if (nrb != ncb) return bad("needs as many block rows as block columns");
        std::vector<const std::vector<int64_t> *> rpart(nrb, nullptr), cpart(ncb, nullptr);
        for (int it = 0; it < nrb; ++it)
            for (int jt = 0; jt < ncb; ++jt) {
                sgm_mat C = A->blocks[(size_t)it * ncb + jt];
                if (!C) continue;
                if (rpart[it] && *rpart[it] != C->row_starts) return bad("the leaves of a block row are partitioned differently");
                if (cpart[jt] && *cpart[jt] != C->col_starts) return bad("the leaves of a block column are partitioned differently");
                rpart[it] = &C->row_starts;
                cpart[jt] = &C->col_starts;
            }
        std::vector<int32_t> lr(1, 0), lc(1, 0);
        for (int it = 0; it < nrb; ++it) {
            if (!rpart[it] || !cpart[it]) return bad("a block row or column without any leaf has no partition");
            if (*rpart[it] != *cpart[it]) return bad("block row i must be partitioned like block column i");
            lr.push_back(lr.back() + (int32_t)((*rpart[it])[me + 1] - (*rpart[it])[me]));
            lc.push_back(lc.back() + (int32_t)((*cpart[it])[me + 1] - (*cpart[it])[me]));
        }
        A->blk_row_ptr = lr;
        A->blk_col_ptr = lc;
        A->comm = comm;
        nloc_r = lr.back();
        nloc_c = lc.back();
    }
    A->parts.resize(1);
    A->parts[0].n = (int32_t)nloc_r;
    A->parts[0].ncol_own = (int32_t)nloc_c;
    int64_t g = (nloc_r + 4 * kBlock - 1) / (4 * kBlock);
    A->parts[0].dot_grid_override = (int)std::max<int64_t>(1, std::min<int64_t>(g, 2048));
    *out = A;
    return SGM_OK;
}

int sgm_mat_matvec_t(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_t_impl(A, x, y, where, false);
}

int sgm_mat_matvec_t_add(sgm_mat A, const double *x, double *y, int where)
{
    return matvec_t_impl(A, x, y, where, true);
}

int sgm_mat_get(sgm_mat A, const char *name, void *out, size_t bytes, size_t *needed)
{
    SGM_TRY(require_init());
    if (!A || !name) return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: null argument");
    if (A->distributed() || A->fmt == SGM_FMT_COMPOSITE)
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_get: leaf single-GPU matrices only");
    const Part &p = A->parts[0];
    const std::string nm(name);
    std::vector<int32_t> vi;
    std::vector<double> vd;
    const bool ell = A->fmt == SGM_FMT_ELL;
    if (!ell && (nm == "node" || nm == "val")) SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; bool on; ~Release() { if (on) csr_release_arrays(p); } } rel{p, !ell};
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    if (!ell && nm == "ptr") {
        vi.resize((size_t)p.n + 1);
        SGM_HIP(hipMemcpy(vi.data(), p.rowptr, vi.size() * 4, hipMemcpyDeviceToHost));
        for (auto &v : vi) v += 1;
    } else if (!ell && nm == "node") {
        vi.resize((size_t)p.nnz);
        if (p.nnz) SGM_HIP(hipMemcpy(vi.data(), p.col, vi.size() * 4, hipMemcpyDeviceToHost));
        for (auto &v : vi) v += 1;
    } else if (!ell && nm == "val") {
        vd.resize((size_t)p.nnz);
        if (p.nnz) SGM_HIP(hipMemcpy(vd.data(), p.val, vd.size() * 8, hipMemcpyDeviceToHost));
    } else if (ell && nm == "max_d") {
        vi.assign(1, p.max_d);
    } else if (ell && nm == "degrees" && p.edeg) {
        vi.resize((size_t)p.n);
        if (p.n) SGM_HIP(hipMemcpy(vi.data(), p.edeg, vi.size() * 4, hipMemcpyDeviceToHost));
    } else if (ell && (nm == "node" || nm == "val")) {
        const size_t total = (size_t)p.n * p.max_d;       // back to the reference's (max_d, n) order
        if (nm == "node") {
            std::vector<int32_t> t(total);
            if (total) SGM_HIP(hipMemcpy(t.data(), p.ecol, total * 4, hipMemcpyDeviceToHost));
            vi.resize(total);
            for (int32_t i = 0; i < p.n; ++i)
                for (int32_t k = 0; k < p.max_d; ++k) vi[(size_t)i * p.max_d + k] = t[(size_t)k * p.n + i] + 1;
        } else {
            std::vector<double> t(total);
            if (total) SGM_HIP(hipMemcpy(t.data(), p.eval, total * 8, hipMemcpyDeviceToHost));
            vd.resize(total);
            for (int32_t i = 0; i < p.n; ++i)
                for (int32_t k = 0; k < p.max_d; ++k) vd[(size_t)i * p.max_d + k] = t[(size_t)k * p.n + i];
        }
    } else {
        return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: unknown array '%s' for this format", name);
    }
    const size_t sz = vi.size() * 4 + vd.size() * 8;
    if (needed) *needed = sz;
    if (out && sz) {
        if (bytes < sz) return fail(SGM_ERR_BAD_ARG, "sgm_mat_get: buffer too small (%zu < %zu)", bytes, sz);
        memcpy(out, vi.empty() ? (const void *)vd.data() : (const void *)vi.data(), sz);
    }
    return SGM_OK;
}

int sgm_mat_info(sgm_mat A, int32_t *nrow, int32_t *ncol, int64_t *nnz, int32_t *fmt, int64_t *x_len)
{
    if (!A) return fail(SGM_ERR_BAD_ARG, "sgm_mat_info: null matrix");
    if (nrow) *nrow = A->nrow;
    if (ncol) *ncol = A->ncol;
    if (nnz) *nnz = A->nnz;
    if (fmt) *fmt = A->fmt;
    if (x_len) *x_len = A->comm ? A->parts[0].xlen() : A->ncol;
    return SGM_OK;
}

}  // extern "C"
namespace sgm {
// name of the SpMV kernel one part of a CSR / ELLPACK matrix runs with under its current options
void part_kernel_name(const Part &p, int fmt, char *name, size_t len)
{
    if (fmt == SGM_FMT_ELL) {
        if (use_ell_colblock(p)) snprintf(name, len, "k_ellcb<cols=%d,R=%d>", p.cb_cols, p.cb_R);
        else if (use_sliced_ell(p)) snprintf(name, len, "k_csr_sl<W=%d>", p.sw);
        else if (p.ecode && p.opt.ell_offset_dict) snprintf(name, len, "k_ell_do<MDP=%d>", p.emdp);
        else snprintf(name, len, "k_ell_spmv");
    } else if (use_ell_colblock(p)) snprintf(name, len, "k_ellcb<cols=%d,R=%d,csr>", p.cb_cols, p.cb_R);
    else if (use_sliced(p)) snprintf(name, len, "k_csr_sl<W=%d>", p.sw);
    else if (use_slicedb(p)) snprintf(name, len, "k_csr_slb<W=%d>", p.sw);
    else if (use_sliced32(p)) snprintf(name, len, "k_csr_sl32<W=%d>", p.sw);
    else if (use_sell(p)) snprintf(name, len, p.sl_win0 && p.opt.csr_xwindow ? "k_csr_sell<pad=%.3f,xw=%dx%d>" : "k_csr_sell<pad=%.3f>",
                                   p.nnz ? (double)p.sl_total / (double)p.nnz : 1.0, p.sl_span, p.sl_gs);
    else if (use_offset_dict(p)) snprintf(name, len, "k_csr_do<256,%d,CW=1>", do_tile_for(p));
    else if (use_row_owner(p)) snprintf(name, len, "k_csr_do<256,%d,CW=4>", do_tile_for(p));
    else if (use_row_lines(p)) snprintf(name, len, "k_csr_rl");
    else snprintf(name, len, "k_csr_spmv");
}
}  // namespace sgm
extern "C" {

int sgm_mat_kernel(sgm_mat A, char *buf, int len)
{
    if (!A || !buf || len < 1) return fail(SGM_ERR_BAD_ARG, "sgm_mat_kernel: bad argument");
    char name[64];
    if (A->fmt == SGM_FMT_COMPOSITE) snprintf(name, sizeof name, "composite");
    else part_kernel_name(A->parts[0], A->fmt, name, sizeof name);
    snprintf(buf, (size_t)len, "%s", name);
    return SGM_OK;
}

// Bytes by construction: what lives in HBM for this handle, and what ONE
// y = A x moves with the kernel the current options select -- the stored format of that kernel
// (padded slices, codes, row pointers as it reads them), every x entry once, every y entry once.
static int64_t part_resident_bytes(const Part &p)
{
    int64_t b = 0;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    if (p.rowptr) b += 4 * ((int64_t)p.n + 1);
    if (p.col) b += 4 * (p.nnz + 4);
    if (p.val) b += 8 * (p.nnz + 2);
    if (p.code) b += p.nnz + 16;
    if (p.dict) b += 4 * 256;
    if (p.sval) b += 8 * nsl * kSlRows * p.sw;
    if (p.scode) b += 4 * nsl * kSlRows;
    if (p.scol) b += 4 * nsl * kSlRows * p.sw;
    if (p.sbcode) b += nsl * kSlRows * ((p.sw + 7) / 8 * 8);
    if (p.sl_val) b += 12 * p.sl_total + 2 * nsl * kSlRows + 8 * (nsl * (kSlRows / kSellChunk) + 1);
    if (p.ecol) b += 4 * (int64_t)p.n * p.max_d;
    if (p.eval) b += 8 * (int64_t)p.n * p.max_d;
    if (p.edeg) b += 4 * (int64_t)p.n;
    if (p.ecode) b += (int64_t)p.n * p.emdp;
    if (p.xext) b += 8 * p.xlen();
    b += ell_colblock_resident_bytes(p);
    for (const auto &nb : p.nbrs) b += (int64_t)nb.send_count * (nb.send_buf ? 12 : 4);
    return b;
}
static int64_t part_matvec_bytes(const sgm_mat_s *A, const Part &p)
{
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    int64_t m;
    if (A->fmt == SGM_FMT_ELL) {
        if (use_ell_colblock(p)) return ell_colblock_matvec_bytes(p);
        if (use_sliced_ell(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + 4);
        else if (p.ecode && p.opt.ell_offset_dict) m = (int64_t)p.n * (8 * (int64_t)p.max_d + p.emdp);
        else m = 12 * (int64_t)p.n * p.max_d;
    } else if (use_ell_colblock(p)) return ell_colblock_matvec_bytes(p);
    else if (use_sliced(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + 4);
    else if (use_slicedb(p)) m = nsl * kSlRows * (8 * (int64_t)p.sw + (p.sw + 7) / 8 * 8);
    else if (use_sliced32(p)) m = nsl * kSlRows * 12 * (int64_t)p.sw;
    else if (use_sell(p)) {
        m = 12 * p.sl_total + 2 * nsl * kSlRows + 8 * nsl * (kSlRows / kSellChunk);      // slots (entries + padding), positions, chunk offsets
        if (p.sl_win0 && p.opt.csr_xwindow)              // every slice loads its window of x (instead of "every x entry once")
            return m + ((nsl + p.sl_gs - 1) / p.sl_gs) * (8 * (int64_t)p.sl_span + 4) + 8 * (int64_t)p.n;
    }
    else if (use_offset_dict(p)) m = 9 * p.nnz + 4 * ((int64_t)p.n + 1);
    else m = 12 * p.nnz + 4 * ((int64_t)p.n + 1);
    return m + 8 * p.xlen() + 8 * (int64_t)p.n;
}

int sgm_mat_footprint(sgm_mat A, int64_t *resident_bytes, int64_t *matvec_bytes)
{
    if (!A) return fail(SGM_ERR_BAD_ARG, "sgm_mat_footprint: null matrix");
    int64_t res = 0, mv = 0;
    if (A->fmt == SGM_FMT_COMPOSITE) {
        for (sgm_mat C : A->blocks) {
            if (!C) continue;
            int64_t r = 0, m = 0;
            SGM_TRY(sgm_mat_footprint(C, &r, &m));
            mv += m + 8 * (int64_t)C->nrow;       // a block leaf adds onto y: one more read of its rows
        }
        mv += 8 * (int64_t)A->nrow;               // y = 0
    } else {
        for (const Part &p : A->parts) { res += part_resident_bytes(p); mv += part_matvec_bytes(A, p); }
        if (A->T) { int64_t r = 0; SGM_TRY(sgm_mat_footprint(A->T, &r, nullptr)); res += r + 4 * A->nnz; }
    }
    if (resident_bytes) *resident_bytes = res;
    if (matvec_bytes) *matvec_bytes = mv;
    return SGM_OK;
}

int sgm_mat_destroy(sgm_mat A)
{
    if (!A) return SGM_OK;
    for (auto &p : A->parts) free_part(p);
    if (A->T) sgm_mat_destroy(A->T);
    dfree(A->tperm);
    delete A;
    return SGM_OK;
}

}  // extern "C"
