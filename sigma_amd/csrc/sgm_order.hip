// Re-orderings of the matrix graph and permutation of a CSR matrix.
//   breadth_first_search / greedy_coloring / greedy_color_ordering   src/graph/permutations.f90:22-205
//   cs_matrix%left_permute / %right_permute   src/matrix/formats/cs_matrices.f90:471-490
//       -> graph_leftperm / graph_rightperm   default_sparse_matrix_kernels.f90:234-277
//       -> cs_graph_left_permute / _right_permute   src/graph/formats/cs_graphs.f90:499-571
// The three graph routines are queue-driven sequential algorithms whose result depends on the
// visiting order (the colouring also on running per-colour tallies), so they run on the host
// over a downloaded copy of the index arrays -- index work at setup, bit-exact, like the ILDU
// factorisation.  The permutation of the matrix itself is device work: row lengths scattered to
// their new places, a prefix sum (hipCUB), row segments copied in stored order (so a permuted
// row sums the same terms in the same order as before), columns renumbered in place; the device
// formats (offset dictionary ...) are then rebuilt.  What this buys on the hot path: ILDU(0)
// of a colour-ordered matrix has as many dependency levels as colours (2 for the 5-point grid
// instead of nx+ny-1), so its triangular solves run as a few full-width launches.
#include "sgm_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <vector>

namespace sgm {
int rebuild_csr_formats(Part &p);          // sgm_spmv.hip
int sgm_invalidate_transpose(sgm_mat A);   // sgm_spmv.hip
}
using namespace sgm;

namespace {

int host_graph(sgm_mat A, const char *who, std::vector<int32_t> &ptr, std::vector<int32_t> &node)
{
    if (!A) return fail(SGM_ERR_BAD_ARG, "%s: null matrix", who);
    if (A->fmt != SGM_FMT_CSR || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "%s: single-GPU CSR matrices only", who);
    if (A->nrow != A->ncol) return fail(SGM_ERR_BAD_ARG, "%s: the matrix graph must be square", who);
    const Part &p = A->parts[0];
    ptr.resize((size_t)p.n + 1);
    node.resize((size_t)p.nnz);
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    SGM_HIP(hipMemcpy(ptr.data(), p.rowptr, ptr.size() * 4, hipMemcpyDeviceToHost));       // 0-based on the device
    if (p.nnz) SGM_HIP(hipMemcpy(node.data(), p.col, node.size() * 4, hipMemcpyDeviceToHost));
    return SGM_OK;
}

// permutations.f90:83-157 on 0-based arrays; colours stay 1-based like the reference's
int32_t greedy_coloring_host(int32_t n, const std::vector<int32_t> &ptr, const std::vector<int32_t> &node,
                             int32_t *colors)
{
    int32_t d = 0;
    for (int32_t i = 0; i < n; ++i) d = std::max(d, ptr[i + 1] - ptr[i]);
    std::vector<int32_t> queue((size_t)std::max(n, 1)), neighbor_colors((size_t)d + 2, 0), color_totals((size_t)d + 2, 0);
    int32_t head = 0, tail = 0, used = 0;
    for (int32_t i = 0; i < n; ++i) colors[i] = -1;
    if (n > 0) { queue[tail++] = 0; colors[0] = 0; }
    while (tail > head) {
        std::fill(neighbor_colors.begin(), neighbor_colors.end(), 0);
        const int32_t i = queue[head++];
        for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) {
            const int32_t j = node[k], c = colors[j];
            if (c > 0) neighbor_colors[c - 1]++;
            else if (c == -1) { queue[tail++] = j; colors[j] = 0; }
        }
        int32_t color = 0, min_occupancy = n + 1;
        for (int32_t k = 1; k <= used; ++k)
            if (color_totals[k - 1] > 0 && color_totals[k - 1] < min_occupancy && neighbor_colors[k - 1] == 0) {
                color = k;
                min_occupancy = color_totals[k - 1];
            }
        if (color == 0) color = ++used;
        colors[i] = color;
        color_totals[color - 1]++;
    }
    return used;
}

__global__ void k_perm_lengths(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ p1,
                               int32_t *__restrict__ len2)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len2[p1[i] - 1] = rowptr[i + 1] - rowptr[i];
}
// one wave per row: the row's entries move to the new row's segment in stored order
__global__ void k_perm_rows(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ p1,
                            const int32_t *__restrict__ rowptr2, const int32_t *__restrict__ col,
                            const double *__restrict__ val, int32_t *__restrict__ col2, double *__restrict__ val2)
{
    const int32_t i = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int32_t s = rowptr[i], e = rowptr[i + 1], d = rowptr2[p1[i] - 1];
    for (int32_t k = s + lane; k < e; k += 64) {
        col2[d + (k - s)] = col[k];
        val2[d + (k - s)] = val[k];
    }
}
__global__ void k_perm_cols(int64_t nnz, int32_t *__restrict__ col, const int32_t *__restrict__ p1)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) col[k] = p1[col[k]] - 1;
}
__global__ void k_check_perm(int32_t n, const int32_t *__restrict__ p1, int32_t *__restrict__ seen, int *bad)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t v = p1[i];
    if (v < 1 || v > n || atomicAdd(&seen[v - 1], 1) != 0) *bad = 1;
}

// stage p on the device and make sure it is a permutation of 1..n
int stage_perm(const char *who, int32_t n, const int32_t *p, int where, int32_t **dp)
{
    if (where != SGM_HOST && where != SGM_DEVICE) return fail(SGM_ERR_BAD_ARG, "%s: bad `where`", who);
    SGM_TRY(dalloc(dp, (size_t)std::max(n, 1)));
    hipStream_t st = g_rt.stream;
    if (n) SGM_HIP(hipMemcpyAsync(*dp, p, (size_t)n * 4, where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    int32_t *seen = nullptr;
    int *bad = nullptr, hbad = 0;
    SGM_TRY(dalloc(&seen, (size_t)std::max(n, 1)));
    SGM_TRY(dalloc(&bad, 1));
    SGM_HIP(hipMemsetAsync(seen, 0, (size_t)std::max(n, 1) * 4, st));
    SGM_HIP(hipMemsetAsync(bad, 0, 4, st));
    if (n) hipLaunchKernelGGL(k_check_perm, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)*dp, seen, bad);
    SGM_HIP(hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(seen); dfree(bad);
    if (hbad) { dfree(*dp); *dp = nullptr; return fail(SGM_ERR_BAD_ARG, "%s: p is not a permutation of 1..%d", who, n); }
    return SGM_OK;
}

}  // namespace

extern "C" {

int sgm_graph_bfs_order(sgm_mat A, int32_t *p_out)
{
    SGM_TRY(require_init());
    std::vector<int32_t> ptr, node;
    SGM_TRY(host_graph(A, "sgm_graph_bfs_order", ptr, node));
    if (!p_out) return fail(SGM_ERR_BAD_ARG, "sgm_graph_bfs_order: null output");
    const int32_t n = A->nrow;
    std::vector<int32_t> queue((size_t)std::max(n, 1));
    int32_t head = 0, tail = 0, num = 0;
    for (int32_t i = 0; i < n; ++i) p_out[i] = -1;
    if (n > 0) queue[tail++] = 0;
    while (tail > head) {                                   // permutations.f90:44-72
        const int32_t i = queue[head++];
        p_out[i] = ++num;
        for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) {
            const int32_t j = node[k];
            if (p_out[j] == -1) { queue[tail++] = j; p_out[j] = 0; }
        }
    }
    return SGM_OK;
}

int sgm_graph_greedy_coloring(sgm_mat A, int32_t *colors_out, int32_t *num_colors)
{
    SGM_TRY(require_init());
    std::vector<int32_t> ptr, node;
    SGM_TRY(host_graph(A, "sgm_graph_greedy_coloring", ptr, node));
    if (!colors_out) return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_coloring: null output");
    const int32_t used = greedy_coloring_host(A->nrow, ptr, node, colors_out);
    if (num_colors) *num_colors = used;
    return SGM_OK;
}

int sgm_graph_greedy_color_order(sgm_mat A, int32_t *p_out, int32_t *ptrs_out, int32_t ptrs_len, int32_t *num_colors)
{
    SGM_TRY(require_init());
    std::vector<int32_t> ptr, node;
    SGM_TRY(host_graph(A, "sgm_graph_greedy_color_order", ptr, node));
    if (!p_out || !num_colors) return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_color_order: null output");
    const int32_t n = A->nrow;
    const int32_t nc = greedy_coloring_host(n, ptr, node, p_out);
    for (int32_t i = 0; i < n; ++i)
        if (p_out[i] < 1)      // the reference indexes ptrs(0) here (permutations.f90:184-186)
            return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_color_order: vertex %d is not reachable from vertex 1", i + 1);
    if (ptrs_out && ptrs_len < nc + 1)
        return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_color_order: ptrs needs %d entries", nc + 1);
    std::vector<int32_t> ptrs((size_t)nc + 1, 0), added((size_t)nc + 1, 0);
    for (int32_t i = 0; i < n; ++i) ptrs[p_out[i]]++;                  // permutations.f90:183-191
    ptrs[0] = 1;
    for (int32_t c = 1; c <= nc; ++c) ptrs[c] += ptrs[c - 1];
    for (int32_t i = 0; i < n; ++i) {                                  // :194-201
        const int32_t c = p_out[i];
        p_out[i] = ptrs[c - 1] + added[c - 1]++;
    }
    if (ptrs_out) std::copy(ptrs.begin(), ptrs.end(), ptrs_out);
    *num_colors = nc;
    return SGM_OK;
}

int sgm_mat_left_permute(sgm_mat A, const int32_t *p, int where)
{
    SGM_TRY(require_init());
    if (!A || !p) return fail(SGM_ERR_BAD_ARG, "sgm_mat_left_permute: null argument");
    if (A->fmt != SGM_FMT_CSR || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_left_permute: single-GPU CSR matrices only");
    Part &pt = A->parts[0];
    const int32_t n = pt.n;
    hipStream_t st = g_rt.stream;
    int32_t *dp = nullptr;
    SGM_TRY(stage_perm("sgm_mat_left_permute", n, p, where, &dp));
    int32_t *len2 = nullptr, *rowptr2 = nullptr, *col2 = nullptr;
    double *val2 = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = dalloc(&len2, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&rowptr2, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&col2, (size_t)pt.nnz + 2);
    if (rc == SGM_OK) rc = dalloc(&val2, (size_t)pt.nnz + 2);
    if (rc == SGM_OK) {
        (void)hipMemsetAsync(len2, 0, ((size_t)n + 1) * 4, st);
        (void)hipMemsetAsync(col2 + pt.nnz, 0, 8, st);
        (void)hipMemsetAsync(val2 + pt.nnz, 0, 16, st);
        if (n) hipLaunchKernelGGL(k_perm_lengths, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n,
                                  (const int32_t *)pt.rowptr, (const int32_t *)dp, len2);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, len2, rowptr2, n + 1, st);
        if (hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)) != hipSuccess) rc = fail(SGM_ERR_HIP, "sgm_mat_left_permute: scan workspace");
    }
    if (rc == SGM_OK) {
        (void)hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, len2, rowptr2, n + 1, st);
        if (n) hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)(((int64_t)n * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n,
                                  (const int32_t *)pt.rowptr, (const int32_t *)dp, (const int32_t *)rowptr2,
                                  (const int32_t *)pt.col, (const double *)pt.val, col2, val2);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
            rc = fail(SGM_ERR_HIP, "sgm_mat_left_permute: kernels failed");
    }
    if (tmp) (void)hipFree(tmp);
    dfree(len2); dfree(dp);
    if (rc != SGM_OK) { dfree(rowptr2); dfree(col2); dfree(val2); return rc; }
    dfree(pt.rowptr); dfree(pt.col); dfree(pt.val);
    pt.rowptr = rowptr2; pt.col = col2; pt.val = val2;
    SGM_TRY(sgm_invalidate_transpose(A));
    return rebuild_csr_formats(pt);
}

int sgm_mat_right_permute(sgm_mat A, const int32_t *p, int where)
{
    SGM_TRY(require_init());
    if (!A || !p) return fail(SGM_ERR_BAD_ARG, "sgm_mat_right_permute: null argument");
    if (A->fmt != SGM_FMT_CSR || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_right_permute: single-GPU CSR matrices only");
    Part &pt = A->parts[0];
    int32_t *dp = nullptr;
    SGM_TRY(stage_perm("sgm_mat_right_permute", A->ncol, p, where, &dp));
    if (pt.nnz) hipLaunchKernelGGL(k_perm_cols, dim3(vec_grid(pt.nnz)), dim3(kBlock), 0, g_rt.stream, pt.nnz, pt.col, (const int32_t *)dp);
    const bool ok = hipStreamSynchronize(g_rt.stream) == hipSuccess && hipGetLastError() == hipSuccess;
    dfree(dp);
    if (!ok) return fail(SGM_ERR_HIP, "sgm_mat_right_permute: kernel failed");
    SGM_TRY(sgm_invalidate_transpose(A));
    return rebuild_csr_formats(pt);
}

}  // extern "C"
