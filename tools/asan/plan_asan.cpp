// TEST INFRASTRUCTURE ONLY -- the sanitizer build of the library's host-only index work.
// sigma_amd/csrc/sgm_plan_host.hpp (the halo / exchange planners of the row-partitioned path, the nnz-balanced row split, the
// slice schedule: the statements libsigma_hip.so itself runs) compiled by g++ with -fsanitize=address,undefined behind the
// same C-ABI entry points, so that tests/test_dist_cpu.py and tests/test_cabi_cpu.py can be run against it
// (tests/test_asan_cpu.py).  No HIP: nothing here touches a GPU, and nothing of it is shipped.
#include "../../sigma_amd/csrc/sgm_plan_host.hpp"

#include <cstdarg>
#include <cstdio>
#include <string>

namespace sgm {
static thread_local std::string g_err;
int fail(int code, const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace sgm

using namespace sgm;

extern "C" {

const char *sgm_last_error(void) { return g_err.c_str(); }

int sgm_halo_plan_host(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node, int32_t *node_local, int32_t *halo_cols,
                       int32_t *n_halo)
{
    return host_halo_plan_host(n_own, col_begin, nnz, node, node_local, halo_cols, n_halo);
}
int sgm_dist_plan_host(int32_t rank, int32_t nranks, const int64_t *row_starts, int32_t n_halo, const int32_t *halo_cols,
                       int32_t *want, int32_t *want_off, int32_t *req)
{
    return host_dist_plan_host(rank, nranks, row_starts, n_halo, halo_cols, want, want_off, req);
}
int sgm_dist_neighbors_host(int32_t rank, int32_t nranks, const int32_t *want_all, int32_t *peer, int32_t *send_count,
                            int32_t *recv_count, int32_t *recv_offset, int32_t *n_nbrs)
{
    return host_dist_neighbors_host(rank, nranks, want_all, peer, send_count, recv_count, recv_offset, n_nbrs);
}
int sgm_partition_links_host(int32_t nparts, const int64_t *row_starts, const int32_t *ptr, const int32_t *node, int32_t *n_links,
                             int32_t *sender, int32_t *receiver, int32_t *recv_offset, int32_t *count, int32_t *idx_concat,
                             int64_t idx_capacity, int64_t *idx_needed)
{
    return host_partition_links_host(nparts, row_starts, ptr, node, n_links, sender, receiver, recv_offset, count, idx_concat,
                                     idx_capacity, idx_needed);
}
int sgm_partition_rows_by_nnz(int32_t nrow, const int32_t *ptr, int32_t nparts, int32_t align, int64_t *row_starts)
{
    return host_partition_rows_by_nnz(nrow, ptr, nparts, align, row_starts);
}
int sgm_ell_degrees_host(int32_t n, int32_t max_d, const int32_t *node, int32_t *deg) { return host_ell_degrees_host(n, max_d, node, deg); }
int sgm_left_permute_rows_host(int32_t n, const int32_t *p, const int32_t *ptr, const int32_t *node, const double *val, int64_t r0,
                               int64_t r1, int32_t *lptr, int32_t *lnode, double *lval, int64_t capacity, int64_t *needed)
{
    return host_left_permute_rows_host(n, p, ptr, node, val, r0, r1, lptr, lnode, lval, capacity, needed);
}
int sgm_slice_sched_host(int64_t n_slices, int64_t period_rows, int32_t grid, int32_t band_slices, int32_t *tab_out,
                         int64_t capacity, int32_t *iters_out)
{
    return host_slice_sched_host(n_slices, period_rows, grid, band_slices, tab_out, capacity, iters_out);
}

}  // extern "C"
