// Host-only index work of the library: the halo / exchange planners of the row-partitioned path and the slice schedule of
// the sliced kernels.  Plain C++17 -- no HIP call, no HIP header -- so that the SAME statements the product runs are
//   * driven bit for bit by the CPU test suite through the C ABI (tests/test_dist_cpu.py, tests/test_cabi_cpu.py), and
//   * compiled by g++ with -fsanitize=address,undefined into tools/asan/libsgm_plan_asan.so (`make -C tools/asan`), which
//     the same tests then run against (tests/test_asan_cpu.py; SURVEY section 5: the reference's Debug flags,
//     /root/reference/CMakeLists.txt:8-12).
// sgm_dist.hip / sgm_spmv.hip include this file; the extern "C" entry points forward to the host_* functions below.
#pragma once
#include "../../include/sigma_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace sgm {

int fail(int code, const char *fmt, ...);        // sgm_runtime.hip (tools/asan/plan_asan.cpp in the sanitizer build)
#ifndef SGM_TRY
#define SGM_TRY(expr) do { const int rc__ = (expr); if (rc__ != SGM_OK) return rc__; } while (0)
#endif
constexpr int kPlanSliceRows = 512;              // rows of a slice of the sliced kernels (== kSlRows, sgm_internal.hpp)

// ------------------------------------------------------------------ halo / exchange planning
// Everything in this section is host-only index work (no HIP call): it is exported so that the
// CPU test suite drives the SAME code the RCCL path runs (tests/test_dist_cpu.py, world_size-2
// gloo), and sgm_csr_create_partitioned / sgm_csr_create_dist are both built on it.

// Sorted unique list of the non-owned columns + renumbering to [owned | halo].
inline void halo_plan(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node,
                      int32_t *node_local, std::vector<int32_t> &halo)
{
    const int64_t lo = col_begin + 1, hi = col_begin + n_own;     // owned 1-based range
    halo.clear();
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t c = node[k];
        if (c < lo || c > hi) halo.push_back((int32_t)c);
    }
    std::sort(halo.begin(), halo.end());
    halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
    for (int64_t k = 0; k < nnz; ++k) {
        const int64_t c = node[k];
        if (c >= lo && c <= hi) {
            node_local[k] = (int32_t)(c - col_begin);
        } else {
            const auto it = std::lower_bound(halo.begin(), halo.end(), (int32_t)c);
            node_local[k] = n_own + 1 + (int32_t)(it - halo.begin());
        }
    }
}

inline int owner_of(const int64_t *row_starts, int nparts, int64_t col0 /*0-based*/)
{
    const int64_t *it = std::upper_bound(row_starts, row_starts + nparts + 1, col0);
    return (int)(it - row_starts) - 1;
}

// What one rank asks of the others: want[q] = how many of my halo entries rank q owns (the halo
// list is sorted, owners are contiguous row blocks, so those entries are the run
// [want_off[q], want_off[q+1]) of the halo), req[t] = index of halo entry t in ITS OWNER's
// local numbering (0-based) -- the list the owner gathers from when it sends to me.
inline int dist_plan(int rank, int nranks, const int64_t *row_starts, const std::vector<int32_t> &halo,
                     std::vector<int32_t> &want, std::vector<int32_t> &want_off, std::vector<int32_t> &req)
{
    want.assign((size_t)nranks, 0);
    want_off.assign((size_t)nranks + 1, 0);
    req.resize(halo.size());
    const int64_t ntot = row_starts[nranks];
    for (size_t t = 0; t < halo.size(); ++t) {
        const int64_t c0 = (int64_t)halo[t] - 1;
        if (c0 < 0 || c0 >= ntot) return fail(SGM_ERR_BAD_ARG, "column %lld outside 1..%lld", (long long)halo[t], (long long)ntot);
        const int q = owner_of(row_starts, nranks, c0);
        if (q == rank) return fail(SGM_ERR_BAD_ARG, "halo entry %lld is owned by this rank", (long long)halo[t]);
        want[q]++;
        req[t] = (int32_t)(c0 - row_starts[q]);
    }
    for (int q = 0; q < nranks; ++q) want_off[q + 1] = want_off[q] + want[q];
    return SGM_OK;
}

struct NbrPlan { int peer; int32_t send_count, recv_count, recv_offset; };
// Neighbour table of `rank` from the all-gathered want matrix (row q = rank q's want[]).
inline void dist_neighbors(int rank, int nranks, const int32_t *want_all, std::vector<NbrPlan> &out)
{
    out.clear();
    int32_t off = 0;
    for (int q = 0; q < nranks; ++q) {
        const int32_t i_want = want_all[(size_t)rank * nranks + q];       // I receive this many from q
        const int32_t they_want = want_all[(size_t)q * nranks + rank];    // q receives this many of mine
        if (q != rank && (i_want || they_want)) out.push_back(NbrPlan{q, they_want, i_want, off});
        off += i_want;
    }
}

// All links of an in-process partition: the same dist_plan, run for every receiver.
struct Link { int sender, receiver; int32_t recv_offset; std::vector<int32_t> idx; };
inline int partition_links(int nparts, const int64_t *row_starts, const std::vector<std::vector<int32_t>> &halos,
                           std::vector<Link> &links)
{
    links.clear();
    std::vector<int32_t> want, want_off, req;
    for (int ip = 0; ip < nparts; ++ip) {
        SGM_TRY(dist_plan(ip, nparts, row_starts, halos[ip], want, want_off, req));
        for (int q = 0; q < nparts; ++q) {
            if (!want[q]) continue;
            Link l;
            l.sender = q;
            l.receiver = ip;
            l.recv_offset = want_off[q];
            l.idx.assign(req.begin() + want_off[q], req.begin() + want_off[q + 1]);
            links.push_back(std::move(l));
        }
    }
    return SGM_OK;
}


inline int host_halo_plan_host(int32_t n_own, int64_t col_begin, int64_t nnz, const int32_t *node,
                       int32_t *node_local, int32_t *halo_cols, int32_t *n_halo)
{
    if (n_own < 0 || nnz < 0 || (nnz && (!node || !node_local)) || !n_halo)
        return fail(SGM_ERR_BAD_ARG, "sgm_halo_plan_host: bad argument");
    std::vector<int32_t> halo;
    halo_plan(n_own, col_begin, nnz, node, node_local, halo);
    *n_halo = (int32_t)halo.size();
    if (halo_cols) std::copy(halo.begin(), halo.end(), halo_cols);
    return SGM_OK;
}

inline int host_dist_plan_host(int32_t rank, int32_t nranks, const int64_t *row_starts, int32_t n_halo,
                       const int32_t *halo_cols, int32_t *want, int32_t *want_off, int32_t *req)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !row_starts || n_halo < 0 || (n_halo && !halo_cols) || !want || !want_off)
        return fail(SGM_ERR_BAD_ARG, "sgm_dist_plan_host: bad argument");
    std::vector<int32_t> halo(halo_cols, halo_cols + n_halo), w, wo, r;
    SGM_TRY(dist_plan(rank, nranks, row_starts, halo, w, wo, r));
    std::copy(w.begin(), w.end(), want);
    std::copy(wo.begin(), wo.end(), want_off);
    if (req) std::copy(r.begin(), r.end(), req);
    return SGM_OK;
}

inline int host_dist_neighbors_host(int32_t rank, int32_t nranks, const int32_t *want_all, int32_t *peer,
                            int32_t *send_count, int32_t *recv_count, int32_t *recv_offset, int32_t *n_nbrs)
{
    if (nranks < 1 || rank < 0 || rank >= nranks || !want_all || !n_nbrs)
        return fail(SGM_ERR_BAD_ARG, "sgm_dist_neighbors_host: bad argument");
    std::vector<NbrPlan> nb;
    dist_neighbors(rank, nranks, want_all, nb);
    *n_nbrs = (int32_t)nb.size();
    for (size_t i = 0; i < nb.size(); ++i) {
        if (peer) peer[i] = nb[i].peer;
        if (send_count) send_count[i] = nb[i].send_count;
        if (recv_count) recv_count[i] = nb[i].recv_count;
        if (recv_offset) recv_offset[i] = nb[i].recv_offset;
    }
    return SGM_OK;
}

inline int host_partition_links_host(int32_t nparts, const int64_t *row_starts, const int32_t *ptr, const int32_t *node,
                             int32_t *n_links, int32_t *sender, int32_t *receiver, int32_t *recv_offset,
                             int32_t *count, int32_t *idx_concat, int64_t idx_capacity, int64_t *idx_needed)
{
    if (nparts < 1 || !row_starts || !ptr || !n_links) return fail(SGM_ERR_BAD_ARG, "sgm_partition_links_host: bad argument");
    std::vector<std::vector<int32_t>> halos((size_t)nparts);
    for (int ip = 0; ip < nparts; ++ip) {
        const int64_t r0 = row_starts[ip], r1 = row_starts[ip + 1];
        const int64_t k0 = ptr[r0] - 1, k1 = ptr[r1] - 1;
        std::vector<int32_t> lnode((size_t)std::max<int64_t>(k1 - k0, 1));
        halo_plan((int32_t)(r1 - r0), r0, k1 - k0, node + k0, lnode.data(), halos[ip]);
    }
    std::vector<Link> links;
    SGM_TRY(partition_links(nparts, row_starts, halos, links));
    int64_t total = 0;
    for (auto &l : links) total += (int64_t)l.idx.size();
    *n_links = (int32_t)links.size();
    if (idx_needed) *idx_needed = total;
    if (!sender) return SGM_OK;                       // sizing call
    if (idx_concat && idx_capacity < total) return fail(SGM_ERR_BAD_ARG, "sgm_partition_links_host: idx buffer too small");
    int64_t off = 0;
    for (size_t i = 0; i < links.size(); ++i) {
        sender[i] = links[i].sender;
        if (receiver) receiver[i] = links[i].receiver;
        if (recv_offset) recv_offset[i] = links[i].recv_offset;
        if (count) count[i] = (int32_t)links[i].idx.size();
        if (idx_concat) std::copy(links[i].idx.begin(), links[i].idx.end(), idx_concat + off);
        off += (int64_t)links[i].idx.size();
    }
    return SGM_OK;
}

inline int host_partition_rows_by_nnz(int32_t nrow, const int32_t *ptr, int32_t nparts, int32_t align, int64_t *row_starts)
{
    if (nrow < 0 || !ptr || nparts < 1 || !row_starts) return fail(SGM_ERR_BAD_ARG, "sgm_partition_rows_by_nnz: bad argument");
    if (align < 2) align = 2;                         // 16-byte vector accesses need even row boundaries
    if (align & 1) align += 1;
    // weight of the rows [0, r): the bytes of B_csr they account for (12 per entry, 20 per row)
    auto weight = [&](int64_t r) { return 12 * ((int64_t)ptr[r] - 1) + 20 * r; };
    const int64_t total = weight(nrow);
    row_starts[0] = 0;
    for (int p = 1; p < nparts; ++p) {
        const int64_t target = total / nparts * p + total % nparts * p / nparts;
        int64_t lo = row_starts[p - 1], hi = nrow;        // first row whose prefix weight reaches the target
        while (lo < hi) {
            const int64_t mid = (lo + hi) / 2;
            if (weight(mid) < target) lo = mid + 1; else hi = mid;
        }
        int64_t r = (lo + align / 2) / align * align;      // nearest multiple of `align`
        r = std::max<int64_t>(r, row_starts[p - 1]);
        row_starts[p] = std::min<int64_t>(r, nrow);
    }
    row_starts[nparts] = nrow;
    return SGM_OK;
}


// ------------------------------------------------------------------ ELLPACK degrees, rows of a permuted matrix
// degrees(i) of an ELLPACK row as the reference holds it (ellpack_graphs.f90:14-21): add_edge / graph_build set the whole rest of
// the row to the neighbour just added (`g%node(d+1:, i) = j`, :164,:394-397) and never store a neighbour twice, so the last slot
// holds the last real neighbour and its FIRST occurrence is slot degrees(i); an empty row keeps node(:, i) = 0.  (The device twin
// is k_ell_degrees, sgm_mat.hip.)  node: (max_d, n) as the Fortran holds it, 1-based.
inline int host_ell_degrees_host(int32_t n, int32_t max_d, const int32_t *node, int32_t *deg)
{
    if (n < 0 || max_d < 0 || (n && max_d && !node) || (n && !deg)) return fail(SGM_ERR_BAD_ARG, "sgm_ell_degrees_host: bad argument");
    for (int32_t i = 0; i < n; ++i) {
        int32_t d = 0;
        if (max_d) {
            const int32_t *row = node + (size_t)i * max_d;
            const int32_t last = row[max_d - 1];
            if (last != 0)
                for (d = 1; d < max_d && row[d - 1] != last; ++d) {}
        }
        deg[i] = d;
    }
    return SGM_OK;
}
// Rows [r0, r1) (0-based) of A%left_permute(p) (cs_matrices.f90:471-478: row i becomes row p(i), its entries in their stored
// order) cut out of the whole matrix (1-based ptr / node as the reference holds them): what a rank keeps of a permuted matrix
// distributed over ranks.  lptr: r1 - r0 + 1 entries, 1-based; lnode / lval: `capacity` entries; *needed = entries of the rows.
inline int host_left_permute_rows_host(int32_t n, const int32_t *p, const int32_t *ptr, const int32_t *node, const double *val,
                                       int64_t r0, int64_t r1, int32_t *lptr, int32_t *lnode, double *lval, int64_t capacity,
                                       int64_t *needed)
{
    if (n < 0 || !p || !ptr || r0 < 0 || r1 < r0 || r1 > n || !lptr) return fail(SGM_ERR_BAD_ARG, "sgm_left_permute_rows_host: bad argument");
    std::vector<int32_t> pinv((size_t)std::max(n, 1), 0);
    for (int32_t i = 0; i < n; ++i) {
        const int32_t t = p[i];
        if (t < 1 || t > n || pinv[(size_t)t - 1]) return fail(SGM_ERR_BAD_ARG, "left_permute: p is not a permutation of 1..%d (p(%d) = %d)", n, i + 1, t);
        pinv[(size_t)t - 1] = i + 1;
    }
    lptr[0] = 1;
    for (int64_t k = r0; k < r1; ++k) {
        const int64_t old = (int64_t)pinv[(size_t)k] - 1;
        lptr[(size_t)(k - r0) + 1] = lptr[(size_t)(k - r0)] + (ptr[(size_t)old + 1] - ptr[(size_t)old]);
    }
    const int64_t nnz = (int64_t)lptr[(size_t)(r1 - r0)] - 1;
    if (needed) *needed = nnz;
    if (!lnode) return SGM_OK;                       // sizing call
    if (capacity < nnz || (nnz && (!node || !val || !lval))) return fail(SGM_ERR_BAD_ARG, "sgm_left_permute_rows_host: %lld entries do not fit %lld", (long long)nnz, (long long)capacity);
    for (int64_t k = r0; k < r1; ++k) {
        const int64_t old = (int64_t)pinv[(size_t)k] - 1;
        const int64_t s = (int64_t)ptr[(size_t)old] - 1, e = (int64_t)ptr[(size_t)old + 1] - 1, dd = (int64_t)lptr[(size_t)(k - r0)] - 1;
        std::copy(node + s, node + e, lnode + dd);
        std::copy(val + s, val + e, lval + dd);
    }
    return SGM_OK;
}

// ---- slice schedule ---------------------------------------------------------------------------------
// A 3-D grid's rows reference x a whole plane away (offset +-D, D >> one slice).  With slices handed out round-robin or
// block-cyclic, the slices D rows apart -- which read the same x lines -- run on different XCDs, so every x line enters
// three L2s (464^3: slice s and s + 420.5 land 4 XCDs apart).  The schedule cuts the period D into NB bands (NB a multiple
// of 8, bands of about `slice_sched_band` slices); band(s) = floor(NB * frac((512 s + 256) / D)), XCD x walks bands
// x, x + 8, ... one after the other, each in ascending slice order: a slice and its +-D neighbours sit one band width apart
// in the SAME XCD's sequence, inside or next to the window of slices that XCD has in flight.  Workgroup b (XCD b % 8,
// the hardware's round-robin) takes positions b / 8, b / 8 + grid / 8, ... of its XCD's sequence: tab[it * grid + b].
// Only the ORDER of whole slices changes: every row is still summed by one lane in stored order.
inline void slice_sched_table(int64_t nsl, int64_t period_rows, int grid, int band_slices, std::vector<int32_t> &tab, int &iters)
{
    const double P = (double)period_rows / kPlanSliceRows;
    const int NB = 8 * std::max(1, (int)std::ceil(P / (8.0 * std::max(1, band_slices))));
    std::vector<int32_t> band((size_t)nsl);
    std::vector<int64_t> cnt((size_t)NB + 1, 0);
    for (int64_t sl = 0; sl < nsl; ++sl) {
        const double t = ((double)sl * kPlanSliceRows + kPlanSliceRows / 2) / (double)period_rows;
        int b = (int)((t - std::floor(t)) * NB);
        b = std::min(std::max(b, 0), NB - 1);
        band[(size_t)sl] = b;
        ++cnt[(size_t)b + 1];
    }
    // XCD x's sequence = bands x, x + 8, ... end to end; start[b] = position of band b's first slice inside it
    std::vector<int64_t> start((size_t)NB, 0), len(8, 0);
    for (int x = 0; x < 8; ++x)
        for (int b = x; b < NB; b += 8) { start[(size_t)b] = len[x]; len[x] += cnt[(size_t)b + 1]; }
    const int64_t L = grid / 8;
    const int64_t longest = *std::max_element(len.begin(), len.end());
    iters = (int)((longest + L - 1) / L);
    tab.assign((size_t)iters * grid, -1);
    for (int64_t sl = 0; sl < nsl; ++sl) {
        const int b = band[(size_t)sl], x = b & 7;
        const int64_t q = start[(size_t)b]++;
        tab[(size_t)((q / L) * grid + (q % L) * 8 + x)] = (int32_t)sl;
    }
}

inline int host_slice_sched_host(int64_t n_slices, int64_t period_rows, int32_t grid, int32_t band_slices,
                                    int32_t *tab_out, int64_t capacity, int32_t *iters_out)
{
    if (n_slices < 1 || n_slices > INT32_MAX || period_rows < 1 || grid < 8 || grid % 8 || !iters_out)
        return fail(SGM_ERR_BAD_ARG, "sgm_slice_sched_host: n_slices %lld, period %lld, grid %d (a multiple of 8)",
                    (long long)n_slices, (long long)period_rows, grid);
    std::vector<int32_t> tab;
    int iters = 0;
    slice_sched_table(n_slices, period_rows, grid, band_slices, tab, iters);
    *iters_out = iters;
    if (tab_out) {
        if (capacity < (int64_t)tab.size()) return fail(SGM_ERR_BAD_ARG, "sgm_slice_sched_host: capacity %lld < %zu", (long long)capacity, tab.size());
        memcpy(tab_out, tab.data(), tab.size() * sizeof(int32_t));
    }
    return SGM_OK;
}


}  // namespace sgm
