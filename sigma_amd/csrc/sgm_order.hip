// Re-orderings of the matrix graph and permutation of a CSR matrix.
//   breadth_first_search / greedy_coloring / greedy_color_ordering   src/graph/permutations.f90:22-205
//   cs_matrix%left_permute / %right_permute   src/matrix/formats/cs_matrices.f90:471-490
//       -> graph_leftperm / graph_rightperm   default_sparse_matrix_kernels.f90:234-277
//       -> cs_graph_left_permute / _right_permute   src/graph/formats/cs_graphs.f90:499-571
// The breadth-first numbering runs on the device (level-synchronous; the FIFO order of the
// reference is reproduced exactly, see k_bfs_*).  The greedy colouring is order-dependent AND
// depends on running per-colour tallies (every decision reads the global state left by the previous
// one), so it runs on the host over a downloaded copy of the index arrays -- index work at setup,
// bit-exact, like the ILDU factorisation.  The permutation of the matrix itself is device work: row lengths scattered to
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
void free_part(Part &p);                   // sgm_spmv.hip
int rebuild_ell_formats(Part &p);          // sgm_spmv.hip
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
    SGM_TRY(csr_need_arrays(p));
    if (p.nnz) SGM_HIP(hipMemcpy(node.data(), p.col, node.size() * 4, hipMemcpyDeviceToHost));
    csr_release_arrays(p);
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

// ---- breadth-first numbering, level-synchronous, in the reference's FIFO order ---------------
// The queue order of breadth_first_search (permutations.f90:44-72) is: level by level; inside a
// level, by (queue position of the parent that enqueued the vertex, slot of the vertex in the
// parent's neighbour list), i.e. the FIRST candidate slot that names an unvisited vertex when the
// frontier's neighbour lists are laid end to end.  That is a parallel computation: expand the
// frontier (prefix sum of degrees), atomicMin the slot number per vertex, keep the winners in slot
// order (stream compaction).
__global__ void k_bfs_degrees(int32_t m, const int32_t *__restrict__ frontier, const int32_t *__restrict__ rowptr,
                              int32_t *__restrict__ deg)
{
    const int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < m) { const int32_t u = frontier[r]; deg[r] = rowptr[u + 1] - rowptr[u]; }
    if (r == m) deg[r] = 0;                      // so that the exclusive sum's last entry is the total
}
// one wave per frontier vertex: its neighbour list goes to cand[off .. off+deg); unvisited ones bid
__global__ void k_bfs_expand(int32_t m, const int32_t *__restrict__ frontier, const int32_t *__restrict__ rowptr,
                             const int32_t *__restrict__ col, const int32_t *__restrict__ off,
                             const int32_t *__restrict__ p, int32_t *__restrict__ cand, int32_t *__restrict__ first)
{
    const int32_t r = (int32_t)(((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= m) return;
    const int32_t u = frontier[r], s0 = rowptr[u], d = rowptr[u + 1] - s0, o = off[r];
    for (int32_t k = lane; k < d; k += 64) {
        const int32_t j = col[s0 + k];
        cand[o + k] = j;
        if (p[j] == -1) atomicMin(&first[j], o + k);
    }
}
__global__ void k_bfs_flags(int32_t e, const int32_t *__restrict__ cand, const int32_t *__restrict__ p,
                            const int32_t *__restrict__ first, uint8_t *__restrict__ flag)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < e) { const int32_t j = cand[s]; flag[s] = p[j] == -1 && first[j] == s; }
}
__global__ void k_bfs_number(int32_t m, const int32_t *__restrict__ frontier, int32_t base, int32_t *__restrict__ p)
{
    const int32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) p[frontier[t]] = base + t + 1;
}
__global__ void k_fill_i32(int64_t n, int32_t *a, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = v;
}

__global__ void k_perm_lengths(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ p1,
                               int32_t *__restrict__ len2)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len2[p1[i] - 1] = rowptr[i + 1] - rowptr[i];
}
__global__ void k_perm_degrees(int32_t n, const int32_t *__restrict__ deg, const int32_t *__restrict__ p1, int32_t *__restrict__ deg2)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) deg2[p1[i] - 1] = deg[i];
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
// ELLPACK, slot-major device layout (slot k of row i at k*n + i): row i -> row p(i) in every slot
__global__ void k_ell_perm_rows(int32_t n, int32_t max_d, const int32_t *__restrict__ p1, const int32_t *__restrict__ ecol,
                                const double *__restrict__ eval, const int32_t *__restrict__ edeg,
                                int32_t *__restrict__ ecol2, double *__restrict__ eval2, int32_t *__restrict__ edeg2)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t d = p1[i] - 1;
    for (int32_t k = 0; k < max_d; ++k) {
        ecol2[(int64_t)k * n + d] = ecol[(int64_t)k * n + i];
        eval2[(int64_t)k * n + d] = eval[(int64_t)k * n + i];
    }
    if (edeg) edeg2[d] = edeg[i];
}
// ellpack_graph_right_permute: every real neighbour j -> p(j) (a 0 in the reference = -1 here stays)
__global__ void k_ell_perm_cols(int64_t total, int32_t *__restrict__ ecol, const int32_t *__restrict__ p1)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < total; k += stride) { const int32_t c = ecol[k]; if (c >= 0) ecol[k] = p1[c] - 1; }
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

// ---- greedy colouring on the device, for the graphs where it is a parallel computation -----------------
// greedy_coloring (permutations.f90:83-157) colours the vertices in breadth-first queue order from vertex 1; each takes,
// among the colours in use that none of its already coloured neighbours has, the one with the fewest members, else a new
// one.  In general every decision reads the tallies the previous one left: sequential.  But when
//   (a) every vertex is reached from vertex 1,
//   (b) every neighbour j /= i of a vertex i sits an ODD number of breadth-first levels away from it, and
//   (c) every vertex but the first has a neighbour on a lower level
// (a bipartite, structurally symmetric graph: every 5- / 7-point grid, holes and all) the result is forced: by induction
// over the queue order a vertex on an even level finds all its coloured neighbours in colour 2 and at least one of them
// there, so colour 1 is its only candidate (and vice versa; the second vertex of the queue opens colour 2): colour =
// 1 + (level mod 2), whatever the tallies.  Levels are a level-synchronous sweep, (a)-(c) are checked inside it; if one
// fails the caller runs the sequential host pass -- which also stays as the checker of this one (tests, SGM_COLOR_HOST=1).
__global__ __launch_bounds__(256) void k_lvl_step(int32_t L, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                  int32_t *level, const int32_t *__restrict__ cur, int32_t *__restrict__ nxt,
                                                  int32_t *cnt /* ring of 4 */, int32_t *state /* [0] bad, [1] vertices reached, [2] first empty level */)
{
    const int32_t m = cnt[L & 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        cnt[(L + 2) & 3] = 0;                      // (nobody reads or appends to that slot during this launch)
        if (m == 0) atomicMin(&state[2], L);
        else atomicAdd(&state[1], m);
    }
    const int32_t stride = gridDim.x * blockDim.x;
    for (int32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < m; t += stride) {
        const int32_t i = cur[t];
        bool lower = L == 0, odd = true;
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            const int32_t j = col[k];
            if (j == i) continue;                  // (the diagonal: its own colour is 0 = "queued" when a vertex looks, :119-128)
            int32_t lj = level[j];
            if (lj == -1) {
                const int32_t old = atomicCAS(&level[j], -1, L + 1);
                if (old == -1) nxt[atomicAdd(&cnt[(L + 1) & 3], 1)] = j;
                lj = L + 1;
            }
            odd = odd && (((lj - L) & 1) != 0);
            lower = lower || lj < L;
        }
        if (!odd || !lower) state[0] = 1;
    }
}
__global__ void k_color_flags(int32_t n, const int32_t *__restrict__ level, int32_t *__restrict__ is1)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) is1[i] = (level[i] & 1) == 0;
}
// colours[i] = 1 + level mod 2, or the ordering p(i) = position when the vertices are sorted by colour, ties by index
__global__ void k_color_out(int32_t n, int32_t n1, const int32_t *__restrict__ level, const int32_t *__restrict__ before1,
                            int32_t *__restrict__ colors, int32_t *__restrict__ p)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool c1 = (level[i] & 1) == 0;
    if (colors) colors[i] = c1 ? 1 : 2;
    if (p) p[i] = c1 ? 1 + before1[i] : n1 + 1 + (i - before1[i]);
}

// ---- the same colours without walking the levels one after another ---------------------------------------------------
// On a structurally SYMMETRIC graph (c) always holds and (b) says "bipartite": then the colour of a vertex is 1 + the parity
// of its distance from vertex 1, and that parity needs no breadth-first search -- it is what a union-find with one parity bit
// per link maintains when every edge (i, j) is recorded as "i and j differ".  All edges at once, any order (the sweep
// above costs one launch per level: 6323 of them on a 3162 x 3162 grid):
//   word P[v] = parent << 1 | (parity of v relative to that parent); a root points at itself, parity 0
//   * every stored word is a TRUE relation between v and one of its ancestors and stays true for ever (a vertex that has a
//     parent never becomes a root again; roots are only ever hooked under a root of SMALLER index: no cycles, and the root of
//     a finished component is its smallest vertex) -- so stale reads (the XCDs' L2s are not coherent with each other within a
//     launch) only make a climb stop early, and the hook itself is a compare-and-swap at device scope whose returned word,
//     when it fails, is the true link to climb on from (strictly smaller index every time: bounded);
//   * path halving writes true relations over true relations (never over a root).
// Afterwards, in launches of their own (coherent): every vertex climbs to its root -- all roots 0 <=> connected (a) -- and
// every edge is checked both ways: its ends differ in parity (b), and (j, i) is stored where (i, j) is (symmetry).  Any
// failure: *ok stays false and the level sweep above (then the host pass) decides.
__device__ inline uint32_t uf_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void uf_st(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// climb from x (parity px of the starting vertex relative to x) to the root as this thread sees it, halving the path
__device__ inline void uf_climb(uint32_t *P, uint32_t &x, uint32_t &px)
{
    for (;;) {
        const uint32_t w = uf_ld(P + x), p = w >> 1;
        if (p == x) return;
        const uint32_t wp = uf_ld(P + p), g = wp >> 1;
        if (g == p) { px ^= w & 1u; x = p; return; }
        uf_st(P + x, (g << 1) | ((w ^ wp) & 1u));
        px ^= (w ^ wp) & 1u;
        x = g;
    }
}
__global__ void k_uf_init(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, uint32_t *__restrict__ P)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t w = (uint32_t)v << 1;
    for (int32_t k = rowptr[v]; k < rowptr[v + 1]; ++k)
        if (col[k] < v) { w = ((uint32_t)col[k] << 1) | 1u; break; }
    P[v] = w;
}
__global__ void k_uf_union(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, uint32_t *P, int32_t *state)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool first = true;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
        const int32_t j = col[k];
        if (j >= i) continue;                        // every edge once, from its larger end (the symmetry check follows)
        if (first) { first = false; continue; }      // (k_uf_init's link)
        uint32_t u = (uint32_t)i, v = (uint32_t)j, pu = 0, pv = 0;
        for (;;) {
            uf_climb(P, u, pu);
            uf_climb(P, v, pv);
            if (u == v) { if (pu == pv) state[0] = 1; break; }        // an odd cycle
            if (u < v) { const uint32_t t = u; u = v; v = t; const uint32_t tp = pu; pu = pv; pv = tp; }
            const uint32_t old = atomicCAS(P + u, u << 1, (v << 1) | (pu ^ pv ^ 1u));
            if (old == (u << 1)) break;
            pu ^= old & 1u;                          // somebody hooked u first: its true link
            u = old >> 1;
        }
    }
}
__global__ void k_uf_flatten(int32_t n, uint32_t *P, int32_t *__restrict__ level, int32_t *state)
{
    const int32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    uint32_t x = (uint32_t)v, px = 0;
    uf_climb(P, x, px);
    level[v] = (int32_t)px;
    if (x != 0u) state[0] = 1;                       // not reached from vertex 1
}
__global__ void k_uf_verify(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, const int32_t *__restrict__ level,
                            int32_t *state)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t li = level[i];
    bool good = true;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1] && good; ++k) {
        const int32_t j = col[k];
        if (j == i) continue;
        good = level[j] != li;
        bool back = false;
        for (int32_t m = rowptr[j]; m < rowptr[j + 1] && !back; ++m) back = col[m] == i;
        good = good && back;
    }
    if (!good) state[0] = 1;
}

// *ok = false: the graph is not of the kind above (nothing written).  colors_dev / p_dev (either may be null): device
// arrays of n int32; *n1 = vertices of colour 1 (0-based position where colour 2 starts)
int greedy_coloring_device(const Part &pt, int32_t *colors_dev, int32_t *p_dev, int32_t *n1_out, int32_t *ncolors, bool *ok)
{
    *ok = false;
    const int32_t n = pt.n;
    if (n < 1 || pt.opt.coloring_pass == 2) return SGM_OK;
    SGM_TRY(csr_need_arrays(pt));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{pt};
    hipStream_t st = g_rt.stream;
    int32_t *level = nullptr, *fr[2] = {nullptr, nullptr}, *cnt = nullptr, *state = nullptr, *before = nullptr;
    void *tmp = nullptr;
    struct Scratch { int32_t **a, **b, **c, **d, **e, **f; void **g; ~Scratch() { dfree(*a); dfree(*b); dfree(*c); dfree(*d); dfree(*e); dfree(*f); if (*g) (void)hipFree(*g); } }
        guard{&level, &fr[0], &fr[1], &cnt, &state, &before, &tmp};
    SGM_TRY(dalloc(&level, (size_t)n));
    SGM_TRY(dalloc(&fr[0], (size_t)n));
    SGM_TRY(dalloc(&fr[1], (size_t)n));
    SGM_TRY(dalloc(&cnt, 4));
    SGM_TRY(dalloc(&state, 4));
    const int32_t init_cnt[4] = {1, 0, 0, 0}, init_state[4] = {0, 0, INT32_MAX, 0}, zero = 0;
    int32_t hstate[4] = {0, 0, INT32_MAX, 0};
    const int gridn = (n + kBlock - 1) / kBlock;
    bool parity_done = false;
    // (the union-find pass verifies symmetry by scanning row j for every edge (i, j): quadratic in the degree -- a hub of a
    //  million leaves would cost 1e12 steps.  Rows beyond 64 entries leave it to the level sweep / the linear host pass.)
    if (pt.opt.coloring_pass == 0 && pt.max_row <= 64) {
        uint32_t *P = reinterpret_cast<uint32_t *>(fr[0]);
        SGM_HIP(hipMemcpyAsync(state, init_state, sizeof init_state, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_uf_init, dim3(gridn), dim3(kBlock), 0, st, n, (const int32_t *)pt.rowptr, (const int32_t *)pt.col, P);
        hipLaunchKernelGGL(k_uf_union, dim3(gridn), dim3(kBlock), 0, st, n, (const int32_t *)pt.rowptr, (const int32_t *)pt.col, P, state);
        hipLaunchKernelGGL(k_uf_flatten, dim3(gridn), dim3(kBlock), 0, st, n, P, level, state);
        hipLaunchKernelGGL(k_uf_verify, dim3(gridn), dim3(kBlock), 0, st, n, (const int32_t *)pt.rowptr, (const int32_t *)pt.col,
                           (const int32_t *)level, state);
        SGM_HIP(hipGetLastError());
        SGM_HIP(hipMemcpyAsync(hstate, state, sizeof hstate, hipMemcpyDeviceToHost, st));
        SGM_HIP(hipStreamSynchronize(st));
        parity_done = hstate[0] == 0;
        hstate[0] = 0;
    }
    if (!parity_done) {
        hipLaunchKernelGGL(k_fill_i32, dim3(vec_grid(n)), dim3(kBlock), 0, st, (int64_t)n, level, -1);
        SGM_HIP(hipMemcpyAsync(cnt, init_cnt, sizeof init_cnt, hipMemcpyHostToDevice, st));
        SGM_HIP(hipMemcpyAsync(state, init_state, sizeof init_state, hipMemcpyHostToDevice, st));
        SGM_HIP(hipMemcpyAsync(fr[0], &zero, 4, hipMemcpyHostToDevice, st));         // the queue starts with vertex 1, level 0
        SGM_HIP(hipMemcpyAsync(level, &zero, 4, hipMemcpyHostToDevice, st));
        // one launch per level, 64 levels between two looks at the state; a fixed grid walks any frontier
        for (int32_t L = 0; hstate[2] == INT32_MAX && hstate[0] == 0 && L <= n; L += 64) {
            for (int32_t l = L; l < L + 64; ++l)
                hipLaunchKernelGGL(k_lvl_step, dim3(128), dim3(256), 0, st, l, (const int32_t *)pt.rowptr, (const int32_t *)pt.col, level,
                                   (const int32_t *)fr[l & 1], fr[(l + 1) & 1], cnt, state);
            SGM_HIP(hipMemcpyAsync(hstate, state, sizeof hstate, hipMemcpyDeviceToHost, st));
            SGM_HIP(hipStreamSynchronize(st));
        }
        SGM_HIP(hipGetLastError());
        if (hstate[0] != 0 || hstate[1] != n) return SGM_OK;          // not that kind of graph / not connected: the host pass decides
    }
    SGM_TRY(dalloc(&before, (size_t)n + 1));
    hipLaunchKernelGGL(k_color_flags, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)level, before);
    size_t tb = 0;
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, before, before, n + 1, st));
    SGM_HIP(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    SGM_HIP(hipMemsetAsync(before + n, 0, 4, st));
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, before, before, n + 1, st));
    int32_t n1 = 0;
    SGM_HIP(hipMemcpyAsync(&n1, before + n, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    hipLaunchKernelGGL(k_color_out, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, n1, (const int32_t *)level,
                       (const int32_t *)before, colors_dev, p_dev);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));
    if (n1_out) *n1_out = n1;
    if (ncolors) *ncolors = n1 < n ? 2 : 1;
    *ok = true;
    return SGM_OK;
}

}  // namespace

namespace sgm {
// greedy_color_ordering with the permutation left ON THE DEVICE (the reordering preconditioner, sgm_pc.hip): *dp = n int32,
// 1-based like the reference's p; ptrs = first position of every colour (num_colors + 1 entries, 1-based)
int color_order_device(sgm_mat A, int32_t **dp, std::vector<int32_t> &ptrs)
{
    *dp = nullptr;
    if (!A || A->fmt != SGM_FMT_CSR || A->distributed() || A->nrow != A->ncol)
        return fail(SGM_ERR_UNSUPPORTED, "colour ordering: single-GPU square CSR matrices only");
    const Part &pt = A->parts[0];
    const int32_t n = pt.n;
    SGM_TRY(dalloc(dp, (size_t)std::max(n, 1)));
    bool ok = false;
    int32_t n1 = 0, nc = 0;
    int rc = greedy_coloring_device(pt, nullptr, *dp, &n1, &nc, &ok);
    if (rc == SGM_OK && ok) {
        ptrs.assign({1, n1 + 1});
        if (nc == 2) ptrs.push_back(n + 1);
        return SGM_OK;
    }
    if (rc == SGM_OK) {
        std::vector<int32_t> hp((size_t)std::max(n, 1)), hptrs((size_t)n + 2, 0);
        int32_t hnc = 0;
        rc = sgm_graph_greedy_color_order(A, hp.data(), hptrs.data(), n + 2, &hnc);
        if (rc == SGM_OK) {
            ptrs.assign(hptrs.begin(), hptrs.begin() + hnc + 1);
            if (n && hipMemcpy(*dp, hp.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(SGM_ERR_HIP, "colour ordering: upload failed");
        }
    }
    if (rc != SGM_OK) { dfree(*dp); *dp = nullptr; }
    return rc;
}

// ---- row blocks of a partition: local orderings (the reordering preconditioner on parts / ranks, sgm_pc.hip) -----------------
namespace {
__global__ void k_blk_count(int32_t n, int32_t own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col, int32_t *__restrict__ cnt)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int32_t c = 0;
    if (i < n)
        for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) c += col[k] < own;
    cnt[i] = c;
}
__global__ void k_blk_fill(int32_t n, int32_t own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                           const double *__restrict__ val, const int32_t *__restrict__ rowptr2, int32_t *__restrict__ col2, double *__restrict__ val2)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t d = rowptr2[i];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (col[k] < own) { col2[d] = col[k]; val2[d] = val[k]; ++d; }
}
__global__ void k_perm_cols_own(int64_t nnz, int32_t own, int32_t *__restrict__ col, const int32_t *__restrict__ p1)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) { const int32_t c = col[k]; if (c < own) col[k] = p1[c] - 1; }
}
__global__ void k_map_idx(int32_t count, const int32_t *__restrict__ src, const int32_t *__restrict__ p1, int32_t *__restrict__ dst)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) dst[j] = p1[src[j]] - 1;
}
// ... into the slot the receiver's re-ordered halo gives entry j of the link (order[j], halo_attach_order below)
__global__ void k_map_idx_ordered(int32_t count, const int32_t *__restrict__ src, const int32_t *__restrict__ p1,
                                  const int32_t *__restrict__ order, int32_t *__restrict__ dst)
{
    const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) dst[order[j]] = p1[src[j]] - 1;
}
// halo columns of a permuted part renumbered: slot h -> hmap[h]
__global__ void k_remap_halo_cols(int64_t nnz, int32_t own, int32_t *__restrict__ col, const int32_t *__restrict__ hmap)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) { const int32_t c = col[k]; if (c >= own) col[k] = own + hmap[c - own]; }
}
// key[h] = the smallest PERMUTED index of a row that references halo slot h (INT32_MAX: no row does)
__global__ void k_halo_attach_key(int32_t n, int32_t own, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                  const int32_t *__restrict__ p1, int32_t *__restrict__ key)
{
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t r = p1[i] - 1;
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k)
        if (col[k] >= own) atomicMin(key + (col[k] - own), r);
}
// exclusive scan of n + 1 int32 counts in place
int scan_counts(int32_t *a, int32_t n1)
{
    size_t tb = 0;
    void *tmp = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, a, a, n1, g_rt.stream);
    SGM_HIP(hipMalloc(&tmp, std::max<size_t>(tb, 16)));
    const hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tb, a, a, n1, g_rt.stream);
    const hipError_t e2 = hipStreamSynchronize(g_rt.stream);
    (void)hipFree(tmp);
    SGM_HIP(e);
    SGM_HIP(e2);
    return SGM_OK;
}
}  // namespace

int diag_block_plain(const Part &p, sgm_mat *out)
{
    *out = nullptr;
    SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{p};
    hipStream_t st = g_rt.stream;
    const int32_t n = p.n, own = p.n_halo == 0 ? INT32_MAX : p.ncol_own;
    sgm_mat C = new sgm_mat_s;
    struct Guard { sgm_mat &C; ~Guard() { if (C) sgm_mat_destroy(C); } } guard{C};
    C->fmt = SGM_FMT_CSR; C->nrow = C->ncol = n;
    C->parts.resize(1);
    Part &q = C->parts[0];
    q.opt.csr_offset_dict = 0; q.opt.csr_sliced = 0; q.opt.csr_sell = 0; q.opt.csr_lean = 0; q.opt.slice_sched = 0;
    q.opt.coloring_pass = p.opt.coloring_pass;
    q.n = n; q.ncol_own = n; q.n_halo = 0;
    SGM_TRY(dalloc(&q.rowptr, (size_t)n + 1));
    hipLaunchKernelGGL(k_blk_count, dim3((n + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, own, (const int32_t *)p.rowptr,
                       (const int32_t *)p.col, q.rowptr);
    SGM_TRY(scan_counts(q.rowptr, n + 1));
    int32_t tot = 0;
    SGM_HIP(hipMemcpy(&tot, q.rowptr + n, 4, hipMemcpyDeviceToHost));
    q.nnz = C->nnz = tot;
    q.max_row = p.max_row;
    SGM_TRY(dalloc(&q.col, (size_t)tot + 4));
    SGM_TRY(dalloc(&q.val, (size_t)tot + 2));
    SGM_HIP(hipMemsetAsync(q.col + tot, 0, 16, st));
    SGM_HIP(hipMemsetAsync(q.val + tot, 0, 16, st));
    if (n) hipLaunchKernelGGL(k_blk_fill, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, own, (const int32_t *)p.rowptr,
                              (const int32_t *)p.col, (const double *)p.val, (const int32_t *)q.rowptr, q.col, q.val);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));
    *out = C;
    C = nullptr;
    return SGM_OK;
}

int permuted_part(const Part &p, const int32_t *p1, Part &q, const int32_t *hmap, const std::vector<int32_t *> *send_order)
{
    SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{p};
    hipStream_t st = g_rt.stream;
    const int32_t n = p.n;
    free_part(q);
    q.opt = p.opt;
    q.n = n; q.ncol_own = p.ncol_own; q.n_halo = p.n_halo; q.nnz = p.nnz; q.row_begin = p.row_begin;
    q.int_lo = q.int_hi = 0;
    int32_t *len2 = nullptr;
    struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } tmp{len2};
    SGM_TRY(dalloc(&len2, (size_t)n + 1));
    SGM_TRY(dalloc(&q.rowptr, (size_t)n + 1));
    SGM_TRY(dalloc(&q.col, (size_t)p.nnz + 4));
    SGM_TRY(dalloc(&q.val, (size_t)p.nnz + 2));
    SGM_HIP(hipMemsetAsync(len2, 0, ((size_t)n + 1) * 4, st));
    SGM_HIP(hipMemsetAsync(q.col + p.nnz, 0, 16, st));
    SGM_HIP(hipMemsetAsync(q.val + p.nnz, 0, 16, st));
    if (n) hipLaunchKernelGGL(k_perm_lengths, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)p.rowptr, p1, len2);
    SGM_TRY(scan_counts(len2, n + 1));
    SGM_HIP(hipMemcpyAsync(q.rowptr, len2, ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, st));
    if (n) hipLaunchKernelGGL(k_perm_rows, dim3((unsigned)(((int64_t)n * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, n,
                              (const int32_t *)p.rowptr, p1, (const int32_t *)q.rowptr, (const int32_t *)p.col, (const double *)p.val,
                              q.col, q.val);
    if (p.nnz) hipLaunchKernelGGL(k_perm_cols_own, dim3(vec_grid(p.nnz)), dim3(kBlock), 0, st, p.nnz, p.n_halo == 0 ? INT32_MAX : p.ncol_own,
                                  q.col, p1);
    // the halo slots in the order of the permuted rows they attach to (halo_attach_order): a stencil's halo columns are then a
    // constant offset away from their rows again, and the part keeps the 4-bit dictionary form in the permuted order
    if (hmap && p.n_halo > 0 && p.nnz) hipLaunchKernelGGL(k_remap_halo_cols, dim3(vec_grid(p.nnz)), dim3(kBlock), 0, st, p.nnz, p.ncol_own, q.col, hmap);
    if (p.edeg) {                  // ELLPACK rows over ranks: the degrees follow their rows (the entries of a row keep their stored order)
        SGM_TRY(dalloc(&q.edeg, (size_t)std::max(n, 1)));
        if (n) hipLaunchKernelGGL(k_perm_degrees, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, (const int32_t *)p.edeg, p1, q.edeg);
    }
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));
    SGM_TRY(rebuild_csr_formats(q));
    if (p.xext) SGM_TRY(dalloc(&q.xext, (size_t)q.xlen()));
    size_t inb = 0;
    for (const HaloNbr &nb : p.nbrs) {
        HaloNbr m = nb;
        m.send_idx = nullptr; m.send_buf = nullptr;
        q.nbrs.push_back(m);                              // (owned by q from here on: free_part releases what follows)
        HaloNbr &o = q.nbrs.back();
        const int32_t *ord = send_order && inb < send_order->size() ? (*send_order)[inb] : nullptr;
        ++inb;
        if (nb.send_idx) {
            SGM_TRY(dalloc(&o.send_idx, (size_t)std::max(nb.send_count, 1)));
            const dim3 g((nb.send_count + kBlock - 1) / kBlock);
            if (nb.send_count && ord)        // the receiver re-ordered its halo: entry j of the link goes to its slot ord[j]
                hipLaunchKernelGGL(k_map_idx_ordered, g, dim3(kBlock), 0, st, nb.send_count, (const int32_t *)nb.send_idx, p1, ord, o.send_idx);
            else if (nb.send_count)
                hipLaunchKernelGGL(k_map_idx, g, dim3(kBlock), 0, st, nb.send_count, (const int32_t *)nb.send_idx, p1, o.send_idx);
        }
        if (nb.send_buf) SGM_TRY(dalloc(&o.send_buf, (size_t)std::max(nb.send_count, 1)));
    }
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));
    return SGM_OK;
}

// The order a permuted part wants its halo slots in: inside every neighbour's segment [off, off + count) of the halo, by the
// permuted index of the first row that references the slot (ties and unreferenced slots: as they were).  hmap_host[h] = the new
// slot of old slot h.  Index work only: the VALUES a slot receives and the order a row adds its entries in do not change, so
// every product on the part keeps its bits.
int halo_attach_order(const Part &p, const int32_t *p1, const std::vector<std::pair<int32_t, int32_t>> &segments,
                      std::vector<int32_t> &hmap_host)
{
    const int32_t nh = p.n_halo;
    hmap_host.resize((size_t)nh);
    for (int32_t h = 0; h < nh; ++h) hmap_host[(size_t)h] = h;
    if (nh == 0 || p.n == 0) return SGM_OK;
    SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{p};
    hipStream_t st = g_rt.stream;
    int32_t *key = nullptr;
    SGM_TRY(dalloc(&key, (size_t)nh));
    struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } tmp{key};
    SGM_HIP(hipMemsetAsync(key, 0x7f, (size_t)nh * 4, st));          // 0x7f7f7f7f: beyond every row index
    hipLaunchKernelGGL(k_halo_attach_key, dim3((p.n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, p.n, p.ncol_own, (const int32_t *)p.rowptr,
                       (const int32_t *)p.col, p1, key);
    std::vector<int32_t> hk((size_t)nh);
    SGM_HIP(hipMemcpyAsync(hk.data(), key, (size_t)nh * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> idx;
    for (const auto &sg : segments) {
        const int32_t off = sg.first, cnt = sg.second;
        if (off < 0 || cnt < 0 || (int64_t)off + cnt > nh) return fail(SGM_ERR_BAD_ARG, "halo_attach_order: a segment lies outside the halo");
        idx.resize((size_t)cnt);
        for (int32_t t = 0; t < cnt; ++t) idx[(size_t)t] = off + t;
        std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return hk[(size_t)a] < hk[(size_t)b]; });
        for (int32_t t = 0; t < cnt; ++t) hmap_host[(size_t)idx[(size_t)t]] = off + t;
    }
    return SGM_OK;
}

}  // namespace sgm

namespace sgm {
// (sgm_dist.hip) the same operations on a matrix distributed over ranks: the graph gathered onto every rank, rows moved between ranks
int permute_dist(sgm_mat A, const int32_t *p_host_global, bool left);
int gathered_graph(sgm_mat A, sgm_mat *out);
}  // namespace sgm

extern "C" {

int sgm_graph_bfs_order(sgm_mat A, int32_t *p_out)
{
    SGM_TRY(require_init());
    if (!A || !p_out) return fail(SGM_ERR_BAD_ARG, "sgm_graph_bfs_order: null argument");
    if (A->fmt == SGM_FMT_CSR && A->comm) {          // over ranks: the whole graph on every rank, the single-GPU pass, the same p everywhere
        sgm_mat G = nullptr;
        SGM_TRY(gathered_graph(A, &G));
        const int rc = sgm_graph_bfs_order(G, p_out);
        sgm_mat_destroy(G);
        return rc;
    }
    if (A->fmt != SGM_FMT_CSR || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "sgm_graph_bfs_order: CSR matrices on one GPU or distributed over ranks (not in-process partitions)");
    if (A->nrow != A->ncol) return fail(SGM_ERR_BAD_ARG, "sgm_graph_bfs_order: the matrix graph must be square");
    const Part &pt = A->parts[0];
    const int32_t n = pt.n;
    if (n == 0) return SGM_OK;
    SGM_TRY(csr_need_arrays(pt));          // the neighbour lists (a part that kept only its sliced form rebuilds them)
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{pt};
    hipStream_t st = g_rt.stream;
    const size_t ne = (size_t)std::max<int64_t>(pt.nnz, 1);
    int32_t *p = nullptr, *first = nullptr, *fr[2] = {nullptr, nullptr}, *deg = nullptr, *off = nullptr, *cand = nullptr, *cnt = nullptr;
    uint8_t *flag = nullptr;
    void *tmp = nullptr;
    size_t tb1 = 0, tb2 = 0;
    int rc = dalloc(&p, (size_t)n);
    if (rc == SGM_OK) rc = dalloc(&first, (size_t)n);
    if (rc == SGM_OK) rc = dalloc(&fr[0], (size_t)n);
    if (rc == SGM_OK) rc = dalloc(&fr[1], (size_t)n);
    if (rc == SGM_OK) rc = dalloc(&deg, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&off, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&cand, ne);
    if (rc == SGM_OK) rc = dalloc(&flag, ne);
    if (rc == SGM_OK) rc = dalloc(&cnt, 1);
    if (rc == SGM_OK) {
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb1, deg, off, n + 1, st);
        (void)hipcub::DeviceSelect::Flagged(nullptr, tb2, cand, flag, fr[0], cnt, (int)std::min<size_t>(ne, INT32_MAX), st);
        if (hipMalloc(&tmp, std::max<size_t>(std::max(tb1, tb2), 16)) != hipSuccess) rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: workspace");
    }
    if (rc == SGM_OK) {
        hipLaunchKernelGGL(k_fill_i32, dim3(vec_grid(n)), dim3(kBlock), 0, st, (int64_t)n, p, -1);
        hipLaunchKernelGGL(k_fill_i32, dim3(vec_grid(n)), dim3(kBlock), 0, st, (int64_t)n, first, INT32_MAX);
        const int32_t zero = 0;
        (void)hipMemcpyAsync(fr[0], &zero, 4, hipMemcpyHostToDevice, st);          // the queue starts with vertex 1
        int32_t m = 1, base = 0, levels = 0;
        int cur = 0;
        bool on_host = false;
        hipLaunchKernelGGL(k_bfs_number, dim3(1), dim3(kBlock), 0, st, m, (const int32_t *)fr[0], base, p);
        base = 1;
        while (m > 0 && rc == SGM_OK) {
            hipLaunchKernelGGL(k_bfs_degrees, dim3((m + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, m,
                               (const int32_t *)fr[cur], (const int32_t *)pt.rowptr, deg);
            (void)hipcub::DeviceScan::ExclusiveSum(tmp, tb1, deg, off, m + 1, st);
            int32_t e = 0;
            (void)hipMemcpyAsync(&e, off + m, 4, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: level failed"); break; }
            if (e == 0) break;
            hipLaunchKernelGGL(k_bfs_expand, dim3((unsigned)(((int64_t)m * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, m,
                               (const int32_t *)fr[cur], (const int32_t *)pt.rowptr, (const int32_t *)pt.col,
                               (const int32_t *)off, (const int32_t *)p, cand, first);
            hipLaunchKernelGGL(k_bfs_flags, dim3((e + kBlock - 1) / kBlock), dim3(kBlock), 0, st, e, (const int32_t *)cand,
                               (const int32_t *)p, (const int32_t *)first, flag);
            (void)hipcub::DeviceSelect::Flagged(tmp, tb2, cand, flag, fr[cur ^ 1], cnt, e, st);
            int32_t m2 = 0;
            (void)hipMemcpyAsync(&m2, cnt, 4, hipMemcpyDeviceToHost, st);
            if (hipStreamSynchronize(st) != hipSuccess) { rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: level failed"); break; }
            cur ^= 1;
            m = m2;
            if (m > 0) {
                hipLaunchKernelGGL(k_bfs_number, dim3((m + kBlock - 1) / kBlock), dim3(kBlock), 0, st, m,
                                   (const int32_t *)fr[cur], base, p);
                base += m;
            }
            // A level costs two host round trips (~65 us).  Graphs of large diameter (grids: thousands
            // of levels of a few thousand vertices) are faster in the sequential queue loop: when the
            // frontiers stay small, the search continues on the host from the current frontier, which
            // IS the queue at this point (same order, same numbering).
            ++levels;
            if (m > 0 && levels >= 16 && (int64_t)base < (int64_t)levels * 4096) { on_host = true; break; }
        }
        // (the last level's numbering kernel may still be in flight on the library's NON-BLOCKING stream when the loop is left for
        //  the host continuation: a blocking copy on the null stream does not wait for it.  Found in round 6 by three rank
        //  processes sharing a GPU -- a quiet GPU finishes the kernel before the copy starts --: the visiting numbers came back one
        //  level short and the host queue then numbered that level after the next one)
        if (rc == SGM_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: the last level failed");
        if (rc == SGM_OK && (hipMemcpy(p_out, p, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipGetLastError() != hipSuccess))
            rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: copy back failed");
        if (rc == SGM_OK && on_host) {
            std::vector<int32_t> ptr((size_t)n + 1), node((size_t)pt.nnz), queue((size_t)n);
            bool ok = hipMemcpy(ptr.data(), pt.rowptr, ptr.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
            if (pt.nnz) ok = ok && hipMemcpy(node.data(), pt.col, node.size() * 4, hipMemcpyDeviceToHost) == hipSuccess;
            ok = ok && hipMemcpy(queue.data(), fr[cur], (size_t)m * 4, hipMemcpyDeviceToHost) == hipSuccess;
            if (!ok) rc = fail(SGM_ERR_HIP, "sgm_graph_bfs_order: hand-over to the host failed");
            // the frontier's vertices are numbered already; they sit in the queue waiting to be expanded
            int32_t head = 0, tail = m, num = base;
            while (rc == SGM_OK && tail > head) {                       // permutations.f90:44-72
                const int32_t i = queue[head++];
                if (p_out[i] <= 0) p_out[i] = ++num;
                for (int32_t k = ptr[i]; k < ptr[i + 1]; ++k) {
                    const int32_t j = node[k];
                    if (p_out[j] == -1) { queue[tail++] = j; p_out[j] = 0; }
                }
            }
        }
    }
    if (tmp) (void)hipFree(tmp);
    dfree(p); dfree(first); dfree(fr[0]); dfree(fr[1]); dfree(deg); dfree(off); dfree(cand); dfree(flag); dfree(cnt);
    return rc;
}

// device path of the two colouring entry points: fills the host array from the device result; *done = false: take the host pass
static int try_device_coloring(sgm_mat A, const char *who, bool ordering, int32_t *out_host, int32_t *n1, int32_t *nc, bool *done)
{
    *done = false;
    if (!A) return fail(SGM_ERR_BAD_ARG, "%s: null matrix", who);
    if (A->fmt != SGM_FMT_CSR || A->distributed() || A->nrow != A->ncol) return SGM_OK;      // (the host pass reports it)
    const Part &pt = A->parts[0];
    if (pt.n < 1) return SGM_OK;
    int32_t *d = nullptr;
    SGM_TRY(dalloc(&d, (size_t)pt.n));
    bool ok = false;
    int rc = greedy_coloring_device(pt, ordering ? nullptr : d, ordering ? d : nullptr, n1, nc, &ok);
    if (rc == SGM_OK && ok && hipMemcpy(out_host, d, (size_t)pt.n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(SGM_ERR_HIP, "%s: copy back failed", who);
    dfree(d);
    *done = rc == SGM_OK && ok;
    return rc;
}

int sgm_graph_greedy_coloring(sgm_mat A, int32_t *colors_out, int32_t *num_colors)
{
    SGM_TRY(require_init());
    if (!colors_out) return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_coloring: null output");
    if (A && A->fmt == SGM_FMT_CSR && A->comm) {     // over ranks: the whole graph on every rank, the same colours everywhere
        sgm_mat G = nullptr;
        SGM_TRY(gathered_graph(A, &G));
        const int rc = sgm_graph_greedy_coloring(G, colors_out, num_colors);
        sgm_mat_destroy(G);
        return rc;
    }
    {   // graphs whose colouring is forced (bipartite, symmetric, connected from vertex 1): on the device
        bool done = false;
        int32_t n1 = 0, nc = 0;
        SGM_TRY(try_device_coloring(A, "sgm_graph_greedy_coloring", false, colors_out, &n1, &nc, &done));
        if (done) { if (num_colors) *num_colors = nc; return SGM_OK; }
    }
    std::vector<int32_t> ptr, node;
    SGM_TRY(host_graph(A, "sgm_graph_greedy_coloring", ptr, node));
    const int32_t used = greedy_coloring_host(A->nrow, ptr, node, colors_out);
    if (num_colors) *num_colors = used;
    return SGM_OK;
}

int sgm_graph_greedy_color_order(sgm_mat A, int32_t *p_out, int32_t *ptrs_out, int32_t ptrs_len, int32_t *num_colors)
{
    SGM_TRY(require_init());
    if (!p_out || !num_colors) return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_color_order: null output");
    if (A && A->fmt == SGM_FMT_CSR && A->comm) {
        sgm_mat G = nullptr;
        SGM_TRY(gathered_graph(A, &G));
        const int rc = sgm_graph_greedy_color_order(G, p_out, ptrs_out, ptrs_len, num_colors);
        sgm_mat_destroy(G);
        return rc;
    }
    {
        bool done = false;
        int32_t n1 = 0, nc = 0;
        SGM_TRY(try_device_coloring(A, "sgm_graph_greedy_color_order", true, p_out, &n1, &nc, &done));
        if (done) {
            if (ptrs_out && ptrs_len < nc + 1) return fail(SGM_ERR_BAD_ARG, "sgm_graph_greedy_color_order: ptrs needs %d entries", nc + 1);
            if (ptrs_out) { ptrs_out[0] = 1; ptrs_out[1] = n1 + 1; if (nc == 2) ptrs_out[2] = A->nrow + 1; }
            *num_colors = nc;
            return SGM_OK;
        }
    }
    std::vector<int32_t> ptr, node;
    SGM_TRY(host_graph(A, "sgm_graph_greedy_color_order", ptr, node));
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
    A->version += 1;
    A->pattern_version += 1;
    if (A->fmt == SGM_FMT_CSR && A->comm) {          // over ranks: p is the GLOBAL permutation, the same on every rank (collective)
        const int64_t ng = A->nrow;
        std::vector<int32_t> hp((size_t)std::max<int64_t>(ng, 1));
        if (where == SGM_DEVICE) SGM_HIP(hipMemcpy(hp.data(), p, (size_t)ng * 4, hipMemcpyDeviceToHost));
        else std::copy(p, p + ng, hp.begin());
        return permute_dist(A, hp.data(), true);
    }
    if ((A->fmt != SGM_FMT_CSR && A->fmt != SGM_FMT_ELL) || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_left_permute: CSR / ELLPACK matrices on one GPU, CSR matrices distributed over ranks");
    Part &pt = A->parts[0];
    const int32_t n = pt.n;
    hipStream_t st = g_rt.stream;
    int32_t *dp = nullptr;
    SGM_TRY(stage_perm("sgm_mat_left_permute", n, p, where, &dp));
    if (A->fmt == SGM_FMT_ELL) {         // ellpack_matrices.f90:601-619
        const size_t total = (size_t)n * pt.max_d;
        int32_t *ecol2 = nullptr, *edeg2 = nullptr;
        double *eval2 = nullptr;
        int rc = dalloc(&ecol2, std::max<size_t>(total, 1));
        if (rc == SGM_OK) rc = dalloc(&eval2, std::max<size_t>(total, 1));
        if (rc == SGM_OK && pt.edeg) rc = dalloc(&edeg2, (size_t)std::max(n, 1));
        if (rc == SGM_OK && n) {
            hipLaunchKernelGGL(k_ell_perm_rows, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, pt.max_d,
                               (const int32_t *)dp, (const int32_t *)pt.ecol, (const double *)pt.eval,
                               (const int32_t *)pt.edeg, ecol2, eval2, edeg2);
            if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess)
                rc = fail(SGM_ERR_HIP, "sgm_mat_left_permute: kernel failed");
        }
        dfree(dp);
        if (rc != SGM_OK) { dfree(ecol2); dfree(eval2); dfree(edeg2); return rc; }
        dfree(pt.ecol); dfree(pt.eval); dfree(pt.edeg);
        pt.ecol = ecol2; pt.eval = eval2; pt.edeg = edeg2;
        SGM_TRY(sgm_invalidate_transpose(A));
        return rebuild_ell_formats(pt);
    }
    if (int rcn = csr_need_arrays(pt)) { dfree(dp); return rcn; }           // (a part that kept only its sliced form: the CSR-order arrays come back first)
    int32_t *len2 = nullptr, *rowptr2 = nullptr, *col2 = nullptr;
    double *val2 = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    int rc = dalloc(&len2, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&rowptr2, (size_t)n + 1);
    if (rc == SGM_OK) rc = dalloc(&col2, (size_t)pt.nnz + 4);
    if (rc == SGM_OK) rc = dalloc(&val2, (size_t)pt.nnz + 2);
    if (rc == SGM_OK) {
        (void)hipMemsetAsync(len2, 0, ((size_t)n + 1) * 4, st);
        (void)hipMemsetAsync(col2 + pt.nnz, 0, 16, st);
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
    A->version += 1;
    A->pattern_version += 1;
    if (A->fmt == SGM_FMT_CSR && A->comm) {          // over ranks: p is the GLOBAL permutation, the same on every rank (collective)
        const int64_t ng = A->ncol;
        std::vector<int32_t> hp((size_t)std::max<int64_t>(ng, 1));
        if (where == SGM_DEVICE) SGM_HIP(hipMemcpy(hp.data(), p, (size_t)ng * 4, hipMemcpyDeviceToHost));
        else std::copy(p, p + ng, hp.begin());
        return permute_dist(A, hp.data(), false);
    }
    if ((A->fmt != SGM_FMT_CSR && A->fmt != SGM_FMT_ELL) || A->distributed())
        return fail(SGM_ERR_UNSUPPORTED, "sgm_mat_right_permute: CSR / ELLPACK matrices on one GPU, CSR matrices distributed over ranks");
    Part &pt = A->parts[0];
    int32_t *dp = nullptr;
    SGM_TRY(stage_perm("sgm_mat_right_permute", A->ncol, p, where, &dp));
    if (A->fmt == SGM_FMT_ELL) {         // ellpack_graphs.f90:523-541
        const int64_t total = (int64_t)pt.n * pt.max_d;
        if (total) hipLaunchKernelGGL(k_ell_perm_cols, dim3(vec_grid(total)), dim3(kBlock), 0, g_rt.stream, total, pt.ecol, (const int32_t *)dp);
        const bool ok = hipStreamSynchronize(g_rt.stream) == hipSuccess && hipGetLastError() == hipSuccess;
        dfree(dp);
        if (!ok) return fail(SGM_ERR_HIP, "sgm_mat_right_permute: kernel failed");
        SGM_TRY(sgm_invalidate_transpose(A));
        return rebuild_ell_formats(pt);
    }
    if (int rcn = csr_need_arrays(pt)) { dfree(dp); return rcn; }
    if (pt.nnz) hipLaunchKernelGGL(k_perm_cols, dim3(vec_grid(pt.nnz)), dim3(kBlock), 0, g_rt.stream, pt.nnz, pt.col, (const int32_t *)dp);
    const bool ok = hipStreamSynchronize(g_rt.stream) == hipSuccess && hipGetLastError() == hipSuccess;
    dfree(dp);
    if (!ok) return fail(SGM_ERR_HIP, "sgm_mat_right_permute: kernel failed");
    SGM_TRY(sgm_invalidate_transpose(A));
    return rebuild_csr_formats(pt);
}

}  // extern "C"
