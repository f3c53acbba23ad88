// Graph -> matrix build on the device: the index work the reference does
// on the host in  ll_graph%add_edge (src/graph/formats/ll_graphs.f90:355-370: a repeated
// edge is ignored) -> cs_graph_build (src/graph/formats/cs_graphs.f90:109-197: count,
// prefix sum with ptr(1)=1, first-free-slot fill in cursor order) / ellpack_graph_build
// (src/graph/formats/ellpack_graphs.f90:105-170: padding slots repeat the last neighbour)
// -> A%set_value (cs_matrices.f90:840-863, ellpack_matrices.f90:444-466: scan the row, the
// LAST value written to an entry wins).  12.7 s on one CPU core at n = 1e7.
//
// Bit-exact by construction: a STABLE radix sort of the edge list by row keeps the insertion
// order inside every row (= the order the ll_graph cursor hands the edges over); an edge is
// kept iff no earlier edge of its row has the same column; its value is that of the last
// duplicate.  hipCUB's device radix sort / scan do the two library-shaped steps (setup code,
// not the hot path); the row-local passes are plain kernels.
#include "sgm_internal.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace sgm {

__global__ void k_edge_keys(int64_t ne, const int32_t *__restrict__ ei, int32_t *__restrict__ key,
                            int32_t *__restrict__ idx, int32_t *__restrict__ rowcnt, int32_t nrow, int *bad)
{
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < ne; e += stride) {
        const int32_t r = ei[e] - 1;
        if (r < 0 || r >= nrow) { *bad = 1; key[e] = 0; idx[e] = (int32_t)e; continue; }
        key[e] = r;
        idx[e] = (int32_t)e;
        atomicAdd(&rowcnt[r], 1);
    }
}

// one lane per sorted edge: keep = first occurrence of its column inside the row
__global__ void k_mark_first(int64_t ne, const int32_t *__restrict__ skey, const int32_t *__restrict__ perm,
                             const int32_t *__restrict__ rowstart, const int32_t *__restrict__ ej,
                             int32_t *__restrict__ keep, int32_t ncol, int *bad)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < ne; p += stride) {
        const int32_t r = skey[p];
        const int32_t j = ej[perm[p]];
        if (j < 1 || j > ncol) *bad = 1;
        int32_t k = 1;
        for (int64_t q = rowstart[r]; q < p; ++q)
            if (ej[perm[q]] == j) { k = 0; break; }
        keep[p] = k;
    }
}

// kept edge -> its slot; value = the last duplicate's value (set_value: last write wins)
__global__ void k_fill_csr(int64_t ne, const int32_t *__restrict__ skey, const int32_t *__restrict__ perm,
                           const int32_t *__restrict__ rowstart, const int32_t *__restrict__ keep,
                           const int32_t *__restrict__ pos, const int32_t *__restrict__ ej,
                           const double *__restrict__ ev, int32_t *__restrict__ node, double *__restrict__ val)
{
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; p < ne; p += stride) {
        if (!keep[p]) continue;
        const int32_t r = skey[p];
        const int32_t j = ej[perm[p]];
        double v = ev ? ev[perm[p]] : 0.0;
        for (int64_t q = p + 1; q < rowstart[r + 1]; ++q)
            if (ej[perm[q]] == j && ev) v = ev[perm[q]];
        node[pos[p]] = j;
        val[pos[p]] = v;
    }
}

__global__ void k_ptr_from_pos(int32_t nrow, int64_t ne, const int32_t *__restrict__ rowstart,
                               const int32_t *__restrict__ pos, int32_t total, int32_t *__restrict__ ptr)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrow) return;
    const int64_t s = r < nrow ? rowstart[r] : ne;
    ptr[r] = (s < ne ? pos[s] : total) + 1;            // 1-based, ptr(1) = 1
}

// ELLPACK from the CSR result: node(max_d, n) / val(max_d, n), Fortran order; padding slots
// repeat the last neighbour with value 0 (ellpack_graphs.f90:164)
__global__ void k_csr_to_ell(int32_t nrow, int32_t max_d, const int32_t *__restrict__ ptr,
                             const int32_t *__restrict__ node, const double *__restrict__ val,
                             int32_t *__restrict__ enode, double *__restrict__ eval, int32_t *__restrict__ deg)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrow) return;
    const int32_t b = ptr[r] - 1, d = ptr[r + 1] - ptr[r];
    deg[r] = d;
    for (int32_t k = 0; k < max_d; ++k) {
        enode[(int64_t)r * max_d + k] = k < d ? node[b + k] : (d ? node[b + d - 1] : 0);
        eval[(int64_t)r * max_d + k] = k < d ? val[b + k] : 0.0;
    }
}

__global__ void k_max_deg(int32_t nrow, const int32_t *__restrict__ ptr, int32_t *out)
{
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    int32_t d = r < nrow ? ptr[r + 1] - ptr[r] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, d);          // one atomic per wave, not per row
}

// device CSR arrays (1-based, exactly what cs_graph_build + set_value produce) from an edge list
static int build_arrays(int32_t nrow, int32_t ncol, int64_t ne, const int32_t *ei_in, const int32_t *ej_in,
                        const double *ev_in, int where, int32_t **ptr_out, int32_t **node_out, double **val_out,
                        int32_t *nnz_out)
{
    hipStream_t st = g_rt.stream;
    if (ne > INT32_MAX - 8) return fail(SGM_ERR_UNSUPPORTED, "edge list longer than int32");
    int32_t *ei = nullptr, *ej = nullptr;
    double *ev = nullptr;
    const size_t nE = (size_t)std::max<int64_t>(ne, 1);
    const bool host = where == SGM_HOST;
    if (host) {
        SGM_TRY(dalloc(&ei, nE)); SGM_TRY(dalloc(&ej, nE)); SGM_TRY(dalloc(&ev, nE));
        if (ne) {
            SGM_HIP(hipMemcpyAsync(ei, ei_in, (size_t)ne * 4, hipMemcpyHostToDevice, st));
            SGM_HIP(hipMemcpyAsync(ej, ej_in, (size_t)ne * 4, hipMemcpyHostToDevice, st));
            SGM_HIP(hipMemcpyAsync(ev, ev_in, (size_t)ne * 8, hipMemcpyHostToDevice, st));
        }
    } else {
        ei = const_cast<int32_t *>(ei_in); ej = const_cast<int32_t *>(ej_in); ev = const_cast<double *>(ev_in);
    }
    int32_t *key = nullptr, *idx = nullptr, *skey = nullptr, *perm = nullptr, *rowcnt = nullptr, *rowstart = nullptr;
    int32_t *keep = nullptr, *pos = nullptr;
    int *bad = nullptr;
    SGM_TRY(dalloc(&key, nE)); SGM_TRY(dalloc(&idx, nE)); SGM_TRY(dalloc(&skey, nE)); SGM_TRY(dalloc(&perm, nE));
    SGM_TRY(dalloc(&rowcnt, (size_t)nrow + 1)); SGM_TRY(dalloc(&rowstart, (size_t)nrow + 1));
    SGM_TRY(dalloc(&keep, nE + 1)); SGM_TRY(dalloc(&pos, nE + 1)); SGM_TRY(dalloc(&bad, 1));
    SGM_HIP(hipMemsetAsync(rowcnt, 0, ((size_t)nrow + 1) * 4, st));
    SGM_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
    SGM_HIP(hipMemsetAsync(keep, 0, (nE + 1) * 4, st));
    const int g = vec_grid(std::max<int64_t>(ne, 1));
    if (ne) hipLaunchKernelGGL(k_edge_keys, dim3(g), dim3(kBlock), 0, st, ne, (const int32_t *)ei, key, idx, rowcnt, nrow, bad);
    // stable sort by row: insertion order survives inside each row
    size_t tb1 = 0, tb2 = 0, tb3 = 0;
    int bits = 1;
    while ((1ll << bits) < nrow) ++bits;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tb1, key, skey, idx, perm, (int)ne, 0, bits, st);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, rowcnt, rowstart, nrow + 1, st);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb3, keep, pos, (int)ne + 1, st);
    char *tmp = nullptr;
    SGM_TRY(dalloc(&tmp, std::max(tb1, std::max(tb2, tb3)) + 256));
    if (ne) SGM_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb1, key, skey, idx, perm, (int)ne, 0, bits, st));
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb2, rowcnt, rowstart, nrow + 1, st));
    if (ne) hipLaunchKernelGGL(k_mark_first, dim3(g), dim3(kBlock), 0, st, ne, (const int32_t *)skey, (const int32_t *)perm,
                               (const int32_t *)rowstart, (const int32_t *)ej, keep, ncol, bad);
    SGM_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb3, keep, pos, (int)ne + 1, st));
    int32_t total = 0;
    int hbad = 0;
    SGM_HIP(hipMemcpyAsync(&total, pos + ne, 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    int rc = SGM_OK;
    if (hbad) rc = fail(SGM_ERR_BAD_ARG, "edge list holds a vertex outside 1..nrow / 1..ncol");
    int32_t *ptr = nullptr, *node = nullptr;
    double *val = nullptr;
    if (rc == SGM_OK) rc = dalloc(&ptr, (size_t)nrow + 1);
    if (rc == SGM_OK) rc = dalloc(&node, (size_t)std::max(total, 1));
    if (rc == SGM_OK) rc = dalloc(&val, (size_t)std::max(total, 1));
    if (rc == SGM_OK) {
        if (ne) hipLaunchKernelGGL(k_fill_csr, dim3(g), dim3(kBlock), 0, st, ne, (const int32_t *)skey, (const int32_t *)perm,
                                   (const int32_t *)rowstart, (const int32_t *)keep, (const int32_t *)pos,
                                   (const int32_t *)ej, (const double *)ev, node, val);
        hipLaunchKernelGGL(k_ptr_from_pos, dim3((nrow + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nrow, ne,
                           (const int32_t *)rowstart, (const int32_t *)pos, total, ptr);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            rc = fail(SGM_ERR_HIP, "graph build kernels failed");
    }
    dfree(key); dfree(idx); dfree(skey); dfree(perm); dfree(rowcnt); dfree(rowstart); dfree(keep); dfree(pos);
    dfree(bad); dfree(tmp);
    if (host) { dfree(ei); dfree(ej); dfree(ev); }
    if (rc != SGM_OK) { dfree(ptr); dfree(node); dfree(val); return rc; }
    *ptr_out = ptr; *node_out = node; *val_out = val; *nnz_out = total;
    return SGM_OK;
}

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_csr_from_edges(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t ne, const int32_t *ei, const int32_t *ej,
                       const double *ev, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || ne < 0 || (ne && (!ei || !ej || !ev)))
        return fail(SGM_ERR_BAD_ARG, "sgm_csr_from_edges: bad argument");
    int32_t *ptr = nullptr, *node = nullptr, nnz = 0;
    double *val = nullptr;
    SGM_TRY(build_arrays(nrow, ncol, ne, ei, ej, ev, where, &ptr, &node, &val, &nnz));
    const int rc = sgm_csr_create(out, nrow, ncol, nnz, ptr, node, val, SGM_DEVICE);
    dfree(ptr); dfree(node); dfree(val);
    return rc;
}

int sgm_ell_from_edges(sgm_mat *out, int32_t nrow, int32_t ncol, int64_t ne, const int32_t *ei, const int32_t *ej,
                       const double *ev, int where)
{
    SGM_TRY(require_init());
    if (!out || nrow < 0 || ncol < 0 || ne < 0 || (ne && (!ei || !ej || !ev)))
        return fail(SGM_ERR_BAD_ARG, "sgm_ell_from_edges: bad argument");
    int32_t *ptr = nullptr, *node = nullptr, nnz = 0;
    double *val = nullptr;
    SGM_TRY(build_arrays(nrow, ncol, ne, ei, ej, ev, where, &ptr, &node, &val, &nnz));
    hipStream_t st = g_rt.stream;
    int32_t *dmax = nullptr, max_d = 0;
    int rc = dalloc(&dmax, 1);
    if (rc == SGM_OK) {
        (void)hipMemsetAsync(dmax, 0, 4, st);
        if (nrow) hipLaunchKernelGGL(k_max_deg, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nrow, (const int32_t *)ptr, dmax);
        (void)hipMemcpyAsync(&max_d, dmax, 4, hipMemcpyDeviceToHost, st);
        if (hipStreamSynchronize(st) != hipSuccess) rc = fail(SGM_ERR_HIP, "k_max_deg failed");
    }
    int32_t *enode = nullptr, *deg = nullptr;
    double *eval = nullptr;
    const size_t total = (size_t)nrow * max_d;
    if (rc == SGM_OK) rc = dalloc(&enode, total);
    if (rc == SGM_OK) rc = dalloc(&eval, total);
    if (rc == SGM_OK) rc = dalloc(&deg, (size_t)nrow);
    if (rc == SGM_OK && nrow)
        hipLaunchKernelGGL(k_csr_to_ell, dim3((nrow + kBlock - 1) / kBlock), dim3(kBlock), 0, st, nrow, max_d,
                           (const int32_t *)ptr, (const int32_t *)node, (const double *)val, enode, eval, deg);
    if (rc == SGM_OK) rc = sgm_ell_create(out, nrow, ncol, max_d, enode, eval, SGM_DEVICE);
    if (rc == SGM_OK) {
        (*out)->parts[0].edeg = deg;           // kept for sgm_mat_get("degrees")
        deg = nullptr;
    }
    dfree(ptr); dfree(node); dfree(val); dfree(dmax); dfree(enode); dfree(eval); dfree(deg);
    return rc;
}

}  // extern "C"
