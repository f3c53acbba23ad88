// The layouts a CSR / ELLPACK part gets at create: the sliced forms and their offset dictionaries, SELL with
// the slices' windows of x, the lean residency, built on the device; rebuilt after set_values / permutations.
#include "sgm_spmv_select.hpp"

namespace sgm {

// ---- kernels of the upload checks and of the layout builders ----------------------------------------------------------
__global__ void k_dec1(int32_t *a, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] -= 1;
}

// Validation of the index arrays a caller hands to sgm_csr_create (the reference prints and exits on a bad
// index only where it happens to look, sparse_matrix_interfaces.f90:663-687; a wild `node` here would be a
// GPU memory fault inside the product).  bad[0] = first row i (0-based) whose pointers are malformed --
// ptr(1) /= 1, ptr(i+1) < ptr(i), ptr(n+1) - 1 /= nnz (reported as row n) --, bad[1] = first entry k (0-based) whose
// 1-based column lies outside 1..ncols.  Both start at INT64_MAX; the create reads them at the synchronisation
// it makes anyway.  The pointer pass runs on the 1-based upload BEFORE k_dec1 (it reads a neighbour); the
// column pass is the decrement itself.
__global__ void k_check_ptr1(const int32_t *__restrict__ ptr1, int64_t n, int64_t nnz, unsigned long long *bad)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i <= n; i += stride) {
        const int32_t a = ptr1[i];
        bool ok = true;
        if (i == 0) ok = a == 1;
        if (i < n) ok = ok && ptr1[i + 1] >= a;
        else ok = ok && (int64_t)a - 1 == nnz;
        if (!ok) atomicMin(bad, (unsigned long long)i);
    }
}

__global__ void k_dec1_check_cols(int32_t *a, int64_t nnz, int64_t ncols, unsigned long long *bad)
{
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nnz; k += stride) {
        const int32_t c = a[k];
        if (c < 1 || c > ncols) atomicMin(bad + 1, (unsigned long long)k);
        a[k] = c - 1;
    }
}

__global__ void k_ell_transpose(const int32_t *__restrict__ node, const double *__restrict__ val,
                                int32_t *__restrict__ ecol, double *__restrict__ eval,
                                int32_t n, int32_t max_d, int32_t ncol, unsigned long long *bad)
{
    // in: (max_d, n) column-major = row i contiguous; out: slot-major [k*n + i]
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)n * max_d;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < total; t += stride) {
        const int32_t k = (int32_t)(t / n), i = (int32_t)(t % n);
        // an empty row keeps node = 0 in the reference (it then reads x(0): README.md:71-73);
        // clamp so that the padding product 0.0 * x(1) stays inside the vector
        if (node) {
            const int32_t c = node[(int64_t)i * max_d + k];
            // (validation of sgm_ell_create's input: 0 is the reference's empty-row marker, anything else must be a column)
            if (bad && (c < 0 || c > ncol)) atomicMin(bad, (unsigned long long)((int64_t)i * max_d + k));
            ecol[t] = max(c - 1, 0);
        }
        if (val) eval[t] = val[(int64_t)i * max_d + k];
    }
}

// setup: the window of columns every 512-row slice gathers from (its first column rounded down to even, and the span to the
// last one); the longest span of the part by atomicMax
__global__ __launch_bounds__(256) void k_sell_window(int64_t nsl, int gs, const int64_t *__restrict__ off, const int32_t *__restrict__ scol,
                                                     int32_t *__restrict__ win0, int32_t *__restrict__ max_span)
{
    __shared__ int32_t lo_s[4], hi_s[4];
    const int64_t ngr = (nsl + gs - 1) / gs;
    for (int64_t sl = blockIdx.x; sl < ngr; sl += gridDim.x) {            // (sl: group of gs slices)
        const int64_t a = off[sl * gs * (kSlRows / kSellChunk)], b = off[min((sl + 1) * gs, nsl) * (kSlRows / kSellChunk)];
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t k = a + threadIdx.x; k < b; k += 256) {
            const int32_t c = scol[k];
            if (c >= 0) { lo = min(lo, c); hi = max(hi, c); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { lo_s[threadIdx.x >> 6] = lo; hi_s[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int t = 1; t < 4; ++t) { lo_s[0] = min(lo_s[0], lo_s[t]); hi_s[0] = max(hi_s[0], hi_s[t]); }
            const int32_t l = hi_s[0] < 0 ? 0 : lo_s[0] & ~1;
            win0[sl] = l;
            if (hi_s[0] >= 0) atomicMax(max_span, hi_s[0] - l + 1);
        }
    }
}

// setup: positions of a window's rows sorted by length (longest first, ties by row: the sort is a pure function of the row
// lengths), the chunks' widths (as entry counts, to be prefix-summed), ...
__global__ __launch_bounds__(256) void k_sell_sort(int32_t n, const int32_t *__restrict__ rowptr, uint16_t *__restrict__ perm,
                                                   int64_t *__restrict__ wid)
{
    __shared__ int32_t len[kSellSigma];
    __shared__ int32_t first_len[kSellSigma / kSellChunk];
    const int64_t nwin = ((int64_t)n + kSellSigma - 1) / kSellSigma;
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);       // chunks that exist (whole slices)
    for (int64_t win = blockIdx.x; win < nwin; win += gridDim.x) {
        __syncthreads();
        for (int r = threadIdx.x; r < kSellSigma; r += blockDim.x) {
            const int64_t row = win * kSellSigma + r;
            len[r] = row < n ? rowptr[row + 1] - rowptr[row] : -1;        // (rows past the end sort last)
        }
        if (threadIdx.x < kSellSigma / kSellChunk) first_len[threadIdx.x] = 0;
        __syncthreads();
        for (int r = threadIdx.x; r < kSellSigma; r += blockDim.x) {
            const int32_t l = len[r];
            int rank = 0;
            for (int j = 0; j < kSellSigma; ++j) rank += (len[j] > l || (len[j] == l && j < r)) ? 1 : 0;
            const int64_t pos = win * kSellSigma + rank;
            if (pos / kSellChunk < nch) perm[pos] = l >= 0 ? (uint16_t)r : (uint16_t)0xffffu;
            if (rank % kSellChunk == 0) first_len[rank / kSellChunk] = l > 0 ? l : 0;
        }
        __syncthreads();
        if (threadIdx.x < kSellSigma / kSellChunk) {
            const int64_t c = win * (kSellSigma / kSellChunk) + threadIdx.x;
            if (c < nch) wid[c] = (int64_t)((first_len[threadIdx.x] + 1) / 2 * 2) * kSellChunk;
        }
    }
}

// ... and the chunks' slots filled from the CSR arrays (val only: a value update)
__global__ __launch_bounds__(256) void k_sell_fill(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const int64_t *__restrict__ off,
                                                   const uint16_t *__restrict__ perm, int32_t *__restrict__ scol, double *__restrict__ sval)
{
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nch; chunk += (int64_t)gridDim.x * 4) {
        const int64_t o0 = off[chunk];
        const int32_t W = (int32_t)((off[chunk + 1] - o0) / kSellChunk);
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            const uint16_t pr = perm[chunk * kSellChunk + q];
            int32_t k = 0, ke = 0;
            if (pr != 0xffffu) {
                const int64_t row = (chunk / (kSellSigma / kSellChunk)) * kSellSigma + pr;
                k = rowptr[row]; ke = rowptr[row + 1];
            }
            for (int32_t u = 0; u < W; ++u) {
                const bool has = k + u < ke;
                sval[o0 + (int64_t)u * kSellChunk + q] = has ? val[k + u] : 0.0;
                if (scol) scol[o0 + (int64_t)u * kSellChunk + q] = has ? col[k + u] : -1;
            }
        }
    }
}

// 1-byte codes in CSR order -> the chunked sliced layout of k_csr_slb (255 where a row has no entry in the slot)
__global__ __launch_bounds__(256) void k_slb_pack_codes(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                        const uint8_t *__restrict__ code, uint8_t *__restrict__ sbcode)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            const int WC = (W + 7) / 8 * 8;          // code bytes per row
            uint8_t *dst = sbcode + sl * (int64_t)WC * kSlRows + (int64_t)r * 8;
            for (int u = 0; u < WC; ++u) dst[(int64_t)(u >> 3) * kSlRows * 8 + (u & 7)] = (u < W && k + u < ke) ? code[k + u] : (uint8_t)255;
        }
}

// columns in CSR order -> sliced layout (-1 where a row has no entry in the slot; whole slices)
__global__ __launch_bounds__(256) void k_sl_pack_cols(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col, int32_t *__restrict__ scol)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            int32_t *dst = scol + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = k + u < ke ? col[k + u] : -1;
        }
}

// values in CSR order -> sliced layout (at upload and after every value update)
__global__ __launch_bounds__(256) void k_sl_pack(int32_t n, int32_t W, const int32_t *__restrict__ rowptr,
                                                 const double *__restrict__ val, double *__restrict__ sval)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            int32_t k = 0, ke = 0;
            if (row < n) { k = rowptr[row]; ke = rowptr[row + 1]; }
            double *dst = sval + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = k + u < ke ? val[k + u] : 0.0;
        }
}

// the same for an ELLPACK matrix (slot-major eval, stride n): all max_d slots of a row are entries
__global__ __launch_bounds__(256) void k_sl_pack_ell(int32_t n, int32_t W, int32_t max_d, const double *__restrict__ eval,
                                                     double *__restrict__ sval)
{
    const int64_t nsl = ((int64_t)n + kSlRows - 1) / kSlRows;
    for (int64_t sl = blockIdx.x; sl < nsl; sl += gridDim.x)
        for (int r = threadIdx.x; r < kSlRows; r += blockDim.x) {
            const int64_t row = sl * kSlRows + r;
            double *dst = sval + sl * (int64_t)W * kSlRows + r;
            for (int u = 0; u < W; ++u) dst[(int64_t)u * kSlRows] = (row < n && u < max_d) ? eval[(int64_t)u * n + row] : 0.0;
        }
}

// ---- "csr_lean": the sliced form as the ONLY resident layout -----------------------------------------------------
// C2 kept 1.13 GB resident for a kernel that reads 0.44 GB of it: CSR-order values (400 MB), int32 columns (200), 1-byte
// codes (50) beside the sliced values + code words.  Slot u of a row in the sliced form IS the row's u-th stored entry
// (k_dict_encode / k_sl_pack), so the three arrays are a pure function of (rowptr, scode, dict, sval): they are released
// once the sliced form stands and rebuilt by k_sl_unpack for whoever reads them.
__global__ __launch_bounds__(256) void k_sl_unpack(int32_t n, int32_t W, const int32_t *__restrict__ rowptr, const uint32_t *__restrict__ scode,
                                                   const int32_t *__restrict__ dict, const double *__restrict__ sval,
                                                   int32_t *__restrict__ col, double *__restrict__ val, uint8_t *__restrict__ code)
{
    __shared__ int32_t dl[16];
    if (threadIdx.x < 16) dl[threadIdx.x] = dict[threadIdx.x];
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t cw = scode[i];
        int32_t k = rowptr[i];
        const double *src = sval + ((int64_t)(i >> 9) * W) * kSlRows + (i & (kSlRows - 1));
        for (int u = 0; u < W; ++u) {
            const uint32_t c = (cw >> (4 * u)) & 15u;
            if (c == 15u) break;
            if (col) col[k] = i + dl[c];
            if (val) val[k] = src[(int64_t)u * kSlRows];
            if (code) code[k] = (uint8_t)c;
            ++k;
        }
    }
}
// the same out of the SELL-128-512 form: position q of chunk c holds row perm(c, q); slot u is its u-th stored entry
__global__ __launch_bounds__(256) void k_sell_unpack(int32_t n, const int32_t *__restrict__ rowptr, const int64_t *__restrict__ off,
                                                     const uint16_t *__restrict__ perm, const int32_t *__restrict__ scol,
                                                     const double *__restrict__ sval, int32_t *__restrict__ col, double *__restrict__ val)
{
    const int64_t nch = (((int64_t)n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nch; chunk += (int64_t)gridDim.x * 4) {
        const int64_t o0 = off[chunk];
        for (int h = 0; h < 2; ++h) {
            const int q = lane + 64 * h;
            const uint16_t pr = perm[chunk * kSellChunk + q];
            if (pr == 0xffffu) continue;
            const int64_t row = (chunk / (kSellSigma / kSellChunk)) * kSellSigma + pr;
            const int32_t k = rowptr[row], len = rowptr[row + 1] - k;
            for (int32_t u = 0; u < len; ++u) {
                if (col) col[k + u] = scol[o0 + (int64_t)u * kSellChunk + q];
                if (val) val[k + u] = sval[o0 + (int64_t)u * kSellChunk + q];
            }
        }
    }
}
void csr_go_lean(Part &p)
{
    if (!lean_applies(p)) return;
    dfree(p.val); dfree(p.col); dfree(p.code);
    p.val = nullptr; p.col = nullptr; p.code = nullptr;
    p.lean = true;
}
int csr_need_arrays(const Part &cp)
{
    Part &p = const_cast<Part &>(cp);
    if (!p.lean || (p.val && p.col && (p.code || lean_sell(p)))) return SGM_OK;
    hipStream_t st = g_rt.stream;
    const int64_t nnz = p.nnz;
    const bool mk_val = !p.val, mk_col = !p.col, mk_code = !p.code;
    if (mk_col) { SGM_TRY(dalloc(&p.col, (size_t)nnz + 4)); SGM_HIP(hipMemsetAsync(p.col + nnz, 0, 4 * sizeof(int32_t), st)); }
    if (mk_val) { SGM_TRY(dalloc(&p.val, (size_t)nnz + 2)); SGM_HIP(hipMemsetAsync(p.val + nnz, 0, 2 * sizeof(double), st)); }
    if (lean_sell(p)) {           // (no dictionary: no byte codes to bring back)
        const int64_t nch = (((int64_t)p.n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
        if (p.n > 0 && (mk_col || mk_val))
            hipLaunchKernelGGL(k_sell_unpack, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, st, p.n,
                               (const int32_t *)p.rowptr, (const int64_t *)p.sl_off, (const uint16_t *)p.sl_perm, (const int32_t *)p.sl_col,
                               (const double *)p.sl_val, mk_col ? p.col : nullptr, mk_val ? p.val : nullptr);
        SGM_HIP(hipGetLastError());
        SGM_HIP(hipStreamSynchronize(st));     // (readers may use blocking copies, which do not order against this stream)
        return SGM_OK;
    }
    if (mk_code) { SGM_TRY(dalloc(&p.code, (size_t)nnz + 32)); SGM_HIP(hipMemsetAsync(p.code + nnz, 0, 32, st)); }
    if (p.n > 0)
        hipLaunchKernelGGL(k_sl_unpack, dim3((unsigned)std::min<int64_t>(((int64_t)p.n + 255) / 256, 4096)), dim3(256), 0, st, p.n, p.sw,
                           (const int32_t *)p.rowptr, (const uint32_t *)p.scode, (const int32_t *)p.dict, (const double *)p.sval,
                           mk_col ? p.col : nullptr, mk_val ? p.val : nullptr, mk_code ? p.code : nullptr);
    SGM_HIP(hipGetLastError());
    SGM_HIP(hipStreamSynchronize(st));         // (readers may use blocking copies, which do not order against this stream)
    return SGM_OK;
}
// a CSR-order value buffer to write new values into (they are then packed into the sliced form): allocated, not unpacked
int lean_val_buffer(Part &p)
{
    if (!p.lean || p.val) return SGM_OK;
    SGM_TRY(dalloc(&p.val, (size_t)p.nnz + 2));
    SGM_HIP(hipMemsetAsync(p.val + p.nnz, 0, 2 * sizeof(double), g_rt.stream));
    return SGM_OK;
}
void csr_release_arrays(const Part &cp)
{
    Part &p = const_cast<Part &>(cp);
    if (!p.lean || !lean_applies(p)) return;         // (option switched off meanwhile: what was rebuilt stays)
    (void)hipStreamSynchronize(g_rt.stream);         // whoever asked for them has queued its reads on the stream
    dfree(p.val); dfree(p.col); dfree(p.code);
    p.val = nullptr; p.col = nullptr; p.code = nullptr;
}

// refresh the sliced copy of the values (no-op for parts without one)
int pack_sliced(Part &p)
{
    if (p.sl_val && p.n > 0) {          // SELL-128-512: the values of every slot again (columns stay)
        const int64_t nch = (((int64_t)p.n + kSlRows - 1) / kSlRows) * (kSlRows / kSellChunk);
        hipLaunchKernelGGL(k_sell_fill, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, g_rt.stream, p.n,
                           (const int32_t *)p.rowptr, (const int32_t *)nullptr, (const double *)p.val, (const int64_t *)p.sl_off,
                           (const uint16_t *)p.sl_perm, (int32_t *)nullptr, p.sl_val);
        SGM_HIP(hipGetLastError());
    }
    if ((!p.scode && !p.scol && !p.sbcode) || p.n == 0) return SGM_OK;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows;
    if (p.ecol)
        hipLaunchKernelGGL(k_sl_pack_ell, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, g_rt.stream, p.n, p.sw,
                           p.max_d, (const double *)p.eval, p.sval);
    else
        hipLaunchKernelGGL(k_sl_pack, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, g_rt.stream, p.n, p.sw,
                           (const int32_t *)p.rowptr, (const double *)p.val, p.sval);
    SGM_HIP(hipGetLastError());
    return SGM_OK;
}

// Offset dictionary of a row block (host index work at setup): distinct (col - row) values in
// order of first appearance; gives up (p.code stays null) beyond 255 distinct offsets.
// ptr1/node1: optional 1-based host copies (otherwise the device arrays are read back).
__global__ void k_fill32(int64_t n, int32_t *a, int32_t v)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = v;
}

// ---- offset dictionary + sliced codes, built on the device ---------------------------------------
// pass 1: the set of distinct (col - row) offsets (open-addressing table of 1024 slots in global
// memory, atomicCAS insert; more than 255 live keys = overflow) and the longest row
constexpr int kDictSlots = 1024;
constexpr int32_t kDictEmpty = INT32_MIN;
__global__ __launch_bounds__(256) void k_dict_collect(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                      int32_t *table, int *count, int *max_row)
{
    // Distinct offsets are collected per workgroup in an LDS hash table first and only its (few) entries go to the
    // global table at the end: every thread inserting its first row's offsets straight into the global table was
    // 2.6 M same-address atomics at n = 1e7 (14 ms of a 24 ms create).
    constexpr int kLocal = 512;
    __shared__ int32_t ltab[kLocal];
    __shared__ int lcount;
    for (int t = threadIdx.x; t < kLocal; t += 256) ltab[t] = kDictEmpty;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    int mr = 0;
    int32_t mine[8];                        // the offsets this lane met last (stencil rows repeat them)
#pragma unroll
    for (int t = 0; t < 8; ++t) mine[t] = kDictEmpty;
    int next = 0;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t s0 = rowptr[i], e = rowptr[i + 1];
        mr = max(mr, e - s0);
        for (int32_t k = s0; k < e; ++k) {
            const int32_t off = col[k] - i;
            bool known = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) known = known || mine[t] == off;
            if (known) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t == next) mine[t] = off;
            next = (next + 1) & 7;
            if (*(volatile int *)&lcount > 255) break;                        // this workgroup alone overflows the dictionary
            uint32_t h = ((uint32_t)off * 2654435761u) >> 23;                // 9 bits
            for (int probe = 0; probe < kLocal; ++probe) {
                const int32_t prev = atomicCAS(&ltab[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(&lcount, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kLocal - 1);
            }
        }
    }
    __syncthreads();
    if (lcount > 255) {
        if (threadIdx.x == 0) atomicAdd(count, 256);                          // overflow: more than 255 distinct offsets
    } else {
        for (int t = threadIdx.x; t < kLocal; t += 256) {
            const int32_t off = ltab[t];
            if (off == kDictEmpty || *(volatile int *)count > 255) continue;
            uint32_t h = ((uint32_t)off * 2654435761u) >> 22;                // 10 bits
            for (int probe = 0; probe < kDictSlots; ++probe) {
                const int32_t prev = atomicCAS(&table[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(count, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kDictSlots - 1);
            }
        }
    }
    // one atomic per workgroup for the longest row
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mr = max(mr, __shfl_xor(mr, off, 64));
    __shared__ int wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mr;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(max_row, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}
// pass 2: 1-byte code of every entry (binary search in the sorted dictionary, held in LDS) and, when
// asked for, the row's word of 4-bit codes (15 = no entry)
__global__ __launch_bounds__(256) void k_dict_encode(int32_t n, const int32_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ dict, int ndict, uint8_t *__restrict__ code,
                                                     uint32_t *__restrict__ scode)
{
    __shared__ int32_t dl[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) dl[t] = t < ndict ? dict[t] : INT32_MAX;
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int32_t s0 = rowptr[i], e = rowptr[i + 1];
        uint32_t cw = 0xffffffffu;
        for (int32_t k = s0; k < e; ++k) {
            const int32_t off = col[k] - i;
            int lo = 0, hi = ndict - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (dl[mid] < off) lo = mid + 1; else hi = mid; }
            code[k] = (uint8_t)lo;
            if (scode && k - s0 < 8) cw = (cw & ~(15u << (4 * (k - s0)))) | ((uint32_t)lo << (4 * (k - s0)));
        }
        if (scode) scode[i] = cw;
    }
}

// The far offset most rows carry (a 3-D grid's plane stride, in rows), for the slice schedule: the largest |offset| that
// at least a quarter of the rows of a 512-row sample from the middle of the part use.  `codes` = the sample's dictionary
// codes (any order; 255 and codes >= ndict are ignored), `rows` = rows sampled.
static int32_t far_offset_of_sample(const std::vector<uint8_t> &codes, int64_t rows, const std::vector<int32_t> &dict, int ndict)
{
    std::vector<int64_t> freq(256, 0);
    for (uint8_t c : codes) ++freq[c];
    int64_t far = 0;
    for (int c = 0; c < ndict && c < 255; ++c)
        if (4 * freq[(size_t)c] >= rows) far = std::max<int64_t>(far, std::llabs((long long)dict[(size_t)c]));
    return (int32_t)std::min<int64_t>(far, INT32_MAX);
}
// SELL-128-512 of a part (see k_csr_sell): built for matrices without an offset dictionary that the uniform sliced form
// does not take, when sorting the rows of a slice keeps the padding below 30 % (a handful of very long rows among short
// ones -- an arrow matrix -- would blow their chunks up: those matrices stay with the CSR kernels)
static void free_sell(Part &p)
{
    dfree(p.sl_val); dfree(p.sl_col); dfree(p.sl_perm); dfree(p.sl_off); dfree(p.sl_win0);
    p.sl_val = nullptr; p.sl_col = nullptr; p.sl_perm = nullptr; p.sl_off = nullptr; p.sl_total = 0;
    p.sl_win0 = nullptr; p.sl_span = 0;
}
static int build_sell(Part &p)
{
    free_sell(p);
    if (!p.opt.csr_sliced || !p.opt.csr_sell || p.ecol || p.n < 1 || p.nnz < 4 * (int64_t)p.n || p.max_row < 1) return SGM_OK;
    // rows of up to 48 entries stay with the row-owner kernel UNLESS the slices' windows of x fit the LDS (decided below): its
    // tiles hold consecutive rows, whose x gathers share more L1 lines than a sorted chunk's (banded 20..40 entries per row:
    // 719-753 us against 793-819 here without the window -- and 553 with it; from 33..64 on SELL wins either way:
    // 620-650 against 757-794, 64..128: 649 against 792-816, 150..300: 750 against 922)
    // A few rows far longer than the rest: their chunk's lanes would walk max_row slots alone, every eight of them two
    // dependent round trips to memory with nothing to hide them (one row of 1000 entries among 250,000 of 8: 166 us against 33
    // without it; tools/probes/long_row_probe.py) -- the streaming kernel's owner lane adds a long row out of LDS at the pace of
    // the additions themselves.
    if (p.max_row > 256 && (int64_t)p.max_row * p.n > 16 * p.nnz && p.opt.csr_sell < 2) return SGM_OK;
    const bool short_rows = p.max_row <= 48 && p.opt.csr_sell < 2;
    if (short_rows && (!p.opt.csr_xwindow || p.n_halo != 0 || p.max_row < 8)) return SGM_OK;
    hipStream_t st = g_rt.stream;
    const int64_t nsl = ((int64_t)p.n + kSlRows - 1) / kSlRows, nch = nsl * (kSlRows / kSellChunk);
    SGM_TRY(dalloc(&p.sl_perm, (size_t)nsl * kSlRows));
    SGM_TRY(dalloc(&p.sl_off, (size_t)nch + 1));
    SGM_HIP(hipMemsetAsync(p.sl_off + nch, 0, sizeof(int64_t), st));
    const int64_t nwin = ((int64_t)p.n + kSellSigma - 1) / kSellSigma;
    hipLaunchKernelGGL(k_sell_sort, dim3((unsigned)std::min<int64_t>(nwin, 65536)), dim3(256), 0, st, p.n, (const int32_t *)p.rowptr,
                       p.sl_perm, p.sl_off);
    void *tmp = nullptr;
    size_t tb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, tb, p.sl_off, p.sl_off, (int)(nch + 1), st);
    if (hipMalloc(&tmp, std::max<size_t>(tb, 16)) != hipSuccess) { free_sell(p); return fail(SGM_ERR_ALLOC, "SELL build: scan workspace"); }
    (void)hipcub::DeviceScan::ExclusiveSum(tmp, tb, p.sl_off, p.sl_off, (int)(nch + 1), st);
    int64_t total = 0;
    const hipError_t e1 = hipMemcpyAsync(&total, p.sl_off + nch, sizeof(int64_t), hipMemcpyDeviceToHost, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e1 != hipSuccess || e2 != hipSuccess) { free_sell(p); return fail(SGM_ERR_HIP, "SELL build: scan failed"); }
    if (total <= 0 || (double)total > 1.30 * (double)p.nnz) { free_sell(p); return SGM_OK; }
    p.sl_total = total;
    int rc = dalloc(&p.sl_val, (size_t)total + 2);
    if (rc == SGM_OK) rc = dalloc(&p.sl_col, (size_t)total + 2);
    if (rc != SGM_OK) { free_sell(p); return rc; }
    hipLaunchKernelGGL(k_sell_fill, dim3((unsigned)std::min<int64_t>((nch + 3) / 4, 65536)), dim3(256), 0, st, p.n,
                       (const int32_t *)p.rowptr, (const int32_t *)p.col, (const double *)p.val, (const int64_t *)p.sl_off,
                       (const uint16_t *)p.sl_perm, p.sl_col, p.sl_val);
    SGM_HIP(hipGetLastError());
    // the windows of x the slices gather from: where every one of them fits the LDS the kernel stages it there (XW).  Parts
    // with halo columns are left out (their windows span the halo region, and their row ranges are cut by slices).
    p.sl_gs = 1;
    if (p.n_halo == 0) {
        int32_t *mx = nullptr;
        SGM_TRY(dalloc(&p.sl_win0, (size_t)nsl));
        SGM_TRY(dalloc(&mx, 1));
        struct Tmp { int32_t *&a; ~Tmp() { dfree(a); } } tmpmx{mx};
        constexpr size_t kLdsCap = (size_t)152 * 1024;
        auto windows = [&](int gs, int32_t *span_out) -> int {
            SGM_HIP(hipMemsetAsync(mx, 0, 4, st));
            hipLaunchKernelGGL(k_sell_window, dim3((unsigned)std::min<int64_t>(nsl, 65536)), dim3(256), 0, st, nsl, gs, (const int64_t *)p.sl_off,
                               (const int32_t *)p.sl_col, p.sl_win0, mx);
            int32_t span = 0;
            SGM_HIP(hipMemcpyAsync(&span, mx, 4, hipMemcpyDeviceToHost, st));
            SGM_HIP(hipStreamSynchronize(st));
            *span_out = (span + 2) & ~1;                            // (even, and one spare entry for an odd tail)
            return SGM_OK;
        };
        int32_t span = 0;
        SGM_TRY(windows(1, &span));
        // a window beyond 72 KiB leaves room for ONE workgroup per CU: let it be a 512-thread one over two slices
        if ((size_t)span * 8 > (size_t)72 * 1024) {
            int32_t span2 = 0;
            SGM_TRY(windows(2, &span2));
            if ((size_t)span2 * 8 <= kLdsCap) { span = span2; p.sl_gs = 2; }
            else if ((size_t)span * 8 <= kLdsCap) SGM_TRY(windows(1, &span));       // (back to one slice per window)
        }
        // worth it when the window is re-used: a slice's rows must reference its columns several times over
        if (span < 2 || (size_t)span * 8 > kLdsCap || (double)span * (double)((nsl + p.sl_gs - 1) / p.sl_gs) > 0.5 * (double)total) {
            dfree(p.sl_win0); p.sl_win0 = nullptr; span = 0; p.sl_gs = 1;
        }
        p.sl_span = span;
    }
    if (short_rows && !p.sl_win0) { free_sell(p); return SGM_OK; }        // (short rows without a window: the row-owner kernel)
    SGM_HIP(hipStreamSynchronize(st));
    csr_go_lean(p);
    return SGM_OK;
}

static int detect_sched_period_csr(Part &p, const std::vector<int32_t> &dict)
{
    p.sched_period = 0;
    if (!p.code || p.n < 64 * kSlRows) return SGM_OK;
    const int32_t R = kSlRows, mid = (p.n / 2) / kSlRows * kSlRows;
    std::vector<int32_t> rp((size_t)R + 1);
    SGM_HIP(hipMemcpy(rp.data(), p.rowptr + mid, ((size_t)R + 1) * 4, hipMemcpyDeviceToHost));
    const int64_t cnt = (int64_t)rp[(size_t)R] - rp[0];
    if (cnt <= 0) return SGM_OK;
    std::vector<uint8_t> codes((size_t)cnt);
    SGM_HIP(hipMemcpy(codes.data(), p.code + rp[0], (size_t)cnt, hipMemcpyDeviceToHost));
    p.sched_period = far_offset_of_sample(codes, R, dict, p.ndict);
    return SGM_OK;
}

// Offset dictionary of a row block (index work at setup, on the device): the distinct (col - row)
// values in ascending order; gives up (p.code stays null) beyond 255 distinct offsets.
static int build_offset_dict(Part &p, const int32_t *, const int32_t *)
{
    const int64_t nnz = p.nnz;
    const int32_t n = p.n;
    if (nnz == 0 || n == 0) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *table = nullptr;
    int *cnt = nullptr;             // [0] distinct offsets, [1] longest row
    SGM_TRY(dalloc(&table, (size_t)kDictSlots));
    SGM_TRY(dalloc(&cnt, 2));
    hipLaunchKernelGGL(k_fill32, dim3(kDictSlots / 256), dim3(256), 0, st, (int64_t)kDictSlots, table, kDictEmpty);
    SGM_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(int), st));
    const int grid = (int)std::min<int64_t>(((int64_t)n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_dict_collect, dim3(grid), dim3(256), 0, st, n, (const int32_t *)p.rowptr, (const int32_t *)p.col, table,
                       cnt, cnt + 1);
    std::vector<int32_t> htab(kDictSlots);
    int hcnt[2] = {0, 0};
    SGM_HIP(hipMemcpyAsync(htab.data(), table, kDictSlots * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(hcnt, cnt, sizeof hcnt, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(table); dfree(cnt);
    p.max_row = hcnt[1];
    // without a dictionary (option off at creation, or more than 255 offsets): try the int32 sliced form
    auto sliced32 = [&]() -> int {
        // scattered columns (x far beyond the L2s' reach, no offset structure): the column-blocked two-phase form the ELLPACK
        // matrices of that kind get (sgm_ellcb.hip) -- every gather an LDS access; same products, same order of additions
        SGM_TRY(build_ell_colblock(p));
        if (p.cb_P) return SGM_OK;
        const int W = p.max_row <= 3 ? 3 : p.max_row <= 5 ? 5 : p.max_row <= 7 ? 7 : p.max_row <= 8 ? 8 : p.max_row <= 12 ? 12
                    : p.max_row <= 16 ? 16 : p.max_row <= 20 ? 20 : p.max_row <= 24 ? 24 : p.max_row <= 28 ? 28 : 32;
        if (!p.opt.csr_sliced || p.max_row < 1 || p.max_row > 32 || (double)W * n > 1.25 * (double)nnz) return build_sell(p);
        const size_t rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scol, rows_padded * W));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        p.sw = W;
        hipLaunchKernelGGL(k_sl_pack_cols, dim3((unsigned)std::min<size_t>(rows_padded / kSlRows, 65536)), dim3(256), 0, g_rt.stream,
                           n, W, (const int32_t *)p.rowptr, (const int32_t *)p.col, p.scol);
        SGM_HIP(hipGetLastError());
        return pack_sliced(p);
    };
    if (!p.opt.csr_offset_dict || hcnt[0] > 255) return sliced32();
    std::vector<int32_t> dict;
    for (int32_t v : htab) if (v != kDictEmpty) dict.push_back(v);
    std::sort(dict.begin(), dict.end());
    p.ndict = (int32_t)dict.size();
    p.dict_reach = 0;
    for (int32_t v : dict) p.dict_reach = std::max(p.dict_reach, v < 0 ? -v : v);
    dict.resize(256, 0);
    SGM_TRY(dalloc(&p.code, (size_t)nnz + 32));
    SGM_TRY(dalloc(&p.dict, (size_t)256));
    SGM_HIP(hipMemcpyAsync(p.dict, dict.data(), 256 * 4, hipMemcpyHostToDevice, st));
    SGM_HIP(hipMemsetAsync(p.code + nnz, 0, 32, st));
    // sliced form: short rows, few offsets, little padding
    const int W = p.max_row <= 3 ? 3 : p.max_row <= 5 ? 5 : p.max_row <= 7 ? 7 : 8;
    const bool sliced = p.opt.csr_sliced && p.ndict <= 15 && p.max_row >= 1 && p.max_row <= 8 && (double)W * n <= 1.25 * (double)nnz;
    size_t rows_padded = 0;
    if (sliced) {
        rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scode, rows_padded));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        hipLaunchKernelGGL(k_fill32, dim3(vec_grid(rows_padded)), dim3(256), 0, st, (int64_t)rows_padded,
                           reinterpret_cast<int32_t *>(p.scode), (int32_t)-1);
        p.sw = W;
    }
    hipLaunchKernelGGL(k_dict_encode, dim3(grid), dim3(256), 0, st, n, (const int32_t *)p.rowptr, (const int32_t *)p.col,
                       (const int32_t *)p.dict, p.ndict, p.code, sliced ? p.scode : nullptr);
    SGM_HIP(hipGetLastError());
    if (sliced) SGM_TRY(pack_sliced(p));
    // longer rows (9..32 entries), <= 255 offsets, little padding: slot-major slices with 1-byte codes (k_csr_slb)
    int Wb = 0;                                               // value slots: the smallest instantiated width that holds the longest row
#define PICK(WW) if (!Wb && p.max_row <= WW) Wb = WW;
    SGM_SLB_WIDTHS(PICK)
#undef PICK
    const int Wc = (Wb + 7) / 8 * 8;                          // code bytes per row in eights
    if (!sliced && p.opt.csr_sliced && Wb && p.ndict <= 255 && p.max_row > 8 && (double)Wb * n <= 1.35 * (double)nnz) {
        rows_padded = ((size_t)n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.sbcode, rows_padded * Wc));
        SGM_TRY(dalloc(&p.sval, rows_padded * Wb));
        p.sw = Wb;
        hipLaunchKernelGGL(k_slb_pack_codes, dim3((unsigned)std::min<size_t>(rows_padded / kSlRows, 65536)), dim3(256), 0, st, n, Wb,
                           (const int32_t *)p.rowptr, (const uint8_t *)p.code, p.sbcode);
        SGM_HIP(hipGetLastError());
        SGM_TRY(pack_sliced(p));
    }
    SGM_HIP(hipStreamSynchronize(st));       // `dict` (host staging of the upload) goes out of scope
    if (p.scode || p.sbcode) SGM_TRY(detect_sched_period_csr(p, dict));
    // Rows too long or too uneven for the sliced forms: the SELL form (+ the slices' windows of x) where its own rules accept
    // the matrix, instead of the row-owner kernel with 1-byte codes -- whose owner lanes walk whole rows: dense bands and
    // blocks of 60..200 entries per row ran at 0.01-0.06 of the roofline with it (tools/perf_survey.py: n = 60301, 199 per
    // row: 559 us), an order of magnitude behind SELL on the same matrices.
    if (!p.scode && !p.sbcode) {
        SGM_TRY(build_sell(p));
        if (p.sl_val) {              // (the part is now what a matrix without a dictionary is: nothing reads the byte codes again)
            dfree(p.code); dfree(p.dict);
            p.code = nullptr; p.dict = nullptr;
            p.ndict = 0; p.dict_reach = 0;
        }
    }
    csr_go_lean(p);
    return SGM_OK;
}

// ELLPACK twin of build_offset_dict: codes for ALL max_d slots of every row (padding slots
// carry the last neighbour, so their offsets are already in the dictionary), row-major with
// the row padded to 4 / 8 / 16 bytes.  Skipped for max_d > 16 or > 255 distinct offsets.
// (device kernels of the ELLPACK twin: offsets of ALL max_d slots, slot-major columns)
__global__ __launch_bounds__(256) void k_ell_dict_collect(int32_t n, int32_t max_d, const int32_t *__restrict__ ecol, int32_t *table,
                                                          int *count)
{
    // per-workgroup LDS table first, its entries to the global table at the end (see k_dict_collect)
    constexpr int kLocal = 512;
    __shared__ int32_t ltab[kLocal];
    __shared__ int lcount;
    for (int t = threadIdx.x; t < kLocal; t += 256) ltab[t] = kDictEmpty;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    int32_t mine[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) mine[t] = kDictEmpty;
    int next = 0;
    bool over = false;
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n && !over; i += gridDim.x * blockDim.x)
        for (int32_t k = 0; k < max_d; ++k) {
            const int32_t off = ecol[(int64_t)k * n + i] - i;
            bool known = false;
#pragma unroll
            for (int t = 0; t < 8; ++t) known = known || mine[t] == off;
            if (known) continue;
#pragma unroll
            for (int t = 0; t < 8; ++t) if (t == next) mine[t] = off;
            next = (next + 1) & 7;
            if (*(volatile int *)&lcount > 255 || *(volatile int *)count > 255) { over = true; break; }
            uint32_t h = ((uint32_t)off * 2654435761u) >> 23;
            for (int probe = 0; probe < kLocal; ++probe) {
                const int32_t prev = atomicCAS(&ltab[h], kDictEmpty, off);
                if (prev == kDictEmpty) { atomicAdd(&lcount, 1); break; }
                if (prev == off) break;
                h = (h + 1) & (kLocal - 1);
            }
        }
    __syncthreads();
    if (lcount > 255) {
        if (threadIdx.x == 0) atomicAdd(count, 256);
        return;
    }
    for (int t = threadIdx.x; t < kLocal; t += 256) {
        const int32_t off = ltab[t];
        if (off == kDictEmpty || *(volatile int *)count > 255) continue;
        uint32_t h = ((uint32_t)off * 2654435761u) >> 22;
        for (int probe = 0; probe < kDictSlots; ++probe) {
            const int32_t prev = atomicCAS(&table[h], kDictEmpty, off);
            if (prev == kDictEmpty) { atomicAdd(count, 1); break; }
            if (prev == off) break;
            h = (h + 1) & (kDictSlots - 1);
        }
    }
}
__global__ __launch_bounds__(256) void k_ell_dict_encode(int32_t n, int32_t max_d, int32_t mdp, const int32_t *__restrict__ ecol,
                                                         const int32_t *__restrict__ dict, int ndict, uint8_t *__restrict__ ecode,
                                                         uint32_t *__restrict__ scode)
{
    __shared__ int32_t dl[256];
    for (int t = threadIdx.x; t < 256; t += blockDim.x) dl[t] = t < ndict ? dict[t] : INT32_MAX;
    __syncthreads();
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t cw = 0xffffffffu;
        for (int32_t k = 0; k < max_d; ++k) {
            const int32_t off = ecol[(int64_t)k * n + i] - i;
            int lo = 0, hi = ndict - 1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (dl[mid] < off) lo = mid + 1; else hi = mid; }
            ecode[(int64_t)i * mdp + k] = (uint8_t)lo;
            if (scode && k < 8) cw = (cw & ~(15u << (4 * k))) | ((uint32_t)lo << (4 * k));
        }
        if (scode) scode[i] = cw;
    }
}

int build_ell_offset_dict(Part &p)
{
    if (p.n == 0 || p.max_d == 0 || p.max_d > 16) return SGM_OK;
    hipStream_t st = g_rt.stream;
    int32_t *table = nullptr;
    int *cnt = nullptr;
    SGM_TRY(dalloc(&table, (size_t)kDictSlots));
    SGM_TRY(dalloc(&cnt, 1));
    hipLaunchKernelGGL(k_fill32, dim3(kDictSlots / 256), dim3(256), 0, st, (int64_t)kDictSlots, table, kDictEmpty);
    SGM_HIP(hipMemsetAsync(cnt, 0, sizeof(int), st));
    const int grid = (int)std::min<int64_t>(((int64_t)p.n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_ell_dict_collect, dim3(grid), dim3(256), 0, st, p.n, p.max_d, (const int32_t *)p.ecol, table, cnt);
    std::vector<int32_t> htab(kDictSlots);
    int hcnt = 0;
    SGM_HIP(hipMemcpyAsync(htab.data(), table, kDictSlots * 4, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipMemcpyAsync(&hcnt, cnt, sizeof hcnt, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));
    dfree(table); dfree(cnt);
    if (hcnt > 255) return SGM_OK;              // too many offsets: the int32 slot-major kernel
    std::vector<int32_t> dict;
    for (int32_t v : htab) if (v != kDictEmpty) dict.push_back(v);
    std::sort(dict.begin(), dict.end());
    const int ndict = (int)dict.size();
    p.dict_reach = 0;
    for (int32_t v : dict) p.dict_reach = std::max(p.dict_reach, v < 0 ? -v : v);
    dict.resize(256, 0);
    const int mdp = p.max_d <= 4 ? 4 : p.max_d <= 8 ? 8 : 16;
    const size_t code_bytes = (size_t)p.n * mdp + 16;
    SGM_TRY(dalloc(&p.ecode, code_bytes));
    if (!p.dict) SGM_TRY(dalloc(&p.dict, (size_t)256));
    SGM_HIP(hipMemsetAsync(p.ecode, 0, code_bytes, st));
    SGM_HIP(hipMemcpyAsync(p.dict, dict.data(), 256 * 4, hipMemcpyHostToDevice, st));
    p.emdp = mdp;
    // sliced form (see k_csr_sl): every one of the max_d slots is an entry (padding slots keep their
    // 0.0 * x(last neighbour) term, like the reference), so the CSR kernel applies as it is
    const bool sliced = p.opt.csr_sliced && ndict <= 15 && p.max_d >= 1 && p.max_d <= 8;
    if (sliced) {
        const int W = p.max_d <= 3 ? 3 : p.max_d <= 5 ? 5 : p.max_d <= 7 ? 7 : 8;
        const size_t rows_padded = ((size_t)p.n + kSlRows - 1) / kSlRows * kSlRows;
        SGM_TRY(dalloc(&p.scode, rows_padded));
        SGM_TRY(dalloc(&p.sval, rows_padded * W));
        hipLaunchKernelGGL(k_fill32, dim3(vec_grid(rows_padded)), dim3(256), 0, st, (int64_t)rows_padded,
                           reinterpret_cast<int32_t *>(p.scode), (int32_t)-1);
        p.sw = W;
    }
    hipLaunchKernelGGL(k_ell_dict_encode, dim3(grid), dim3(256), 0, st, p.n, p.max_d, mdp, (const int32_t *)p.ecol,
                       (const int32_t *)p.dict, ndict, p.ecode, sliced ? p.scode : nullptr);
    SGM_HIP(hipGetLastError());
    if (sliced) SGM_TRY(pack_sliced(p));
    SGM_HIP(hipStreamSynchronize(st));
    p.sched_period = 0;
    if (sliced && p.n >= 64 * kSlRows) {       // the slice schedule's period, from a 512-row sample (see far_offset_of_sample)
        const int32_t mid = (p.n / 2) / kSlRows * kSlRows;
        std::vector<uint8_t> rows((size_t)kSlRows * mdp), codes;
        SGM_HIP(hipMemcpy(rows.data(), p.ecode + (size_t)mid * mdp, rows.size(), hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < kSlRows; ++i)
            for (int32_t k = 0; k < p.max_d; ++k) codes.push_back(rows[(size_t)i * mdp + k]);
        p.sched_period = far_offset_of_sample(codes, kSlRows, dict, ndict);
    }
    return SGM_OK;
}

// Upload one CSR row block.  ptr1 is 1-based local (n+1), node1 is 1-based and already
// renumbered to [owned | halo]; `where` says where the three arrays live.  With `validate` the
// index arrays are checked on the device as they are converted (k_check_ptr1, k_dec1_check_cols):
// a malformed pointer array is SGM_ERR_BAD_ARG, a pointer array that does not end at nnz or a
// column outside 1..ncol_own+n_halo is SGM_ERR_DIMS, each naming the first offending row --
// never a memory fault inside a later product.
int build_csr_part(Part &p, int32_t n, int32_t ncol_own, int32_t n_halo, int64_t nnz,
                   const int32_t *ptr1, const int32_t *node1, const double *val, int where, bool validate)
{
    p.n = n;
    p.ncol_own = ncol_own;
    p.n_halo = n_halo;
    p.nnz = nnz;
    SGM_TRY(dalloc(&p.rowptr, (size_t)n + 1));
    SGM_TRY(dalloc(&p.col, (size_t)nnz + 4));          // k_csr_rl reads whole 16-byte pieces of col
    SGM_TRY(dalloc(&p.val, (size_t)nnz + 2));
    hipStream_t st = g_rt.stream;
    const hipMemcpyKind kind = where == SGM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    unsigned long long *bad = nullptr;                 // validation verdict: {first bad row, first bad entry}
    if (validate) {
        SGM_TRY(dalloc(&bad, 2));
        SGM_HIP(hipMemsetAsync(bad, 0xff, 2 * sizeof(unsigned long long), st));
    }
    SGM_HIP(hipMemsetAsync(p.col + nnz, 0, 4 * sizeof(int32_t), st));
    SGM_HIP(hipMemsetAsync(p.val + nnz, 0, 2 * sizeof(double), st));
    SGM_HIP(hipMemcpyAsync(p.rowptr, ptr1, ((size_t)n + 1) * sizeof(int32_t), kind, st));
    if (nnz) {
        SGM_HIP(hipMemcpyAsync(p.col, node1, (size_t)nnz * sizeof(int32_t), kind, st));
        SGM_HIP(hipMemcpyAsync(p.val, val, (size_t)nnz * sizeof(double), kind, st));
    }
    if (validate)
        hipLaunchKernelGGL(k_check_ptr1, dim3(vec_grid(n + 1)), dim3(kBlock), 0, st, (const int32_t *)p.rowptr, (int64_t)n, nnz, bad);
    hipLaunchKernelGGL(k_dec1, dim3(vec_grid(n + 1)), dim3(kBlock), 0, st, p.rowptr, (int64_t)n + 1);
    if (nnz) {
        if (validate)
            hipLaunchKernelGGL(k_dec1_check_cols, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, p.col, nnz, (int64_t)ncol_own + n_halo, bad);
        else
            hipLaunchKernelGGL(k_dec1, dim3(vec_grid(nnz)), dim3(kBlock), 0, st, p.col, nnz);
    }
    SGM_HIP(hipGetLastError());
    unsigned long long hbad[2] = {~0ull, ~0ull};
    if (validate) SGM_HIP(hipMemcpyAsync(hbad, bad, sizeof hbad, hipMemcpyDeviceToHost, st));
    SGM_HIP(hipStreamSynchronize(st));   // host staging buffers may go away after return
    dfree(bad);
    if (hbad[0] != ~0ull) {
        const int64_t i = (int64_t)hbad[0];
        int32_t v[2] = {0, 0};            // (0-based by now)
        SGM_HIP(hipMemcpy(v, p.rowptr + i, (i < n ? 2 : 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
        if (i == n)
            return fail(SGM_ERR_DIMS, "csr create: ptr(%lld) - 1 = %lld entries, but nnz = %lld", (long long)n + 1, (long long)v[0],
                        (long long)nnz);
        if (i == 0 && v[0] != 0) return fail(SGM_ERR_BAD_ARG, "csr create: ptr(1) = %d, expected 1 (1-based row pointers)", v[0] + 1);
        return fail(SGM_ERR_BAD_ARG, "csr create: row pointers decrease at row %lld: ptr(%lld) = %d > ptr(%lld) = %d", (long long)i + 1,
                    (long long)i + 1, v[0] + 1, (long long)i + 2, v[1] + 1);
    }
    if (hbad[1] != ~0ull) {
        const int64_t k = (int64_t)hbad[1];
        int32_t c = 0;
        SGM_HIP(hipMemcpy(&c, p.col + k, sizeof c, hipMemcpyDeviceToHost));
        std::vector<int32_t> hp((size_t)n + 1);          // error path only: the row that holds entry k
        SGM_HIP(hipMemcpy(hp.data(), p.rowptr, hp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        const int64_t row = std::upper_bound(hp.begin(), hp.end(), (int32_t)k) - hp.begin();      // 1-based
        return fail(SGM_ERR_DIMS, "csr create: node(%lld) = %d in row %lld is outside 1..%lld", (long long)k + 1, c + 1, (long long)row,
                    (long long)ncol_own + n_halo);
    }
    return build_offset_dict(p, where == SGM_HOST ? ptr1 : nullptr, where == SGM_HOST ? node1 : nullptr);
}

// A plain-CSR copy of a single-part CSR matrix (device to device; no derived SpMV format): scratch for setup work that wants to
// permute a matrix without touching the caller's (the reordering preconditioner, sgm_pc.hip)
int clone_csr_plain(sgm_mat A, sgm_mat *out)
{
    *out = nullptr;
    if (!A || A->fmt != SGM_FMT_CSR || A->parts.size() != 1 || A->comm)
        return fail(SGM_ERR_UNSUPPORTED, "clone_csr_plain: single-GPU CSR matrices only");
    const Part &p = A->parts[0];
    SGM_TRY(csr_need_arrays(p));
    struct Release { const Part &p; ~Release() { csr_release_arrays(p); } } rel{p};
    sgm_mat C = new sgm_mat_s;
    C->fmt = SGM_FMT_CSR; C->nrow = A->nrow; C->ncol = A->ncol; C->nnz = A->nnz;
    C->parts.resize(1);
    Part &q = C->parts[0];
    q.opt.csr_offset_dict = 0; q.opt.csr_sliced = 0; q.opt.csr_sell = 0; q.opt.csr_lean = 0; q.opt.slice_sched = 0;
    q.n = p.n; q.ncol_own = p.ncol_own; q.n_halo = 0; q.nnz = p.nnz; q.max_row = p.max_row;
    hipStream_t st = g_rt.stream;
    int rc = dalloc(&q.rowptr, (size_t)p.n + 1);
    if (rc == SGM_OK) rc = dalloc(&q.col, (size_t)p.nnz + 4);
    if (rc == SGM_OK) rc = dalloc(&q.val, (size_t)p.nnz + 2);
    if (rc != SGM_OK) { sgm_mat_destroy(C); return rc; }
    (void)hipMemcpyAsync(q.rowptr, p.rowptr, ((size_t)p.n + 1) * 4, hipMemcpyDeviceToDevice, st);
    (void)hipMemsetAsync(q.col + p.nnz, 0, 16, st);
    (void)hipMemsetAsync(q.val + p.nnz, 0, 16, st);
    if (p.nnz) {
        (void)hipMemcpyAsync(q.col, p.col, (size_t)p.nnz * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(q.val, p.val, (size_t)p.nnz * 8, hipMemcpyDeviceToDevice, st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { sgm_mat_destroy(C); return fail(SGM_ERR_HIP, "clone_csr_plain: copy failed"); }
    *out = C;
    return SGM_OK;
}

// after a change of the index arrays (matrix permutation): drop and rebuild the derived formats
int rebuild_csr_formats(Part &p)
{
    SGM_TRY(csr_need_arrays(p));           // col / val are what the formats are rebuilt from
    p.lean = false;
    dfree(p.code); dfree(p.dict); dfree(p.sval); dfree(p.scode); dfree(p.scol); dfree(p.sbcode);
    p.code = nullptr; p.dict = nullptr; p.sval = nullptr; p.scode = nullptr; p.scol = nullptr; p.sbcode = nullptr;
    free_sell(p);
    free_ell_colblock(p);          // (a scattered matrix's column-blocked form: the new order may have an offset dictionary instead)
    p.ndict = 0; p.dict_reach = 0; p.sw = 0; p.max_row = 0; p.sched_period = 0;
    free_slice_sched(p);
    return build_offset_dict(p, nullptr, nullptr);
}
int rebuild_ell_formats(Part &p)
{
    dfree(p.ecode); dfree(p.dict); dfree(p.sval); dfree(p.scode);
    p.ecode = nullptr; p.dict = nullptr; p.dict_reach = 0; p.emdp = 0; p.sval = nullptr; p.scode = nullptr; p.sw = 0; p.sched_period = 0;
    free_slice_sched(p);
    SGM_TRY(build_ell_offset_dict(p));
    SGM_TRY(build_ell_colblock(p));
    return refresh_ell_colblock_values(p);
}
int sgm_invalidate_transpose(sgm_mat A)
{
    if (A->T) { sgm_mat_destroy(A->T); A->T = nullptr; }
    dfree(A->tperm);
    A->tperm = nullptr;
    A->t_stale = true;
    return SGM_OK;
}

void free_part(Part &p)
{
    dfree(p.rowptr); dfree(p.col); dfree(p.val); dfree(p.code); dfree(p.dict); dfree(p.sval); dfree(p.scode); dfree(p.scol); dfree(p.sbcode); dfree(p.ecol); dfree(p.eval); dfree(p.edeg); dfree(p.ecode); dfree(p.xext);
    for (auto &nb : p.nbrs) { dfree(nb.send_idx); dfree(nb.send_buf); }
    free_slice_sched(p);
    free_ell_colblock(p);
    free_sell(p);
    p = Part();
}

}  // namespace sgm
