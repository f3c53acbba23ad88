// Runtime plumbing of libsigma_hip.so: device selection, stream, errors, HBM buffers.
#include "sgm_internal.hpp"

#include <algorithm>

namespace sgm {

std::string g_err;
Runtime g_rt;
Options g_opt;
Heartbeat g_hb;
extern int g_force_collectives;     // sgm_dist.hip

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

bool trace_on()
{
    static const bool on = getenv("SGM_TRACE") != nullptr;
    return on;
}

int require_init()
{
    if (g_rt.ready) return SGM_OK;
    return sgm_init(-1);
}

int finish()
{
    if (g_rt.async) return SGM_OK;
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

// Large transfers between PAGEABLE host memory and the device: the runtime's own staging moves them at 1-2 GB/s; two pinned
// 32 MiB buffers, the host memcpy of one chunk overlapping the DMA of the other, move them at the host memcpy rate.
// Synchronous: the data has arrived when the call returns (and the library's stream has been drained).
int copy_big(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    constexpr size_t kChunk = (size_t)32 << 20;
    if (bytes == 0) return SGM_OK;
    if (bytes < (size_t)4 << 20 || (kind != hipMemcpyHostToDevice && kind != hipMemcpyDeviceToHost)) {
        SGM_HIP(hipMemcpyAsync(dst, src, bytes, kind, g_rt.stream));
        SGM_HIP(hipStreamSynchronize(g_rt.stream));
        return SGM_OK;
    }
    static char *pin[2] = {nullptr, nullptr};
    static hipEvent_t ev[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; ++b)
        if (!pin[b]) {
            SGM_HIP(hipHostMalloc((void **)&pin[b], kChunk, hipHostMallocDefault));
            SGM_HIP(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
        }
    hipStream_t st = g_rt.stream;
    const size_t nchunk = (bytes + kChunk - 1) / kChunk;
    if (kind == hipMemcpyHostToDevice) {
        for (size_t c = 0; c < nchunk; ++c) {
            const int b = (int)(c & 1);
            const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
            if (c >= 2) SGM_HIP(hipEventSynchronize(ev[b]));                  // the DMA that last read this buffer is done
            memcpy(pin[b], (const char *)src + off, sz);
            SGM_HIP(hipMemcpyAsync((char *)dst + off, pin[b], sz, hipMemcpyHostToDevice, st));
            SGM_HIP(hipEventRecord(ev[b], st));
        }
    } else {
        for (size_t c = 0; c <= nchunk; ++c) {
            if (c < nchunk) {
                const int b = (int)(c & 1);
                const size_t off = c * kChunk, sz = std::min(kChunk, bytes - off);
                SGM_HIP(hipMemcpyAsync(pin[b], (const char *)src + off, sz, hipMemcpyDeviceToHost, st));
                SGM_HIP(hipEventRecord(ev[b], st));
            }
            if (c >= 1) {                                                     // drain the previous chunk while this one flies
                const int b = (int)((c - 1) & 1);
                const size_t off = (c - 1) * kChunk, sz = std::min(kChunk, bytes - off);
                SGM_HIP(hipEventSynchronize(ev[b]));
                memcpy((char *)dst + off, pin[b], sz);
            }
        }
    }
    SGM_HIP(hipStreamSynchronize(st));
    return SGM_OK;
}

#define SGM_OPT(group, field) if (!strcmp(name, #field)) return &o.field;
int *mat_option_field(MatOptions &o, const char *name)
{
    SGM_OPT(mat, csr_offset_dict) SGM_OPT(mat, ell_offset_dict) SGM_OPT(mat, csr_row_owner) SGM_OPT(mat, csr_row_lines)
    SGM_OPT(mat, csr_sliced) SGM_OPT(mat, csr_sell) SGM_OPT(mat, csr_xwindow) SGM_OPT(mat, csr_lean) SGM_OPT(mat, ell_colblock) SGM_OPT(mat, ell_colblock_cols)
    SGM_OPT(mat, ell_colblock_rows) SGM_OPT(mat, slice_sched) SGM_OPT(mat, coloring_pass)
    return nullptr;
}
int *solver_option_field(SolverOptions &o, const char *name)
{
    SGM_OPT(solver, cg_small) SGM_OPT(solver, bicgstab_small) SGM_OPT(solver, krylov_graph) SGM_OPT(solver, dot_order)
    SGM_OPT(solver, gmres_cgs2) SGM_OPT(solver, dist_halo_fused) SGM_OPT(solver, coop_spin_limit) SGM_OPT(solver, cg_coop_variant)
    SGM_OPT(solver, reorder_solve)
    return nullptr;
}
int *pc_option_field(PcOptions &o, const char *name)
{
    SGM_OPT(pc, ildu_strips) SGM_OPT(pc, ildu_rows) SGM_OPT(pc, pipeline_spin_limit) SGM_OPT(pc, ildu_reorder)
    return nullptr;
}
#undef SGM_OPT

int normalise_option(const char *name, int value, int *out)
{
    int v = value;
    if (!strcmp(name, "ell_colblock_cols")) v = std::min(kEllcbMaxCols, std::max(2, value)) & ~1;
    else if (!strcmp(name, "ell_colblock_rows")) v = value == 512 ? 512 : value == 256 ? 256 : 0;
    else if (!strcmp(name, "ildu_reorder")) v = value != 0;
    else if (!strcmp(name, "csr_sell")) v = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(name, "csr_lean") || !strcmp(name, "csr_xwindow")) v = value != 0;
    else if (!strcmp(name, "slice_sched")) v = value <= 0 ? 0 : value == 1 ? 1 : std::max(4, value);
    else if (!strcmp(name, "krylov_graph")) v = value <= 0 ? 0 : value == 1 ? 1 : std::max(16, (value + 15) / 16 * 16);
    else if (!strcmp(name, "cg_small")) v = std::max(0, value);
    else if (!strcmp(name, "dist_halo_fused") || !strcmp(name, "reorder_solve")) v = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(name, "coop_spin_limit")) v = std::max(0, value);
    else if (!strcmp(name, "gmres_cgs2")) v = value != 0;
    else if (!strcmp(name, "cg_coop_variant")) {
        const int r = value & 15;
        if (value < 0 || value > 31 || (r != 0 && r != 1 && r != 2 && r != 4 && r != 8))
            return fail(SGM_ERR_BAD_ARG, "option cg_coop_variant: rows per thread 0 (by size), 1, 2, 4 or 8, + 16 = no one-XCD variant");
    }
    else if (!strcmp(name, "coloring_pass")) v = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(name, "pipeline_spin_limit")) v = std::max(0, value);
    else if (!strcmp(name, "dot_order") && value != 0 && value != 1)
        return fail(SGM_ERR_BAD_ARG, "option dot_order is 0 (tree) or 1 (the reference's sequential order)");
    *out = v;
    return SGM_OK;
}

}  // namespace sgm

using namespace sgm;

extern "C" {

int sgm_init(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(SGM_ERR_NO_DEVICE,
                    "sgm_init: no HIP device visible (%s); this library has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0) {
        if (g_rt.ready) return SGM_OK;
        device = 0;
    }
    if (device >= count) return fail(SGM_ERR_BAD_ARG, "sgm_init: device %d of %d", device, count);
    if (g_rt.ready && g_rt.device == device) return SGM_OK;
    SGM_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SGM_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fprintf(stderr, "[sigma_hip] warning: built for gfx950, running on %s\n", prop.gcnArchName);
    g_rt.num_cu = prop.multiProcessorCount;
    if (!g_rt.own_stream) SGM_HIP(hipStreamCreateWithFlags(&g_rt.own_stream, hipStreamNonBlocking));
    g_rt.stream = g_rt.own_stream;
    g_rt.device = device;
    g_rt.ready = true;
    return SGM_OK;
}

int sgm_finalize(void)
{
    if (!g_rt.ready) return SGM_OK;
    (void)hipStreamSynchronize(g_rt.stream);
    if (g_rt.own_stream) (void)hipStreamDestroy(g_rt.own_stream);
    g_rt = Runtime();
    return SGM_OK;
}

const char *sgm_last_error(void) { return g_err.c_str(); }

int sgm_set_stream(void *s)
{
    SGM_TRY(require_init());
    g_rt.stream = s ? (hipStream_t)s : g_rt.own_stream;
    return SGM_OK;
}

int sgm_set_async(int on)
{
    g_rt.async = on != 0;
    return SGM_OK;
}

int sgm_synchronize(void)
{
    SGM_TRY(require_init());
    SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return SGM_OK;
}

int sgm_set_option(const char *name, int value)
{
    if (!name) return fail(SGM_ERR_BAD_ARG, "sgm_set_option: null name");
    if (!strcmp(name, "dist_force_collectives")) { g_force_collectives = value != 0; return SGM_OK; }      // (process-wide by nature)
    int v = 0;
    SGM_TRY(normalise_option(name, value, &v));
    int *f = mat_option_field(g_opt.mat, name);
    if (!f) f = solver_option_field(g_opt.solver, name);
    if (!f) f = pc_option_field(g_opt.pc, name);
    if (!f) return fail(SGM_ERR_BAD_ARG, "sgm_set_option: unknown option '%s'", name);
    *f = v;
    return SGM_OK;
}

int sgm_heartbeat(int64_t *out6, char *phase_name, int len)
{
    static const char *const names[HB_PHASES] = {
        "idle (outside the library, or in a call without collectives)", "create_dist: all-gather of the want matrix / swap of the request lists",
        "product: halo send/recv group being posted", "dot: all-reduce being posted", "solver: queueing a batch of iterations",
        "solver: waiting for the queued batch (stream synchronisation; a collective that never completes hangs HERE)",
        "dot_order=1: running sum travelling rank to rank", "distributed transpose: entry exchange"};
    const int32_t ph = g_hb.phase;
    if (out6) {
        out6[0] = ph; out6[1] = g_hb.beats; out6[2] = g_hb.iteration; out6[3] = g_hb.halo_posts; out6[4] = g_hb.allreduce_posts;
        out6[5] = g_hb.solves;
    }
    if (phase_name && len > 0) snprintf(phase_name, (size_t)len, "%s", ph >= 0 && ph < HB_PHASES ? names[ph] : "?");
    return SGM_OK;
}

int sgm_malloc(void **p, size_t bytes)
{
    SGM_TRY(require_init());
    if (!p) return fail(SGM_ERR_BAD_ARG, "sgm_malloc: null out pointer");
    char *q = nullptr;
    SGM_TRY(dalloc(&q, bytes));
    *p = q;
    return SGM_OK;
}

int sgm_free(void *p)
{
    if (p) SGM_HIP(hipFree(p));
    return SGM_OK;
}

int sgm_memcpy(void *dst, const void *src, size_t bytes, int kind)
{
    SGM_TRY(require_init());
    hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice
                      : kind == 1 ? hipMemcpyDeviceToHost
                                  : hipMemcpyDeviceToDevice;
    SGM_HIP(hipMemcpyAsync(dst, src, bytes, k, g_rt.stream));
    if (kind != 2) SGM_HIP(hipStreamSynchronize(g_rt.stream));
    return finish();
}

}  // extern "C"
