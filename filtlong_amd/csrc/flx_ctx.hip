// flx_ctx.hip — context, error handling, timing, layout helpers of libfiltlong_hip.so.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "flx_internal.h"
#include <vector>

static std::string g_create_error;

int flx_fail(flx_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_error = buf;
    return code;
}

extern "C" int flx_abi_version(void) { return FLX_ABI_VERSION; }

extern "C" int flx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}
extern "C" const char *flx_version(void) { return "filtlong-amd 0.1 (hot path of Filtlong v0.3.1; gfx950)"; }

extern "C" const char *flx_last_phred_kernel(const flx_ctx *ctx) { return ctx ? ctx->last_phred_kernel : ""; }
extern "C" int flx_last_kmer_locus(const flx_ctx *ctx) { return ctx && ctx->last_kmer_locus ? 1 : 0; }
extern "C" int flx_last_kmer_fold_grid(const flx_ctx *ctx) { return ctx && ctx->last_kmer_fold_grid ? 1 : 0; }
extern "C" const char *flx_last_kmer_cover(const flx_ctx *ctx) { return ctx ? ctx->last_kmer_cover : ""; }
extern "C" int64_t flx_last_kmer_handed_over(flx_ctx *ctx) {
    if (!ctx) return -1;
    if (!ctx->last_kmer_redo || !ctx->last_kmer_redo_n) return 0;
    std::vector<uint8_t> marks(ctx->last_kmer_redo_n);
    if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess ||
        hipMemcpy(marks.data(), ctx->last_kmer_redo, marks.size(), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    int64_t n = 0;
    for (uint8_t m : marks) n += m != 0;
    return n;
}

extern "C" const char *flx_last_error(const flx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// Phred byte -> quality, reference src/read.cpp:270-273.  Evaluated on the HOST with the host libm so
// that the table is bit-identical to what the CPU reference computes on the same box (glibc pow is
// not correctly rounded).  The call goes through a volatile pointer so the compiler
// cannot rewrite pow(10, x) into exp10(x).
static void build_phred_lut(double *lut257) {
    double (*volatile powfn)(double, double) = pow;
    for (int b = 0; b < 256; ++b) {
        int q = (int)(signed char)(unsigned char)b - 33;
        lut257[b] = 1.0 - powfn(10.0, -q / 10.0);
    }
    lut257[256] = 0.0;
}

// Every FLX_* environment variable the library reads (README.md lists what each one does) — most of them select a second
// implementation for the tests.  A name under one of the library's OWN prefixes (FLX_KMER_, FLX_PHRED_, FLX_RANK_, FLX_RCCL_, FLX_API_)
// that is none of these is refused once, here: a mistyped switch in a user's shell would otherwise be ignored silently, i.e. run
// another kernel than the one asked for.  Any other FLX_ name — another tool's, a stale export — is none of this library's business
// and is ignored, as the reference would ignore it (round-5 review: a foreign FLX_FOO must not stop a drop-in binary).
// (FLX_CLI_* belong to the command line, which checks its own.  tests/test_abi.py holds this list against the getenv calls in the sources.)
static const char *const kKnownEnv[] = {
    "FLX_API_TIMING", "FLX_KMER_COVER", "FLX_KMER_FOLD", "FLX_KMER_FOLD_EVENTS", "FLX_KMER_FOLD_GRID", "FLX_KMER_FOLD_STREAMS", "FLX_KMER_LOCUS",
    "FLX_KMER_LOCUS_BUILD", "FLX_KMER_PAIRTABLE", "FLX_KMER_PREFILTER", "FLX_KMER_SAFE1", "FLX_KMER_TEXT_ORDER", "FLX_PHRED_KERNEL",
    "FLX_PHRED_TABLES", "FLX_RANK_EXACT", "FLX_RANK_SORT", "FLX_RCCL_LIB",
};
static const char *const kOwnPrefixes[] = {"FLX_KMER_", "FLX_PHRED_", "FLX_RANK_", "FLX_RCCL_", "FLX_API_"};
extern char **environ;
static int check_environment() {
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "FLX_", 4) != 0) continue;
        const char *eq = strchr(*e, '=');
        const std::string name(*e, eq ? (size_t)(eq - *e) : strlen(*e));
        bool own = false;
        for (const char *pre : kOwnPrefixes) own = own || name.compare(0, strlen(pre), pre) == 0;
        if (!own) continue;
        bool known = false;
        for (const char *k : kKnownEnv) known = known || name == k;
        if (!known)
            return flx_fail(nullptr, FLX_ERR_INVALID, "unknown environment variable %s: not one of this library's switches (README.md lists them; "
                                                      "a mistyped one would silently select another kernel)", name.c_str());
    }
    return FLX_OK;
}

extern "C" int flx_ctx_create(int device_ordinal, flx_ctx **out) {
    if (!out) return flx_fail(nullptr, FLX_ERR_INVALID, "flx_ctx_create: out is NULL");
    *out = nullptr;
    FLX_CHECK(check_environment());
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return flx_fail(nullptr, FLX_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                        e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= n)
        return flx_fail(nullptr, FLX_ERR_INVALID, "device ordinal %d out of range (have %d)", device_ordinal, n);
    flx_ctx *ctx = new flx_ctx();
    ctx->device = device_ordinal;
    FLX_HIP(nullptr, hipSetDevice(device_ordinal));
    FLX_HIP(nullptr, hipGetDeviceProperties(&ctx->prop, device_ordinal));
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = ctx->prop.gcnArchName;
        delete ctx;
        return flx_fail(nullptr, FLX_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                        device_ordinal, arch.c_str());
    }
    FLX_HIP(nullptr, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    build_phred_lut(ctx->h_lut_q);
    FLX_HIP(nullptr, hipMalloc((void **)&ctx->d_lut_q, 257 * sizeof(double)));
    FLX_HIP(nullptr, hipMalloc((void **)&ctx->d_lut_d, 257 * sizeof(double)));
    FLX_HIP(nullptr, hipMemcpy(ctx->d_lut_q, ctx->h_lut_q, 257 * sizeof(double), hipMemcpyHostToDevice));
    *out = ctx;
    return FLX_OK;
}

extern "C" void flx_ctx_destroy(flx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)flx_comm_destroy(ctx);
    for (auto &t : ctx->timed) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->d_lut_q) (void)hipFree(ctx->d_lut_q);
    if (ctx->d_lut_d) (void)hipFree(ctx->d_lut_d);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    for (void *w : ctx->ws)
        if (w) (void)hipFree(w);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int flx_ctx_set_stream(flx_ctx *ctx, void *hip_stream) {
    if (!ctx) return FLX_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return FLX_OK;
}

extern "C" int flx_ctx_synchronize(flx_ctx *ctx) {
    if (!ctx) return FLX_ERR_INVALID;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLX_OK;
}

extern "C" int flx_ctx_device_info(const flx_ctx *ctx, char *name, size_t name_cap, int *n_cu, uint64_t *hbm_bytes) {
    if (!ctx) return FLX_ERR_INVALID;
    if (name && name_cap) {
        snprintf(name, name_cap, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = ctx->prop.totalGlobalMem;
    return FLX_OK;
}

// The division table D[c] = Q[c] / double(window_size): IEEE division, done once per window size on
// the host (exact, identical to the per-step `qualities[i] / window_size` of src/read.cpp:228-229,
// where window_size is a size_t converted to double).
int flx_ensure_lut_d(flx_ctx *ctx, int window_size) {
    if (ctx->lut_d_ws == window_size) return FLX_OK;
    double d[257];
    volatile double ws = (double)(size_t)window_size;
    for (int b = 0; b < 257; ++b) d[b] = ctx->h_lut_q[b] / ws;
    FLX_HIP(ctx, hipMemcpyAsync(ctx->d_lut_d, d, sizeof d, hipMemcpyHostToDevice, ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->lut_d_ws = window_size;
    return FLX_OK;
}

// ------------------------------------------------------------------------------------------ timing
static hipEvent_t take_event(flx_ctx *ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void flx_time_begin(flx_ctx *ctx, const char *name) {
    if (!ctx->timing) return;
    flx_timed_launch t;
    t.name = name;
    t.start = take_event(ctx);
    t.stop = take_event(ctx);
    (void)hipEventRecord(t.start, ctx->stream);
    ctx->timed_open.push_back(ctx->timed.size());
    ctx->timed.push_back(t);
}

void flx_time_end(flx_ctx *ctx) {  // closes the innermost open bracket
    if (!ctx->timing || ctx->timed_open.empty()) return;
    (void)hipEventRecord(ctx->timed[ctx->timed_open.back()].stop, ctx->stream);
    ctx->timed_open.pop_back();
}

extern "C" int flx_timing_enable(flx_ctx *ctx, int on) {
    if (!ctx) return FLX_ERR_INVALID;
    ctx->timing = on != 0;
    return FLX_OK;
}

extern "C" int flx_timing_reset(flx_ctx *ctx) {
    if (!ctx) return FLX_ERR_INVALID;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->timed) {
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->timed.clear();
    ctx->timed_open.clear();
    return FLX_OK;
}

extern "C" int flx_timing_get(flx_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches) {
    if (!ctx) return FLX_ERR_INVALID;
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    uint64_t n = 0;
    size_t pl = prefix ? strlen(prefix) : 0;
    for (auto &t : ctx->timed) {
        if (pl && strncmp(t.name, prefix, pl) != 0) continue;
        float ms = 0.f;
        FLX_HIP(ctx, hipEventElapsedTime(&ms, t.start, t.stop));
        tot += ms;
        ++n;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = n;
    return FLX_OK;
}

// ------------------------------------------------------------------------------------------ memory
int flx_scratch(flx_ctx *ctx, size_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipMalloc(&ctx->scratch, want);
        if (e != hipSuccess) return flx_fail(ctx, FLX_ERR_NOMEM, "device scratch of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return FLX_OK;
}

int flx_workspace(flx_ctx *ctx, int slot, size_t bytes, void **out) {
    if (slot < 0 || slot >= 3) return flx_fail(ctx, FLX_ERR_INVALID, "workspace slot %d", slot);
    if (bytes > ctx->ws_bytes[slot]) {
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws[slot]) (void)hipFree(ctx->ws[slot]);
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
        const size_t want = bytes + bytes / 16 + 4096;
        hipError_t e = hipMalloc(&ctx->ws[slot], want);
        if (e != hipSuccess) return flx_fail(ctx, FLX_ERR_NOMEM, "device workspace of %zu bytes: %s", want, hipGetErrorString(e));
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return FLX_OK;
}

int flx_dalloc(flx_ctx *ctx, flx_dbuf &b, size_t bytes) {
    hipError_t e = hipMalloc(&b.p, bytes ? bytes : 16);
    if (e != hipSuccess) return flx_fail(ctx, FLX_ERR_NOMEM, "device allocation of %zu bytes: %s", bytes, hipGetErrorString(e));
    return FLX_OK;
}

// ------------------------------------------------------------------------------------------ layout
extern "C" int flx_plane_layout(const int32_t *lengths, uint64_t n_reads, uint64_t *offsets, uint64_t *plane_bytes) {
    if ((!lengths && n_reads) || !plane_bytes) return FLX_ERR_INVALID;
    uint64_t off = 0;
    for (uint64_t i = 0; i < n_reads; ++i) {
        if (lengths[i] < 0) return FLX_ERR_INVALID;
        // Reads start 16-byte aligned (the kernels' only requirement); reads of at least 1 KiB start on a 128-byte line, so
        // that the Phred kernel's 128-byte chunks are whole lines (each line crosses the fabric exactly once; < 6 % padding).
        if (lengths[i] >= 1024) off = (off + 127u) & ~(uint64_t)127u;
        if (offsets) offsets[i] = off;
        off += ((uint64_t)lengths[i] + 15u) & ~(uint64_t)15u;
    }
    *plane_bytes = off ? off : 16;
    return FLX_OK;
}

extern "C" int flx_length_order(const int32_t *lengths, uint64_t n_reads, uint32_t *order) {
    if ((!lengths || !order) && n_reads) return FLX_ERR_INVALID;
    if (n_reads > 0xffffffffull) return FLX_ERR_INVALID;
    std::iota(order, order + n_reads, 0u);
    std::stable_sort(order, order + n_reads, [lengths](uint32_t a, uint32_t b) { return lengths[a] > lengths[b]; });
    return FLX_OK;
}
