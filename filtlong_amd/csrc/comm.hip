// comm.hip — multi-GPU exchange of the global stage over RCCL (xGMI), owned by the library.
//
// One process per GPU; every process holds one flx_ctx with one RCCL communicator.  The reference has no counterpart (it
// is a single process, src/main.cpp:37-321); what is exchanged is what main.cpp:169-261 needs from ALL reads2 entries:
//   * one small sum first: every rank's reads2 count + the bases of the reads passing the hard cut-offs (main.cpp:222-226);
//   * ONE all-gather of the mean qualities (the statistics of main.cpp:170-196 are order-dependent folds over all of
//     them; 8 bytes per entry; always ncclAllGather — unequal shards (children, a remainder) are padded to the largest
//     count and packed by one kernel);
//   * the 8 selection histograms (257 x u64) are ncclAllReduce'd on the device, on the context's stream, between the
//     histogram kernel and the kernel that picks the next key byte — no host synchronisation per pass;
//   * ONE small host-side sum after the selection: band sizes + the boundary-audit candidates themselves (up to 32 per rank
//     ride along; a larger band — a big group of near-equal scores — takes a second exchange);
//   * only when the reference's own std::sort order over all reads has to decide (NaN scores, equal scores straddling
//     the cut): all-gather of window / length / passed as well and the single-GPU stage replicated on every rank.
//
// RCCL is loaded with dlopen on first use (FLX_RCCL_LIB, else librccl.so.1 — the copy a host process such as PyTorch may
// already have loaded — else /opt/rocm/lib/librccl.so.1), so single-GPU users never touch it and the library does not
// force a second RCCL into a process that brings its own.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>
#include <rccl/rccl.h>

#include "flx_internal.h"
#include "rank_internal.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
RcclApi g_rccl;

int load_rccl(flx_ctx *ctx) {
    if (g_rccl.handle) return FLX_OK;
    const char *cands[] = {getenv("FLX_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *c : cands) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return flx_fail(ctx, FLX_ERR_STATE, "cannot load RCCL (librccl.so.1): %s", dlerror());
#define FLX_SYM(field, name)                                                                  \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                  \
    if (!g_rccl.field) return flx_fail(ctx, FLX_ERR_STATE, "RCCL symbol %s not found", name);
    FLX_SYM(GetUniqueId, "ncclGetUniqueId")
    FLX_SYM(CommInitRank, "ncclCommInitRank")
    FLX_SYM(CommDestroy, "ncclCommDestroy")
    FLX_SYM(AllReduce, "ncclAllReduce")
    FLX_SYM(AllGather, "ncclAllGather")
    FLX_SYM(Broadcast, "ncclBroadcast")
    FLX_SYM(GroupStart, "ncclGroupStart")
    FLX_SYM(GroupEnd, "ncclGroupEnd")
    FLX_SYM(GetErrorString, "ncclGetErrorString")
#undef FLX_SYM
    g_rccl.handle = h;
    return FLX_OK;
}

#define FLX_NCCL(ctx, call)                                                                                     \
    do {                                                                                                        \
        ncclResult_t r__ = (call);                                                                              \
        if (r__ != ncclSuccess)                                                                                 \
            return flx_fail((ctx), FLX_ERR_STATE, "%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r__), __FILE__, \
                            __LINE__);                                                                          \
    } while (0)

// This RCCL build prints a banner ("RCCL version : ... Librccl path : ...") on STDOUT when a communicator is created — and
// stdout is the data channel of the host this library serves (the reference writes the surviving reads there,
// src/main.cpp:263-313).  File descriptor 1 points at /dev/null while RCCL initialises.
struct StdoutSilencer {
    int saved = -1;
    StdoutSilencer() {
        fflush(stdout);
        const int nul = open("/dev/null", O_WRONLY);
        if (nul < 0) return;
        saved = dup(1);
        if (saved >= 0) dup2(nul, 1);
        close(nul);
    }
    ~StdoutSilencer() {
        if (saved < 0) return;
        fflush(stdout);
        dup2(saved, 1);
        close(saved);
    }
};

}  // namespace

struct flx_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    void *stage = nullptr;  // small device staging buffer for host-side sums
    size_t stage_bytes = 0;
    void *gather = nullptr;  // all-gathered arrays (grow-only)
    size_t gather_bytes = 0;
};

extern "C" int flx_comm_unique_id(flx_ctx *ctx, void *id_out) {
    if (!ctx || !id_out) return FLX_ERR_INVALID;
    FLX_CHECK(load_rccl(ctx));
    ncclUniqueId id;
    {
        StdoutSilencer quiet;
        FLX_NCCL(ctx, g_rccl.GetUniqueId(&id));
    }
    static_assert(sizeof id == FLX_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof id);
    return FLX_OK;
}

extern "C" int flx_comm_init(flx_ctx *ctx, const void *id_in, int rank, int world) {
    if (!ctx || !id_in) return FLX_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world) return flx_fail(ctx, FLX_ERR_INVALID, "bad rank %d / world %d", rank, world);
    if (ctx->comm) return flx_fail(ctx, FLX_ERR_STATE, "communicator already initialised");
    FLX_CHECK(load_rccl(ctx));
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    flx_comm *c = new flx_comm();
    c->rank = rank;
    c->world = world;
    ncclResult_t r;
    {
        StdoutSilencer quiet;
        r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    }
    if (r != ncclSuccess) {
        delete c;
        return flx_fail(ctx, FLX_ERR_STATE, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
    ctx->comm = c;
    return FLX_OK;
}

extern "C" int flx_comm_destroy(flx_ctx *ctx) {
    if (!ctx) return FLX_ERR_INVALID;
    flx_comm *c = ctx->comm;
    if (!c) return FLX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm);
    if (c->stage) (void)hipFree(c->stage);
    if (c->gather) (void)hipFree(c->gather);
    delete c;
    ctx->comm = nullptr;
    return FLX_OK;
}

extern "C" int flx_comm_rank(const flx_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->rank : 0; }
extern "C" int flx_comm_world(const flx_ctx *ctx) { return ctx && ctx->comm ? ctx->comm->world : 1; }

int flx_comm_allreduce_u64_dev(flx_ctx *ctx, uint64_t *d_buf, uint64_t count) {
    flx_comm *c = ctx->comm;
    if (!c) return flx_fail(ctx, FLX_ERR_STATE, "no communicator");
    flx_time_begin(ctx, "flx_comm_allreduce_dev");  // (inside the selection's own bracket, flx_rank_select: brackets nest)
    ncclResult_t r = g_rccl.AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, c->comm, ctx->stream);
    flx_time_end(ctx);
    if (r != ncclSuccess) return flx_fail(ctx, FLX_ERR_STATE, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    return FLX_OK;
}

static int comm_stage(flx_ctx *ctx, size_t bytes) {  // the small device staging buffer of host-visible sums (grow-only)
    flx_comm *c = ctx->comm;
    if (bytes > c->stage_bytes) {
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_bytes = 0;
        const size_t want = std::max<size_t>(bytes * 2, 1 << 16);
        FLX_HIP(ctx, hipMalloc(&c->stage, want));
        c->stage_bytes = want;
    }
    return FLX_OK;
}

int flx_comm_allreduce_u64_host(flx_ctx *ctx, uint64_t *buf, uint64_t count) {
    flx_comm *c = ctx->comm;
    if (!c) return flx_fail(ctx, FLX_ERR_STATE, "no communicator");
    const size_t bytes = count * 8;
    FLX_CHECK(comm_stage(ctx, bytes));
    flx_time_begin(ctx, "flx_comm_sum_host");  // a host-visible sum: copy in, all-reduce, copy out, and the host waits for it
    hipError_t e = hipMemcpyAsync(c->stage, buf, bytes, hipMemcpyHostToDevice, ctx->stream);
    ncclResult_t r = e == hipSuccess ? g_rccl.AllReduce(c->stage, c->stage, count, ncclUint64, ncclSum, c->comm, ctx->stream) : ncclSuccess;
    if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(buf, c->stage, bytes, hipMemcpyDeviceToHost, ctx->stream);
    flx_time_end(ctx);
    if (r != ncclSuccess) return flx_fail(ctx, FLX_ERR_STATE, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
    FLX_HIP(ctx, e);
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLX_OK;
}

extern "C" int flx_comm_sum_u64(flx_ctx *ctx, uint64_t *buf, uint64_t count) {
    if (!ctx || (!buf && count)) return FLX_ERR_INVALID;
    if (!ctx->comm || count == 0) return FLX_OK;  // single rank: the sum is the value itself
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    return flx_comm_allreduce_u64_host(ctx, buf, count);
}

// all-gather with per-rank counts: rank r's `elems[r]` elements of `esize` bytes land at offset sum(elems[0..r)) of recv.
// Always ONE ncclAllGather (RCCL's own multi-ring schedule over the xGMI links).  Equal counts (a batch sharded evenly,
// the benchmark's case): straight into recv.  Unequal counts (children, a remainder): every rank contributes the LARGEST
// count (the tail of a shorter shard is padding, never read), the slots land in `d_padded` and one kernel packs them.
struct SegTable {
    uint64_t begin[65];  // begin[r] = first packed element of rank r; begin[world] = total
    int world;
};

__global__ void __launch_bounds__(256) k_pack_segments(const uint8_t *padded, uint8_t *packed, SegTable t, uint64_t slot_elems,
                                                       uint32_t esize) {
    const uint64_t total = t.begin[t.world];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        int r = 0;
        while (i >= t.begin[r + 1]) ++r;
        const uint64_t src = (uint64_t)r * slot_elems + (i - t.begin[r]);
        if (esize == 8) ((uint64_t *)packed)[i] = ((const uint64_t *)padded)[src];
        else if (esize == 4) ((uint32_t *)packed)[i] = ((const uint32_t *)padded)[src];
        else packed[i] = padded[src];
    }
}

static int allgather_v(flx_ctx *ctx, const void *d_send, void *d_recv, void *d_padded, void *d_send_pad,
                       const std::vector<uint64_t> &elems, size_t esize) {
    flx_comm *c = ctx->comm;
    if (c->world > 64) return flx_fail(ctx, FLX_ERR_INVALID, "at most 64 ranks");
    bool equal = true;
    uint64_t slot = 0, total = 0;
    for (int r = 0; r < c->world; ++r) {
        equal = equal && elems[r] == elems[0];
        slot = std::max<uint64_t>(slot, elems[r]);
        total += elems[r];
    }
    if (total == 0) return FLX_OK;
    if (equal) {
        FLX_NCCL(ctx, g_rccl.AllGather(d_send, d_recv, (size_t)slot * esize, ncclUint8, c->comm, ctx->stream));
        return FLX_OK;
    }
    // a shorter shard sends from a staging slot of full size (reading past the end of the caller's array is not ours to do)
    const void *send = d_send;
    if (elems[c->rank] < slot) {
        if (elems[c->rank])
            FLX_HIP(ctx, hipMemcpyAsync(d_send_pad, d_send, (size_t)elems[c->rank] * esize, hipMemcpyDeviceToDevice, ctx->stream));
        send = d_send_pad;
    }
    FLX_NCCL(ctx, g_rccl.AllGather(send, d_padded, (size_t)slot * esize, ncclUint8, c->comm, ctx->stream));
    SegTable t;
    t.world = c->world;
    t.begin[0] = 0;
    for (int r = 0; r < c->world; ++r) t.begin[r + 1] = t.begin[r] + elems[r];
    const unsigned grid = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((total + 255) / 256, 4096));
    hipLaunchKernelGGL(k_pack_segments, dim3(grid), dim3(256), 0, ctx->stream, (const uint8_t *)d_padded, (uint8_t *)d_recv, t, slot,
                       (uint32_t)esize);
    return FLX_OK;
}

extern "C" int flx_rank_and_cut_comm_dev(flx_ctx *ctx, uint64_t n_local, const void *d_mean_q, const void *d_window_q,
                                         const void *d_length, void *d_passed, double lw, double mw, double ww,
                                         int target_bases_set, int64_t target_bases, int keep_percent_set,
                                         double keep_percent, int64_t total_bases, void *d_final_score,
                                         flx_cut_report *rep) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!rep) return flx_fail(ctx, FLX_ERR_INVALID, "report must not be NULL");
    flx_comm *c = ctx->comm;
    if (!c)  // no communicator: the plain single-GPU stage
        return flx_rank_and_cut_dev(ctx, n_local, d_mean_q, d_window_q, d_length, d_passed, lw, mw, ww, target_bases_set,
                                    target_bases, keep_percent_set, keep_percent, total_bases, d_final_score, rep);
    if (n_local && (!d_mean_q || !d_window_q || !d_length || !d_passed)) return flx_fail(ctx, FLX_ERR_INVALID, "NULL local array");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;

    // shard sizes (reads2 entries per rank: children make them unequal) and the bases of the reads that pass the hard
    // cut-offs (main.cpp:222-226), all ranks, in ONE host-visible sum
    std::vector<uint64_t> counts((size_t)c->world + 1, 0);
    {
        const size_t bytes = counts.size() * 8;
        FLX_CHECK(comm_stage(ctx, bytes));
        uint64_t *d_stage = (uint64_t *)c->stage;
        counts[c->rank] = n_local;
        FLX_HIP(ctx, hipMemcpyAsync(d_stage, counts.data(), bytes, hipMemcpyHostToDevice, st));
        if (n_local) FLX_CHECK(flx_passed_bases_async(ctx, n_local, (const int32_t *)d_length, (const uint8_t *)d_passed, d_stage + c->world));
        flx_time_begin(ctx, "flx_comm_counts");
        ncclResult_t r = g_rccl.AllReduce(d_stage, d_stage, counts.size(), ncclUint64, ncclSum, c->comm, st);
        flx_time_end(ctx);
        if (r != ncclSuccess) return flx_fail(ctx, FLX_ERR_STATE, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r));
        FLX_HIP(ctx, hipMemcpyAsync(counts.data(), d_stage, bytes, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
    }
    const uint64_t passed_bases = counts[c->world];
    counts.resize((size_t)c->world);
    uint64_t n_total = 0, first = 0, slot = 0;
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) first = n_total;
        n_total += counts[r];
        slot = std::max<uint64_t>(slot, counts[r]);
    }
    if (n_total > 0xffffffffull) return flx_fail(ctx, FLX_ERR_INVALID, "at most 2^32-1 reads");

    // gather buffer: [mean f64 | window f64 | length i32 | passed u8] x n_total (only the first part is filled unless the
    // replicated fallback is needed) + the padded slots of an all-gather with unequal counts + one send slot
    const size_t rec_bytes = (n_total * 21 + 255) & ~(size_t)255;
    const size_t pad_bytes = ((size_t)c->world * slot * 8 + 255) & ~(size_t)255;
    // (+ the final scores and the report of the replicated fallback below: nothing is allocated once the collectives of that path
    // have begun — a rank that returned on a failed allocation there would leave its peers inside a broadcast)
    const size_t fs_bytes = d_final_score ? ((n_total * 8 + 255) & ~(size_t)255) : 0;
    const size_t rep_bytes = (sizeof(flx_cut_report) + 8 + 255) & ~(size_t)255;
    const size_t need = rec_bytes + pad_bytes + ((slot * 8 + 64 + 255) & ~(size_t)255) + fs_bytes + rep_bytes;
    if (need > c->gather_bytes) {
        FLX_HIP(ctx, hipStreamSynchronize(st));
        if (c->gather) (void)hipFree(c->gather);
        c->gather = nullptr;
        c->gather_bytes = 0;
        FLX_HIP(ctx, hipMalloc(&c->gather, need + need / 8));
        c->gather_bytes = need + need / 8;
    }
    double *g_mean = (double *)c->gather;
    double *g_win = g_mean + n_total;
    int32_t *g_len = (int32_t *)(g_win + n_total);
    uint8_t *g_pass = (uint8_t *)(g_len + n_total);
    void *g_padded = (char *)c->gather + rec_bytes;
    void *g_sendpad = (char *)g_padded + pad_bytes;
    void *g_fs = (char *)g_sendpad + ((slot * 8 + 64 + 255) & ~(size_t)255);
    void *g_rep = (char *)g_fs + fs_bytes;

    flx_time_begin(ctx, "flx_comm_allgather_means");
    int rc = allgather_v(ctx, d_mean_q, g_mean, g_padded, g_sendpad, counts, 8);
    flx_time_end(ctx);
    FLX_CHECK(rc);

    rc = flx_rank_and_cut_sharded_comm(ctx, n_total, g_mean, first, n_local, (const double *)d_window_q, (const int32_t *)d_length,
                                       (uint8_t *)d_passed, lw, mw, ww, target_bases_set, target_bases, keep_percent_set,
                                       keep_percent, total_bases, d_final_score, c->rank, c->world, passed_bases, rep);
    if (rc != FLX_NEED_REPLICATED) return rc;

    // The reference's own std::sort order over ALL reads decides (NaN scores, an order-dependent tie at the cut).  The records
    // are gathered and RANK 0 ALONE runs the single-GPU stage — its host part is a std::sort of every record on all host
    // threads (rank.hip: exact_host_cut), and the ranks of one node share that host: eight identical sorts side by side took
    // eight times the threads for the same answer — then the flags, the report and (if asked for) the scores are broadcast.
    flx_time_begin(ctx, "flx_comm_allgather_records");
    rc = allgather_v(ctx, d_window_q, g_win, g_padded, g_sendpad, counts, 8);
    if (rc == FLX_OK) rc = allgather_v(ctx, d_length, g_len, g_padded, g_sendpad, counts, 4);
    if (rc == FLX_OK) rc = allgather_v(ctx, d_passed, g_pass, g_padded, g_sendpad, counts, 1);
    flx_time_end(ctx);
    FLX_CHECK(rc);
    int stage_rc = FLX_OK;
    if (c->rank == 0) {  // (errors are recorded, not returned: the head word must reach the peers, who are on their way into the broadcast)
        stage_rc = flx_rank_and_cut_dev(ctx, n_total, g_mean, g_win, g_len, g_pass, lw, mw, ww, target_bases_set, target_bases,
                                        keep_percent_set, keep_percent, total_bases, d_final_score ? g_fs : nullptr, rep);
        int64_t head[1] = {stage_rc};
        if (hipMemcpyAsync(g_rep, head, 8, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync((char *)g_rep + 8, rep, sizeof(flx_cut_report), hipMemcpyHostToDevice, st) != hipSuccess) {
            if (stage_rc == FLX_OK) stage_rc = flx_fail(ctx, FLX_ERR_HIP, "the outcome of the global stage could not be staged for the broadcast");
        }
    }
    {
        flx_time_scope tb(ctx, "flx_comm_broadcast_outcome");
        FLX_NCCL(ctx, g_rccl.Broadcast(g_rep, g_rep, sizeof(flx_cut_report) + 8, ncclUint8, 0, c->comm, st));
        if (n_total) FLX_NCCL(ctx, g_rccl.Broadcast(g_pass, g_pass, n_total, ncclUint8, 0, c->comm, st));
        if (d_final_score && n_total) FLX_NCCL(ctx, g_rccl.Broadcast(g_fs, g_fs, n_total * 8, ncclUint8, 0, c->comm, st));
    }
    {
        int64_t head[1] = {0};
        FLX_HIP(ctx, hipMemcpyAsync(head, g_rep, 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(rep, (char *)g_rep + 8, sizeof(flx_cut_report), hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipStreamSynchronize(st));
        if (c->rank == 0 && stage_rc != FLX_OK) return stage_rc;
        if (head[0] != FLX_OK) return flx_fail(ctx, (int)head[0], "the global stage failed on rank 0");
    }
    if (n_local) {
        FLX_HIP(ctx, hipMemcpyAsync(d_passed, g_pass + first, n_local, hipMemcpyDeviceToDevice, st));
        if (d_final_score)
            FLX_HIP(ctx, hipMemcpyAsync(d_final_score, (const double *)g_fs + first, n_local * 8, hipMemcpyDeviceToDevice, st));
    }
    FLX_HIP(ctx, hipStreamSynchronize(st));
    return FLX_OK;
}

// host arrays (the CLI's reads2 scalars): staged like flx_rank_and_cut
extern "C" int flx_rank_and_cut_comm(flx_ctx *ctx, uint64_t n_local, const double *mean_q, const double *window_q,
                                     const int32_t *length, uint8_t *passed, double lw, double mw, double ww,
                                     int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                                     int64_t total_bases, double *final_score, flx_cut_report *rep) {
    if (!ctx) return FLX_ERR_INVALID;
    if (n_local && (!mean_q || !window_q || !length || !passed)) return flx_fail(ctx, FLX_ERR_INVALID, "NULL input array");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_dbuf d_mean, d_win, d_len, d_pass, d_fs;
    FLX_CHECK(flx_dalloc(ctx, d_mean, n_local * 8));
    FLX_CHECK(flx_dalloc(ctx, d_win, n_local * 8));
    FLX_CHECK(flx_dalloc(ctx, d_len, n_local * 4));
    FLX_CHECK(flx_dalloc(ctx, d_pass, n_local));
    if (final_score) FLX_CHECK(flx_dalloc(ctx, d_fs, n_local * 8));
    if (n_local) {
        FLX_HIP(ctx, hipMemcpyAsync(d_mean.p, mean_q, n_local * 8, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_win.p, window_q, n_local * 8, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_len.p, length, n_local * 4, hipMemcpyHostToDevice, ctx->stream));
        FLX_HIP(ctx, hipMemcpyAsync(d_pass.p, passed, n_local, hipMemcpyHostToDevice, ctx->stream));
    }
    FLX_CHECK(flx_rank_and_cut_comm_dev(ctx, n_local, d_mean.p, d_win.p, d_len.p, d_pass.p, lw, mw, ww, target_bases_set, target_bases,
                                        keep_percent_set, keep_percent, total_bases, final_score ? d_fs.p : nullptr, rep));
    if (n_local) {
        FLX_HIP(ctx, hipMemcpyAsync(passed, d_pass.p, n_local, hipMemcpyDeviceToHost, ctx->stream));
        if (final_score) FLX_HIP(ctx, hipMemcpyAsync(final_score, d_fs.p, n_local * 8, hipMemcpyDeviceToHost, ctx->stream));
        FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FLX_OK;
}
