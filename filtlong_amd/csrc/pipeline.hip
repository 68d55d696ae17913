// pipeline.hip — streaming ingest into the per-read scoring (seam 2): pinned, double-buffered chunks.
//
// Replaces the pass-1 loop of the reference's main() (src/main.cpp:70-127: one Read::Read per kseq record, everything the
// record needs dropped again before the next one) for hosts whose input does not fit the GPU — or the host — at once:
// the caller packs one CHUNK of reads at a time into a pinned staging buffer the pipeline hands out; `submit` starts the
// H2D copy on a copy stream and returns, a worker thread scores the chunk (flx_score_batch_dev on the context's stream) as
// soon as its copy has landed and appends the per-read results to host arrays.  Two slots: while chunk k is copied and
// scored, the caller parses and packs chunk k+1 into the other slot.  Between the chunks only per-read scalars survive —
// mean, window, pass flag, first/last and the children's ranges and scores — exactly what the reference keeps per Read.
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>

#include "flx_internal.h"
#include "kmerset.h"

namespace {

struct Slot {
    uint8_t *h_plane = nullptr;  // pinned
    uint8_t *d_plane = nullptr;
    uint64_t *h_off = nullptr;   // pinned: offsets | lengths | order
    int32_t *h_len = nullptr;
    uint32_t *h_ord = nullptr;
    uint64_t *d_off = nullptr;
    int32_t *d_len = nullptr;
    uint32_t *d_ord = nullptr;
    double *d_mean = nullptr, *d_win = nullptr;
    uint8_t *d_pass = nullptr;
    int32_t *d_first = nullptr, *d_last = nullptr;
    uint64_t *d_coff = nullptr;
    hipEvent_t copied = nullptr;
    uint64_t n = 0, plane_bytes = 0;
    bool busy = false;  // submitted, not yet scored
};

}  // namespace

struct flx_pipeline {
    flx_ctx *ctx = nullptr;
    const flx_kmerset *set = nullptr;
    flx_params params;
    uint64_t cap_bytes = 0, cap_reads = 0;
    Slot slot[2];
    int cur = 0;            // the slot the caller is packing
    bool handed_out = false;
    hipStream_t copy_stream = nullptr;
    // children staging on the device (grown on demand, worker only)
    uint64_t child_cap = 0;
    int32_t *d_crng = nullptr;
    double *d_cmean = nullptr, *d_cwin = nullptr;
    uint8_t *d_cpass = nullptr;

    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<int> queue;
    bool stop = false;
    int error = FLX_OK;
    std::string error_msg;

    // results, in submission order
    std::vector<double> mean_q, window_q, child_mean_q, child_window_q;
    std::vector<uint8_t> passed, child_passed;
    std::vector<int32_t> first, last, child_ranges;
    std::vector<uint64_t> child_offsets;  // global CSR: [n + 1]
    double h2d_s = 0, score_s = 0;
};

namespace {

int alloc_slot(flx_pipeline *p, Slot &s) {
    flx_ctx *ctx = p->ctx;
    const uint64_t nr = p->cap_reads;
    FLX_HIP(ctx, hipHostMalloc((void **)&s.h_plane, p->cap_bytes, hipHostMallocDefault));
    FLX_HIP(ctx, hipHostMalloc((void **)&s.h_off, nr * 16 + 64, hipHostMallocDefault));
    s.h_len = (int32_t *)(s.h_off + nr);
    s.h_ord = (uint32_t *)(s.h_len + nr);
    FLX_HIP(ctx, hipMalloc((void **)&s.d_plane, p->cap_bytes));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_off, nr * 16 + 64));
    s.d_len = (int32_t *)(s.d_off + nr);
    s.d_ord = (uint32_t *)(s.d_len + nr);
    FLX_HIP(ctx, hipMalloc((void **)&s.d_mean, nr * 8));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_win, nr * 8));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_pass, nr));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_first, nr * 4));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_last, nr * 4));
    FLX_HIP(ctx, hipMalloc((void **)&s.d_coff, (nr + 1) * 8));
    FLX_HIP(ctx, hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
    return FLX_OK;
}

void free_slot(Slot &s) {
    if (s.h_plane) (void)hipHostFree(s.h_plane);
    if (s.h_off) (void)hipHostFree(s.h_off);
    for (void *d : {(void *)s.d_plane, (void *)s.d_off, (void *)s.d_mean, (void *)s.d_win, (void *)s.d_pass, (void *)s.d_first,
                    (void *)s.d_last, (void *)s.d_coff})
        if (d) (void)hipFree(d);
    if (s.copied) (void)hipEventDestroy(s.copied);
    s = Slot();
}

int grow_children(flx_pipeline *p, uint64_t want) {
    flx_ctx *ctx = p->ctx;
    for (void *d : {(void *)p->d_crng, (void *)p->d_cmean, (void *)p->d_cwin, (void *)p->d_cpass})
        if (d) (void)hipFree(d);
    p->d_crng = nullptr; p->d_cmean = nullptr; p->d_cwin = nullptr; p->d_cpass = nullptr;
    p->child_cap = 0;
    FLX_HIP(ctx, hipMalloc((void **)&p->d_crng, want * 8));
    FLX_HIP(ctx, hipMalloc((void **)&p->d_cmean, want * 8));
    FLX_HIP(ctx, hipMalloc((void **)&p->d_cwin, want * 8));
    FLX_HIP(ctx, hipMalloc((void **)&p->d_cpass, want));
    p->child_cap = want;
    return FLX_OK;
}

// worker: score one submitted slot and append its results
int score_slot(flx_pipeline *p, Slot &s) {
    flx_ctx *ctx = p->ctx;
    const uint64_t n = s.n;
    FLX_HIP(ctx, hipStreamWaitEvent(ctx->stream, s.copied, 0));
    const bool kmer_mode = p->set && flx_kmerset_size(p->set) > 0;
    const bool want_children = kmer_mode && (p->params.trim || p->params.split_set);
    flx_scores dev;
    for (;;) {
        memset(&dev, 0, sizeof dev);
        dev.mean_q = s.d_mean; dev.window_q = s.d_win; dev.passed = s.d_pass;
        dev.first = s.d_first; dev.last = s.d_last;
        if (want_children) {
            if (p->child_cap == 0) FLX_CHECK(grow_children(p, std::max<uint64_t>(1024, 4 * p->cap_reads)));
            dev.child_offsets = s.d_coff;
            dev.child_ranges = p->d_crng; dev.child_mean_q = p->d_cmean; dev.child_window_q = p->d_cwin;
            dev.child_passed = p->d_cpass; dev.child_capacity = p->child_cap;
        }
        const int rc = flx_score_batch_dev(ctx, p->set, s.d_plane, s.plane_bytes, s.d_off, s.d_len, s.d_ord, n, &p->params, &dev);
        if (rc == FLX_ERR_CAPACITY && dev.n_children > p->child_cap) {
            FLX_CHECK(grow_children(p, dev.n_children + dev.n_children / 4));
            continue;
        }
        FLX_CHECK(rc);
        break;
    }
    const size_t at = p->mean_q.size();
    p->mean_q.resize(at + n); p->window_q.resize(at + n); p->passed.resize(at + n);
    p->first.resize(at + n); p->last.resize(at + n);
    hipStream_t st = ctx->stream;
    if (n) {
        FLX_HIP(ctx, hipMemcpyAsync(&p->mean_q[at], s.d_mean, n * 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->window_q[at], s.d_win, n * 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->passed[at], s.d_pass, n, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->first[at], s.d_first, n * 4, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->last[at], s.d_last, n * 4, hipMemcpyDeviceToHost, st));
    }
    const uint64_t nc = want_children ? dev.n_children : 0;
    const uint64_t cbase = p->child_offsets.back();
    const size_t oat = p->child_offsets.size();  // == at + 1
    p->child_offsets.resize(oat + n);
    std::vector<uint64_t> coff(n + 1, 0);
    if (want_children && n) FLX_HIP(ctx, hipMemcpyAsync(coff.data(), s.d_coff, (n + 1) * 8, hipMemcpyDeviceToHost, st));
    const size_t cat = p->child_mean_q.size();
    p->child_ranges.resize(2 * (cat + nc)); p->child_mean_q.resize(cat + nc); p->child_window_q.resize(cat + nc);
    p->child_passed.resize(cat + nc);
    if (nc) {
        FLX_HIP(ctx, hipMemcpyAsync(&p->child_ranges[2 * cat], p->d_crng, nc * 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->child_mean_q[cat], p->d_cmean, nc * 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->child_window_q[cat], p->d_cwin, nc * 8, hipMemcpyDeviceToHost, st));
        FLX_HIP(ctx, hipMemcpyAsync(&p->child_passed[cat], p->d_cpass, nc, hipMemcpyDeviceToHost, st));
    }
    FLX_HIP(ctx, hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n; ++i) p->child_offsets[oat + i] = cbase + coff[i + 1];
    return FLX_OK;
}

void worker_main(flx_pipeline *p) {
    (void)hipSetDevice(p->ctx->device);
    for (;;) {
        int k;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv.wait(lk, [&] { return p->stop || !p->queue.empty(); });
            if (p->queue.empty()) return;  // stop requested and drained
            k = p->queue.front();
        }
        int rc = FLX_OK;
        {
            std::unique_lock<std::mutex> lk(p->mu);
            rc = p->error;
        }
        if (rc == FLX_OK) {
            rc = score_slot(p, p->slot[k]);
        }
        {
            std::unique_lock<std::mutex> lk(p->mu);
            if (rc != FLX_OK && p->error == FLX_OK) {
                p->error = rc;
                p->error_msg = p->ctx->err;
            }
            p->queue.pop_front();
            p->slot[k].busy = false;
        }
        p->cv.notify_all();
    }
}

}  // namespace

extern "C" int flx_pipeline_create(flx_ctx *ctx, const flx_kmerset *set, const flx_params *params, uint64_t chunk_plane_bytes,
                                   uint64_t chunk_reads, flx_pipeline **out) {
    if (!ctx) return FLX_ERR_INVALID;
    if (!params || !out) return flx_fail(ctx, FLX_ERR_INVALID, "params/out must not be NULL");
    *out = nullptr;
    if (set && !flx_kmerset_is_final(set)) return flx_fail(ctx, FLX_ERR_STATE, "k-mer set is not finalized");
    if (chunk_plane_bytes < 4096 || chunk_reads < 1 || chunk_reads > 0xffffffffull)
        return flx_fail(ctx, FLX_ERR_INVALID, "chunk of %llu bytes / %llu reads", (unsigned long long)chunk_plane_bytes,
                        (unsigned long long)chunk_reads);
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    flx_pipeline *p = new flx_pipeline();
    p->ctx = ctx;
    p->set = set;
    p->params = *params;
    p->cap_bytes = (chunk_plane_bytes + 4095) & ~4095ull;
    p->cap_reads = chunk_reads;
    p->child_offsets.push_back(0);
    int rc = FLX_OK;
    hipError_t e = hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) rc = flx_fail(ctx, FLX_ERR_HIP, "copy stream: %s", hipGetErrorString(e));
    for (int k = 0; k < 2 && rc == FLX_OK; ++k) rc = alloc_slot(p, p->slot[k]);
    if (rc != FLX_OK) {
        for (int k = 0; k < 2; ++k) free_slot(p->slot[k]);
        if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
        delete p;
        return rc;
    }
    p->worker = std::thread(worker_main, p);
    *out = p;
    return FLX_OK;
}

extern "C" int flx_pipeline_reserve(flx_pipeline *p, uint64_t chunk_plane_bytes, uint64_t chunk_reads) {
    if (!p) return FLX_ERR_INVALID;
    flx_ctx *ctx = p->ctx;
    if (p->handed_out) return flx_fail(ctx, FLX_ERR_STATE, "flx_pipeline_reserve between flx_pipeline_next_buffer and flx_pipeline_submit");
    const uint64_t want_bytes = std::max<uint64_t>(p->cap_bytes, (chunk_plane_bytes + 4095) & ~4095ull);
    const uint64_t want_reads = std::max<uint64_t>(p->cap_reads, chunk_reads);
    if (want_reads > 0xffffffffull) return flx_fail(ctx, FLX_ERR_INVALID, "chunk of %llu reads", (unsigned long long)want_reads);
    if (want_bytes == p->cap_bytes && want_reads == p->cap_reads) return FLX_OK;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv.wait(lk, [&] { return p->queue.empty(); });  // both slots idle: the worker has appended their results
        if (p->error != FLX_OK) {
            ctx->err = p->error_msg;
            return p->error;
        }
    }
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FLX_HIP(ctx, hipStreamSynchronize(p->copy_stream));
    for (int k = 0; k < 2; ++k) free_slot(p->slot[k]);
    p->cap_bytes = want_bytes;
    p->cap_reads = want_reads;
    for (int k = 0; k < 2; ++k) {
        const int rc = alloc_slot(p, p->slot[k]);
        if (rc != FLX_OK) return rc;
    }
    return FLX_OK;
}

extern "C" int flx_pipeline_next_buffer(flx_pipeline *p, uint8_t **plane, uint64_t *capacity_bytes, uint64_t *capacity_reads) {
    if (!p || !plane) return FLX_ERR_INVALID;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return !p->slot[p->cur].busy; });  // the other slot may still be in flight
    if (p->error != FLX_OK) {
        p->ctx->err = p->error_msg;
        return p->error;
    }
    p->handed_out = true;
    *plane = p->slot[p->cur].h_plane;
    if (capacity_bytes) *capacity_bytes = p->cap_bytes;
    if (capacity_reads) *capacity_reads = p->cap_reads;
    return FLX_OK;
}

extern "C" int flx_pipeline_submit(flx_pipeline *p, uint64_t plane_bytes, const uint64_t *offsets, const int32_t *lengths,
                                   uint64_t n_reads) {
    if (!p) return FLX_ERR_INVALID;
    flx_ctx *ctx = p->ctx;
    if (!p->handed_out) return flx_fail(ctx, FLX_ERR_STATE, "flx_pipeline_submit without flx_pipeline_next_buffer");
    if (plane_bytes > p->cap_bytes || n_reads > p->cap_reads || (plane_bytes & 15))
        return flx_fail(ctx, FLX_ERR_INVALID, "chunk of %llu bytes / %llu reads exceeds the pipeline's slots",
                        (unsigned long long)plane_bytes, (unsigned long long)n_reads);
    if (n_reads && (!offsets || !lengths)) return flx_fail(ctx, FLX_ERR_INVALID, "offsets/lengths must not be NULL");
    Slot &s = p->slot[p->cur];
    for (uint64_t i = 0; i < n_reads; ++i)
        if (lengths[i] < 0 || (offsets[i] & 15) || offsets[i] + (((uint64_t)lengths[i] + 15) & ~15ull) > plane_bytes)
            return flx_fail(ctx, FLX_ERR_INVALID, "read %llu: offset must be 16-byte aligned and inside the chunk", (unsigned long long)i);
    p->handed_out = false;
    s.n = n_reads;
    s.plane_bytes = plane_bytes;
    if (n_reads) {
        memcpy(s.h_off, offsets, n_reads * 8);
        memcpy(s.h_len, lengths, n_reads * 4);
        flx_length_order(s.h_len, n_reads, s.h_ord);  // longest first: the 64 reads of a wavefront finish together
    }
    (void)hipSetDevice(ctx->device);
    if (plane_bytes) FLX_HIP(ctx, hipMemcpyAsync(s.d_plane, s.h_plane, plane_bytes, hipMemcpyHostToDevice, p->copy_stream));
    if (n_reads) {
        FLX_HIP(ctx, hipMemcpyAsync(s.d_off, s.h_off, n_reads * 8, hipMemcpyHostToDevice, p->copy_stream));
        FLX_HIP(ctx, hipMemcpyAsync(s.d_len, s.h_len, n_reads * 4, hipMemcpyHostToDevice, p->copy_stream));
        FLX_HIP(ctx, hipMemcpyAsync(s.d_ord, s.h_ord, n_reads * 4, hipMemcpyHostToDevice, p->copy_stream));
    }
    FLX_HIP(ctx, hipEventRecord(s.copied, p->copy_stream));
    {
        std::unique_lock<std::mutex> lk(p->mu);
        s.busy = true;
        p->queue.push_back(p->cur);
        p->cur ^= 1;
    }
    p->cv.notify_all();
    return FLX_OK;
}

extern "C" int flx_pipeline_finish(flx_pipeline *p, flx_scores *all, uint64_t *n_reads) {
    if (!p || !all) return FLX_ERR_INVALID;
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv.wait(lk, [&] { return p->queue.empty(); });
    if (p->error != FLX_OK) {
        p->ctx->err = p->error_msg;
        return p->error;
    }
    memset(all, 0, sizeof *all);
    all->mean_q = p->mean_q.data(); all->window_q = p->window_q.data(); all->passed = p->passed.data();
    all->first = p->first.data(); all->last = p->last.data();
    all->child_offsets = p->child_offsets.data();
    all->child_ranges = p->child_ranges.data(); all->child_mean_q = p->child_mean_q.data();
    all->child_window_q = p->child_window_q.data(); all->child_passed = p->child_passed.data();
    all->child_capacity = p->child_mean_q.size();
    all->n_children = p->child_mean_q.size();
    if (n_reads) *n_reads = p->mean_q.size();
    return FLX_OK;
}

extern "C" void flx_pipeline_destroy(flx_pipeline *p) {
    if (!p) return;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        p->stop = true;
    }
    p->cv.notify_all();
    if (p->worker.joinable()) p->worker.join();
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    if (p->copy_stream) {
        (void)hipStreamSynchronize(p->copy_stream);
        (void)hipStreamDestroy(p->copy_stream);
    }
    for (int k = 0; k < 2; ++k) free_slot(p->slot[k]);
    for (void *d : {(void *)p->d_crng, (void *)p->d_cmean, (void *)p->d_cwin, (void *)p->d_cpass})
        if (d) (void)hipFree(d);
    delete p;
}
