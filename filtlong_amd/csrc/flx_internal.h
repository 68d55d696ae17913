// flx_internal.h — shared internals of libfiltlong_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/filtlong_hip.h"

struct flx_comm;  // comm.hip: the RCCL communicator of this rank

struct flx_timed_launch {
    const char *name;
    hipEvent_t start, stop;
};

struct flx_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    std::string err;
    bool last_kmer_fold_grid = false;    // ... and its window folds on the integer grid (score_kmer.hip: GridTab)
    const uint8_t *last_kmer_redo = nullptr;  // the marks of the last k_kmer_cover_q run (device, in workspace 0 until the next scoring call) ...
    uint64_t last_kmer_redo_n = 0;            // ... and how many there are
    const char *last_kmer_cover = "";    // which coverage kernel the last k-mer scoring call ran: "q" (cover_queue.hip), "w", "v2"
    bool last_kmer_locus = false;        // the last k-mer scoring call ran with the assembly text (kmerset.h: flx_locus)
    const char *last_phred_kernel = "";  // which Phred kernel the last scoring call launched (flx_last_phred_kernel)

    // Phred LUTs: lut_q[c] = 1 - 10^(-(c-33)/10) for the signed-char value of byte c, built on the
    // host with the host libm (the same one the CPU reference uses); lut_d = lut_q / window_size.
    // Entry 256 is the "no base" entry (0.0): adding it is an exact no-op.
    double h_lut_q[257];
    double *d_lut_q = nullptr;  // [257]
    double *d_lut_d = nullptr;  // [257], rebuilt when window_size changes
    int lut_d_ws = -1;

    flx_comm *comm = nullptr;  // multi-GPU: set by flx_comm_init

    // timing
    bool timing = false;
    std::vector<flx_timed_launch> timed;
    std::vector<size_t> timed_open;  // brackets that are open, innermost last (they nest: a collective inside the selection's bracket)
    std::vector<hipEvent_t> event_pool;

    // reusable device scratch (grown on demand)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // grow-only workspaces of the k-mer scoring path (kept between calls: a 12 GB hipMalloc + hipFree per batch costs
    // more than the fold kernels)
    void *ws[3] = {nullptr, nullptr, nullptr};
    size_t ws_bytes[3] = {0, 0, 0};
};

int flx_fail(flx_ctx *ctx, int code, const char *fmt, ...);

#define FLX_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return flx_fail((ctx), FLX_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                                   \
    } while (0)

#define FLX_CHECK(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != FLX_OK) return rc__; \
    } while (0)

// timing brackets on the context's stream: call begin before a launch and end right after it (brackets nest) ...
void flx_time_begin(flx_ctx *ctx, const char *name);
void flx_time_end(flx_ctx *ctx);
// ... or as a scope, wherever a FLX_HIP / FLX_CHECK between the two can return: an early return must not leave a bracket open
// (the next flx_time_end would close the wrong one; advisor, round 5)
struct flx_time_scope {
    flx_ctx *ctx;
    bool open;
    flx_time_scope(flx_ctx *c, const char *name) : ctx(c), open(true) { flx_time_begin(c, name); }
    flx_time_scope(const flx_time_scope &) = delete;
    void end() {
        if (open) flx_time_end(ctx);
        open = false;
    }
    ~flx_time_scope() { end(); }
};

// grow-only scratch on the device
int flx_scratch(flx_ctx *ctx, size_t bytes, void **out);
// grow-only workspace `slot` (0: per-read arrays of the k-mer path, 1: its coverage bit plane); valid until the next call
// with the same slot
int flx_workspace(flx_ctx *ctx, int slot, size_t bytes, void **out);

// simple device buffer owned by a call (freed in destructor)
struct flx_dbuf {
    void *p = nullptr;
    ~flx_dbuf() {
        if (p) (void)hipFree(p);
    }
    template <typename T>
    T *as() { return (T *)p; }
};
int flx_dalloc(flx_ctx *ctx, flx_dbuf &b, size_t bytes);

// comm.hip: element-wise sums over all ranks of the context's communicator
int flx_comm_allreduce_u64_dev(flx_ctx *ctx, uint64_t *d_buf, uint64_t count);   // device buffer, in stream order
int flx_comm_allreduce_u64_host(flx_ctx *ctx, uint64_t *buf, uint64_t count);    // host buffer, synchronous

// internal launchers -------------------------------------------------------------------------
struct flx_score_out_dev {  // device pointers
    double *mean_q;
    double *window_q;
    uint8_t *passed;
};

int flx_ensure_lut_d(flx_ctx *ctx, int window_size);

int flx_launch_score_phred(flx_ctx *ctx, const uint8_t *d_plane, uint64_t plane_bytes, const uint64_t *d_offsets,
                           const int32_t *d_lengths, const uint32_t *d_order, uint64_t n_reads,
                           const flx_params *p, flx_score_out_dev out);
