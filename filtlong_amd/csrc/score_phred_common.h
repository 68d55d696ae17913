// score_phred_common.h — pieces shared by the Phred scoring kernels (score_phred.hip: LDS-ring and direct kernels;
// score_phred_regs.hip: register-history kernel).
#pragma once

#include "flx_internal.h"

namespace flx_phred {

constexpr int LUT_PAD = 264;  // doubles per plain table in LDS (257 used)

struct PhredArgs {
    const uint8_t *plane;
    const uint64_t *offsets;
    const int32_t *lengths;
    const uint32_t *order;
    uint64_t n_reads;
    const double *lut_q;
    const double *lut_d;
    int ws;
    int n_slots;  // ring slots of CH bytes: ceil(ws / CH) + 1
    int stride;   // bytes per read row in the ring (odd multiple of 16: conflict-free b128 rows)
    double ws_d;
    double clamp;  // 0.5 / ws   (src/read.cpp:233)
    flx_params p;
    double *mean_q;
    double *window_q;
    uint8_t *passed;
    unsigned int *ticket;  // ring kernel: next group of 64 reads (persistent waves)
    unsigned int n_groups; // ceil(n_reads / 64)
    unsigned int *redo_count;  // register kernel, bank-private tables: reads holding a byte >= 128 ...
    uint32_t *redo_list;       // ... are re-scored by the direct kernel (their ids are appended here)
};

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ uint32_t byte_of(const uint4 &v, int i) {
    const uint32_t d = (i >> 2) == 0 ? v.x : (i >> 2) == 1 ? v.y : (i >> 2) == 2 ? v.z : v.w;
    return (d >> (8 * (i & 3))) & 0xffu;
}

// byte `b` of dword x, times 8: the LDS byte offset of that Phred value's table entry.  One SDWA shift.
__device__ __forceinline__ uint32_t lut_addr(uint32_t x, int b) {
    uint32_t r;
    const uint32_t three = 3;
    switch (b) {
        case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(x)); break;
        case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(three), "v"(x)); break;
        case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(three), "v"(x)); break;
        default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(three), "v"(x)); break;
    }
    return r;
}

// 8-byte LDS table load at a byte offset (the offset already carries the x8 scaling)
__device__ __forceinline__ double lds_f64(const double *table, uint32_t byte_off) {
    return *reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(table) + byte_off);
}

// hard cut-offs, src/read.cpp:64-73 (pre-normalisation 0-100 values; NaN compares false)
__device__ __forceinline__ uint8_t hard_cutoffs(const flx_params &p, int L, double mean, double window) {
    bool ok = true;
    if (p.min_length_set && L < p.min_length) ok = false;
    else if (p.max_length_set && L > p.max_length) ok = false;
    else if (p.min_mean_q_set && mean < p.min_mean_q) ok = false;
    else if (p.min_window_q_set && window < p.min_window_q) ok = false;
    return ok ? 1 : 0;
}

__device__ __forceinline__ void finish_read(const PhredArgs &a, uint32_t rid, int L, double s, double mn) {
    const double mean = 100.0 * s / (double)L;  // (100*s)/n, src/read.cpp:212; L == 0 -> NaN
    double window;
    if (L <= a.ws) window = mean;               // src/read.cpp:217-218
    else {
        if (mn < a.clamp) mn = 0.0;             // src/read.cpp:233-234
        window = 100.0 * mn;
    }
    a.mean_q[rid] = mean;
    a.window_q[rid] = window;
    a.passed[rid] = hard_cutoffs(a.p, L, mean, window);
}


}  // namespace flx_phred

// score_phred_regs.hip: the register-history kernel.  *launched = false when the window size has no instantiation
// (the caller then uses the LDS-ring kernel).
int flx_launch_score_phred_regs(flx_ctx *ctx, flx_phred::PhredArgs a, bool *launched);
// score_phred_regs.hip: any window size, both window edges streamed from global memory (used where the LDS ring does not fit)
int flx_launch_score_phred_stream(flx_ctx *ctx, flx_phred::PhredArgs a);
// score_phred_regs.hip: any window size, the trailing edge as a second LDS-DMA stream (nothing of the window stays on chip)
int flx_launch_score_phred_dual(flx_ctx *ctx, flx_phred::PhredArgs a);
