// cover_common.h — device helpers and the argument block shared by the wave-level coverage kernels of k-mer mode
// (score_kmer.hip: k_kmer_cover_w, the kernel of rounds 3-5; cover_queue.hip: k_kmer_cover_q, round 6).
// Reference semantics: src/read.cpp:43-58 (rolling 2-bit 16-mer, one set lookup per position, bases i-15..i marked on a hit).
#pragma once
#include "flx_internal.h"
#include "kmerset.h"

#ifndef FLX_COVER_THREADS
#define FLX_COVER_THREADS 256  // threads per workgroup of the wave-level cover kernels (their waves are independent)
#endif
#ifndef FLX_LOCUS_SEEDS
#define FLX_LOCUS_SEEDS 4  // seed attempts per span of the locus path
#endif
#ifndef FLX_LOCUS_TAIL
#define FLX_LOCUS_TAIL 3  // lanes without a known member behind the last one that has one, from which the span seeds again
#endif
// 8 waves per SIMD (63 registers instead of 68): 14.8 -> 14.3 ms per 1e10 positions; 9 and 10 are slower again (profiles/r04_microbench.txt)
#ifndef FLX_COVER_WAVES_PER_EU
#define FLX_COVER_WAVES_PER_EU 8
#endif
#define FLX_COVER_OCC __attribute__((amdgpu_waves_per_eu(FLX_COVER_WAVES_PER_EU, FLX_COVER_WAVES_PER_EU)))

// The 2-bit codes of the four bases of a dword (src/kmers.cpp:176-196: C/c 1, G/g 2, T/t 3, anything else 0) packed into 8
// bits, first base (lowest byte) in the top two.  Branch free (a switch per base compiles into divergent control flow — half
// of the cover kernel's run time once) and, since the cover kernel turned out to be bound by its vector instructions (round 4:
// 0.81 per position, a quarter of them here), by table: bits 1..3 of a letter tell A, C, T and G apart (0, 1, 2, 3 — in either
// case), v_perm_b32 looks up the letter that index stands for and the byte is that letter or it is none of them; a second
// v_perm_b32 turns the index into the code and v_dot4_u32_u8 packs the four.  12 instructions per dword (three SWAR comparisons: 30).
__device__ __forceinline__ uint32_t codes4(uint32_t w) {
    const uint32_t idx = (w >> 1) & 0x07070707u;
    const uint32_t letter = __builtin_amdgcn_perm(0u, 0x47544341u, idx);  // A C T G for 0 1 2 3, 0x00 for 4..7
    const uint32_t d = (letter ^ w) & 0xDFDFDFDFu;                         // a zero byte: that letter, upper or lower case
    const uint32_t nz = ((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d;             // bit 7 of every byte that is NOT zero
    const uint32_t sel = ((nz >> 5) & 0x04040404u) | idx;                  // anything else: an index from 4 on
    const uint32_t code = __builtin_amdgcn_perm(0u, 0x02030100u, sel);     // A 0, C 1, T 3, G 2; 0 from 4 on
    return __builtin_amdgcn_udot4(code, 0x01041040u, 0u, false);           // byte 0 * 64 + byte 1 * 16 + byte 2 * 4 + byte 3
}

// 16 bytes of the read plane.  The plane is streamed once and the coverage rows are written once: non-temporal, so that they do
// not push the prefilter out of the L2 (14.26 -> 13.8 ms per 1e10 positions, profiles/r04_microbench.txt)
__device__ __forceinline__ uint4 flx_plane16(const uint8_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 v;
    v.x = __builtin_nontemporal_load(&q->x); v.y = __builtin_nontemporal_load(&q->y);
    v.z = __builtin_nontemporal_load(&q->z); v.w = __builtin_nontemporal_load(&q->w);
    return v;
}
// A lane's left / right neighbour's value, with lane 0's / lane 63's coming from a wave-uniform carry: ONE DPP move (wave_shr:1 /
// wave_shl:1: a lane without a source keeps the destination's old value, which is set to the carry).  __shfl_up + `if (lane == 0)`
// compiles to ds_bpermute_b32 + v_cndmask_b32 with the lane mask held in an SGPR pair — twenty of those per span kept four such pairs
// alive in a kernel that is short of scalar registers (78 at 8 waves per SIMD: they were spilled to vector lanes and read back with
// two v_readlane each), and sent thirty operations per span through the LDS crossbar (round 5; tools/rejected/score_kmer_ablations.patch
// has the old form as FLX_COVER_NO_DPP).
__device__ __forceinline__ uint32_t flx_from_left(uint32_t x, uint32_t lane0) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane0, (int)x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t flx_from_right(uint32_t x, uint32_t lane63) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lane63, (int)x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

struct CoverArgs {
    const uint8_t *plane;
    const uint64_t *offsets;
    const int32_t *lengths;
    const uint32_t *order;
    uint64_t n_reads;
    const uint8_t *exact15;
    const uint8_t *pre11;
    flx_locus loc;
    uint32_t *cov;
    const uint64_t *cov_off;
    int32_t *count, *first, *last;
    // k_kmer_cover_q: a mark per slot of the processing order — the reads the first kernel hands to the one with a diagonal per lane
    uint8_t *redo;
};
// (a pointer out of an integer: without the global address space on it every access would be a flat load with a 64-bit address
// built in vector registers — one more vector instruction per access)
#define FLX_GLOBAL_PTR(elem) const elem __attribute__((address_space(1))) *
#define FLX_KARG_PTR(elem, field) ((FLX_GLOBAL_PTR(elem))(uint64_t)(uintptr_t)(a.field))

// cover_queue.hip: the cover kernel of round 6 (sets with a text); returns a HIP launch error through the context
// (every_read_to_second: FLX_KMER_COVER=q2, tests — every read goes straight to the kernel with a diagonal per lane)
int flx_cover_queue_launch(flx_ctx *ctx, const CoverArgs &args, bool has_prefilter, unsigned grid, bool every_read_to_second);
