// synth.hip — deterministic synthetic Phred planes generated directly in HBM (bench / test support).
// Same integer-only definition as filtlong_amd/synth.py and oracle/synth.h; stands in
// for the reference's test/make_synthetic_reads.py, which needs PBSIM/wgsim.
#include "flx_internal.h"

namespace {

__device__ __forceinline__ uint64_t mix(uint64_t seed, uint64_t stream, uint64_t read, uint64_t pos) {
    uint64_t z = seed ^ (stream * 0x9E3779B97F4A7C15ULL) ^ (read * 0xBF58476D1CE4E5B9ULL) ^ (pos * 0x94D049BB133111EBULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// one block per read-chunk: grid.x walks reads, each thread writes 16 bytes (4 hashes) per step
// quality profile: per-read centre mu = mu_lo + hash % mu_span, per base q = clamp(mu + a % j1 - j1 / 2 + b % 9 - 4, 1, q_max).
// profile 0: centre 8..25, jitter +-4 +-4, q <= 60; profile 1 ("wide", a realistic ONT spread): centre 3..44,
// jitter +-6 +-4, q <= 50 — a wave's 64 reads then touch ~50 distinct table entries instead of ~34.
__global__ void __launch_bounds__(256) k_synth_qual(uint64_t seed, uint8_t *plane, const uint64_t *offsets,
                                                    const int32_t *lengths, const uint64_t *read_ids, uint64_t n,
                                                    int mu_lo, int mu_span, int j1, int q_max) {
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint64_t gid = read_ids ? read_ids[r] : r;
        const int L = lengths[r];
        const int mu = mu_lo + (int)(mix(seed, 2, gid, 0) % (uint64_t)mu_span);
        uint8_t *dst = plane + offsets[r];
        const int L16 = (L + 15) & ~15;
        for (int p0 = threadIdx.x * 16; p0 < L16; p0 += 256 * 16) {
            uint32_t w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint64_t h = mix(seed, 3, gid, (uint64_t)(p0 >> 2) + g);
                uint32_t packed = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t f = (uint32_t)(h >> (16 * b)) & 0xffffu;
                    int q = mu + (int)((f & 0xff) % (uint32_t)j1) - j1 / 2 + (int)((f >> 8) % 9) - 4;
                    q = q < 1 ? 1 : (q > q_max ? q_max : q);
                    const int pos = p0 + g * 4 + b;
                    const uint32_t byte = pos < L ? (uint32_t)(q + 33) : 0u;  // padding bytes are zero
                    packed |= byte << (8 * b);
                }
                w[g] = packed;
            }
            *reinterpret_cast<uint4 *>(dst + p0) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// Long reads for the k-mer configurations: read = forward substring of the reference genome
// starting at mix(START) % (ref_len - L), per-read substitution rate (mix(ERATE) % 13) percent applied per base by
// an integer threshold, and for 30 % of the reads longer than 3000 one 800-base random junk block at
// 500 + mix(JUNK,1) % (L - 2000)  (exercises window = 0, --trim and --split).
__global__ void __launch_bounds__(256) k_synth_seq(uint64_t seed, uint8_t *plane, const uint64_t *offsets,
                                                   const int32_t *lengths, const uint64_t *read_ids, uint64_t n,
                                                   const uint8_t *ref, uint64_t ref_len, int profile) {
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint64_t gid = read_ids ? read_ids[r] : r;
        const int L = lengths[r];
        uint8_t *dst = plane + offsets[r];
        const uint64_t start = ref_len > (uint64_t)L ? mix(seed, 6, gid, 0) % (ref_len - (uint64_t)L) : 0;
        const uint32_t erate = (uint32_t)(mix(seed, 7, gid, 0) % 13);
        const bool junk = L > 3000 && (mix(seed, 9, gid, 0) % 10) < 3;
        const int jstart = junk ? 500 + (int)(mix(seed, 9, gid, 1) % (uint64_t)(L - 2000)) : -1;
        const bool unrelated = profile == 2 && (mix(seed, 11, gid, 0) % 10) < 3;  // profile 2: 30 % of the reads are random bases
        const int L16 = (L + 15) & ~15;
        for (int p = threadIdx.x; p < L16; p += 256) {
            uint8_t c = 0;
            if (p < L) {
                if (unrelated || (junk && p >= jstart && p < jstart + 800)) {
                    const uint64_t h = mix(seed, 4, gid, (uint64_t)p >> 5);
                    c = "ACGT"[(h >> (2 * (p & 31))) & 3];
                } else {
                    c = ref[(start + (uint64_t)p) % ref_len];
                    const uint64_t h = mix(seed, 8, gid, (uint64_t)p >> 2);
                    const uint32_t f = (uint32_t)(h >> (16 * (p & 3))) & 0xffffu;
                    if ((f & 0xff) % 100 < erate) c = "ACGT"[(f >> 8) & 3];
                }
            }
            dst[p] = c;
        }
    }
}

// Profile 1 (oracle/synth.h: flx_synth_seq_read): a third of the errors insertions and a third deletions of 1-3 bases — one possible
// indel per block of 8 read bases, the reference offset of a block = what the blocks before it consumed.  One workgroup per read:
// every thread owns a run of consecutive blocks, sums what they consume, the 256 sums are scanned in LDS, then the thread walks its
// run again and writes its bases.
struct IndelBlock { bool has, del; int size, off, sp, consumed; };
__device__ __forceinline__ IndelBlock indel_block(uint64_t seed, uint64_t gid, uint64_t block, uint32_t erate) {
    const uint64_t g = mix(seed, 10, gid, block);
    IndelBlock b;
    b.has = (uint32_t)((g & 0xffffu) % 300u) < 16u * erate;
    b.del = ((g >> 16) & 1u) != 0;
    b.size = 1 + (int)((g >> 17) % 3u);
    b.off = (int)((g >> 20) & 7u);
    b.sp = min(b.size, 8 - b.off);
    b.consumed = !b.has ? 8 : (b.del ? 8 + b.size : 8 - b.sp);
    return b;
}
__global__ void __launch_bounds__(256) k_synth_seq_indels(uint64_t seed, uint8_t *plane, const uint64_t *offsets,
                                                          const int32_t *lengths, const uint64_t *read_ids, uint64_t n,
                                                          const uint8_t *ref, uint64_t ref_len) {
    __shared__ unsigned long long run_sum[256];
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint64_t gid = read_ids ? read_ids[r] : r;
        const int L = lengths[r];
        uint8_t *dst = plane + offsets[r];
        const uint64_t start = ref_len > (uint64_t)L ? mix(seed, 6, gid, 0) % (ref_len - (uint64_t)L) : 0;
        const uint32_t erate = (uint32_t)(mix(seed, 7, gid, 0) % 13);
        const bool junk = L > 3000 && (mix(seed, 9, gid, 0) % 10) < 3;
        const int jstart = junk ? 500 + (int)(mix(seed, 9, gid, 1) % (uint64_t)(L - 2000)) : -1;
        const int nblk = ((L + 15) & ~15) >> 3;  // blocks of 8 bytes up to the padded end of the row
        const int per = (nblk + 255) / 256;
        const int b0 = min((int)threadIdx.x * per, nblk), b1 = min(b0 + per, nblk);
        unsigned long long mine = 0;
        for (int b = b0; b < b1; ++b) mine += (unsigned long long)indel_block(seed, gid, (uint64_t)b, erate).consumed;
        __syncthreads();
        run_sum[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long acc = 0;
            for (int t = 0; t < 256; ++t) {
                const unsigned long long v = run_sum[t];
                run_sum[t] = acc;
                acc += v;
            }
        }
        __syncthreads();
        unsigned long long base0 = run_sum[threadIdx.x];
        for (int b = b0; b < b1; ++b) {
            const IndelBlock k = indel_block(seed, gid, (uint64_t)b, erate);
            uint64_t packed = 0;
            for (int j = 0; j < 8; ++j) {
                const int p = 8 * b + j;
                uint8_t c = 0;
                if (p < L) {
                    const bool inserted = k.has && !k.del && j >= k.off && j < k.off + k.sp;
                    if (inserted || (junk && p >= jstart && p < jstart + 800)) {
                        const uint64_t h = mix(seed, 4, gid, (uint64_t)p >> 5);
                        c = "ACGT"[(h >> (2 * (p & 31))) & 3];
                    } else {
                        long long rr = (long long)base0 + j;
                        if (k.has && !k.del && j >= k.off + k.sp) rr -= k.sp;
                        if (k.has && k.del && j >= k.off) rr += k.size;
                        c = ref[(start + (uint64_t)rr) % ref_len];
                        const uint64_t h = mix(seed, 8, gid, (uint64_t)p >> 2);
                        const uint32_t f = (uint32_t)(h >> (16 * (p & 3))) & 0xffffu;
                        if ((f & 0x3fffu) % 300u < erate) c = "ACGT"[(f >> 14) & 3];
                    }
                }
                packed |= (uint64_t)c << (8 * j);
            }
            *reinterpret_cast<uint64_t *>(dst + 8 * b) = packed;
            base0 += (unsigned long long)k.consumed;
        }
    }
}

}  // namespace

extern "C" int flx_synth_seq_profile_dev(flx_ctx *ctx, uint64_t seed, int profile, void *d_plane, uint64_t plane_bytes, const void *d_offsets,
                                         const void *d_lengths, const void *d_read_ids, uint64_t n_reads, const void *d_ref,
                                         uint64_t ref_len) {
    if (!ctx) return FLX_ERR_INVALID;
    if (profile < 0 || profile > 2) return flx_fail(ctx, FLX_ERR_INVALID, "unknown sequence profile %d", profile);
    (void)plane_bytes;
    if (n_reads == 0) return FLX_OK;
    if (!d_ref || ref_len == 0) return flx_fail(ctx, FLX_ERR_INVALID, "reference genome required");
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<uint64_t>(n_reads, 65535ull * 16);
    if (profile == 1)
        hipLaunchKernelGGL(k_synth_seq_indels, dim3(grid), dim3(256), 0, ctx->stream, seed, (uint8_t *)d_plane, (const uint64_t *)d_offsets,
                           (const int32_t *)d_lengths, (const uint64_t *)d_read_ids, n_reads, (const uint8_t *)d_ref, ref_len);
    else
        hipLaunchKernelGGL(k_synth_seq, dim3(grid), dim3(256), 0, ctx->stream, seed, (uint8_t *)d_plane, (const uint64_t *)d_offsets,
                           (const int32_t *)d_lengths, (const uint64_t *)d_read_ids, n_reads, (const uint8_t *)d_ref, ref_len, profile);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLX_OK;
}

extern "C" int flx_synth_seq_dev(flx_ctx *ctx, uint64_t seed, void *d_plane, uint64_t plane_bytes, const void *d_offsets,
                                 const void *d_lengths, const void *d_read_ids, uint64_t n_reads, const void *d_ref,
                                 uint64_t ref_len) {
    return flx_synth_seq_profile_dev(ctx, seed, 0, d_plane, plane_bytes, d_offsets, d_lengths, d_read_ids, n_reads, d_ref, ref_len);
}

extern "C" int flx_synth_qual_profile_dev(flx_ctx *ctx, uint64_t seed, int profile, void *d_plane, uint64_t plane_bytes,
                                          const void *d_offsets, const void *d_lengths, const void *d_read_ids,
                                          uint64_t n_reads) {
    if (!ctx) return FLX_ERR_INVALID;
    if (profile != 0 && profile != 1) return flx_fail(ctx, FLX_ERR_INVALID, "unknown quality profile %d", profile);
    const int mu_lo = profile ? 3 : 8, mu_span = profile ? 42 : 18, j1 = profile ? 13 : 9, q_max = profile ? 50 : 60;
    (void)plane_bytes;
    if (n_reads == 0) return FLX_OK;
    FLX_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned grid = (unsigned)std::min<uint64_t>(n_reads, 65535ull * 16);
    hipLaunchKernelGGL(k_synth_qual, dim3(grid), dim3(256), 0, ctx->stream, seed, (uint8_t *)d_plane,
                       (const uint64_t *)d_offsets, (const int32_t *)d_lengths, (const uint64_t *)d_read_ids, n_reads, mu_lo, mu_span,
                       j1, q_max);
    FLX_HIP(ctx, hipGetLastError());
    FLX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FLX_OK;
}

extern "C" int flx_synth_qual_dev(flx_ctx *ctx, uint64_t seed, void *d_plane, uint64_t plane_bytes,
                                  const void *d_offsets, const void *d_lengths, const void *d_read_ids,
                                  uint64_t n_reads) {
    return flx_synth_qual_profile_dev(ctx, seed, 0, d_plane, plane_bytes, d_offsets, d_lengths, d_read_ids, n_reads);
}
