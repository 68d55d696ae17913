/* synth.h — deterministic synthetic-read generator (TEST / BENCH INFRASTRUCTURE).
 *
 * Integer-only per-base hashing so that the host (numpy in filtlong_amd/synth.py,
 * C here) and the device generator (filtlong_amd/csrc/synth.hip) produce identical
 * bytes.  Follows SURVEY.md §8(d): splitmix64 finaliser over
 *   seed ^ stream*0x9E3779B97F4A7C15 ^ read*0xBF58476D1CE4E5B9 ^ pos*0x94D049BB133111EB,
 * master seed 20250919.  Replaces test/make_synthetic_reads.py of the reference,
 * which shells out to PBSIM/wgsim at hard-coded paths
 * (reference test/make_synthetic_reads.py:25-26,66-70) and cannot run here.
 */
#ifndef FLX_SYNTH_H
#define FLX_SYNTH_H

#include <stdint.h>

#define FLX_SYNTH_SEED 20250919ULL

enum {
    FLX_STREAM_LEN = 1,   /* read lengths (host only)            */
    FLX_STREAM_MU = 2,    /* per-read Phred centre               */
    FLX_STREAM_QUAL = 3,  /* per-base Phred jitter (4 bases/hash) */
    FLX_STREAM_BASE = 4,  /* random bases (32 bases/hash)        */
    FLX_STREAM_REF = 5,   /* reference genome bases              */
    FLX_STREAM_START = 6, /* read start in the reference         */
    FLX_STREAM_ERATE = 7, /* per-read substitution rate          */
    FLX_STREAM_SUB = 8,   /* per-base substitution draw          */
    FLX_STREAM_JUNK = 9,  /* junk block presence/offset          */
    FLX_STREAM_INDEL = 10, /* profile 1: one possible insertion / deletion per block of 8 read bases */
    FLX_STREAM_UNREL = 11  /* profile 2: reads unrelated to the reference */
};

static inline uint64_t flx_mix(uint64_t seed, uint64_t stream, uint64_t read, uint64_t pos) {
    uint64_t z = seed ^ (stream * 0x9E3779B97F4A7C15ULL) ^ (read * 0xBF58476D1CE4E5B9ULL) ^
                 (pos * 0x94D049BB133111EBULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* per-read Phred centre, 8..25 */
static inline int flx_synth_mu(uint64_t seed, uint64_t read) {
    return 8 + (int)(flx_mix(seed, FLX_STREAM_MU, read, 0) % 18);
}

/* Phred+33 byte of base `pos` of read `read` */
static inline uint8_t flx_synth_qual(uint64_t seed, uint64_t read, uint64_t pos, int mu) {
    uint64_t h = flx_mix(seed, FLX_STREAM_QUAL, read, pos >> 2);
    uint32_t f = (uint32_t)(h >> (16 * (pos & 3))) & 0xffffu;
    int q = mu + (int)((f & 0xff) % 9) - 4 + (int)((f >> 8) % 9) - 4;
    if (q < 1) q = 1;
    if (q > 60) q = 60;
    return (uint8_t)(q + 33);
}

/* random base of stream `stream` (FLX_STREAM_BASE for junk, FLX_STREAM_REF for the genome) */
static inline uint8_t flx_synth_base(uint64_t seed, uint64_t stream, uint64_t read, uint64_t pos) {
    uint64_t h = flx_mix(seed, stream, read, pos >> 5);
    return (uint8_t)"ACGT"[(h >> (2 * (pos & 31))) & 3];
}


/* ---- k-mer-mode long reads (C3 / C4 and the two profiles of round 6) -------------------------------------------------------
 * profile 0: forward substring of the reference at mix(START) % (ref_len - L), per-read error rate e = mix(ERATE) % 13
 *   percent applied per base as substitutions, one 800-base junk block in 30 % of the reads longer than 3000.
 * profile 1 ("indels", the round-5 review's item 2): the same e, but a third of the errors are insertions and a third deletions of
 *   1-3 bases: per base a substitution with probability e / 300, and per BLOCK of 8 read bases one indel with probability 16 e / 300
 *   (type, size 1..3 and offset from the block's hash); an insertion puts random bases into the read and consumes no reference, a
 *   deletion skips reference bases.  The reference offset of a block is the sum of what the blocks before it consumed.
 * profile 2 ("unrelated"): profile 0, but 30 % of the reads are random bases with no relation to the reference (contaminants).
 * filtlong_amd/synth.py (numpy) and csrc/synth.hip (device) follow the same integer definition byte for byte. */
typedef struct { int has, del, size, off, sp, consumed; } flx_synth_indel;
static inline flx_synth_indel flx_synth_indel_block(uint64_t seed, uint64_t read, uint64_t block, uint32_t erate) {
    const uint64_t g = flx_mix(seed, FLX_STREAM_INDEL, read, block);
    flx_synth_indel b;
    b.has = (uint32_t)((g & 0xffffu) % 300u) < 16u * erate;
    b.del = (int)((g >> 16) & 1u);
    b.size = 1 + (int)((g >> 17) % 3u);
    b.off = (int)((g >> 20) & 7u);
    b.sp = b.size < 8 - b.off ? b.size : 8 - b.off; /* an insertion ends with its block */
    b.consumed = !b.has ? 8 : (b.del ? 8 + b.size : 8 - b.sp);
    return b;
}
/* the whole read (sequential: a block's reference offset depends on the blocks before it); out[length] */
static inline void flx_synth_seq_read(uint64_t seed, int profile, uint64_t read, int length, const uint8_t *ref, uint64_t ref_len,
                                      uint8_t *out) {
    const uint64_t L = (uint64_t)length;
    const uint64_t start = ref_len > L ? flx_mix(seed, FLX_STREAM_START, read, 0) % (ref_len - L) : 0;
    const uint32_t erate = (uint32_t)(flx_mix(seed, FLX_STREAM_ERATE, read, 0) % 13);
    const int junk = length > 3000 && (flx_mix(seed, FLX_STREAM_JUNK, read, 0) % 10) < 3;
    const int64_t js = junk ? 500 + (int64_t)(flx_mix(seed, FLX_STREAM_JUNK, read, 1) % (uint64_t)(length - 2000)) : -1;
    const int unrelated = profile == 2 && (flx_mix(seed, FLX_STREAM_UNREL, read, 0) % 10) < 3;
    uint64_t base0 = 0;
    flx_synth_indel blk = {0, 0, 0, 0, 0, 8};
    for (int64_t p = 0; p < length; ++p) {
        uint8_t c;
        int64_t r = p;
        int inserted = 0;
        if (profile == 1) {
            const int j = (int)(p & 7);
            if (j == 0) {
                if (p) base0 += (uint64_t)blk.consumed;
                blk = flx_synth_indel_block(seed, read, (uint64_t)p >> 3, erate);
            }
            if (blk.has && !blk.del && j >= blk.off && j < blk.off + blk.sp) inserted = 1;
            r = (int64_t)base0 + j;
            if (blk.has && !blk.del && j >= blk.off + blk.sp) r -= blk.sp;
            if (blk.has && blk.del && j >= blk.off) r += blk.size;
        }
        if (unrelated || inserted || (junk && p >= js && p < js + 800)) {
            c = flx_synth_base(seed, FLX_STREAM_BASE, read, (uint64_t)p);
        } else {
            c = ref[(start + (uint64_t)r) % ref_len];
            const uint64_t h = flx_mix(seed, FLX_STREAM_SUB, read, (uint64_t)p >> 2);
            const uint32_t f = (uint32_t)(h >> (16 * (p & 3))) & 0xffffu;
            if (profile == 1) {
                if ((f & 0x3fffu) % 300u < erate) c = (uint8_t)"ACGT"[(f >> 14) & 3];
            } else if ((f & 0xff) % 100 < erate) {
                c = (uint8_t)"ACGT"[(f >> 8) & 3];
            }
        }
        out[p] = c;
    }
}

#endif
