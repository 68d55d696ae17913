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
    FLX_STREAM_JUNK = 9   /* junk block presence/offset          */
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

#endif
