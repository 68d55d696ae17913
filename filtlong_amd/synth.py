"""Deterministic synthetic long-read generator (host side, numpy).

Same integer-only definition as ``oracle/synth.h`` (C) and ``filtlong_amd/csrc/synth.hip``
(device), so all three produce identical bytes.  It stands in for the
reference's ``test/make_synthetic_reads.py``, which shells out to PBSIM/wgsim at hard-coded
paths (reference test/make_synthetic_reads.py:25-26,66-70) and cannot run here.

Only read *lengths* use floating point (gamma(k=4), mean 10 kbp) and are therefore produced
on the host only and uploaded.
"""
import numpy as np

SEED = 20250919

STREAM_LEN, STREAM_MU, STREAM_QUAL, STREAM_BASE = 1, 2, 3, 4
STREAM_REF, STREAM_START, STREAM_ERATE, STREAM_SUB, STREAM_JUNK = 5, 6, 7, 8, 9
STREAM_INDEL, STREAM_UNREL = 10, 11

_M = np.uint64
_C1 = _M(0x9E3779B97F4A7C15)
_C2 = _M(0xBF58476D1CE4E5B9)
_C3 = _M(0x94D049BB133111EB)


def mix(seed, stream, read, pos):
    """splitmix64 finaliser over seed ^ stream*C1 ^ read*C2 ^ pos*C3 (vectorised, uint64)."""
    with np.errstate(over="ignore"):
        z = (_M(seed) ^ (np.asarray(stream, dtype=np.uint64) * _C1) ^ (np.asarray(read, dtype=np.uint64) * _C2)
             ^ (np.asarray(pos, dtype=np.uint64) * _C3))
        z = (z ^ (z >> _M(30))) * _C2
        z = (z ^ (z >> _M(27))) * _C3
        return z ^ (z >> _M(31))


def lengths(n, first=0, seed=SEED, fixed=None):
    """Read lengths: clamp(round(2500 * sum_{j<4} -ln u_j), 200, 200000); int32[n]."""
    if fixed is not None:
        return np.full(n, fixed, dtype=np.int32)
    reads = np.arange(first, first + n, dtype=np.uint64)
    g = np.zeros(n, dtype=np.float64)
    for j in range(4):
        h = mix(seed, STREAM_LEN, reads, j)
        u = ((h >> _M(11)).astype(np.float64) + 0.5) / 9007199254740992.0
        g += -np.log(u)
    return np.clip(np.rint(2500.0 * g), 200, 200000).astype(np.int32)


# quality profiles (mu_lo, mu_span, jitter1, q_max): 0 = SURVEY §8(d); 1 = "wide" (centre Q3..Q44, q <= 50); csrc/synth.hip
PROFILES = {0: (8, 18, 9, 60), 1: (3, 42, 13, 50)}


def mu(read, seed=SEED, profile=0):
    lo, span, _, _ = PROFILES[profile]
    return (lo + (mix(seed, STREAM_MU, read, 0) % _M(span))).astype(np.int64)


def qual_read(read, length, seed=SEED, profile=0):
    """Phred+33 bytes of one read (uint8[length])."""
    _, _, j1, q_max = PROFILES[profile]
    pos = np.arange(length, dtype=np.uint64)
    h = mix(seed, STREAM_QUAL, read, pos >> _M(2))
    f = (h >> (_M(16) * (pos & _M(3)))) & _M(0xFFFF)
    m = int(mu(np.uint64(read), seed, profile))
    q = m + (f & _M(0xFF)).astype(np.int64) % j1 - j1 // 2 + (f >> _M(8)).astype(np.int64) % 9 - 4
    return (np.clip(q, 1, q_max) + 33).astype(np.uint8)


def bases_read(stream, read, start, length, seed=SEED):
    """Random bases of a stream (uint8[length], ASCII ACGT)."""
    pos = np.arange(start, start + length, dtype=np.uint64)
    h = mix(seed, stream, read, pos >> _M(5))
    idx = (h >> (_M(2) * (pos & _M(31)))) & _M(3)
    return np.frombuffer(b"ACGT", dtype=np.uint8)[idx.astype(np.int64)]


def seq_read(read, length, ref, seed=SEED, profile=0):
    """k-mer-mode long read (uint8[length]) drawn from reference genome `ref` (uint8 array of ASCII ACGT); same definition as
    flx_synth_seq_read in oracle/synth.h and k_synth_seq in csrc/synth.hip.  profile 0: SURVEY §8(d) (substitutions only);
    1: a third of the errors insertions and a third deletions of 1-3 bases; 2: 30 % of the reads unrelated to the reference."""
    L = int(length)
    ref_len = len(ref)
    start = int(mix(seed, STREAM_START, read, 0) % _M(ref_len - L)) if ref_len > L else 0
    erate = int(mix(seed, STREAM_ERATE, read, 0) % _M(13))
    pos = np.arange(L, dtype=np.uint64)
    if profile == 2 and int(mix(seed, STREAM_UNREL, read, 0) % _M(10)) < 3:
        return bases_read(STREAM_BASE, read, 0, L, seed)
    h = mix(seed, STREAM_SUB, read, pos >> _M(2))
    f = (h >> (_M(16) * (pos & _M(3)))) & _M(0xFFFF)
    if profile == 1:
        nblk = (L + 7) // 8
        g = mix(seed, STREAM_INDEL, read, np.arange(nblk, dtype=np.uint64))
        has = ((g & _M(0xFFFF)) % _M(300)) < _M(16 * erate)
        dele = ((g >> _M(16)) & _M(1)).astype(bool)
        size = 1 + ((g >> _M(17)) % _M(3)).astype(np.int64)
        off = ((g >> _M(20)) & _M(7)).astype(np.int64)
        sp = np.minimum(size, 8 - off)
        consumed = np.where(~has, 8, np.where(dele, 8 + size, 8 - sp))
        base0 = np.concatenate([[0], np.cumsum(consumed)[:-1]]) if nblk else np.zeros(0, np.int64)
        b = (pos >> _M(3)).astype(np.int64)
        j = (pos & _M(7)).astype(np.int64)
        ins = has[b] & ~dele[b]
        dl = has[b] & dele[b]
        inserted = ins & (j >= off[b]) & (j < off[b] + sp[b])
        r = base0[b] + j - np.where(ins & (j >= off[b] + sp[b]), sp[b], 0) + np.where(dl & (j >= off[b]), size[b], 0)
        out = ref[(start + r) % ref_len].copy()
        sub = (f & _M(0x3FFF)) % _M(300) < _M(erate)
        out[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[((f >> _M(14)) & _M(3)).astype(np.int64)][sub]
        rnd = bases_read(STREAM_BASE, read, 0, L, seed)
        out[inserted] = rnd[inserted]
    else:
        out = ref[(start + pos.astype(np.int64)) % ref_len].copy()
        sub = (f & _M(0xFF)) % _M(100) < _M(erate)
        out[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[((f >> _M(8)) & _M(3)).astype(np.int64)][sub]
    if L > 3000 and int(mix(seed, STREAM_JUNK, read, 0) % _M(10)) < 3:
        js = 500 + int(mix(seed, STREAM_JUNK, read, 1) % _M(L - 2000))
        je = min(js + 800, L)
        out[js:je] = bases_read(STREAM_BASE, read, js, je - js, seed)
    return out
