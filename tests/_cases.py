"""Seeded test inputs shared by the golden-vector generator (tests/golden/make_golden.py), the
oracle tests and the GPU parity tests.  Inputs are regenerated deterministically; only the
reference's *outputs* are stored under tests/golden/.
"""
import os

import numpy as np

from filtlong_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = os.path.join(GOLDEN, "ref_fixtures")  # copies of the reference's small test data files

EDGE_LENGTHS = [0, 1, 2, 15, 16, 17, 31, 32, 33, 99, 100, 101, 249, 250, 251, 252, 255, 256, 257, 265, 266, 267,
                499, 500, 501, 511, 512, 513, 1000, 1023, 1024, 1025, 4095, 4096, 4097]


def phred_reads(seed=7, n_gamma=40, weird=True):
    """(name, seq, qual) triples for the Phred-only path: edge lengths + gamma lengths (+ arbitrary bytes)."""
    reads = []
    rid = 0
    for L in EDGE_LENGTHS:
        q = synth.qual_read(rid, L, seed)
        reads.append(("e%d_%d" % (rid, L), b"A" * L, q.tobytes()))
        rid += 1
    for L in synth.lengths(n_gamma, first=1000, seed=seed):
        q = synth.qual_read(rid, int(L), seed)
        reads.append(("g%d" % rid, b"C" * int(L), q.tobytes()))
        rid += 1
    if weird:
        rng = np.random.RandomState(seed)
        for L in (300, 777, 2048, 5000):
            # every byte value, including < 33 (negative q) and >= 128 (negative signed char); SURVEY §7.2
            q = rng.randint(0, 256, size=L).astype(np.uint8)
            reads.append(("w%d" % rid, b"G" * L, q.tobytes()))
            rid += 1
        # constant-quality and ramp reads
        reads.append(("k%d" % rid, b"T" * 1200, bytes([33 + 20]) * 1200)); rid += 1
        reads.append(("z%d" % rid, b"T" * 900, bytes([33]) * 900)); rid += 1   # q = 0 -> quality 0.0 exactly
        reads.append(("r%d" % rid, b"T" * 1500, bytes([33 + (i % 41) for i in range(1500)]))); rid += 1
    return reads


PHRED_PARAM_SETS = [
    dict(),
    dict(window_size=1),
    dict(window_size=16),
    dict(window_size=100, min_length=200),
    dict(window_size=1000, min_mean_q=85.0),
    dict(window_size=251, min_window_q=60.0, max_length=20000),
    dict(window_size=4096),
]


def synth_reference(n_contigs=3, contig_len=20000, seed=11):
    """Random reference contigs (list of bytes)."""
    return [synth.bases_read(synth.STREAM_REF, c, 0, contig_len, seed).tobytes() for c in range(n_contigs)]


def fasta_bytes(contigs, width=70, prefix="contig"):
    out = []
    for i, c in enumerate(contigs):
        out.append((">%s_%d some comment" % (prefix, i + 1)).encode())
        for j in range(0, len(c), width):
            out.append(c[j:j + width])
    return b"\n".join(out) + b"\n"


_COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def kmer_reads(contigs, seed=13, n=60):
    """Long reads drawn from the reference with substitutions, junk blocks, N's, lowercase, reverse strands.

    Exercises coverage marking, first/last covered base, trim and split (reference src/read.cpp:43-142).
    """
    rng = np.random.RandomState(seed)
    reads = []
    lens = list(EDGE_LENGTHS[:12]) + [int(x) for x in np.clip(synth.lengths(n, first=5000, seed=seed) // 4, 40, 12000)]
    for i, L in enumerate(lens):
        c = contigs[i % len(contigs)]
        L = min(L, len(c) - 1)
        s = int(rng.randint(0, len(c) - L)) if L > 0 else 0
        r = bytearray(c[s:s + L])
        if i % 3 == 1:
            r = bytearray(revcomp(bytes(r)))
        err = [0.0, 0.01, 0.03, 0.08, 0.15][i % 5]
        if L:
            mask = rng.random_sample(L) < err
            sub = rng.randint(0, 4, size=L)
            for p in np.nonzero(mask)[0]:
                r[p] = b"ACGT"[sub[p]]
        if L > 1200 and i % 2 == 0:  # junk block(s)
            for _ in range(1 + i % 3):
                jl = int(rng.randint(30, 700))
                js = int(rng.randint(0, L - jl))
                r[js:js + jl] = bytes(rng.choice(list(b"ACGT"), size=jl).astype(np.uint8))
        if L > 100 and i % 7 == 0:  # N's and lowercase (src/kmers.cpp:176-220: anything else -> 0)
            for p in rng.randint(0, L, size=5):
                r[p] = ord("N")
            r[10:40] = bytes(r[10:40]).lower()
        if L > 400 and i % 4 == 3:  # junk head / tail for --trim
            h = int(rng.randint(1, 60)); t = int(rng.randint(1, 60))
            r[:h] = bytes(rng.choice(list(b"ACGT"), size=h).astype(np.uint8))
            r[L - t:] = bytes(rng.choice(list(b"ACGT"), size=t).astype(np.uint8))
        q = synth.qual_read(9000 + i, L, seed).tobytes()
        reads.append(("k%d" % i, bytes(r), q))
    # a read with no covered base at all, longer than any split value: (0,L) bad range, no children
    reads.append(("junk_all", bytes(rng.choice(list(b"ACGT"), size=3000).astype(np.uint8)),
                  synth.qual_read(9999, 3000, seed).tobytes()))
    return reads


KMER_PARAM_SETS = [
    dict(),
    dict(trim=True),
    dict(split=100),
    dict(trim=True, split=250),
    dict(trim=True, split=40, window_size=100, min_length=100),
    dict(split=1),
    dict(trim=True, split=16, min_mean_q=50.0),
    dict(trim=True, split=1000000),
]


def short_read_pairs(contigs, seed=17, depth=12, rl=100):
    """Error-free 100 bp pairs covering the contigs at ~depth x (so most 16-mers pass the >= 4 copies rule,
    reference src/kmers.cpp:142-166), plus a tail of low-coverage regions that do not."""
    rng = np.random.RandomState(seed)
    r1, r2 = [], []
    for ci, c in enumerate(contigs):
        n = len(c) * depth // (2 * rl)
        if ci == len(contigs) - 1:
            n //= 6  # low coverage contig: many 16-mers seen 1-3 times only
        for _ in range(n):
            s = int(rng.randint(0, len(c) - 450))
            r1.append(c[s:s + rl])
            r2.append(revcomp(c[s + 350:s + 350 + rl]))
    return r1, r2


def fastq_bytes(seqs, prefix="sr", qchar=b"I"):
    out = []
    for i, s in enumerate(seqs):
        out += [("@%s_%d" % (prefix, i)).encode(), s, b"+", qchar * len(s)]
    return b"\n".join(out) + b"\n"


def long_fastq_bytes(reads):
    out = []
    for name, s, q in reads:
        out += [b"@" + name.encode() + b" extra comment", s, b"+", q]
    return b"\n".join(out) + b"\n"


def fasta_cr_at_eof_bytes(reads):
    """FASTA whose last line is a bare carriage return without a newline ("...\\n\\r<EOF>"): kseq keeps that '\\r' as the last
    base of the last record — ks_getuntil2 returns at the end of the stream before its '\\r' strip (reference
    src/kseq.h:141-146) — while every '\\r' in front of a newline is dropped."""
    out = bytearray()
    for name, s, _q in reads:
        out += b">" + name.encode() + b"\r\n" + s + b"\r\n"
    return bytes(out[:-2]) + b"\n\r"


def c1_fastq_bytes(n=10_000, length=5000, seed=synth.SEED):
    """BASELINE.json configs[0] (C1): n reads x 5 kbp, Phred-only, as a FASTQ file (seq is irrelevant in Phred mode)."""
    seq = (b"ACGT" * (length // 4 + 1))[:length]
    out = []
    for i in range(n):
        out += [("@c1_%d" % i).encode(), seq, b"+", synth.qual_read(i, length, seed).tobytes()]
    return b"\n".join(out) + b"\n"


# ------------------------------------------------------------------------------------------------------------
# engineered Bloom-filter false positive (reference src/kmers.cpp:142-166 + src/bloom_filter.h)
# ------------------------------------------------------------------------------------------------------------
BLOOM_SALTS = [0x1B5793D2, 0x81BDFA38, 0xEB8E30D5, 0x45B52496, 0x85C1FE3C, 0x3DACB627, 0x78776869, 0x94A40D1E,
               0x5F9BB638, 0x40FB59D5, 0x8174BDB2, 0x0B466EAA, 0x209D29A7]
BLOOM_BITS = 1917295480
_M32 = 0xFFFFFFFF


def bloom_index(kmer, i):
    s = BLOOM_SALTS[i]
    h = s ^ (~(((s << 11) & _M32) + (kmer ^ (s >> 5))) & _M32)
    return (h & _M32) % BLOOM_BITS


def bloom_preimage(index, i=0):
    """A 16-mer whose i-th Bloom hash lands on bit `index` (hash_ap is invertible for 4-byte keys)."""
    s = BLOOM_SALTS[i]
    h = index  # index < table size < 2^32, so h = index is a valid hash value
    x = (~(h ^ s)) & _M32                 # = (s << 11) + (kmer ^ (s >> 5))   (mod 2^32)
    return (((x - ((s << 11) & _M32)) & _M32) ^ (s >> 5)) & _M32


def kmer_to_seq(k):
    return bytes(b"ACGT"[(k >> (30 - 2 * j)) & 3] for j in range(16))


def bloom_fp_case():
    """Short-read 'files' (lists of 16 bp reads) in which the 16-mer TARGET is seen exactly 3 times but enters the set
    because every one of its 13 Bloom bits was already set by other 16-mers at its first sighting (so it starts
    counting at 2), while CONTROL — also seen exactly 3 times — does not.  Returns (file1, file2, target, control)."""
    target = 0x1B2D3F4A
    control = 0x6C0FFEE5
    # helper j reaches the target's j-th bit through a DIFFERENT hash function (salt j+1), so it is another 16-mer
    helpers = [bloom_preimage(bloom_index(target, j), (j + 1) % 13) for j in range(13)]
    assert all(bloom_index(h, (j + 1) % 13) == bloom_index(target, j) for j, h in enumerate(helpers))
    assert target not in helpers and len(set(helpers)) == 13
    file1 = [kmer_to_seq(h) for h in helpers]                 # first sightings of the helpers: they insert
    file1 += [kmer_to_seq(target), kmer_to_seq(control)]      # 1st sighting of both
    file2 = [kmer_to_seq(target), kmer_to_seq(control)] * 2   # 2nd and 3rd sightings
    return file1, file2, target, control


def odd_fastq_bytes(reads, width=61):
    """The same records in the formats klib's kseq tolerates (reference src/kseq.h:176-224): sequences and qualities
    wrapped over several lines, CRLF line ends on some records, blank lines between records, tabs before comments."""
    out = bytearray()
    for i, (name, s, q) in enumerate(reads):
        nl = b"\r\n" if i % 2 else b"\n"
        out += b"@" + name.encode() + (b"\tcomment with spaces" if i % 3 == 0 else b"") + nl
        for j in range(0, len(s), width):
            out += s[j:j + width] + nl
        out += b"+" + (name.encode() if i % 4 == 1 else b"") + nl
        for j in range(0, len(q), width + 7):
            out += q[j:j + width + 7] + nl
        if i % 5 == 2:
            out += b"\n"
    return bytes(out)
