"""HIP k-mer path vs the reference's golden vectors and the oracle — bit exact.

seam 1: reference 16-mer set (assembly rule, >= 4 copies rule, assembly-then-short-reads, size, membership);
seam 2: coverage marking, mean/window (drift included), first/last, bad ranges -> child ranges, child scores.
"""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

import _cases
import _e2e_checks
import _oracle
import _pipeline
from filtlong_amd import api

pytestmark = pytest.mark.gpu
FIX = _cases.FIXTURES


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def be(ctx):
    return _pipeline.HipBackend(ctx)


@pytest.fixture(scope="module")
def synth(be):
    contigs = _cases.synth_reference()
    r1, r2 = _cases.short_read_pairs(contigs)
    return {"contigs": contigs, "sr": [r1, r2],
            "asm": be.kmers(assembly=contigs), "short": be.kmers(short_files=[r1, r2]),
            "both": be.kmers(assembly=contigs, short_files=[r1])}


@pytest.fixture(scope="module")
def fixture_sets(be):
    asm = [s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, "test_reference.fasta"))]
    sr = [[s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, f))]
          for f in ("test_reference_1.fastq.gz", "test_reference_2.fastq.gz")]
    return {"asm": be.kmers(assembly=asm), "short": be.kmers(short_files=sr), "asm_seqs": asm, "sr_seqs": sr}


def test_set_sizes_match_reference(fixture_sets, synth):
    assert len(fixture_sets["asm"]) == 199964      # "1 contig, 199,964 16-mers"   
    assert len(fixture_sets["short"]) == 204833    # "40,000 reads, 204,833 16-mers"
    gold = json.load(gzip.open(os.path.join(_cases.GOLDEN, "probe_synth_kmer.json.gz"), "rt"))["__sets__"]
    assert len(synth["asm"]) == gold["asm"]["n_present"]
    assert len(synth["short"]) == gold["short"]["n_present"]


def test_membership_matches_reference(fixture_sets, synth):
    """is_kmer_present over (every 16-mer seen at least once + 20k random ones): same answers as the reference."""
    gold = json.load(gzip.open(os.path.join(_cases.GOLDEN, "probe_synth_kmer.json.gz"), "rt"))["__sets__"]
    srcs = {"asm": (synth["asm"], synth["contigs"]), "short": (synth["short"], synth["sr"][0] + synth["sr"][1]),
            "fix_asm": (fixture_sets["asm"], fixture_sets["asm_seqs"]),
            "fix_short": (fixture_sets["short"], fixture_sets["sr_seqs"][0] + fixture_sets["sr_seqs"][1])}
    rng = np.random.RandomState(5)
    for mode in ("asm", "short", "fix_asm", "fix_short"):  # same order / rng stream as make_golden.py
        ks, src = srcs[mode]
        seen = _oracle.KmerSet(); seen.add_assembly(src)
        q = np.unique(np.concatenate([seen.dump(), rng.randint(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)]))
        assert hashlib.sha256(q.tobytes()).hexdigest() == gold[mode]["queries_sha256"]
        pres = q[ks.is_kmer_present(q)]
        assert len(pres) == gold[mode]["n_present"] == len(ks)
        assert hashlib.sha256(pres.tobytes()).hexdigest() == gold[mode]["present_sha256"]


def test_encoders_non_acgt_and_lowercase(ctx, be):
    """'N' acts as A on the forward strand and as T on the reverse strand (src/kmers.cpp:176-220)."""
    seqs = [b"ACGTNNNNacgtACGTTTGGCCAANNAC", b"nnnnnnnnnnnnnnnnnnnn", b"ACGTACGTACGTACG", b"A" * 16]
    ks = be.kmers(assembly=seqs)
    orc = _oracle.KmerSet(); orc.add_assembly(seqs)
    want = orc.dump()
    assert len(ks) == len(want)
    assert ks.is_kmer_present(want).all()


def check_reads(got, gold_reads, where):
    for g, o in zip(gold_reads, got):
        for k in ("mean_q", "window_q"):
            w = float.fromhex(g[k])
            assert (w == o[k]) or (np.isnan(w) and np.isnan(o[k])), "%s %s %s: ref %s hip %s" % (where, g["name"], k, g[k], float(o[k]).hex())
        assert g["passed"] == o["passed"] and g["first"] == o["first"] and g["last"] == o["last"], (where, g["name"])
        assert [tuple(x) for x in g["child_ranges"]] == o["child_ranges"], (where, g["name"])
        for gc, oc in zip(g["children"], o["children"]):
            for k in ("mean_q", "window_q"):
                w = float.fromhex(gc[k])
                assert (w == oc[k]) or (np.isnan(w) and np.isnan(oc[k])), "%s %s child %s" % (where, g["name"], k)
            assert gc["passed"] == oc["passed"] and gc["length"] == oc["length"]


def test_reference_fixtures_probe(be, fixture_sets):
    gold = json.load(open(os.path.join(_cases.GOLDEN, "probe_fixtures.json")))
    n = 0
    for key, case in gold.items():
        fx, mode, _ = key.split("|", 2)
        if mode == "phred":
            continue
        reads = _oracle.read_fastx(os.path.join(FIX, fx))
        got = be.score(reads, case["params"], fixture_sets[mode])
        check_reads(got, case["reads"], key)
        n += 1
    assert n == 3 * 2 * 11


def test_synth_kmer_probe(be, synth):
    gold = json.load(gzip.open(os.path.join(_cases.GOLDEN, "probe_synth_kmer.json.gz"), "rt"))
    reads = _cases.kmer_reads(synth["contigs"])
    n = 0
    for key, case in gold.items():
        if key == "__sets__":
            continue
        mode = key.split("|")[0]
        got = be.score(reads, case["params"], synth[mode])
        check_reads(got, case["reads"], key)
        n += sum(len(r["children"]) for r in case["reads"])
    assert n > 500


def test_empty_set_falls_back_to_phred(ctx, be):
    """A reference that yields no 16-mer leaves Kmers::empty() true -> Phred mode (src/read.cpp:35)."""
    ks = be.kmers(assembly=[b"ACGTACGT"])  # shorter than 16
    assert ks.empty()
    reads = _cases.phred_reads()[:20]
    got = be.score(reads, {}, ks)
    p = _oracle.make_params()
    for (n, s, q), o in zip(reads, got):
        w = _oracle.score_read(None, q, p)
        assert (w["mean_q"] == o["mean_q"]) or (np.isnan(w["mean_q"]) and np.isnan(o["mean_q"]))


def test_call_order_is_enforced(ctx):
    ks = api.Kmers(ctx)
    ks.add_read_fastqs([[b"ACGTACGTACGTACGTACGT"]])
    with pytest.raises(api.FlxError):
        ks.add_assembly_fasta([b"ACGTACGTACGTACGTACGT"])


def test_e2e_all_goldens(be):
    """Every end-to-end golden of the real reference binary (Phred, -a, -1/-2, --trim, --split, weights, cut-offs)."""
    n = _e2e_checks.check_all(be)
    assert n >= 50


def test_device_seq_generator_matches_numpy(ctx):
    import torch
    from filtlong_amd import synth
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, 60000)
    lens = np.array([16, 200, 3500, 9000, 4097, 31, 1, 8, 9, 30000], dtype=np.int32)
    ids = np.array([0, 7, 11, 123456789, 3, 2, 5, 6, 8, 40], dtype=np.uint64)
    plane, offsets, _ = api.pack_reads([b"\0" * int(L) for L in lens])
    d_plane = torch.zeros(plane.nbytes, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_len = torch.from_numpy(lens).cuda()
    d_ids = torch.from_numpy(ids.astype(np.int64)).cuda()
    d_ref = torch.from_numpy(ref).cuda()
    torch.cuda.synchronize()
    for profile in (0, 1, 2):  # SURVEY §8(d); a third of the errors insertions and a third deletions; 30 % unrelated reads
        d_plane.fill_(0xEE)
        ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), plane.nbytes, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(),
                          len(lens), d_ref.data_ptr(), len(ref), profile=profile)
        got = d_plane.cpu().numpy()
        for i, L in enumerate(lens):
            o = int(offsets[i])
            assert (got[o:o + L] == synth.seq_read(int(ids[i]), int(L), ref, profile=profile)).all(), (profile, i)
            assert (got[o + L:o + ((L + 15) & ~15)] == 0).all(), (profile, i)  # the row's padding


def test_bloom_false_positive_rule(be):
    """Engineered Bloom false positive (see tests/test_oracle_golden.py): the device set must contain the 3-sighting
    16-mer whose bits were pre-set, and not the control — exactly like the real reference (tests/golden/bloom_fp.json)."""
    gold = json.load(open(os.path.join(_cases.GOLDEN, "bloom_fp.json")))
    f1, f2, target, control = _cases.bloom_fp_case()
    ks = be.kmers(short_files=[f1, f2])
    q = np.array([int(k) for k in gold["present"]], dtype=np.uint32)
    got = ks.is_kmer_present(q)
    assert [bool(x) for x in got] == [gold["present"][str(int(k))] for k in q]
    assert len(ks) == 1
    # same reads, but the target's first sighting comes BEFORE the helpers: no false positive, empty set
    ks2 = be.kmers(short_files=[[_cases.kmer_to_seq(target)] + f1[:13], f2[:1] + f2[2:3]])
    assert len(ks2) == 0 and not ks2.is_kmer_present(np.array([target], dtype=np.uint32))[0]


def test_long_reads_vs_oracle(be, synth):
    """Reads far longer than one coverage span (4096 positions) and than the window: 30-120 kbp reads stitched from the
    reference with substitutions and junk, plain and with --trim --split."""
    from filtlong_amd import synth as S
    rng = np.random.RandomState(3)
    ref = np.frombuffer(b"".join(synth["contigs"]), dtype=np.uint8)
    reads = []
    for i, L in enumerate([30000, 65536, 65537, 100001, 120000, 4096, 4097, 8192]):
        r = np.concatenate([ref, ref, ref])[:L].copy() if L > len(ref) else ref[:L].copy()
        if L > len(ref):  # tile the reference to reach the length
            r = np.resize(ref, L).copy()
        sub = rng.random_sample(L) < 0.03
        r[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, int(sub.sum()))]
        for _ in range(3):
            js = int(rng.randint(0, L - 900)); jl = int(rng.randint(20, 900))
            r[js:js + jl] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, jl)]
        reads.append(("long%d" % i, r.tobytes(), S.qual_read(i, L).tobytes()))
    orc = _oracle.KmerSet(); orc.add_assembly(synth["contigs"])
    for pkw in (dict(), dict(trim=True, split=300), dict(split=64, window_size=1000)):
        got = be.score(reads, pkw, synth["asm"])
        p = _oracle.make_params(**pkw)
        for (name, s, q), o in zip(reads, got):
            w = _oracle.score_read(s, q, p, orc, cap=65536)
            assert w["mean_q"] == o["mean_q"] and w["window_q"] == o["window_q"], (name, pkw)
            assert (w["first"], w["last"], w["passed"]) == (o["first"], o["last"], o["passed"])
            assert w["child_ranges"] == o["child_ranges"], (name, pkw)
            for wc, oc in zip(w["children"], o["children"]):
                assert wc["mean_q"] == oc["mean_q"] and wc["window_q"] == oc["window_q"] and wc["passed"] == oc["passed"]


def test_every_byte_value_in_reads(be, synth):
    """The device's base encoder is branch free (three SWAR byte comparisons per dword, case folded by clearing bit 5):
    exactly 'C' 'c' 'G' 'g' 'T' 't' may code 1/2/3 and every other byte value — 'A', 'N', IUPAC letters, digits, bytes
    that differ from a base only in bit 7 or bit 4, 0x00, 0xff — codes 0 like the reference's switch (src/kmers.cpp:176-196)."""
    from filtlong_amd import synth as S
    rng = np.random.RandomState(11)
    ref = np.frombuffer(b"".join(synth["contigs"]), dtype=np.uint8)
    reads = []
    lookalikes = np.array([0xC3, 0xE3, 0x53, 0x03, 0xC7, 0xE7, 0x57, 0x07, 0xD4, 0xF4, 0x44, 0x14, 0x41, 0x61, 0x4E, 0x6E, 0, 255],
                          dtype=np.uint8)   # C/G/T with bit 7 set, bit 4 flipped, bits 6-5 cleared; A, a, N, n
    for i, L in enumerate([300, 1000, 4096, 5000, 20000]):
        s0 = int(rng.randint(0, len(ref) - L))
        r = ref[s0:s0 + L].copy()
        pos = rng.randint(0, L, max(4, L // 40))
        r[pos] = rng.randint(0, 256, len(pos)).astype(np.uint8) if i % 2 == 0 else lookalikes[rng.randint(0, len(lookalikes), len(pos))]
        lower = rng.randint(0, L - 64)
        r[lower:lower + 64] = np.frombuffer(bytes(r[lower:lower + 64]).lower(), dtype=np.uint8)
        reads.append(("b%d" % i, r.tobytes(), S.qual_read(100 + i, L).tobytes()))
    # one read made of every byte value in turn, and one of the look-alikes only: no 16-mer of theirs may be found by accident
    reads.append(("allbytes", bytes(range(256)) * 8, S.qual_read(200, 2048).tobytes()))
    reads.append(("lookalikes", bytes(lookalikes.tolist()) * 100, S.qual_read(201, 1800).tobytes()))
    orc = _oracle.KmerSet(); orc.add_assembly(synth["contigs"])
    for pkw in (dict(), dict(trim=True, split=50)):
        got = be.score(reads, pkw, synth["asm"])
        p = _oracle.make_params(**pkw)
        for (name, s, q), o in zip(reads, got):
            w = _oracle.score_read(s, q, p, orc, cap=65536)
            assert w["mean_q"] == o["mean_q"] and w["window_q"] == o["window_q"], (name, pkw, w["mean_q"], o["mean_q"])
            assert (w["first"], w["last"], w["passed"]) == (o["first"], o["last"], o["passed"]), (name, pkw)
            assert w["child_ranges"] == o["child_ranges"], (name, pkw)


def test_prefilter_and_fold_variants_give_identical_results(ctx, be, synth, monkeypatch):
    """The L2 prefilter (kmerset finalize) and the word-level child passes are pure accelerations: without the prefilter
    (FLX_KMER_PREFILTER=0, the path large sets take) and with the bit-level fold (FLX_KMER_FOLD=bits) every output field
    of the synthetic k-mer reads is the same."""
    reads = _cases.kmer_reads(synth["contigs"])
    pkw = dict(trim=True, split=100)
    base = be.score(reads, pkw, be.kmers(assembly=synth["contigs"]))
    monkeypatch.setenv("FLX_KMER_PREFILTER", "0")
    monkeypatch.setenv("FLX_KMER_FOLD", "bits")
    alt = be.score(reads, pkw, be.kmers(assembly=synth["contigs"]))
    def bits(v):  # NaN-safe, bit-exact comparison key (a zero-length read has NaN qualities)
        if isinstance(v, float):
            return np.float64(v).view(np.uint64).item()
        if isinstance(v, dict):
            return {k: bits(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [bits(x) for x in v]
        return v

    assert len(base) == len(alt)
    for (name, _s, _q), a, b in zip(reads, base, alt):
        assert bits(a) == bits(b), name
    # the two data paths of the fold kernels: coverage rows through the per-lane LDS ring (default) and both window edges
    # streamed from global memory (what windows too long for the ring use); also with windows that need the longer rings
    monkeypatch.delenv("FLX_KMER_PREFILTER")
    monkeypatch.delenv("FLX_KMER_FOLD")
    # ... and the three implementations of the children: one lane per child (default: ranges pass, children sorted by length,
    # the parent's recurrence on the child's slice of the row), inside the read's lane at word level (FLX_KMER_FOLD=words), bit
    # by bit (FLX_KMER_FOLD=bits)
    ks = be.kmers(assembly=synth["contigs"])
    for ws in (250, 31, 1, 1500, 5000):
        for split in (100, 32, 1000):
            pk = dict(pkw, window_size=ws, split=split)
            ring = be.score(reads, pk, ks)
            monkeypatch.setenv("FLX_KMER_FOLD_STREAMS", "global")
            glob_ = be.score(reads, pk, ks)
            monkeypatch.delenv("FLX_KMER_FOLD_STREAMS")
            monkeypatch.setenv("FLX_KMER_FOLD_EVENTS", "1")  # the steady state by events (positions where the window's edges differ)
            events = be.score(reads, pk, ks)
            monkeypatch.delenv("FLX_KMER_FOLD_EVENTS")
            for (name, _s, _q), a, b in zip(reads, ring, events):
                assert bits(a) == bits(b), (name, ws, split, "events")
            monkeypatch.setenv("FLX_KMER_FOLD", "words")
            words = be.score(reads, pk, ks)
            monkeypatch.setenv("FLX_KMER_FOLD", "bits")
            bitl = be.score(reads, pk, ks)
            monkeypatch.delenv("FLX_KMER_FOLD")
            assert sum(len(a["child_ranges"]) for a in ring) > 0
            for (name, _s, _q), a, b, c, d in zip(reads, ring, glob_, words, bitl):
                assert bits(a) == bits(b), (name, ws, split, "global streams")
                assert bits(a) == bits(c), (name, ws, split, "words")
                assert bits(a) == bits(d), (name, ws, split, "bits")
    monkeypatch.setenv("FLX_KMER_PREFILTER", "0")
    monkeypatch.setenv("FLX_KMER_FOLD", "bits")
    # the two implementations of the coverage kernel: the wave-level one (pair tables, outermost-member search, far-first spans;
    # default) and round 2's workgroup-per-read kernel (one bitmap lookup per candidate run end) — same coverage, every field
    monkeypatch.delenv("FLX_KMER_PREFILTER")
    monkeypatch.delenv("FLX_KMER_FOLD")
    monkeypatch.setenv("FLX_KMER_COVER", "v2")
    for ks_kw in (dict(assembly=synth["contigs"]), dict(short_files=synth["sr"])):
        monkeypatch.setenv("FLX_KMER_COVER", "v2")
        old_k = be.score(reads, pkw, be.kmers(**ks_kw))
        assert ctx.last_kmer_cover() == "v2"
        monkeypatch.setenv("FLX_KMER_COVER", "w")  # the wave-level kernel of rounds 3-5
        wave_k = be.score(reads, pkw, be.kmers(**ks_kw))
        assert ctx.last_kmer_cover() == "w"
        monkeypatch.delenv("FLX_KMER_COVER")
        new_k = be.score(reads, pkw, be.kmers(**ks_kw))
        assert ctx.last_kmer_cover() == "q"  # round 6: phases with a queue in LDS between them (cover_queue.hip)
        for (name, _s, _q), a, b, c in zip(reads, old_k, new_k, wave_k):
            assert bits(a) == bits(b), name
            assert bits(c) == bits(b), (name, "w")


def test_reads2_gather_matches_oracle(ctx, be, synth):
    """flx_reads2_gather / _dev (src/main.cpp:138-147) against the oracle's loop: parents replaced in place by their children,
    a child's length = end - start; without children the reads themselves; capacity errors report the size needed."""
    import ctypes as C
    from filtlong_amd import _lib
    reads = _cases.kmer_reads(synth["contigs"])
    for pkw, ks in ((dict(trim=True, split=100), synth["asm"]), (dict(split=40), synth["short"]), (dict(), synth["asm"]),
                    (dict(trim=True), None)):
        p = api.make_params(**pkw)
        strings = [(seq if ks is not None else qual) for _, seq, qual in reads]
        plane, offsets, lengths = api.pack_reads(strings)
        sc = ctx.score_reads(plane, offsets, lengths, p, kmers=ks, order=api.length_order(lengths))
        got = ctx.reads2_gather(lengths, sc)
        want = _oracle.reads2_gather(lengths, sc)
        nc = len(sc["child_mean_q"])
        assert (nc > 0) == (ks is not None and bool(pkw))
        for k in ("mean_q", "window_q"):
            assert (got[k].view(np.uint64) == want[k].view(np.uint64)).all(), (pkw, k)
        for k in ("length", "passed", "parent", "child"):
            assert (got[k] == want[k]).all(), (pkw, k)
        n_parents = int((np.diff(sc["child_offsets"].astype(np.int64)) > 0).sum())
        assert len(got["mean_q"]) == len(reads) - n_parents + nc
        if nc:
            # capacity too small: FLX_ERR_CAPACITY and the size needed
            s = _lib.Scores()
            keep = {k: np.ascontiguousarray(sc[k]) for k in sc}
            s.mean_q, s.window_q, s.passed, s.child_offsets = (keep["mean_q"].ctypes.data, keep["window_q"].ctypes.data,
                                                               keep["passed"].ctypes.data, keep["child_offsets"].ctypes.data)
            s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (
                keep["child_ranges"].ctypes.data, keep["child_mean_q"].ctypes.data, keep["child_window_q"].ctypes.data,
                keep["child_passed"].ctypes.data)
            s.n_children = s.child_capacity = nc
            small = len(got["mean_q"]) - 1
            o = [np.zeros(small, t) for t in (np.float64, np.float64, np.int32, np.uint8)]
            n2 = C.c_uint64()
            rc = ctx.L.flx_reads2_gather(ctx.h, len(reads), lengths.ctypes.data, C.byref(s), small, o[0].ctypes.data,
                                         o[1].ctypes.data, o[2].ctypes.data, o[3].ctypes.data, None, None, C.byref(n2))
            assert rc == 5 and n2.value == small + 1
    # no reads at all
    empty = {k: np.zeros(0, t) for k, t in (("mean_q", np.float64), ("window_q", np.float64), ("passed", np.uint8),
                                            ("child_ranges", np.int32), ("child_mean_q", np.float64),
                                            ("child_window_q", np.float64), ("child_passed", np.uint8))}
    empty["child_offsets"] = np.zeros(1, np.uint64)
    assert len(ctx.reads2_gather(np.zeros(0, np.int32), empty)["mean_q"]) == 0


def test_cover_kernel_boundaries_vs_oracle(be, synth):
    """The wave-level coverage kernel at its seams, against the oracle: read lengths around its lane (16) and span (1024) sizes, a
    single error / a run of errors placed exactly at lane and span boundaries, isolated members (one clean 16-mer in noise),
    clean reads (the far-first mode), homopolymers and tandem repeats (every position the same few 16-mers), the reverse strand,
    non-ACGT bytes — each with and without the prefilter, and with --trim / --split so that first / last / children are checked."""
    import os
    contigs = synth["contigs"]
    c0 = contigs[0]
    rng = np.random.RandomState(99)
    reads = []

    def add(name, seq):
        reads.append((name, bytes(seq), b"I" * len(seq)))

    for L in (15, 16, 17, 18, 31, 32, 33, 47, 48, 1007, 1008, 1009, 1023, 1024, 1025, 1039, 1040, 1041, 2047, 2048, 2049, 3071, 3072, 3073, 5000):
        add("clean_%d" % L, c0[100:100 + L])
    for pos in (0, 1, 14, 15, 16, 17, 30, 31, 32, 1007, 1008, 1022, 1023, 1024, 1025, 1038, 1039, 1040, 2047, 2048, 2999):
        r = bytearray(c0[500:3500])
        r[pos] = ord("ACGT"[("ACGT".index(chr(r[pos])) + 1) % 4])  # one substitution exactly there
        add("sub_at_%d" % pos, r)
        r = bytearray(c0[500:3500])
        for k in range(pos, min(pos + 40, len(r)), 3):  # a burst of errors starting there
            r[k] = ord("ACGT"[("ACGT".index(chr(r[k])) + 2) % 4])
        add("burst_at_%d" % pos, r)
    for pos in (0, 5, 16, 1000, 1008, 1015, 1024, 2031):  # one clean 16-mer (or 17, 31, 32 clean bases) in random noise
        for keep in (16, 17, 31, 32):
            r = bytearray(rng.choice(list(b"ACGT"), size=2100).astype(np.uint8).tobytes())
            r[pos:pos + keep] = c0[7000 + pos:7000 + pos + keep]
            add("island_%d_%d" % (pos, keep), r)
    add("homopolymer", b"A" * 2500)
    add("tandem2", b"AC" * 1300)
    add("tandem3", b"ACG" * 900)
    add("noise", rng.choice(list(b"ACGT"), size=3000).astype(np.uint8).tobytes())
    add("revcomp", _cases.revcomp(c0[2000:4500]))
    r = bytearray(c0[9000:12000]); r[100] = ord("N"); r[1024] = ord("n"); r[1500:1510] = b"NNNNNNNNNN"; r[2000:2040] = bytes(r[2000:2040]).lower()
    add("non_acgt", r)
    # a reference that holds the homopolymer and the repeats as well
    ref = contigs + [b"A" * 300, b"AC" * 200, c0[100:400] + b"ACG" * 100]
    orc = _oracle.KmerSet(); orc.add_assembly(ref)
    for pf in ("1", "0"):
        os.environ["FLX_KMER_PREFILTER"] = pf
        try:
            ks = be.kmers(assembly=ref)
        finally:
            del os.environ["FLX_KMER_PREFILTER"]
        assert len(ks) == len(orc)
        for pkw in (dict(), dict(trim=True, split=20), dict(trim=True, split=300, window_size=40)):
            got = be.score(reads, pkw, ks)
            p = _oracle.make_params(**pkw)
            for (name, seq, q), o in zip(reads, got):
                w = _oracle.score_read(seq, q, p, orc, cap=65536)
                assert w["mean_q"] == o["mean_q"] and w["window_q"] == o["window_q"], (name, pf, pkw, w["mean_q"], o["mean_q"])
                assert (w["first"], w["last"], w["passed"]) == (o["first"], o["last"], o["passed"]), (name, pf, pkw)
                assert w["child_ranges"] == o["child_ranges"], (name, pf, pkw)
                for wc, oc in zip(w["children"], o["children"]):
                    assert wc["mean_q"] == oc["mean_q"] and wc["window_q"] == oc["window_q"] and wc["passed"] == oc["passed"], (name, pf, pkw)


def _text_codes(contigs):
    """The assembly as the locus text sees it (filtlong_amd/csrc/kmerset.h: flx_locus): per contig of >= 16 bases its forward
    strand in forward codes, then its reverse strand in the reverse encoder's codes, reversed.  Returns (codes, copy starts)."""
    fwd = {ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2, ord("T"): 3, ord("t"): 3}
    rev = {ord("G"): 1, ord("g"): 1, ord("C"): 2, ord("c"): 2, ord("A"): 3, ord("a"): 3}
    codes, starts = [], []
    for c in contigs:
        if len(c) < 16:
            continue
        starts.append(len(codes))
        codes += [fwd.get(b, 0) for b in c]
        starts.append(len(codes))
        codes += [rev.get(b, 0) for b in reversed(c)]
    return codes, starts


def test_locus_path_vs_oracle(ctx, be, synth, monkeypatch):
    """Members confirmed along the read's locus in the assembly text (round 4): a locus match is sufficient, never necessary, so
    every field must equal the oracle's whatever the read looks like — clean, reverse strand, across two contigs, hanging over a
    contig's end, with indels (the diagonal is lost and found again), junk in front, N runs on either strand, a 16-mer that occurs
    only at ANOTHER locus, repeats (the seed points at the wrong copy), chimeras, and reads that follow the TEXT across the seam
    between two strand copies (those windows are in no contig).  The same with FLX_KMER_LOCUS=0 and with round 2's kernel."""
    rng = np.random.RandomState(404)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return acgt[rng.randint(0, 4, n)].tobytes()

    c0 = synth["contigs"][0]
    # a contig that holds one 16-mer of c0 with its 9th base changed: the only other place where that 16-mer occurs
    w = bytearray(c0[5000:5016]); w[8] = ord("ACGT"[("ACGT".index(chr(w[8])) + 1) % 4])
    c_second = rnd(700) + bytes(w) + rnd(900)
    c_n = bytearray(rnd(4000)); c_n[1000:1040] = b"N" * 40; c_n[2000] = ord("N"); c_n[2500:2503] = b"nRY"; c_n = bytes(c_n)
    c_rep = rnd(300) + b"A" * 200 + rnd(100) + b"AC" * 150 + rnd(100) + (rnd(23) * 40) + rnd(300)
    c_low = rnd(1500).lower()
    # S1 (kmerset.h: safe1): 48 16-mers of c0 with ONE base changed — every offset, every other base — that are members through this
    # contig only; a read that carries those substitutions differs from the text in exactly one base per window, and exactly the
    # windows that start at the changed 16-mers must not be refuted
    near_at = [(6000 + 40 * k, k % 16, k // 16 + 1) for k in range(48)]
    c_near = b""
    for at, j, x in near_at:
        w1 = bytearray(c0[at:at + 16]); w1[j] = ord("ACGT"[("ACGT".index(chr(w1[j])) + x) % 4])
        c_near += rnd(3) + bytes(w1)
    contigs = list(synth["contigs"]) + [c_second, c_n, b"ACGTACGTACGTACG", c_rep, c_low, rnd(16), rnd(17), c_near]
    codes, starts = _text_codes(contigs)
    dec = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []

    def add(name, seq):
        reads.append((name, bytes(seq), b"I" * len(seq)))

    def mutate(seq, rate):
        r = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
        sub = rng.random_sample(len(r)) < rate
        r[sub] = acgt[rng.randint(0, 4, int(sub.sum()))]
        return r.tobytes()

    add("clean", c0[300:7300])
    add("clean_rev", _cases.revcomp(c0[1000:6000]))
    add("errors_3pct", mutate(c0[2000:12000], 0.03))
    add("errors_10pct_rev", mutate(_cases.revcomp(c0[2000:9000]), 0.10))
    add("two_contigs", c0[-1500:] + contigs[1][:1500] if len(contigs[1]) >= 1500 else c0[-1500:] + c_second)
    add("over_the_start", rnd(700) + c0[:2500])
    add("over_the_end", c0[-2500:] + rnd(700))
    add("whole_contig", c_second)
    add("whole_contig_rev", _cases.revcomp(c_second))
    r = bytearray(c0[3000:9000])
    for at in sorted(rng.randint(50, len(r) - 50, 25), reverse=True):  # indels of 1-3 bases
        if rng.rand() < 0.5:
            del r[at:at + int(rng.randint(1, 4))]
        else:
            r[at:at] = rnd(int(rng.randint(1, 4)))
    add("indels", r)
    add("deletion_37", c0[1000:3100] + c0[3137:6000])

    # round 6: reads whose diagonal moves all the time (cover_queue.hip: the kernel with a diagonal per lane, reached through the
    # hand-over of the first kernel) — an indel every ~25 bases like a nanopore read, on either strand, with substitutions on top,
    # across two contigs, over N runs and repeats, of every size up to 20, tiny reads, reads that are clean in one half (the mode has
    # to come and go), and a drift of more than the 16 bases a lane searches within one span
    def indels(seq, every, max_size=3, sub=0.0):
        r = bytearray(mutate(seq, sub) if sub else bytes(seq))
        n_ev = max(1, len(r) // every)
        for at in sorted(rng.randint(5, max(6, len(r) - 5), n_ev), reverse=True):
            size = int(rng.randint(1, max_size + 1))
            if rng.rand() < 0.5:
                del r[at:at + size]
            else:
                r[at:at] = rnd(size)
        return bytes(r)

    add("indels_dense", indels(c0[2000:9000], 25, sub=0.03))
    add("indels_dense_rev", indels(_cases.revcomp(c0[4000:12000]), 25, sub=0.03))
    add("indels_very_dense", indels(c0[1000:5000], 12))
    add("indels_sizes_to_20", indels(c0[3000:12000], 150, max_size=20))
    add("indels_two_contigs", indels(c0[-2500:] + (contigs[1][:2500] if len(contigs[1]) >= 2500 else c_second), 30))
    add("indels_n_runs", indels(c_n[300:3700], 30))
    add("indels_repeats", indels(c_rep, 20))
    add("indels_short", indels(c0[7000:7200], 20))
    add("indels_tiny", indels(c0[7000:7040], 15))
    add("indels_then_clean", indels(c0[1000:5000], 25) + c0[5000:9000])
    add("clean_then_indels", c0[1000:5000] + indels(c0[5000:9000], 25))
    add("indels_clean_indels_rev", _cases.revcomp(indels(c0[1000:3500], 25) + c0[3500:7000] + indels(c0[7000:9500], 25)))
    r = bytearray(c0[2000:8000])
    for k in range(12):  # twelve deletions of 3 bases inside 600 bases: the diagonal drifts 36 bases within one span
        del r[2000 + 45 * k:2003 + 45 * k]
    add("drift_36_in_a_span", r)
    r = bytearray(c0[2000:8000])
    for k in range(12):
        r[2000 + 48 * k:2000 + 48 * k] = rnd(3)
    add("drift_back_36_in_a_span", r)
    add("indels_junk_indels", indels(c0[1000:4000], 25) + rnd(800) + indels(c0[4800:8000], 25))
    add("junk_first", rnd(2300) + c0[4000:8000])
    add("junk_middle", c0[1000:2500] + rnd(800) + c0[3300:6000])
    add("junk_only", rnd(5000))
    add("n_runs", c_n[500:3500])
    add("n_runs_rev", _cases.revcomp(c_n[500:3500]))
    add("n_as_a", c_n[500:3500].replace(b"N", b"A"))                       # on the forward strand N codes like A ...
    add("n_as_t_rev", _cases.revcomp(c_n[500:3500].replace(b"N", b"T")))   # ... in the reverse k-mers like T (src/kmers.cpp:199-219)
    r = bytearray(c0[4000:6500]); r[1008] = w[8]
    add("second_locus_only", r)                                            # [1000, 1016) is a member only through c_second
    add("second_locus_rev", _cases.revcomp(bytes(r)))
    r = bytearray(c0[5800:8100])
    for at, j, x in near_at:
        r[at - 5800 + j] = ord("ACGT"[("ACGT".index(chr(c0[at + j])) + x) % 4])
    add("one_base_away_members", r)
    add("one_base_away_members_rev", _cases.revcomp(bytes(r)))
    r2 = bytearray(c0[5800:8100])
    for at, j, x in near_at:  # the same places with ANOTHER base: one mismatch per window, none of them a member
        r2[at - 5800 + j] = ord("ACGT"[("ACGT".index(chr(c0[at + j])) + x % 3 + 1) % 4])
    add("one_base_away_others", r2)
    add("repeats", c_rep)
    add("repeats_rev", _cases.revcomp(c_rep))
    add("homopolymer", b"A" * 3000)
    add("lowercase", c_low[100:1400].upper())
    add("lowercase_as_is", c_low[100:1400])
    add("chimera", c0[100:1800] + _cases.revcomp(c_n[200:1900]) + c0[9000:10500])
    add("sixteen", c0[77:93])
    add("fifteen", c0[77:92])
    add("seventeen_rev", _cases.revcomp(c0[77:94]))
    for k, t0 in enumerate(starts[1:8]):  # follow the text across the seam between two strand copies
        lo_, hi_ = max(0, t0 - 700 - 16 * k), min(len(codes), t0 + 900 + k)
        add("across_seam_%d" % k, dec[np.array(codes[lo_:hi_], dtype=np.int64)].tobytes())
        add("across_seam_short_%d" % k, dec[np.array(codes[max(0, t0 - 15 - k):t0 + 17], dtype=np.int64)].tobytes())
    add("long_mixed", mutate(c0[:15000], 0.02) + rnd(1200) + mutate(_cases.revcomp(c0[3000:14000]), 0.06))

    orc = _oracle.KmerSet(); orc.add_assembly(contigs)
    ks = be.kmers(assembly=contigs)
    assert len(ks) == len(orc)

    def bits(v):
        if isinstance(v, float):
            return np.float64(v).view(np.uint64).item()
        if isinstance(v, dict):
            return {k: bits(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [bits(x) for x in v]
        return v

    for pkw in (dict(), dict(trim=True, split=20), dict(trim=True, split=250, window_size=100)):
        got = be.score(reads, pkw, ks)
        assert ctx.last_kmer_locus() and ctx.last_kmer_cover() == "q"
        # (the reads with indels reach the kernel with a diagonal per lane; a read that follows one diagonal never does)
        assert 10 <= ctx.last_kmer_handed_over() < len(reads) - 20, ctx.last_kmer_handed_over()
        p = _oracle.make_params(**pkw)
        for (name, seq, q), o in zip(reads, got):
            wnt = _oracle.score_read(seq, q, p, orc, cap=65536)
            assert wnt["mean_q"] == o["mean_q"] or (np.isnan(wnt["mean_q"]) and np.isnan(o["mean_q"])), (name, pkw, wnt["mean_q"], o["mean_q"])
            assert wnt["window_q"] == o["window_q"] or (np.isnan(wnt["window_q"]) and np.isnan(o["window_q"])), (name, pkw)
            assert (wnt["first"], wnt["last"], wnt["passed"]) == (o["first"], o["last"], o["passed"]), (name, pkw)
            assert wnt["child_ranges"] == o["child_ranges"], (name, pkw)
            for wc, oc in zip(wnt["children"], o["children"]):
                assert wc["mean_q"] == oc["mean_q"] and wc["window_q"] == oc["window_q"] and wc["passed"] == oc["passed"], (name, pkw)
        monkeypatch.setenv("FLX_KMER_LOCUS", "0")
        plain = be.score(reads, pkw, ks)
        assert not ctx.last_kmer_locus()
        monkeypatch.delenv("FLX_KMER_LOCUS")
        monkeypatch.setenv("FLX_KMER_COVER", "v2")
        v2 = be.score(reads, pkw, ks)
        monkeypatch.setenv("FLX_KMER_COVER", "w")
        wv = be.score(reads, pkw, ks)
        assert ctx.last_kmer_cover() == "w" and ctx.last_kmer_locus()
        monkeypatch.setenv("FLX_KMER_COVER", "q2")  # EVERY read through the kernel with a diagonal per lane, not only those handed to it
        q2 = be.score(reads, pkw, ks)
        assert ctx.last_kmer_cover() == "q2" and ctx.last_kmer_handed_over() == len(reads)
        monkeypatch.delenv("FLX_KMER_COVER")
        for (name, _s, _q), a, b, c, d, e2 in zip(reads, got, plain, v2, wv, q2):
            assert bits(a) == bits(b), (name, pkw, "FLX_KMER_LOCUS=0")
            assert bits(a) == bits(c), (name, pkw, "v2")
            assert bits(a) == bits(d), (name, pkw, "w")
            assert bits(a) == bits(e2), (name, pkw, "q2")
    # the same set built without S1: nothing may differ
    monkeypatch.setenv("FLX_KMER_SAFE1", "0")
    ks_plain = be.kmers(assembly=contigs)
    monkeypatch.delenv("FLX_KMER_SAFE1")
    for pkw in (dict(), dict(trim=True, split=20)):
        for (name, _s, _q), a, b in zip(reads, be.score(reads, pkw, ks), be.score(reads, pkw, ks_plain)):
            assert bits(a) == bits(b), (name, pkw, "FLX_KMER_SAFE1=0")
    # the synthetic set, every read, three ways; and a set of assembly + short reads keeps the assembly's text
    sreads = _cases.kmer_reads(synth["contigs"])
    for ks2 in (synth["asm"], synth["both"]):
        a = be.score(sreads, dict(trim=True, split=100), ks2)
        assert ctx.last_kmer_locus()
        monkeypatch.setenv("FLX_KMER_LOCUS", "0")
        b = be.score(sreads, dict(trim=True, split=100), ks2)
        monkeypatch.delenv("FLX_KMER_LOCUS")
        for (name, _s, _q), x, y in zip(sreads, a, b):
            assert bits(x) == bits(y), name
    be.score(sreads, {}, synth["short"])
    assert ctx.last_kmer_locus()  # a short-read set is a text too: its de Bruijn graph cut into paths (pathtext.hip)


def test_path_text_of_short_read_sets_vs_oracle(ctx, be, monkeypatch):
    """A set without an assembly becomes a text by cutting the members' de Bruijn graph into paths (csrc/pathtext.hip): every
    member is a window of exactly one piece — also at branching 15-mers (a segment that occurs three times, so the pairing of
    entering and leaving members is wrong for some passages and the diagonal must be found again inside the span), on cycles
    (tandem repeats: AC.., a 23-mer unit) and on the self-loop of a homopolymer.  Reads through all of that, every field against
    the oracle and against the kernel without the text and round 2's kernel; assembly + short reads in one set as well."""
    rng = np.random.RandomState(808)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rnd(n):
        return acgt[rng.randint(0, 4, n)].tobytes()

    seg = rnd(300)
    unit23 = rnd(23)
    genome = (rnd(3000) + seg + rnd(2500) + seg + rnd(1200) + b"AC" * 120 + rnd(800) + b"A" * 90 + rnd(700) + unit23 * 12 + rnd(1500) +
              _cases.revcomp(seg) + rnd(2000))
    # 100-mers at every 7th position, alternately as they are and reverse-complemented, split over two files: every inner 16-mer
    # is seen about a dozen times
    files = [[], []]
    for k, at in enumerate(range(0, len(genome) - 100, 7)):
        piece = genome[at:at + 100]
        files[k % 2].append(piece if k % 3 else _cases.revcomp(piece))
    orc = _oracle.KmerSet(); orc.add_short_reads(files[0]); orc.add_short_reads(files[1])
    ks = be.kmers(short_files=files)
    assert len(ks) == len(orc) and len(ks) > 20000

    def mutate(seq, rate):
        r = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
        sub = rng.random_sample(len(r)) < rate
        r[sub] = acgt[rng.randint(0, 4, int(sub.sum()))]
        return r.tobytes()

    reads = []

    def add(name, seq):
        reads.append((name, bytes(seq), b"I" * len(seq)))

    add("whole", genome)
    add("whole_rev", _cases.revcomp(genome))
    add("inner", genome[150:-150])
    add("errors_2pct", mutate(genome[100:-100], 0.02))
    add("errors_8pct_rev", mutate(_cases.revcomp(genome[100:-100]), 0.08))
    for k, at in enumerate((2900, 3250, 5700, 6050, 7200, 7450, 8300, 9100, 9400)):  # in and out of the repeats
        add("piece_%d" % k, genome[at:at + 1100 + 37 * k])
        add("piece_rev_%d" % k, _cases.revcomp(genome[at + 5:at + 900 + 11 * k]))
    add("repeat_only", seg + seg + seg)
    add("ac", b"AC" * 700)
    add("poly_a", b"A" * 1500)
    add("poly_t", b"T" * 1500)
    add("unit23", unit23 * 60)
    add("junk", rnd(3000))
    add("junk_then_genome", rnd(1100) + genome[1000:4000])
    add("sixteen", genome[5000:5016])

    def bits(v):
        if isinstance(v, float):
            return np.float64(v).view(np.uint64).item()
        if isinstance(v, dict):
            return {k: bits(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [bits(x) for x in v]
        return v

    both = be.kmers(assembly=[genome[:6000]], short_files=files)
    orc_both = _oracle.KmerSet(); orc_both.add_assembly([genome[:6000]]); orc_both.add_short_reads(files[0]); orc_both.add_short_reads(files[1])
    assert len(both) == len(orc_both)
    # the text at order 24 (default: paths of the sequences' 24-mers) and at order 16 (paths of the members themselves)
    monkeypatch.setenv("FLX_KMER_TEXT_ORDER", "16")
    ks16 = be.kmers(short_files=files)
    monkeypatch.delenv("FLX_KMER_TEXT_ORDER")
    # short sequences (below 24 bases) and a set whose members mostly lie in no 24-mer at all
    tiny = [genome[a:a + 20] for a in range(0, 4000, 3)] * 2
    orc_tiny = _oracle.KmerSet(); orc_tiny.add_short_reads(tiny)
    ks_tiny = be.kmers(short_files=[tiny])
    assert len(ks_tiny) == len(orc_tiny) > 1000
    for kset, oset in ((ks, orc), (both, orc_both), (ks16, orc), (ks_tiny, orc_tiny)):
        for pkw in (dict(), dict(trim=True, split=40), dict(trim=True, split=300, window_size=64)):
            got = be.score(reads, pkw, kset)
            assert ctx.last_kmer_locus()
            p = _oracle.make_params(**pkw)
            for (name, seq, q), o in zip(reads, got):
                w = _oracle.score_read(seq, q, p, oset, cap=65536)
                assert w["mean_q"] == o["mean_q"] and w["window_q"] == o["window_q"], (name, pkw, w["mean_q"], o["mean_q"])
                assert (w["first"], w["last"], w["passed"]) == (o["first"], o["last"], o["passed"]), (name, pkw)
                assert w["child_ranges"] == o["child_ranges"], (name, pkw)
            monkeypatch.setenv("FLX_KMER_LOCUS", "0")
            plain = be.score(reads, pkw, kset)
            assert not ctx.last_kmer_locus()
            monkeypatch.delenv("FLX_KMER_LOCUS")
            monkeypatch.setenv("FLX_KMER_COVER", "v2")
            v2 = be.score(reads, pkw, kset)
            monkeypatch.setenv("FLX_KMER_COVER", "w")
            wv = be.score(reads, pkw, kset)
            monkeypatch.setenv("FLX_KMER_COVER", "q2")
            q2 = be.score(reads, pkw, kset)
            monkeypatch.delenv("FLX_KMER_COVER")
            for (name, _s, _q), a, b, c, d, e2 in zip(reads, got, plain, v2, wv, q2):
                assert bits(a) == bits(b), (name, pkw, "FLX_KMER_LOCUS=0")
                assert bits(a) == bits(c), (name, pkw, "v2")
                assert bits(a) == bits(d), (name, pkw, "w")
                assert bits(a) == bits(e2), (name, pkw, "q2")


def test_set_without_room_for_the_pair_table(ctx, be, synth, monkeypatch):
    """flx_kmerset_finalize without the 1 GiB pair table (the allocation failed; FLX_KMER_PAIRTABLE=0 plays that): no error, the
    set scores through the kernel that asks the bitmap, every field the same (advisor, round 3)."""
    reads = _cases.kmer_reads(synth["contigs"])
    pkw = dict(trim=True, split=100)
    want = be.score(reads, pkw, synth["asm"])
    monkeypatch.setenv("FLX_KMER_PAIRTABLE", "0")
    ks = be.kmers(assembly=synth["contigs"])
    ks2 = be.kmers(short_files=synth["sr"])
    monkeypatch.delenv("FLX_KMER_PAIRTABLE")
    assert len(ks) == len(synth["asm"]) and len(ks2) == len(synth["short"])
    got = be.score(reads, pkw, ks)
    assert not ctx.last_kmer_locus()

    def key(o):
        return (np.float64(o["mean_q"]).view(np.uint64).item(), np.float64(o["window_q"]).view(np.uint64).item(), o["first"], o["last"], o["passed"],
                o["child_ranges"])

    for (name, _s, _q), a, b in zip(reads, want, got):
        assert key(a) == key(b), name
    want2 = be.score(reads, pkw, synth["short"])
    got2 = be.score(reads, pkw, ks2)
    for (name, _s, _q), a, b in zip(reads, want2, got2):
        assert key(a) == key(b), name


def test_integer_grid_folds_vs_oracle(ctx, monkeypatch):
    """Round 5: the window recurrence's steady state on the integer grid (score_kmer.hip: GridTab; tools/sim_fold_grid.cpp is the same
    logic on the host).  Reads whose coverage is engineered to sit ON the regime's edges — junk and clean stretches alternating with
    periods from 16 to 3000 bases, so that the window count hovers around ws / 2, ws / 4, ..., falls to 0 and climbs back, reaches
    the full window and leaves it — against the ORACLE (the plain recurrence on the CPU) bit for bit, parents and children, for the
    default window and for window sizes with one wide group (500, 128: a power of two, d is exact), narrow groups (100), none that
    pays (1000, 31: the FP kernel must have run) — and the grid kernel against the FP kernel (FLX_KMER_FOLD_GRID=0) on all of it."""
    from filtlong_amd import synth as S
    rng = np.random.RandomState(11)
    ref = S.bases_read(S.STREAM_REF, 0, 0, 400_000)
    oset = _oracle.KmerSet()
    oset.add_assembly([ref.tobytes()])
    ks = api.Kmers(ctx)
    ks.add_assembly_fasta([ref.tobytes()])
    ks.finalize()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = []
    periods = [(16, 16), (17, 15), (40, 40), (100, 20), (20, 100), (125, 125), (126, 124), (250, 250), (500, 300), (62, 190), (31, 219),
               (15, 235), (8, 242), (3000, 900), (1000, 16), (249, 1), (1, 16), (64, 64), (33, 31), (700, 700)]
    for k in range(400):
        L = int(rng.randint(600, 9000))
        start = int(rng.randint(0, len(ref) - L))
        seq = ref[start:start + L].copy()
        a, b = periods[k % len(periods)]
        if k >= 2 * len(periods):  # later reads: the same periods with jitter, then sparse substitutions on top
            a, b = max(1, a + int(rng.randint(-3, 4))), max(1, b + int(rng.randint(-3, 4)))
        pos = int(rng.randint(0, a + b))
        while pos < L:
            pos += a
            e = min(L, pos + b)
            if pos < L:
                seq[pos:e] = acgt[rng.randint(0, 4, e - pos)]
            pos = e
        if k % 3 == 2:
            sub = rng.rand(L) < 0.02
            seq[sub] = acgt[rng.randint(0, 4, int(sub.sum()))]
        reads.append(seq.tobytes())
    plane, offsets, lengths = api.pack_reads(reads)
    order = api.length_order(lengths)
    keys = ("mean_q", "window_q", "passed", "first", "last", "child_offsets", "child_ranges", "child_mean_q", "child_window_q", "child_passed")

    def same(a, b, what):
        for k in keys:
            x, y = np.ascontiguousarray(a[k]), np.ascontiguousarray(b[k])
            if x.dtype == np.float64:
                x, y = x.view(np.uint64), np.asarray(y, dtype=np.float64).view(np.uint64)
            assert x.shape == y.shape and (x == np.asarray(y, dtype=x.dtype).reshape(x.shape)).all(), (what, k)

    for ws, grid in ((250, True), (500, True), (128, True), (64, True), (100, None), (96, None), (333, None), (1000, False), (31, False), (2047, None), (7, False)):
        for pkw in (dict(window_size=ws), dict(window_size=ws, trim=True, split=max(32, ws // 2))):
            monkeypatch.delenv("FLX_KMER_FOLD_GRID", raising=False)
            dev = ctx.score_reads(plane, offsets, lengths, api.make_params(**pkw), kmers=ks, order=order)
            if grid is not None:
                assert ctx.last_kmer_fold_grid() == grid, (ws, pkw)
            want = _oracle.score_plane_mt(plane, offsets, lengths, _oracle.make_params(**pkw), kmerset=oset, child_cap=300 * len(reads))
            same(dev, want, ("oracle", ws, sorted(pkw)))
            monkeypatch.setenv("FLX_KMER_FOLD_GRID", "0")
            fp = ctx.score_reads(plane, offsets, lengths, api.make_params(**pkw), kmers=ks, order=order)
            assert not ctx.last_kmer_fold_grid()
            same(dev, fp, ("FP kernel", ws, sorted(pkw)))
    monkeypatch.delenv("FLX_KMER_FOLD_GRID", raising=False)
    ks.close()
