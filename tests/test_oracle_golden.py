"""The CPU oracle (oracle/flx_oracle.cpp) pinned against the real reference.

Golden vectors under tests/golden/ were produced by tests/golden/make_golden.py from
oracle/_ref/ref_probe (our harness linked against the reference's own read.o/kmers.o/
arguments.o/misc.o) and oracle/_ref/filtlong.  Bit-exact comparison (hex floats).
Also checks the known answers recorded in SURVEY.md §8(c).
"""
import gzip
import hashlib
import json
import math
import os

import numpy as np
import pytest

import _cases
import _oracle

G = _cases.GOLDEN
FIX = _cases.FIXTURES


def same_f(a_hex, b):
    a = float.fromhex(a_hex)
    if math.isnan(a):
        return math.isnan(b)
    return a == b and math.copysign(1, a) == math.copysign(1, b)


def check_read(g, o, where):
    assert g["length"] == o["length"], where
    for k in ("length_score", "mean_q", "window_q"):
        assert same_f(g[k], o[k]), "%s %s: ref %s oracle %s" % (where, k, g[k], float(o[k]).hex())
    for k in ("passed", "first", "last"):
        assert g[k] == o[k], "%s %s" % (where, k)


def check_case(gold_case, reads, kmerset):
    p = _oracle.make_params(**gold_case["params"])
    assert len(gold_case["reads"]) == len(reads)
    for g, (name, seq, qual) in zip(gold_case["reads"], reads):
        o = _oracle.score_read(seq, qual, p, kmerset)
        check_read(g, o, name)
        assert [tuple(x) for x in g["bad"]] == o["bad"], name
        assert [tuple(x) for x in g["child_ranges"]] == o["child_ranges"], name
        assert len(g["children"]) == len(o["children"])
        for gc, oc in zip(g["children"], o["children"]):
            check_read(gc, oc, name + " child")
            assert not gc["bad"] and not gc["child_ranges"]  # children never have grandchildren


def test_known_answers_survey_8c():
    L = _oracle.lib()
    # a1 goldens
    assert L.flo_qscore_to_quality(33 + 10).hex() == "0x1.ccccccccccccdp-1"
    assert L.flo_qscore_to_quality(33 + 20).hex() == "0x1.fae147ae147aep-1"
    assert L.flo_qscore_to_quality(33).hex() == "0x0.0p+0"
    assert L.flo_length_score(5000).hex() == "0x1.9000000000000p+5"
    # a17-a19: Bloom parameters, salts, KAT
    nh, bits = _oracle.C.c_uint32(), _oracle.C.c_uint64()
    L.flo_bloom_parameters(100000000, 0.0001, nh, bits)
    assert (nh.value, bits.value) == (13, 1917295480)
    salts = np.zeros(13, dtype=np.uint32)
    L.flo_bloom_salts(0xA5A5A5A5, 13, salts.ctypes.data)
    assert [int(s) for s in salts] == [0x1B5793D2, 0x81BDFA38, 0xEB8E30D5, 0x45B52496, 0x85C1FE3C, 0x3DACB627,
                                       0x78776869, 0x94A40D1E, 0x5F9BB638, 0x40FB59D5, 0x8174BDB2, 0x0B466EAA,
                                       0x209D29A7]
    idx = [L.flo_bloom_hash(0x12345678, int(s)) % 1917295480 for s in salts]
    assert idx == [723850955, 100786422, 542598492, 57963573, 1546795082, 1234273945, 1253286573, 1570870513,
                   1310589454, 1409716892, 1348413024, 1058179296, 589242733]


def test_encoders():
    L = _oracle.lib()
    for ch, f, r in ((b"A", 0, 3), (b"C", 1, 2), (b"G", 2, 1), (b"T", 3, 0), (b"a", 0, 3), (b"c", 1, 2), (b"g", 2, 1),
                     (b"t", 3, 0), (b"N", 0, 0), (b"-", 0, 0)):
        assert L.flo_base_fwd(ch[0]) == f
        assert L.flo_base_rev(ch[0]) == r << 30
    s = b"ACGTACGTTTGGCCAA"
    assert L.flo_start_kmer_fwd(s) == int("".join("{:02b}".format("ACGT".index(c)) for c in s.decode()), 2)
    rc = _cases.revcomp(s)
    assert L.flo_start_kmer_rev(s) == L.flo_start_kmer_fwd(rc)


def _fixture_set(mode):
    if mode == "phred":
        return None
    ks = _oracle.KmerSet()
    if mode == "asm":
        ks.add_assembly([s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, "test_reference.fasta"))])
    else:
        for f in ("test_reference_1.fastq.gz", "test_reference_2.fastq.gz"):
            ks.add_short_reads([s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, f))])
    return ks


@pytest.fixture(scope="module")
def fixture_sets():
    return {m: _fixture_set(m) for m in ("phred", "asm", "short")}


def test_fixture_set_sizes(fixture_sets):
    # reference stderr: "1 contig, 199,964 16-mers" / "40,000 reads, 204,833 16-mers" 
    assert len(fixture_sets["asm"]) == 199964
    assert len(fixture_sets["short"]) == 204833


def test_reference_fixtures_probe(fixture_sets):
    gold = json.load(open(os.path.join(G, "probe_fixtures.json")))
    n = 0
    for key, case in gold.items():
        fx, mode, _ = key.split("|", 2)
        reads = _oracle.read_fastx(os.path.join(FIX, fx))
        check_case(case, reads, fixture_sets[mode])
        n += 1
    assert n == 3 * (1 + 11 + 11)


def test_survey_goldens_bit_level(fixture_sets):
    reads = {n: (s, q) for n, s, q in _oracle.read_fastx(os.path.join(FIX, "test_sort.fastq"))}
    p = _oracle.make_params()
    exp = {"test_sort_1": ("0x1.6765056776ee5p+6", "0x1.656d069f0b576p+6"),
           "test_sort_2": ("0x1.8bebf07f8e0a2p+6", "0x1.8bd0b23c524bfp+6"),
           "test_sort_3": ("0x1.831476491630dp+6", "0x1.82b0ce9fc8fd7p+6")}
    for n, (m, w) in exp.items():
        o = _oracle.score_read(*reads[n], p)
        assert o["mean_q"] == float.fromhex(m) and o["window_q"] == float.fromhex(w)
    exp = {"test_sort_1": ("0x1.9p+6", "0x1.9p+6"), "test_sort_2": ("0x1.7533333333333p+6", "0x1.4666666666666p+6"),
           "test_sort_3": ("0x1.8b47ae147ae14p+6", "0x1.7999999999999p+6")}
    for mode in ("asm", "short"):
        for n, (m, w) in exp.items():
            o = _oracle.score_read(*reads[n], p, fixture_sets[mode])
            assert o["mean_q"] == float.fromhex(m) and o["window_q"] == float.fromhex(w)
            assert (o["first"], o["last"]) == (0, 5000)
    tr = {n: (s, q) for n, s, q in _oracle.read_fastx(os.path.join(FIX, "test_trim.fastq"))}
    o = _oracle.score_read(*tr["test_trim_3"], _oracle.make_params(trim=True), fixture_sets["asm"])
    assert (o["first"], o["last"]) == (0, 970) and o["window_q"] == float.fromhex("0x1.5ffffffffffffp+6")
    o = _oracle.score_read(*tr["test_trim_2"], _oracle.make_params(trim=True), fixture_sets["asm"])
    assert (o["first"], o["last"]) == (20, 701) and o["child_ranges"] == [(20, 701)]


def test_synth_phred_probe():
    gold = json.load(open(os.path.join(G, "probe_synth_phred.json")))
    reads = _cases.phred_reads()
    for key, case in gold.items():
        assert case["kmers_empty"] is True
        check_case(case, reads, None)


@pytest.fixture(scope="module")
def synth_sets():
    contigs = _cases.synth_reference()
    r1, r2 = _cases.short_read_pairs(contigs)
    a = _oracle.KmerSet(); a.add_assembly(contigs)
    s = _oracle.KmerSet(); s.add_short_reads(r1); s.add_short_reads(r2)
    b = _oracle.KmerSet(); b.add_assembly(contigs); b.add_short_reads(r1)
    return contigs, {"asm": a, "short": s, "both": b}


def test_synth_kmer_probe(synth_sets):
    contigs, sets = synth_sets
    gold = json.load(gzip.open(os.path.join(G, "probe_synth_kmer.json.gz"), "rt"))
    reads = _cases.kmer_reads(contigs)
    n_children = 0
    for key, case in gold.items():
        if key == "__sets__":
            continue
        mode = key.split("|")[0]
        assert case["kmers_empty"] is False
        check_case(case, reads, sets[mode])
        n_children += sum(len(r["children"]) for r in case["reads"])
    assert n_children > 500


def test_kmer_set_membership(synth_sets, fixture_sets):
    """Set equality with the reference via is_kmer_present over (all seen 16-mers + 20k random)."""
    contigs, sets = synth_sets
    gold = json.load(gzip.open(os.path.join(G, "probe_synth_kmer.json.gz"), "rt"))["__sets__"]
    r1, r2 = _cases.short_read_pairs(contigs)
    srcs = {"asm": (sets["asm"], contigs), "short": (sets["short"], r1 + r2),
            "fix_asm": (fixture_sets["asm"], [s for _, s, _ in _oracle.read_fastx(os.path.join(FIX, "test_reference.fasta"))]),
            "fix_short": (fixture_sets["short"],
                          [s for f in ("test_reference_1.fastq.gz", "test_reference_2.fastq.gz")
                           for _, s, _ in _oracle.read_fastx(os.path.join(FIX, f))])}
    rng = np.random.RandomState(5)
    for mode in ("asm", "short", "fix_asm", "fix_short"):   # same order as make_golden.py (shared rng)
        ks, src = srcs[mode]
        seen = _oracle.KmerSet(); seen.add_assembly(src)
        q = np.unique(np.concatenate([seen.dump(), rng.randint(0, 2 ** 32, size=20000, dtype=np.uint64).astype(np.uint32)]))
        g = gold[mode]
        assert hashlib.sha256(q.tobytes()).hexdigest() == g["queries_sha256"]
        present = ks.dump()
        pres_q = present[np.isin(present, q)]
        assert len(ks) == g["n_present"]
        assert hashlib.sha256(pres_q.tobytes()).hexdigest() == g["present_sha256"]


def test_bloom_false_positive_rule():
    """A 16-mer seen only 3 times enters the set when its 13 Bloom bits were all set by OTHER 16-mers before its first
    sighting (src/kmers.cpp:148-155); engineered with the invertible 4-byte hash_ap, confirmed by the real reference."""
    gold = json.load(open(os.path.join(G, "bloom_fp.json")))
    f1, f2, target, control = _cases.bloom_fp_case()
    assert gold["present"][str(target)] is True and gold["present"][str(control)] is False
    ks = _oracle.KmerSet()
    ks.add_short_reads(f1)
    ks.add_short_reads(f2)
    for k, v in gold["present"].items():
        assert ks.contains(int(k)) == v
    assert len(ks) == 1


def test_batch_scorer_equals_per_read_calls():
    """flo_score_plane_mt (the oracle on every host thread over a packed batch: whole-population parity at BASELINE size,
    tests/test_gpu_fullsize.py) is flo_score_read per read — the function the goldens above pin — in Phred mode and in k-mer
    mode with children, any number of threads."""
    from filtlong_amd import synth
    n = 150
    lens = np.maximum(synth.lengths(n, first=7) // 8, 1)
    offs = np.concatenate([[0], np.cumsum((lens.astype(np.int64) + 15) // 16 * 16)]).astype(np.uint64)
    # Phred mode
    plane = np.zeros(int(offs[-1]), np.uint8)
    quals = [synth.qual_read(7 + i, int(L)) for i, L in enumerate(lens)]
    for o, q in zip(offs, quals):
        plane[int(o):int(o) + len(q)] = q
    p = _oracle.make_params(min_length=300, min_mean_q=80.0)
    for th in (1, 3):
        got = _oracle.score_plane_mt(plane, offs[:-1], lens, p, threads=th)
        for i, q in enumerate(quals):
            w = _oracle.score_read(None, q.tobytes(), p)
            assert (w["mean_q"], w["window_q"], w["passed"]) == (got["mean_q"][i], got["window_q"][i], got["passed"][i])
        assert got["n_children"] == 0
    # k-mer mode, --trim --split 200
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, 60000)
    ks = _oracle.KmerSet()
    ks.add_assembly([ref.tobytes()])
    seqs = [synth.seq_read(7 + i, int(L), ref) for i, L in enumerate(lens)]
    for o, q in zip(offs, seqs):
        plane[int(o):int(o) + len(q)] = q
    p = _oracle.make_params(trim=True, split=200, window_size=100)
    for th in (1, 4):
        got = _oracle.score_plane_mt(plane, offs[:-1], lens, p, kmerset=ks, threads=th)
        co = got["child_offsets"]
        assert int(co[-1]) == got["n_children"] > 0
        for i, q in enumerate(seqs):
            w = _oracle.score_read(q.tobytes(), None, p, kmerset=ks)
            assert (w["mean_q"], w["window_q"], w["passed"], w["first"], w["last"]) == (
                got["mean_q"][i], got["window_q"][i], got["passed"][i], got["first"][i], got["last"][i])
            a, b = int(co[i]), int(co[i + 1])
            assert [tuple(r) for r in got["child_ranges"][a:b]] == w["child_ranges"]
            assert [(c["mean_q"], c["window_q"], c["passed"]) for c in w["children"]] == list(
                zip(got["child_mean_q"][a:b], got["child_window_q"][a:b], got["child_passed"][a:b]))
