"""The synthetic generator: numpy (filtlong_amd/synth.py) == C (oracle/synth.h)."""
import numpy as np

import _oracle
from filtlong_amd import synth


def test_mix_matches_c():
    L = _oracle.lib()
    for args in [(synth.SEED, 1, 0, 0), (synth.SEED, 3, 12345, 77), (7, 9, 2 ** 40 + 3, 2 ** 33 + 1)]:
        assert int(synth.mix(*args)) == L.flo_synth_mix(*args)


def test_qual_and_bases_match_c():
    L = _oracle.lib()
    for read, n in [(0, 1), (5, 1000), (123456789, 4097)]:
        out = np.zeros(n, dtype=np.uint8)
        L.flo_synth_qual(synth.SEED, read, n, out.ctypes.data)
        assert (out == synth.qual_read(read, n)).all()
        assert out.min() >= 34 and out.max() <= 93
        L.flo_synth_bases(synth.SEED, synth.STREAM_REF, read, 13, n, out.ctypes.data)
        assert (out == synth.bases_read(synth.STREAM_REF, read, 13, n)).all()


def test_lengths_distribution():
    ln = synth.lengths(200000)
    assert ln.min() >= 200 and ln.max() <= 200000
    assert abs(ln.mean() - 10000) < 100 and abs(ln.std() - 5000) < 150
    assert (synth.lengths(10, fixed=5000) == 5000).all()


def test_seq_profiles_match_c():
    """The k-mer-mode long reads, every profile (0: SURVEY §8(d), substitutions only; 1: a third of the errors insertions and a third
    deletions of 1-3 bases; 2: 30 % of the reads unrelated to the reference): numpy == C, byte for byte — and the profiles do what they
    say (the device generator is held against numpy in tests/test_gpu_kmer.py)."""
    L = _oracle.lib()
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, 200000)
    for profile in (0, 1, 2):
        for read in (0, 1, 2, 3, 5, 17, 40, 41, 99, 1234567):
            for n in (1, 7, 8, 9, 200, 3001, 12345):
                out = np.zeros(n, dtype=np.uint8)
                L.flo_synth_seq(synth.SEED, profile, read, n, ref.ctypes.data, len(ref), out.ctypes.data)
                assert (out == synth.seq_read(read, n, ref, profile=profile)).all(), (profile, read, n)
    # profile 1: indels shift the read against the reference (profile 0 never does)
    shifted = 0
    for read in range(30):
        erate = int(synth.mix(synth.SEED, synth.STREAM_ERATE, read, 0) % np.uint64(13))
        a, b = synth.seq_read(read, 2000, ref, profile=0), synth.seq_read(read, 2000, ref, profile=1)
        if erate == 0:
            assert (a == b).all()
        elif (a[-200:] != b[-200:]).mean() > 0.5:
            shifted += 1
    assert shifted >= 15
    # profile 2: about 30 % of the reads share nothing with their profile-0 form beyond chance
    unrelated = sum((synth.seq_read(r, 1000, ref, profile=0) != synth.seq_read(r, 1000, ref, profile=2)).mean() > 0.5 for r in range(200))
    assert 40 <= unrelated <= 80
