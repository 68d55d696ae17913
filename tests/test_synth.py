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
