"""flx_pipeline_* (streaming pass 1, replaces the loop of src/main.cpp:70-127 for inputs larger than memory) against
flx_score_batch on the same reads: every per-read and per-child result bit-identical, in submission order, whatever the
chunking — one read per chunk, ragged chunks, empty chunks, chunks that fill a slot exactly — in both scoring modes."""
import numpy as np
import pytest

import _cases
import _pipeline
from filtlong_amd import api

pytestmark = pytest.mark.gpu

FIELDS = ("mean_q", "window_q", "passed", "first", "last", "child_offsets", "child_ranges", "child_mean_q", "child_window_q",
          "child_passed")


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def chunked(strings, sizes):
    out, at, k = [], 0, 0
    while at < len(strings):
        n = sizes[k % len(sizes)]
        out.append(strings[at:at + n])
        at += n
        k += 1
    return out


def same(whole, streamed, what):
    for f in FIELDS:
        a, b = np.asarray(whole[f]), np.asarray(streamed[f])
        assert a.shape == b.shape, "%s: %s has shape %s, expected %s" % (what, f, b.shape, a.shape)
        if a.dtype == np.float64:
            a, b = a.view(np.uint64), b.view(np.uint64)
        assert (a == b).all(), "%s: %s differs" % (what, f)


CHUNKINGS = {"one-read-chunks": [1], "ragged": [7, 1, 0, 19, 3], "one-chunk": [10 ** 9], "empty-first": [0, 0, 11]}


@pytest.mark.parametrize("chunking", sorted(CHUNKINGS))
@pytest.mark.parametrize("pkw", [{}, {"window_size": 1000}, {"min_length": 1000, "min_mean_q": 60.0, "min_window_q": 40.0}],
                         ids=["defaults", "ws1000", "cutoffs"])
def test_phred_stream_matches_batch(ctx, chunking, pkw):
    quals = [q for _, _, q in _cases.phred_reads()]
    p = api.make_params(**pkw)
    plane, offsets, lengths = api.pack_reads(quals)
    whole = ctx.score_reads(plane, offsets, lengths, p, order=api.length_order(lengths))
    streamed = ctx.score_stream(chunked(quals, CHUNKINGS[chunking]), p, chunk_bytes=1 << 20, chunk_reads=256)
    same(whole, streamed, chunking)


@pytest.mark.parametrize("chunking", sorted(CHUNKINGS))
@pytest.mark.parametrize("pkw", [{}, {"trim": True}, {"split": 100}, {"trim": True, "split": 20, "min_length": 500}],
                         ids=["plain", "trim", "split", "trim-split-minlen"])
def test_kmer_stream_matches_batch(ctx, chunking, pkw):
    be = _pipeline.HipBackend(ctx)
    contigs = _cases.synth_reference()
    ks = be.kmers(assembly=contigs)
    seqs = [s for _, s, _ in _cases.kmer_reads(contigs)]
    p = api.make_params(**pkw)
    plane, offsets, lengths = api.pack_reads(seqs)
    whole = ctx.score_reads(plane, offsets, lengths, p, kmers=ks, order=api.length_order(lengths))
    streamed = ctx.score_stream(chunked(seqs, CHUNKINGS[chunking]), p, kmers=ks, chunk_bytes=1 << 20, chunk_reads=256)
    same(whole, streamed, chunking)
    if "split" in pkw or "trim" in pkw:
        assert len(streamed["child_mean_q"]) > 0
    ks.close()


def test_slot_filled_exactly_and_overflow_refused(ctx):
    p = api.make_params()
    quals = [bytes([33 + (i * 7 + j) % 40 for j in range(64)]) for i in range(64)]  # 64 reads of 64 bytes = 4096 bytes, the smallest slot
    plane, offsets, lengths = api.pack_reads(quals)
    whole = ctx.score_reads(plane, offsets, lengths, p)
    same(whole, ctx.score_stream([quals], p, chunk_bytes=4096, chunk_reads=64), "exact fit")
    with pytest.raises(api.FlxError):
        ctx.score_stream([quals + quals[:1]], p, chunk_bytes=4096, chunk_reads=128)   # one read too many bytes
    with pytest.raises(api.FlxError):
        ctx.score_stream([quals + quals[:1]], p, chunk_bytes=8192, chunk_reads=64)   # one read too many reads
    # flx_pipeline_reserve: the slots grow in the middle of the stream, results of the earlier chunks survive
    grown = ctx.score_stream([quals[:3], quals[3:40], quals[40:] + quals, quals[:1]], p, chunk_bytes=4096, chunk_reads=40, grow=True)
    for f in ("mean_q", "window_q"):
        assert (grown[f][:64].view(np.uint64) == whole[f].view(np.uint64)).all()
        assert (grown[f][64:128].view(np.uint64) == whole[f].view(np.uint64)).all()
    assert len(grown["mean_q"]) == 129
    # the context is usable again after a refused chunk
    same(whole, ctx.score_stream(chunked(quals, [5]), p, chunk_bytes=4096, chunk_reads=64), "after an error")


def test_no_chunks_at_all(ctx):
    out = ctx.score_stream([], api.make_params())
    assert len(out["mean_q"]) == 0 and list(out["child_offsets"]) == [0]
