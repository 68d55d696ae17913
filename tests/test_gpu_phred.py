"""HIP Phred scoring (flx_score_batch) vs the oracle and the reference's golden vectors — bit exact.

Covers: every PHRED_PARAM_SETS window size (ring kernel and the direct fallback), edge lengths
(0, 1, ws-1, ws, ws+1, 16-byte boundaries), arbitrary byte values (negative q), ragged batches,
the reference's own fixture, hard cut-offs, processing order on/off.
"""
import json
import math
import os

import numpy as np
import pytest

import _cases
import _oracle
from filtlong_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context(0)
    yield c
    c.close()


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def assert_same_f64(got, want, what):
    g, w = bits(got), bits(want)
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert (nan_g == nan_w).all(), what + ": NaN pattern differs"
    bad = np.nonzero((g != w) & ~nan_w)[0]
    assert len(bad) == 0, "%s: %d mismatches, first at %d: got %s want %s" % (
        what, len(bad), bad[0], float(got[bad[0]]).hex(), float(want[bad[0]]).hex())


def score_hip(ctx, reads, pkw, use_order=True):
    plane, offsets, lengths = api.pack_reads([q for _, _, q in reads])
    order = api.length_order(lengths) if use_order else None
    return ctx.score_reads(plane, offsets, lengths, api.make_params(**pkw), order=order)


def select_kernel(monkeypatch, kernel):
    """default = register-history kernel where the window size has one (1..623, else the dual-slot kernel), table layout chosen from a
    sample of the data; "plain" / "private" force its table layout; "ring" / "direct" force the older kernels."""
    if kernel in ("ring", "direct", "stream", "dual"):
        monkeypatch.setenv("FLX_PHRED_KERNEL", kernel)
    elif kernel in ("private", "plain"):
        monkeypatch.setenv("FLX_PHRED_TABLES", kernel)


@pytest.mark.parametrize("kernel", ["default", "plain", "private", "ring", "direct", "stream", "dual"])
def test_golden_synth_phred(ctx, kernel, monkeypatch):
    select_kernel(monkeypatch, kernel)
    gold = json.load(open(os.path.join(_cases.GOLDEN, "probe_synth_phred.json")))
    reads = _cases.phred_reads()
    for key, case in gold.items():
        o = score_hip(ctx, reads, case["params"])
        want_m = np.array([float.fromhex(r["mean_q"]) for r in case["reads"]])
        want_w = np.array([float.fromhex(r["window_q"]) for r in case["reads"]])
        assert_same_f64(o["mean_q"], want_m, key + " mean_q")
        assert_same_f64(o["window_q"], want_w, key + " window_q")
        assert list(o["passed"]) == [r["passed"] for r in case["reads"]], key
        assert (o["first"] == -1).all() and (o["last"] == -1).all() and len(o["child_ranges"]) == 0


def test_reference_fixture_phred(ctx):
    reads = _oracle.read_fastx(os.path.join(_cases.FIXTURES, "test_sort.fastq"))
    o = score_hip(ctx, reads, {})
    exp = [("0x1.6765056776ee5p+6", "0x1.656d069f0b576p+6"), ("0x1.8bebf07f8e0a2p+6", "0x1.8bd0b23c524bfp+6"),
           ("0x1.831476491630dp+6", "0x1.82b0ce9fc8fd7p+6")]
    for i, (m, w) in enumerate(exp):
        assert o["mean_q"][i] == float.fromhex(m) and o["window_q"][i] == float.fromhex(w)


@pytest.mark.parametrize("kernel", ["plain", "private"])
def test_regs_kernel_every_window_remainder(ctx, kernel, monkeypatch):
    """Register-history kernel: window sizes around its default range and every remainder (ws % 16 = 0..15 drives the byte funnel and the position of
    the first full window inside a piece), lengths around ws and every 16/64-byte boundary, arbitrary bytes (bank-private
    tables: reads with bytes >= 128 take the redo path), batches that are not a multiple of 64, both processing orders."""
    select_kernel(monkeypatch, kernel)
    rng = np.random.RandomState(31)
    lens = list(range(0, 70)) + list(range(230, 330)) + [383, 384, 385, 511, 512, 513, 1023, 1024, 1025, 5000, 5003]
    lens += [int(x) for x in rng.randint(200, 6000, 150)]
    reads = []
    for i, L in enumerate(lens):
        if i % 9 == 0:
            q = rng.randint(0, 256, size=L).astype(np.uint8)        # any byte
        elif i % 9 == 1:
            q = rng.randint(33, 127, size=L).astype(np.uint8)       # the full FASTQ alphabet
        elif i % 9 == 2 and L > 4:
            # bytes at the top of the 7-bit range (the bank-private tables hold rows 0..127) and just above it, also
            # alone, at either end, or next to 0xff / 0xfe
            q = rng.randint(120, 127, size=L).astype(np.uint8)
            where = (i // 9) % 4
            if where == 0:
                q[L // 2] = 127
            elif where == 1:
                q[0] = 127
            elif where == 2:
                q[-1] = 127
                q[-2] = 0xff
            else:
                q[L // 3] = 0xfe   # 0xfe + 1 = 0xff: still no carry, but >= 128 anyway
        else:
            q = synth.qual_read(5000 + i, int(L), 17)
        reads.append(("x%d" % i, b"", q.tobytes()))
    for ws in list(range(240, 256)) + [47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 191, 192, 200, 256, 257, 300, 319, 320]:
        pkw = dict(window_size=ws)
        p = _oracle.make_params(**pkw)
        want = [_oracle.score_read(None, q, p) for _, _, q in reads]
        for use_order in (True, False):
            o = score_hip(ctx, reads, pkw, use_order)
            assert_same_f64(o["mean_q"], np.array([w["mean_q"] for w in want]), "ws %d mean_q" % ws)
            assert_same_f64(o["window_q"], np.array([w["window_q"] for w in want]), "ws %d window_q" % ws)
            assert (o["passed"] == np.array([w["passed"] for w in want], dtype=np.uint8)).all()


def _instantiation_reads(ws, rng):
    """A small read set built around one window size: lengths at ws and around it, at the 16/64/128-byte boundaries behind
    it (ring piece, LDS-DMA chunk, cache line), one and several ring revolutions later, and a few long ones; bytes from the
    synthetic profile, the FASTQ alphabet, any byte value (>= 128: redo path of the bank-private tables) and the top of the
    7-bit range."""
    lens = {0, 1, 15, 16, 17, max(ws - 1, 0), ws, ws + 1, ws + 2, ws + 15, ws + 16, ws + 17, ws + 31, ws + 33, ws + 63, ws + 64,
            ws + 65, ws + 127, ws + 128, ws + 129, 2 * ws - 1, 2 * ws, 2 * ws + 1, 2 * ws + 16, 3 * ws + 5,
            (ws // 16 + 4) * 16 - 1, (ws // 16 + 4) * 16, (ws // 16 + 4) * 16 + 1, 2 * (ws // 16 + 4) * 16 + 3,
            ((ws + 127) // 128) * 128, ((ws + 127) // 128) * 128 + 128}
    lens = sorted(lens) + [int(x) for x in rng.randint(ws + 1, ws + 700, 10)] + [int(x) for x in rng.randint(2 * ws + 200, 4 * ws + 3000, 8)]
    reads = []
    for i, L in enumerate(lens):
        kind = i % 7
        if kind == 0:
            q = rng.randint(0, 256, size=L).astype(np.uint8)
        elif kind == 1:
            q = rng.randint(33, 127, size=L).astype(np.uint8)
        elif kind == 2 and L > 2:
            q = rng.randint(120, 128, size=L).astype(np.uint8)
            q[(L * 2) // 3] = 0xff if i % 2 else 127
        else:
            q = synth.qual_read(90_000 + 131 * ws + i, int(L), 17 if kind == 3 else 5)
        reads.append(("w%d_%d" % (ws, i), b"", q.tobytes()))
    return reads


REGS_MAX_A = 38  # the register-history kernel serves window sizes 1 .. 623; from 624 on the dual-slot kernel runs (round 4)


def _instantiation_window_sizes():
    """Every instantiation A = ws / 16 = 0..38 of flx_score_phred_regs with the remainders B = ws % 16 in {0, 1, 15}, the
    switch points of the wave occupancy classes (A = 7|8, 15|16) and of the plain-only rings (31|32) with every remainder, and
    the end of the kernel's range (A = 38) — plus the first window sizes of the dual-slot kernel behind it."""
    wss = set()
    for A in range(REGS_MAX_A + 1):
        for B in (0, 1, 15):
            if 16 * A + B >= 1:
                wss.add(16 * A + B)
    for A in (7, 8, 15, 16, 31, 32, 38, 39):
        wss.update(range(16 * A, 16 * A + 16))
    wss.discard(0)
    return sorted(wss)


@pytest.mark.parametrize("part", range(8))
def test_regs_kernel_every_instantiation(ctx, part, monkeypatch):
    """All 39 instantiations of the register-history kernel (round 3 had 63, the rings for ws >= 624 — VGPR +
    AGPR, one wave per SIMD — gave way to the dual-slot kernel): each A is its own unroll / ring-index / register-allocation
    product of the compiler.  For every window size of
    `_instantiation_window_sizes()`: one oracle pass, then the plain and (A < 32) the bank-private tables, both processing
    orders — and the test asserts which kernel ran (the dual-slot kernel from ws 624 on; 1007 / 1008 ride along)."""
    wss = _instantiation_window_sizes() + [1007, 1008]
    mine = wss[part::8]
    rng = np.random.RandomState(4000 + part)
    seen_a = set()
    for ws in mine:
        reads = _instantiation_reads(ws, rng)
        pkw = dict(window_size=ws)
        p = _oracle.make_params(**pkw)
        want = [_oracle.score_read(None, q, p) for _, _, q in reads]
        wm = np.array([w["mean_q"] for w in want])
        ww = np.array([w["window_q"] for w in want])
        wp = np.array([w["passed"] for w in want], dtype=np.uint8)
        layouts = ("plain", "private") if ws < 512 else ("plain",)  # (rings from A = 32 on and the dual-slot kernel: plain tables only)
        for li, layout in enumerate(layouts):
            monkeypatch.setenv("FLX_PHRED_TABLES", layout)
            for use_order in ((True, False) if li == 0 else (True,)):
                o = score_hip(ctx, reads, pkw, use_order)
                k = ctx.last_phred_kernel()
                if ws // 16 <= REGS_MAX_A:
                    assert k == ("flx_score_phred_regs_private" if layout == "private" else "flx_score_phred_regs"), (ws, k)
                else:
                    assert k == "flx_score_phred_dual", (ws, k)
                tag = "ws %d (A %d, B %d) %s order=%s" % (ws, ws // 16, ws % 16, layout, use_order)
                assert_same_f64(o["mean_q"], wm, tag + " mean_q")
                assert_same_f64(o["window_q"], ww, tag + " window_q")
                assert (o["passed"] == wp).all(), tag
        seen_a.add(ws // 16)
    assert len(mine) > 0 and seen_a


def test_regs_instantiation_list_is_complete():
    wss = _instantiation_window_sizes()
    for A in range(REGS_MAX_A + 1):
        have = {w % 16 for w in wss if w // 16 == A}
        assert {1, 15} <= have and (A == 0 or 0 in have), A
    assert 623 in wss and 624 in wss


@pytest.mark.parametrize("ws", [250, 7, 64, 333])
def test_random_batch_vs_oracle(ctx, ws):
    """3000 seeded reads with gamma lengths: ragged waves, many rounds, both orders."""
    n = 3000
    lens = np.maximum(synth.lengths(n, first=777, seed=99) // 3, 1)
    reads = [("r%d" % i, b"", synth.qual_read(10_000 + i, int(L), 99).tobytes()) for i, L in enumerate(lens)]
    pkw = dict(window_size=ws, min_length=500, min_mean_q=80.0, min_window_q=55.0)
    p = _oracle.make_params(**pkw)
    want = [_oracle.score_read(None, q, p) for _, _, q in reads]
    for use_order in (True, False):
        o = score_hip(ctx, reads, pkw, use_order)
        assert_same_f64(o["mean_q"], np.array([w["mean_q"] for w in want]), "mean_q")
        assert_same_f64(o["window_q"], np.array([w["window_q"] for w in want]), "window_q")
        assert (o["passed"] == np.array([w["passed"] for w in want], dtype=np.uint8)).all()


def test_empty_batch_and_all_empty_reads(ctx):
    o = score_hip(ctx, [], {})
    assert len(o["mean_q"]) == 0
    o = score_hip(ctx, [("a", b"", b""), ("b", b"", b"")], dict(min_length=1))
    assert np.isnan(o["mean_q"]).all() and np.isnan(o["window_q"]).all() and list(o["passed"]) == [0, 0]


@pytest.mark.parametrize("profile", [0, 1])
def test_device_generator_matches_numpy(ctx, profile):
    """flx_synth_qual_profile_dev (HBM) == filtlong_amd/synth.py (host), so full-size runs score known bytes."""
    import torch
    lens = np.array([1, 15, 16, 17, 250, 1000, 4097, 33], dtype=np.int32)
    ids = np.array([0, 5, 9, 123456789, 2, 77, 1 << 33, 3], dtype=np.uint64)
    plane, offsets, _ = api.pack_reads([b"\0" * int(L) for L in lens])
    d_plane = torch.zeros(plane.nbytes, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(offsets.astype(np.int64)).cuda()
    d_len = torch.from_numpy(lens).cuda()
    d_ids = torch.from_numpy(ids.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    ctx.synth_qual_dev(synth.SEED, d_plane.data_ptr(), plane.nbytes, d_off.data_ptr(), d_len.data_ptr(),
                       d_ids.data_ptr(), len(lens), profile=profile)
    got = d_plane.cpu().numpy()
    for i, L in enumerate(lens):
        o = int(offsets[i])
        assert (got[o:o + L] == synth.qual_read(int(ids[i]), int(L), profile=profile)).all(), i
        assert (got[o + L:o + ((L + 15) & ~15)] == 0).all()


@pytest.mark.parametrize("kernel", ["default", "stream", "dual"])
def test_randomised_configurations_vs_oracle(ctx, kernel, monkeypatch):
    """40 random (window_size, cut-offs, batch shape) configurations: window sizes around every 16/64-byte boundary and up
    to the ring/direct switch, batches smaller than a wave, reads shorter than the window, unsorted order."""
    select_kernel(monkeypatch, kernel)
    rng = np.random.RandomState(2024)
    ws_pool = [1, 2, 15, 16, 17, 31, 32, 48, 63, 64, 65, 127, 128, 129, 192, 249, 250, 251, 255, 256, 257, 320, 511, 512,
               1000, 1999, 2000, 2047, 2048, 2100, 3000]
    for trial in range(40):
        ws = int(ws_pool[rng.randint(len(ws_pool))]) if trial % 4 else int(rng.randint(1, 2600))
        n = int(rng.choice([1, 3, 63, 64, 65, 130, 300]))
        lens = rng.randint(0, int(rng.choice([40, 300, 3000, 9000])), n)
        reads = [("t%d_%d" % (trial, i), b"", synth.qual_read(trial * 1000 + i, int(L), 5).tobytes()) for i, L in enumerate(lens)]
        pkw = dict(window_size=ws)
        if trial % 3 == 0:
            pkw.update(min_length=int(rng.randint(1, 500)), min_window_q=float(rng.uniform(40, 95)))
        if trial % 5 == 0:
            pkw.update(max_length=int(rng.randint(100, 5000)), min_mean_q=float(rng.uniform(60, 95)))
        p = _oracle.make_params(**pkw)
        want = [_oracle.score_read(None, q, p) for _, _, q in reads]
        o = score_hip(ctx, reads, pkw, use_order=bool(trial % 2))
        assert_same_f64(o["mean_q"], np.array([w["mean_q"] for w in want]), "trial %d ws %d mean" % (trial, ws))
        assert_same_f64(o["window_q"], np.array([w["window_q"] for w in want]), "trial %d ws %d window" % (trial, ws))
        assert (o["passed"] == np.array([w["passed"] for w in want], dtype=np.uint8)).all(), (trial, pkw)


@pytest.mark.parametrize("part", range(4))
def test_dual_slot_kernel_window_sizes(ctx, part, monkeypatch):
    """The dual-slot kernel (both window edges as LDS-DMA streams, any window size): window sizes with every remainder mod 16 and
    mod 64 (the trailing stream's half slots), around the switch points of the other kernels (639 | 640, 1007 | 1008) and far
    beyond them; the read set of the instantiation test (lengths around ws and every 16 / 64 / 128-byte boundary, any byte)."""
    monkeypatch.setenv("FLX_PHRED_KERNEL", "dual")
    wss = sorted(set(list(range(1, 70)) + list(range(120, 136)) + [249, 250, 251, 255, 256, 257, 639, 640, 641, 1000, 1007, 1008, 1009, 1023, 1024,
                                                                     1025, 1500, 2000, 2047, 2048, 2049, 2500, 3000, 4095, 4096, 5000, 10000]))
    rng = np.random.RandomState(6000 + part)
    for ws in wss[part::4]:
        reads = _instantiation_reads(ws, rng)
        if ws >= 2000:
            reads = reads[::2]
        pkw = dict(window_size=ws)
        p = _oracle.make_params(**pkw)
        want = [_oracle.score_read(None, q, p) for _, _, q in reads]
        for use_order in (True, False):
            o = score_hip(ctx, reads, pkw, use_order)
            assert ctx.last_phred_kernel() == "flx_score_phred_dual"
            tag = "dual ws %d order=%s" % (ws, use_order)
            assert_same_f64(o["mean_q"], np.array([w["mean_q"] for w in want]), tag + " mean_q")
            assert_same_f64(o["window_q"], np.array([w["window_q"] for w in want]), tag + " window_q")
            assert (o["passed"] == np.array([w["passed"] for w in want], dtype=np.uint8)).all(), tag
