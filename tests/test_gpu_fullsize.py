"""BASELINE.json full-size configuration on the GPU (C2: 10 M synthetic reads x mean 10 kbp, Phred-only,
--target_bases 50 % of the bases), checked through properties that do not need a full CPU scoring run:

  * per-read exactness on a sample: reads are independent, so the oracle re-scores ~300 of them (regenerated on the
    host from the same integer hash) and must match the device bit for bit — and since round 5 on a whole POPULATION: the
    first 10^6 reads of C2 (10^5 of C3 / C4, children and ranges included) re-scored by the oracle on every host core from
    the device's own plane bytes, every field bit for bit;
  * the global stage is re-run by the ORACLE on the device's 10^7 per-read values (std::sort of 10^7 takes seconds):
    exact statistics and IDENTICAL pass set;
  * threshold structure: every kept read scores >= every dropped-but-passed read; the walk overshoots the target by
    less than the last kept read; kept_bases is the sum of kept lengths;
  * idempotence: a second run gives the same arrays.
"""
import ctypes as C

import numpy as np
import pytest

import _oracle
from filtlong_amd import api, synth

pytestmark = pytest.mark.gpu

N_READS = 10_000_000


def test_c2_full_size_properties():
    import torch
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    n = N_READS
    lengths = synth.lengths(n)
    offsets = np.zeros(n, dtype=np.uint64)
    pb = C.c_uint64()
    ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    order = api.length_order(lengths)
    total = int(lengths.astype(np.int64).sum())
    assert abs(total / n - 10000) < 20
    d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lengths).to(dev)
    d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    d_mean = torch.zeros(n, dtype=torch.float64, device=dev)
    d_win = torch.zeros(n, dtype=torch.float64, device=dev)
    d_pass = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_fs = torch.zeros(n, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx.synth_qual_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n)
    params = api.make_params(min_length=1000)

    def run():
        ctx.score_reads_dev(d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, params,
                            d_mean.data_ptr(), d_win.data_ptr(), d_pass.data_ptr())
        pre = d_pass.cpu().numpy().copy()
        rep = ctx.rank_and_cut_dev(n, d_mean.data_ptr(), d_win.data_ptr(), d_len.data_ptr(), d_pass.data_ptr(),
                                   target_bases=total // 2, total_bases=total, d_final_score=d_fs.data_ptr())
        return pre, d_mean.cpu().numpy().copy(), d_win.cpu().numpy().copy(), d_pass.cpu().numpy().copy(), d_fs.cpu().numpy().copy(), rep

    pre, mean, win, passed, fs, rep = run()

    # 1. sampled per-read exactness
    rng = np.random.RandomState(1)
    sample = np.unique(np.concatenate([rng.randint(0, n, 280), order[:10], order[-10:]]))
    p = _oracle.make_params(min_length=1000)
    for i in sample:
        w = _oracle.score_read(None, synth.qual_read(int(i), int(lengths[i])).tobytes(), p)
        assert w["mean_q"] == mean[i] and w["window_q"] == win[i] and w["passed"] == pre[i], int(i)

    # 1b. WHOLE-POPULATION exactness on the first 10^6 reads (10^10 bases: every lane of ~15 000 wavefronts, ticket scheduler,
    # tail groups and redo list included): the oracle re-scores the device's OWN plane bytes on every host core (flo_score_plane_mt
    # = flo_score_read per read), 10^5 reads at a time, and every mean / window / pass flag must match bit for bit
    # (src/read.cpp:25-73, 208-236)
    n_pop, step = (1_000_000, 100_000) if n >= 1_000_000 else (n, n)
    for a in range(0, n_pop, step):
        b = min(a + step, n_pop)
        lo, hi = int(offsets[a]), int(offsets[b - 1]) + int(lengths[b - 1])
        chunk = d_plane[lo:hi].cpu().numpy()
        w = _oracle.score_plane_mt(chunk, offsets[a:b] - np.uint64(lo), lengths[a:b], p)
        assert (w["mean_q"].view(np.uint64) == mean[a:b].view(np.uint64)).all(), ("mean_q", a)
        assert (w["window_q"].view(np.uint64) == win[a:b].view(np.uint64)).all(), ("window_q", a)
        assert (w["passed"] == pre[a:b]).all(), ("passed", a)
        del chunk, w

    # 2. oracle global stage on the device's per-read values: exact statistics, identical pass set
    want = _oracle.rank_and_cut(mean, win, lengths, pre, target_bases=total // 2, total_bases=total)
    assert rep.mean_quality == want["mean_quality"] and rep.stdev_quality == want["stdev_quality"]
    assert rep.min_z == want["min_z"] and rep.max_z == want["max_z"]
    assert rep.outcome == 3 and rep.target_bases == total // 2 and rep.kept_bases == want["kept_bases"]
    assert (passed == want["passed"]).all()
    assert np.allclose(fs, want["final_score"], rtol=1e-12, atol=0)

    # 3. threshold structure
    kept = passed.astype(bool)
    dropped = pre.astype(bool) & ~kept
    assert int(lengths[kept].astype(np.int64).sum()) == rep.kept_bases >= total // 2
    assert fs[kept].min() >= fs[dropped].max()
    last = np.argmin(np.where(kept, fs, np.inf))
    assert rep.kept_bases - int(lengths[last]) < total // 2   # the read that crossed the target is kept (overshoot)
    assert not passed[~pre.astype(bool)].any()                 # hard cut-offs are never resurrected

    # 4. idempotence
    pre2, mean2, win2, passed2, fs2, rep2 = run()
    assert (mean2.view(np.uint64) == mean.view(np.uint64)).all() and (win2.view(np.uint64) == win.view(np.uint64)).all()
    assert (passed2 == passed).all() and rep2.kept_bases == rep.kept_bases
    ctx.close()


@pytest.mark.parametrize("size", ["mid", "full"])
@pytest.mark.parametrize("short_reads", [False, True], ids=["C3-assembly", "C4-short-reads"])
def test_kmer_mode_properties(short_reads, size):
    """BASELINE.json configs[2]/[3] (k-mer mode, `--trim --split 500` for C4): "mid" = 2x10^5 reads / 2x10^9 bases against a
    1 Mbp reference, "full" = the BASELINE size, 10^7 reads / 10^11 bases against the 5 Mbp reference (denser set, ~10^7
    children).  The oracle builds the same 16-mer set (equal sizes) and re-scores a sample of reads bit for bit (mean,
    window, first/last, every child); the rest is checked structurally (children ordered, disjoint, inside the read, CSR
    sums), and for C4 the word-level and bit-level child passes must agree on every read."""
    import torch
    from filtlong_amd import _lib
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    n, ref_len = (200_000, 1_000_000) if size == "mid" else (10_000_000, 5_000_000)
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, ref_len)
    ks = api.Kmers(ctx)
    oset = _oracle.KmerSet()
    if short_reads:
        npairs = ref_len // 5  # 40x of error-free 100 bp pairs
        starts = (synth.mix(synth.SEED, synth.STREAM_START, np.arange(npairs, dtype=np.uint64) + np.uint64(1 << 40), 0)
                  % np.uint64(ref_len - 450)).astype(np.int64)
        comp = np.zeros(256, dtype=np.uint8)
        comp[list(b"ACGT")] = list(b"TGCA")
        idx = starts[:, None] + np.arange(100)[None, :]
        r1 = [r.tobytes() for r in ref[idx]]
        r2 = [r.tobytes() for r in comp[ref[idx + 350]][:, ::-1]]
        ks.add_read_fastqs([r1, r2])
        oset.add_short_reads(r1)
        oset.add_short_reads(r2)
    else:
        ks.add_assembly_fasta([ref.tobytes()])
        oset.add_assembly([ref.tobytes()])
    ks.finalize()
    assert len(ks) == len(oset) > 0.9 * ref_len

    lengths = synth.lengths(n)
    offsets = np.zeros(n, dtype=np.uint64)
    pb = C.c_uint64()
    ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    order = api.length_order(lengths)
    d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lengths).to(dev)
    d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n,
                      d_ref.data_ptr(), ref_len)
    pkw = dict(trim=True, split=500) if short_reads else dict()
    params = api.make_params(**pkw)
    cap = 4 * n
    t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (
        ("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8), ("first", n, torch.int32),
        ("last", n, torch.int32), ("coff", n + 1, torch.int64), ("crng", 2 * cap, torch.int32), ("cmean", cap, torch.float64),
        ("cwin", cap, torch.float64), ("cpass", cap, torch.uint8))}
    torch.cuda.synchronize()
    s = _lib.Scores()
    s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(),
                                                      t["first"].data_ptr(), t["last"].data_ptr())
    s.child_offsets, s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (
        t["coff"].data_ptr(), t["crng"].data_ptr(), t["cmean"].data_ptr(), t["cwin"].data_ptr(), t["cpass"].data_ptr())
    s.child_capacity = cap
    rc = ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, params, s)
    assert rc == 0
    torch.cuda.synchronize()
    assert ctx.last_kmer_fold_grid()  # (the default at this window size: the folds' steady state on the integer grid)
    if short_reads:
        # one lane per child (default), the children inside their read's lane at word level (FLX_KMER_FOLD=words) and bit by bit
        # (FLX_KMER_FOLD=bits) are three implementations of src/read.cpp:86-141: they must agree on every one of the reads and
        # children of the batch
        import os
        per_child = {k: v.clone() for k, v in t.items()}
        n_word = int(s.n_children)
        for variant in ("words", "bits"):
            os.environ["FLX_KMER_FOLD"] = variant
            try:
                for v in t.values():
                    v.zero_()
                torch.cuda.synchronize()
                assert ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n,
                                          params, s) == 0
                torch.cuda.synchronize()
            finally:
                del os.environ["FLX_KMER_FOLD"]
            assert int(s.n_children) == n_word
            for k in ("mean", "win", "pass", "first", "last", "coff"):
                assert torch.equal(t[k].view(torch.uint8), per_child[k].view(torch.uint8)), (variant, k)
            for k, per in (("crng", 2), ("cmean", 1), ("cwin", 1), ("cpass", 1)):
                assert torch.equal(t[k][:per * n_word].view(torch.uint8), per_child[k][:per * n_word].view(torch.uint8)), (variant, k)
    # the window folds on the integer grid (default at this window size, round 5) and in floating point step by step
    # (FLX_KMER_FOLD_GRID=0) agree on every read and child of the batch
    import os
    on_grid = {k: v.clone() for k, v in t.items()}
    n_grid = int(s.n_children)
    os.environ["FLX_KMER_FOLD_GRID"] = "0"
    try:
        for v in t.values():
            v.zero_()
        torch.cuda.synchronize()
        assert ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n,
                                  params, s) == 0
        torch.cuda.synchronize()
        assert not ctx.last_kmer_fold_grid()
    finally:
        del os.environ["FLX_KMER_FOLD_GRID"]
    assert int(s.n_children) == n_grid
    for k in ("mean", "win", "pass", "first", "last", "coff"):
        assert torch.equal(t[k].view(torch.uint8), on_grid[k].view(torch.uint8)), ("FP folds", k)
    for k, per in (("crng", 2), ("cmean", 1), ("cwin", 1), ("cpass", 1)):
        assert torch.equal(t[k][:per * n_grid].view(torch.uint8), on_grid[k][:per * n_grid].view(torch.uint8)), ("FP folds", k)
    for k, v in on_grid.items():  # (the default kernels' results are what the rest of the test looks at)
        t[k].copy_(v)
    del on_grid
    # the two implementations of the coverage kernel (wave level with pair tables / round 2's workgroup-per-read kernel with one
    # bitmap lookup per candidate run end) agree on every read and child of the batch
    wave_level = {k: v.clone() for k, v in t.items()}
    n_wave = int(s.n_children)
    assert ctx.last_kmer_cover() == "q"  # (round 6: the default is the kernel with the queue in LDS, cover_queue.hip)
    for other in ("w", "v2", "q2"):  # the wave-level kernel of rounds 3-5, round 2's kernel, every read in the launch with a diagonal per lane
        os.environ["FLX_KMER_COVER"] = other
        try:
            for v in t.values():
                v.zero_()
            torch.cuda.synchronize()
            assert ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n,
                                      params, s) == 0
            torch.cuda.synchronize()
            assert ctx.last_kmer_cover() == other
        finally:
            del os.environ["FLX_KMER_COVER"]
        assert int(s.n_children) == n_wave
        for k in ("mean", "win", "pass", "first", "last", "coff"):
            assert torch.equal(t[k].view(torch.uint8), wave_level[k].view(torch.uint8)), (other, k)
        for k, per in (("crng", 2), ("cmean", 1), ("cwin", 1), ("cpass", 1)):
            assert torch.equal(t[k][:per * n_wave].view(torch.uint8), wave_level[k][:per * n_wave].view(torch.uint8)), (other, k)
    for k, v in wave_level.items():
        t[k].copy_(v)
    del wave_level
    mean, win, first, last = (t[k].cpu().numpy() for k in ("mean", "win", "first", "last"))
    coff = t["coff"].cpu().numpy()
    nchild = int(s.n_children)
    crng = t["crng"].cpu().numpy()[:2 * nchild].reshape(-1, 2)
    cmean, cwin, cpass = (t[k].cpu().numpy()[:nchild] for k in ("cmean", "cwin", "cpass"))

    # structure
    assert coff[0] == 0 and coff[-1] == nchild and (np.diff(coff) >= 0).all()
    if not short_reads:
        assert nchild == 0
    else:
        assert nchild > n // 10
        owner = np.repeat(np.arange(n), np.diff(coff))
        assert (crng[:, 0] >= 0).all() and (crng[:, 1] > crng[:, 0]).all() and (crng[:, 1] <= lengths[owner]).all()
        same = owner[1:] == owner[:-1]
        assert (crng[1:, 0][same] > crng[:-1, 1][same]).all()  # ordered, separated by a bad range
    assert ((mean >= 0) & (mean <= 100)).all() and ((win >= 0) & (win <= 100.000001)).all()
    cov = first >= 0
    assert (last[cov] > first[cov]).all() and (last[cov] <= lengths[cov]).all() and (mean[~cov] == 0).all()

    # sampled exactness against the oracle (same generator on the host)
    rng = np.random.RandomState(3)
    sample = np.unique(np.concatenate([rng.randint(0, n, 1000 if size == "full" else 300), order[:3], order[-3:]]))
    p = _oracle.make_params(**pkw)
    for i in sample:
        L = int(lengths[i])
        seq = synth.seq_read(int(i), L, ref).tobytes()
        w = _oracle.score_read(seq, None, p, kmerset=oset)
        assert w["mean_q"] == mean[i] and w["window_q"] == win[i], int(i)
        assert w["first"] == first[i] and w["last"] == last[i], int(i)
        a, b = int(coff[i]), int(coff[i + 1])
        assert len(w["children"]) == b - a, int(i)
        for k, ch in enumerate(w["children"]):
            assert w["child_ranges"][k] == (int(crng[a + k, 0]), int(crng[a + k, 1])), (int(i), k)
            assert ch["mean_q"] == cmean[a + k] and ch["window_q"] == cwin[a + k] and ch["passed"] == cpass[a + k], (int(i), k)

    # WHOLE-POPULATION exactness on the first 10^5 reads (10^9 bases through the reference's own unordered_set lookups, on every
    # host core): the oracle re-scores the device's own plane bytes — every mean, window, first / last, pass flag, child range and
    # child score bit for bit (src/read.cpp:25-144)
    n_pop = min(n, 100_000)
    hi = int(offsets[n_pop - 1]) + int(lengths[n_pop - 1])
    chunk = d_plane[:hi].cpu().numpy()
    w = _oracle.score_plane_mt(chunk, offsets[:n_pop], lengths[:n_pop], p, kmerset=oset)
    del chunk
    dev_pass = t["pass"].cpu().numpy()
    assert (w["mean_q"].view(np.uint64) == mean[:n_pop].view(np.uint64)).all()
    assert (w["window_q"].view(np.uint64) == win[:n_pop].view(np.uint64)).all()
    assert (w["first"] == first[:n_pop]).all() and (w["last"] == last[:n_pop]).all() and (w["passed"] == dev_pass[:n_pop]).all()
    assert (w["child_offsets"] == coff[:n_pop + 1].view(np.uint64)).all()
    nc = w["n_children"]
    assert nc == int(coff[n_pop]) and (short_reads or nc == 0)
    assert (w["child_ranges"] == crng[:nc]).all()
    assert (w["child_mean_q"].view(np.uint64) == cmean[:nc].view(np.uint64)).all()
    assert (w["child_window_q"].view(np.uint64) == cwin[:nc].view(np.uint64)).all()
    assert (w["child_passed"] == cpass[:nc]).all()
    del w

    # The global stage at this size (src/main.cpp:138-147 then 169-261): the device's reads2 gather equals the oracle's loop
    # entry for entry, and on those reads2 arrays the ORACLE (its std::sort over all ~1.2x10^7 entries) gives the same
    # statistics, kept bases and pass set as flx_rank_and_cut_dev.
    cap2 = n + nchild
    r2 = {k: torch.zeros(cap2, dtype=dt, device=dev) for k, dt in (
        ("mean", torch.float64), ("win", torch.float64), ("len", torch.int32), ("pass", torch.uint8), ("parent", torch.int32),
        ("child", torch.int64), ("fs", torch.float64))}
    n2 = ctx.reads2_gather_dev(n, d_len.data_ptr(), s, cap2, r2["mean"].data_ptr(), r2["win"].data_ptr(), r2["len"].data_ptr(),
                               r2["pass"].data_ptr(), r2["parent"].data_ptr(), r2["child"].data_ptr())
    assert n2 == n - int((np.diff(coff) > 0).sum()) + nchild
    sc = {"mean_q": mean, "window_q": win, "passed": t["pass"].cpu().numpy(), "child_offsets": coff.view(np.uint64),
          "child_ranges": crng, "child_mean_q": cmean, "child_window_q": cwin, "child_passed": cpass}
    w2 = _oracle.reads2_gather(lengths, sc)
    assert len(w2["mean_q"]) == n2
    g2 = {k: r2[k].cpu().numpy()[:n2] for k in r2}
    assert (g2["mean"].view(np.uint64) == w2["mean_q"].view(np.uint64)).all()
    assert (g2["win"].view(np.uint64) == w2["window_q"].view(np.uint64)).all()
    assert (g2["len"] == w2["length"]).all() and (g2["pass"] == w2["passed"]).all()
    assert (g2["parent"].view(np.uint32) == w2["parent"]).all() and (g2["child"] == w2["child"]).all()
    total = int(lengths.astype(np.int64).sum())  # original reads, src/main.cpp:89
    rep = ctx.rank_and_cut_dev(n2, r2["mean"].data_ptr(), r2["win"].data_ptr(), r2["len"].data_ptr(), r2["pass"].data_ptr(),
                               target_bases=total // 2, total_bases=total, d_final_score=r2["fs"].data_ptr())
    want = _oracle.rank_and_cut(w2["mean_q"], w2["window_q"], w2["length"], w2["passed"], target_bases=total // 2,
                                total_bases=total)
    assert rep.mean_quality == want["mean_quality"] and rep.stdev_quality == want["stdev_quality"]
    assert rep.min_z == want["min_z"] and rep.max_z == want["max_z"]
    assert rep.outcome == want["outcome"] and rep.target_bases == want["target_bases"] == total // 2
    assert rep.kept_bases == want["kept_bases"]
    final = r2["pass"].cpu().numpy()[:n2]
    assert (final == want["passed"]).all()
    assert np.allclose(r2["fs"].cpu().numpy()[:n2], want["final_score"], rtol=1e-12, atol=0, equal_nan=True)
    ks.close()
    ctx.close()


@pytest.mark.parametrize("profile", [1, 2], ids=["indels", "unrelated"])
def test_kmer_read_profiles_whole_population(profile):
    """Round-5 review, item 2: reads the SURVEY §8(d) generator does not make — profile 1: a third of the errors insertions and a
    third deletions of 1-3 bases (every indel moves the read's diagonal in the set's text), profile 2: 30 % of the reads unrelated
    to the reference (oracle/synth.h: flx_synth_seq_read).  10^5 reads / 10^9 bases against the 5 Mbp reference, assembly set and
    --trim --split 500: the oracle re-scores the device's own plane bytes on every host core — every mean, window, first / last,
    pass flag, child range and child score bit for bit (src/read.cpp:25-144) — and the three coverage kernels agree on every byte."""
    import os
    import torch
    from filtlong_amd import _lib
    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    n, ref_len = 100_000, 5_000_000
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, ref_len)
    ks = api.Kmers(ctx)
    oset = _oracle.KmerSet()
    ks.add_assembly_fasta([ref.tobytes()])
    oset.add_assembly([ref.tobytes()])
    ks.finalize()
    assert len(ks) == len(oset)
    lengths = synth.lengths(n)
    offsets = np.zeros(n, dtype=np.uint64)
    pb = C.c_uint64()
    ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    order = api.length_order(lengths)
    d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lengths).to(dev)
    d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n,
                      d_ref.data_ptr(), ref_len, profile=profile)
    plane = d_plane.cpu().numpy()
    for i in (0, 1, 17, n - 1):  # the device generator wrote what the host definition says
        o, L = int(offsets[i]), int(lengths[i])
        assert (plane[o:o + L] == synth.seq_read(i, L, ref, profile=profile)).all(), i
    cap = 4 * n
    t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (
        ("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8), ("first", n, torch.int32),
        ("last", n, torch.int32), ("coff", n + 1, torch.int64), ("crng", 2 * cap, torch.int32), ("cmean", cap, torch.float64),
        ("cwin", cap, torch.float64), ("cpass", cap, torch.uint8))}
    s = _lib.Scores()
    s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(),
                                                      t["first"].data_ptr(), t["last"].data_ptr())
    s.child_offsets, s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (
        t["coff"].data_ptr(), t["crng"].data_ptr(), t["cmean"].data_ptr(), t["cwin"].data_ptr(), t["cpass"].data_ptr())
    s.child_capacity = cap
    for pkw in (dict(), dict(trim=True, split=500)):
        params = api.make_params(**pkw)
        w = _oracle.score_plane_mt(plane, offsets, lengths, _oracle.make_params(**pkw), kmerset=oset)
        nc = w["n_children"]
        if profile == 2 and not pkw:
            assert 0.25 * n < int((w["mean_q"] < 15.0).sum()) < 0.35 * n  # the unrelated reads: a random 16-mer is a member once in 430 (5 Mbp), ~4 % of their bases are covered
        for cover in (None, "w", "v2", "q2"):  # q2: every read in the launch with a diagonal per lane
            if cover:
                os.environ["FLX_KMER_COVER"] = cover
            try:
                for v in t.values():
                    v.fill_(0x5A if v.dtype == torch.uint8 else 0)
                torch.cuda.synchronize()
                assert ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n,
                                          params, s) == 0
                torch.cuda.synchronize()
                assert ctx.last_kmer_cover() == (cover or "q")
            finally:
                os.environ.pop("FLX_KMER_COVER", None)
            tag = (profile, pkw, cover)
            assert int(s.n_children) == nc, tag
            assert (w["mean_q"].view(np.uint64) == t["mean"].cpu().numpy().view(np.uint64)).all(), tag
            assert (w["window_q"].view(np.uint64) == t["win"].cpu().numpy().view(np.uint64)).all(), tag
            assert (w["first"] == t["first"].cpu().numpy()).all() and (w["last"] == t["last"].cpu().numpy()).all(), tag
            assert (w["passed"] == t["pass"].cpu().numpy()).all(), tag
            assert (w["child_offsets"] == t["coff"].cpu().numpy().view(np.uint64)).all(), tag
            assert (w["child_ranges"] == t["crng"].cpu().numpy()[:2 * nc].reshape(-1, 2)).all(), tag
            assert (w["child_mean_q"].view(np.uint64) == t["cmean"].cpu().numpy()[:nc].view(np.uint64)).all(), tag
            assert (w["child_window_q"].view(np.uint64) == t["cwin"].cpu().numpy()[:nc].view(np.uint64)).all(), tag
            assert (w["child_passed"] == t["cpass"].cpu().numpy()[:nc]).all(), tag
    ks.close()
    ctx.close()
