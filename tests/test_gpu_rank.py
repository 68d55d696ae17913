"""HIP global stage (flx_rank_and_cut: exact statistics, normalise, final score, radix sort, cut, boundary
audit) vs the oracle, which is itself pinned to the reference binary's outputs (tests/test_oracle_e2e.py).
The pass set must be IDENTICAL; device final scores only order reads and are checked to 1e-12 relative.
"""
import numpy as np
import pytest

import _e2e_checks
import _oracle
import _pipeline
from filtlong_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["select", "sort"])
def ctx(request):
    """Both device implementations of the cut: weighted radix SELECT (default) and radix SORT + scan (FLX_RANK_SORT=1)."""
    import os
    if request.param == "sort":
        os.environ["FLX_RANK_SORT"] = "1"
    else:
        os.environ.pop("FLX_RANK_SORT", None)
    c = api.Context(0)
    yield c
    c.close()
    os.environ.pop("FLX_RANK_SORT", None)


def random_reads2(n, seed, dup=0):
    rng = np.random.RandomState(seed)
    mean = rng.uniform(60, 99, n)
    window = mean * rng.uniform(0.3, 1.05, n)
    length = np.clip(rng.gamma(4, 2500, n), 1, None).astype(np.int32)
    passed = (rng.random_sample(n) > 0.1).astype(np.uint8)
    if dup:  # exact duplicates -> exact score ties
        src = rng.randint(0, n, dup)
        dst = rng.randint(0, n, dup)
        mean[dst], window[dst], length[dst] = mean[src], window[src], length[src]
    return mean, window, length, passed


def compare(ctx, mean, window, length, passed, **kw):
    want = _oracle.rank_and_cut(mean, window, length, passed, **{k: v for k, v in kw.items()})
    got = ctx.rank_and_cut(mean, window, length, passed,
                           length_weight=kw.get("lw", 1.0), mean_q_weight=kw.get("mw", 1.0),
                           window_q_weight=kw.get("ww", 1.0), target_bases=kw.get("target_bases"),
                           keep_percent=kw.get("keep_percent"), total_bases=kw.get("total_bases"))
    rep = got["report"]
    assert rep.outcome == want["outcome"]
    assert rep.target_bases == want["target_bases"]
    assert (got["passed"] == want["passed"]).all(), "pass set differs (%d reads)" % int((got["passed"] != want["passed"]).sum())
    if want["outcome"] == 3:
        assert rep.kept_bases == want["kept_bases"]
    # statistics are bit exact
    for a, b in ((rep.mean_quality, want["mean_quality"]), (rep.stdev_quality, want["stdev_quality"]),
                 (rep.min_z, want["min_z"]), (rep.max_z, want["max_z"])):
        assert a == b or (np.isnan(a) and np.isnan(b))
    fs, wf = got["final_score"], want["final_score"]
    ok = ~np.isnan(wf)
    assert np.allclose(fs[ok], wf[ok], rtol=1e-12, atol=0)
    return rep


@pytest.mark.parametrize("n,seed", [(1, 1), (2, 2), (65, 3), (1000, 4), (4097, 5), (100_000, 6)])
def test_random_vs_oracle(ctx, n, seed):
    mean, window, length, passed = random_reads2(n, seed)
    tot = int(length.astype(np.int64).sum())
    for frac in (0.01, 0.33, 0.5, 0.9, 0.999):
        compare(ctx, mean, window, length, passed, target_bases=max(1, int(tot * frac)))
    compare(ctx, mean, window, length, passed, keep_percent=42.5)
    compare(ctx, mean, window, length, passed, keep_percent=80.0, target_bases=tot // 3, lw=2.0, mw=0.5, ww=3.0)
    compare(ctx, mean, window, length, passed)                       # no cut at all
    compare(ctx, mean, window, length, passed, target_bases=tot)     # not enough reads
    compare(ctx, mean, window, length, passed, target_bases=tot - 1, total_bases=tot)  # usually "already below"


def test_million_reads(ctx):
    mean, window, length, passed = random_reads2(1_000_000, 11)
    tot = int(length.astype(np.int64).sum())
    rep = compare(ctx, mean, window, length, passed, target_bases=tot // 2)
    assert rep.exact_fallback == 0


def test_ties_straddling_the_cut(ctx):
    """Duplicate reads give exactly equal scores; wherever such a tie group straddles the cut the reference's
    std::sort tie order decides (main.cpp:247-248) and the library must reproduce it."""
    n = 3000
    mean, window, length, passed = random_reads2(n, 21, dup=2500)
    tot = int(length.astype(np.int64).sum())
    fallbacks = 0
    for t in np.linspace(tot * 0.05, tot * 0.95, 40):
        fallbacks += compare(ctx, mean, window, length, passed, target_bases=int(t)).exact_fallback
    assert fallbacks > 0


def test_equal_scores_with_different_lengths_at_the_cut(ctx):
    """length_weight 0 makes the score independent of the length, so reads with equal qualities but DIFFERENT lengths tie
    exactly.  When the target falls inside such a group the kept set depends on the order inside it (lengths {10, 20}
    entering at target - 15: one order keeps both, the other only the second), i.e. on libstdc++'s unstable std::sort:
    the library must notice that and take the reference's own order (n > 16, so introsort really permutes)."""
    rng = np.random.RandomState(5)
    n = 400
    mean = rng.uniform(60, 99, n)
    window = mean * rng.uniform(0.5, 1.0, n)
    length = rng.randint(500, 3000, n).astype(np.int32)
    for g in range(60):  # tie groups of 2..5 members with different lengths
        members = rng.choice(n, rng.randint(2, 6), replace=False)
        mean[members] = mean[members[0]]
        window[members] = window[members[0]]
    passed = (rng.random_sample(n) > 0.05).astype(np.uint8)
    tot = int(length.astype(np.int64).sum())
    fallbacks = 0
    for t in range(tot // 20, tot - tot // 20, tot // 300):
        fallbacks += compare(ctx, mean, window, length, passed, target_bases=int(t), lw=0.0).exact_fallback
    assert fallbacks > 0


def test_comm_single_rank_matches_plain_stage(ctx):
    """flx_rank_and_cut_comm_dev over a one-rank RCCL communicator (ncclCommInitRank, all-gather of the mean qualities,
    device-side all-reduces inside the selection) gives the result of flx_rank_and_cut_dev."""
    import torch
    mean, window, length, passed = random_reads2(50_000, 17, dup=300)
    tot = int(length.astype(np.int64).sum())
    c2 = api.Context(0)
    try:
        c2.comm_init(c2.comm_unique_id(), 0, 1)
        assert int(c2.comm_sum_u64([5, 7])[1]) == 7
        for target in (tot // 3, tot // 2, int(tot * 0.9)):
            want = ctx.rank_and_cut(mean, window, length, passed, target_bases=target, total_bases=tot)
            d_mean, d_win = torch.from_numpy(mean).cuda(), torch.from_numpy(window).cuda()
            d_len, d_pass = torch.from_numpy(length).cuda(), torch.from_numpy(passed.copy()).cuda()
            torch.cuda.synchronize()
            rep = c2.rank_and_cut_comm_dev(len(mean), d_mean.data_ptr(), d_win.data_ptr(), d_len.data_ptr(), d_pass.data_ptr(),
                                           target_bases=target, total_bases=tot)
            assert (d_pass.cpu().numpy() == want["passed"]).all()
            assert rep.kept_bases == want["report"].kept_bases and rep.mean_quality == want["report"].mean_quality
    finally:
        c2.close()


def test_all_equal_quality_gives_nan_scores(ctx):
    """stdev == 0 -> 0/0 (main.cpp:192-195,206): all scores NaN; must follow the reference's order, not crash."""
    n = 500
    rng = np.random.RandomState(3)
    length = rng.randint(100, 5000, n).astype(np.int32)
    compare(ctx, np.full(n, 77.0), np.full(n, 70.0), length, np.ones(n, np.uint8),
            target_bases=int(length.sum()) // 2)


def test_zero_and_negative_qualities(ctx):
    mean, window, length, passed = random_reads2(5000, 31)
    mean[:50] = 0.0
    window[:50] = 0.0
    mean[50:60] = -3.5  # negative mean quality: possible with Phred bytes below '!' 
    compare(ctx, mean, window, length, passed, target_bases=int(length.astype(np.int64).sum()) // 2)


def test_e2e_phred_goldens(ctx):
    """Reference binary outputs for the Phred-only configurations (fixtures + seeded synthetic)."""
    be = _pipeline.HipBackend(ctx)
    n = _e2e_checks.check_all(be, only=lambda k: k.startswith("synth_phred") or k.startswith("sort|phred"))
    assert n == 10 + 6


def test_exact_serial_statistics_hard_cases(ctx):
    """The device reproduces the reference's two SERIAL FP64 folds bit for bit (main.cpp:173-186), including
    binade crossings, round-to-even ties (addends that are exact multiples of half an ulp of the running sum),
    tiny / huge spreads, negatives and NaN."""
    rng = np.random.RandomState(77)
    cases = []
    cases.append(rng.uniform(0, 100, 300_000))
    cases.append(np.full(200_000, 64.0))                                  # every add is exact; many binade crossings
    cases.append(rng.choice([0.5, 1.5, 64.0, 96.0, 3.0, 0.25, 0.125, 100.0], 250_000))   # ties galore
    cases.append(np.concatenate([rng.uniform(90, 100, 100_000), [2.0 ** -30] * 5000, rng.uniform(0, 1, 50_000)]))
    cases.append(rng.uniform(0, 100, 70_000) * (2.0 ** -200))             # tiny values
    x = rng.uniform(50, 100, 120_000); x[777] = -12.25; x[5000:5010] = -0.5
    cases.append(x)                                                       # negatives -> serial chunks
    cases.append(np.array([77.7]))
    cases.append(rng.uniform(0, 100, 513))
    # exact multiples of 2^-44 around 90: with the running sum near 2^23..2^26 these hit half-ulp ties often
    cases.append(np.round(rng.uniform(80, 100, 400_000) * 2.0 ** 28) / 2.0 ** 28)
    # the walk opens a chunk into per-lane maps for the binade the sum is really in: crossings inside a lane, addends larger
    # than the running sum, long runs of zeros, powers of two over 40 binades, a sum that stays subnormal, infinity
    cases.append(np.concatenate([np.zeros(3000), rng.uniform(0, 100, 2000), np.zeros(5000), rng.uniform(0, 1e-3, 40_000)]))
    cases.append(2.0 ** rng.randint(-20, 21, 150_000).astype(np.float64))
    cases.append(np.concatenate([rng.uniform(0, 1e-6, 30_000), [1e9], rng.uniform(0, 100, 30_000), [1e15], rng.uniform(0, 100, 30_000)]))
    cases.append(np.full(20_000, 5e-324))
    cases.append(np.where(rng.uniform(0, 1, 200_000) < 0.97, 0.0, rng.uniform(0, 100, 200_000)))
    y = rng.uniform(0, 100, 40_000); y[31_000] = np.inf
    cases.append(y)
    cases.append(np.concatenate([np.full(1500, 2.0 ** 52), np.full(70_000, 0.5), np.full(70_000, 1.5)]))  # ties at 2^52..2^53
    for i, mean in enumerate(cases):
        n = len(mean)
        window = mean * 0.9
        length = rng.randint(200, 20000, n).astype(np.int32)
        want = _oracle.rank_and_cut(mean, window, length, np.ones(n, np.uint8))
        got = ctx.rank_and_cut(mean, window, length, np.ones(n, np.uint8))["report"]
        for a, b, what in ((got.mean_quality, want["mean_quality"], "mean"), (got.stdev_quality, want["stdev_quality"], "stdev"),
                           (got.min_z, want["min_z"], "min_z"), (got.max_z, want["max_z"], "max_z")):
            assert a == b or (np.isnan(a) and np.isnan(b)), "case %d %s: %r vs %r" % (i, what, a, b)
    nan_case = rng.uniform(0, 100, 5000); nan_case[1234] = np.nan
    got = ctx.rank_and_cut(nan_case, nan_case, np.full(5000, 100, np.int32), np.ones(5000, np.uint8))["report"]
    assert np.isnan(got.mean_quality) and np.isnan(got.stdev_quality)


def test_empty_and_single(ctx):
    e = np.zeros(0)
    got = ctx.rank_and_cut(e, e, np.zeros(0, np.int32), np.zeros(0, np.uint8), target_bases=100, total_bases=0)
    assert len(got["passed"]) == 0 and got["report"].outcome == 1   # target >= total_bases: "not enough reads"
    compare(ctx, np.array([88.0]), np.array([80.0]), np.array([5000], np.int32), np.array([1], np.uint8), target_bases=1)


def test_runs_on_callers_stream():
    """flx_ctx_set_stream: the library enqueues on the caller's HIP stream (here a torch stream); results unchanged."""
    import torch
    from filtlong_amd import synth
    c = api.Context(0)
    st = torch.cuda.Stream()
    c.set_stream(st.cuda_stream)
    n = 2000
    lens = np.maximum(synth.lengths(n, first=5) // 5, 1)
    quals = [synth.qual_read(5 + i, int(L)).tobytes() for i, L in enumerate(lens)]
    plane, offsets, lengths = api.pack_reads(quals)
    out = c.score_reads(plane, offsets, lengths, api.make_params(), order=api.length_order(lengths))
    p = _oracle.make_params()
    want = np.array([_oracle.score_read(None, q, p)["window_q"] for q in quals[:200]])
    assert (out["window_q"][:200].view(np.uint64) == want.view(np.uint64)).all()
    tot = int(lengths.astype(np.int64).sum())
    got = c.rank_and_cut(out["mean_q"], out["window_q"], lengths, out["passed"], target_bases=tot // 3)
    ref = _oracle.rank_and_cut(out["mean_q"], out["window_q"], lengths, out["passed"], target_bases=tot // 3)
    assert (got["passed"] == ref["passed"]).all()
    c.set_stream(None)
    c.close()
