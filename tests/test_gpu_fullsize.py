"""BASELINE.json full-size configuration on the GPU (C2: 10 M synthetic reads x mean 10 kbp, Phred-only,
--target_bases 50 % of the bases), checked through properties that do not need a full CPU scoring run:

  * per-read exactness on a sample: reads are independent, so the oracle re-scores ~300 of them (regenerated on the
    host from the same integer hash) and must match the device bit for bit;
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
