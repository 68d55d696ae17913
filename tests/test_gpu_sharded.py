"""The SHARDED global stage (flx_rank_and_cut_sharded_dev): reads2 entries split over several ranks, mean
qualities all-gathered, selection histograms all-reduced.  Every rank must end with exactly the pass flags the oracle
(= the reference's main.cpp:169-261) gives for its slice.

One GPU is enough to exercise it: the ranks are threads with their own library context, and the all-reduce the library
calls back into is a barrier-synchronised sum (the C ABI only sees `int reduce(user, buf, count)`).  The torch.distributed
plumbing around it (filtlong_amd/dist.py) runs as two real processes in test_bench_two_processes_one_gpu.
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest
import torch

import _oracle
from filtlong_amd import api
from test_gpu_rank import random_reads2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ThreadAllReduce:
    def __init__(self, world):
        self.bar = threading.Barrier(world, timeout=60)
        self.lock = threading.Lock()
        self.acc = None
        self.calls = 0

    def for_rank(self, rank):
        def reduce(buf):
            with self.lock:
                self.acc = buf.copy() if self.acc is None else self.acc + buf
            self.bar.wait()
            buf[:] = self.acc
            self.bar.wait()
            if rank == 0:
                self.acc = None
                self.calls += 1
            self.bar.wait()
        return reduce


def run_sharded(shards, mean, window, length, passed, **kw):
    """shards: list of (lo, hi).  Returns (flags per rank concatenated, reports, need_replicated flags)."""
    world = len(shards)
    ar = ThreadAllReduce(world)
    dev = torch.device("cuda", 0)
    g_mean = torch.from_numpy(mean).to(dev)
    out = [None] * world
    err = []

    def work(r):
        try:
            lo, hi = shards[r]
            ctx = api.Context(0)
            try:
                w = torch.from_numpy(window[lo:hi].copy()).to(dev)
                l = torch.from_numpy(length[lo:hi].copy()).to(dev)
                p = torch.from_numpy(passed[lo:hi].copy()).to(dev)
                torch.cuda.synchronize()
                rep, need = ctx.rank_and_cut_sharded_dev(len(mean), g_mean.data_ptr(), lo, hi - lo, w.data_ptr(), l.data_ptr(),
                                                         p.data_ptr(), r, world, reduce=ar.for_rank(r), **kw)
                torch.cuda.synchronize()
                out[r] = (p.cpu().numpy(), rep, need)
            finally:
                ctx.close()
        except Exception as e:  # noqa
            err.append(e)
            ar.bar.abort()

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    return out, ar.calls


def check(shards, mean, window, length, passed, **kw):
    okw = dict(kw)
    for a, b in (("length_weight", "lw"), ("mean_q_weight", "mw"), ("window_q_weight", "ww")):
        if a in okw:
            okw[b] = okw.pop(a)
    want = _oracle.rank_and_cut(mean, window, length, passed, **okw)
    out, calls = run_sharded(shards, mean, window, length, passed, **kw)
    needs = [o[2] for o in out]
    assert len(set(needs)) == 1, "ranks disagree on the fallback: %r" % needs
    if needs[0]:
        for (lo, hi), (flags, rep, need) in zip(shards, out):
            assert (flags == passed[lo:hi]).all(), "flags modified before a fallback"
        return "fallback", calls
    for (lo, hi), (flags, rep, need) in zip(shards, out):
        assert rep.outcome == want["outcome"] and rep.target_bases == want["target_bases"]
        assert rep.mean_quality == want["mean_quality"] and rep.stdev_quality == want["stdev_quality"]
        if want["outcome"] == 3:
            assert rep.kept_bases == want["kept_bases"]
        bad = int((flags != want["passed"][lo:hi]).sum())
        assert bad == 0, "rank slice [%d,%d): %d flags differ" % (lo, hi, bad)
    return "ok", calls


def even(n, w):
    return [(n * r // w, n * (r + 1) // w) for r in range(w)]


@pytest.mark.parametrize("n,world,seed", [(1000, 2, 1), (4097, 3, 2), (100_000, 4, 3), (300_000, 8, 4)])
def test_sharded_matches_oracle(n, world, seed):
    mean, window, length, passed = random_reads2(n, seed)
    tot = int(length.astype(np.int64).sum())
    for frac in (0.02, 0.5, 0.97):
        res, calls = check(even(n, world), mean, window, length, passed, target_bases=max(1, int(tot * frac)), total_bases=tot)
        assert res == "ok" and calls <= 12
    assert check(even(n, world), mean, window, length, passed, keep_percent=42.5, total_bases=tot)[0] == "ok"
    assert check(even(n, world), mean, window, length, passed, keep_percent=80.0, target_bases=tot // 3, total_bases=tot,
                 length_weight=2.0, mean_q_weight=0.5, window_q_weight=3.0)[0] == "ok"
    assert check(even(n, world), mean, window, length, passed, total_bases=tot)[0] == "ok"                      # no cut
    assert check(even(n, world), mean, window, length, passed, target_bases=tot, total_bases=tot)[0] == "ok"    # not enough
    assert check(even(n, world), mean, window, length, passed, target_bases=tot - 1, total_bases=tot)[0] == "ok"  # already below


def test_unequal_and_empty_shards():
    n = 20_000
    mean, window, length, passed = random_reads2(n, 9)
    tot = int(length.astype(np.int64).sum())
    for shards in ([(0, 0), (0, n)], [(0, 7), (7, 7), (7, 15_000), (15_000, n)], [(0, n), (n, n), (n, n)]):
        assert check(shards, mean, window, length, passed, target_bases=tot // 2, total_bases=tot)[0] == "ok"


def test_ties_and_nan_fall_back_on_every_rank():
    n = 5000
    mean, window, length, passed = random_reads2(n, 5)
    mean[:] = 90.0                       # stdev == 0 -> NaN scores (main.cpp:192-206)
    tot = int(length.astype(np.int64).sum())
    assert check(even(n, 3), mean, window, length, passed, target_bases=tot // 2, total_bases=tot)[0] == "fallback"
    # all reads identical: whatever the target, equal scores straddle the cut -> the reference's std::sort decides
    mean, window, length, passed = random_reads2(n, 6)
    mean[1:] = mean[0]; window[1:] = window[0]; length[1:] = length[0]
    mean[-1] += 1.0                      # keep stdev > 0
    tot = int(length.astype(np.int64).sum())
    assert check(even(n, 2), mean, window, length, passed, target_bases=tot // 2, total_bases=tot)[0] == "fallback"
    # duplicates away from the cut are no problem; sprinkled everywhere they may or may not straddle it — either outcome
    # must be consistent (check() asserts that) and, when decided on the device, exact
    mean, window, length, passed = random_reads2(50_000, 7, dup=20_000)
    tot = int(length.astype(np.int64).sum())
    for frac in (0.1, 0.5, 0.9):
        check(even(50_000, 4), mean, window, length, passed, target_bases=int(tot * frac), total_bases=tot)


def test_single_rank_equals_unsharded():
    n = 30_000
    mean, window, length, passed = random_reads2(n, 12)
    tot = int(length.astype(np.int64).sum())
    assert check([(0, n)], mean, window, length, passed, target_bases=tot // 3, total_bases=tot)[0] == "ok"


@pytest.mark.parametrize("stage", ["sharded", "replicated", "rccl"])
def test_bench_two_processes_one_gpu(tmp_path, stage):
    """bench.py's N > 1 path as two real processes (gloo, both on GPU 0): every rank's flags equal the slice of the
    single-process run over the same 2 x 20 000 reads.  "rccl" is the driver's default stage (the library's own
    communicator); two ranks on one GPU get the loopback stand-in for RCCL's entry points (tests/shim)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("FLX_RANK_SORT", None)
    if stage == "rccl":
        shim_dir = os.path.join(ROOT, "tests", "shim")
        subprocess.check_call(["make", "-s", "-C", shim_dir])
        env["FLX_RCCL_LIB"] = os.path.join(shim_dir, "libloopback_rccl.so")
    one = str(tmp_path / "one")
    two = str(tmp_path / "two")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--reads", "40000",
                        "--no-cpu-baseline", "--no-extras", "--dump-flags", one], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--reads", "20000", "--backend", "gloo", "--global-stage", stage, "--dump-flags", two],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    import json
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["config"]["reads_total"] == 40000 and j["scaling"] == "weak"
    if stage == "rccl":
        assert "library-owned RCCL communicator" in j["config"]["parallelism"]
    # the self-check behind the timed steps: 2 x 200 000 fresh reads through the same stage == rank 0 alone
    assert j["verify"]["ok"] is True and j["verify"]["ranks"] == 2 and j["verify"]["flags_differing"] == 0, j["verify"]
    assert 0 < j["verify"]["passed_reads"] < j["verify"]["reads"]
    assert j["config"]["allgather_bytes_per_rank"] == {"sent": 8 * 20000, "received": 16 * 20000}
    want = np.load(one + ".rank0.npy")
    got = np.concatenate([np.load(two + ".rank%d.npy" % k) for k in range(2)])
    assert want.shape == got.shape and (want == got).all()
    assert 0 < int(want.sum()) < len(want)


def test_bench_rccl_stage_single_rank(tmp_path):
    """bench.py's default N > 1 path (the library's own RCCL communicator: ncclCommInitRank, all-gather of the mean
    qualities, device-side all-reduces) taken with ONE rank (--force-dist): same flags as the plain single-GPU run."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    env.pop("FLX_RANK_SORT", None)
    outs = []
    for extra, tag in (([], "plain"), (["--force-dist", "--global-stage", "rccl"], "rccl")):
        path = str(tmp_path / tag)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "50000",
                            "--no-cpu-baseline", "--no-extras", "--dump-flags", path] + extra, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        outs.append(np.load(path + ".rank0.npy"))
    assert (outs[0] == outs[1]).all() and 0 < int(outs[0].sum()) < len(outs[0])
    # --verify with one rank: real RCCL (ncclCommInitRank, all-gather, device all-reduces) against the plain stage on fresh reads
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--reads", "50000", "--no-cpu-baseline",
                        "--no-extras", "--force-dist", "--global-stage", "rccl", "--verify", "--verify-reads", "60000"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert j["verify"]["ok"] is True and j["verify"]["reads"] == 60000, j["verify"]


@pytest.mark.parametrize("shim", [True, False], ids=["library-communicator", "torch-sharded"])
def test_bench_starts_its_own_ranks(tmp_path, shim):
    """`python3 bench.py --gpus 2 ...` exactly as the driver types it (no torch.distributed.run in front, no WORLD_SIZE in
    the environment): bench.py starts the two ranks itself, rank 0 prints ONE JSON line with n_gpus 2, and the flags of both
    ranks equal the single-process run over the same 2 x 20 000 reads.  Two ranks on one GPU: the launch backend falls to
    gloo by itself; with FLX_RCCL_LIB the library's own communicator runs over the loopback stand-in (tests/shim)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FLX_RANK_SORT")}
    if shim:
        shim_dir = os.path.join(ROOT, "tests", "shim")
        subprocess.check_call(["make", "-s", "-C", shim_dir])
        env["FLX_RCCL_LIB"] = os.path.join(shim_dir, "libloopback_rccl.so")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--reads", "40000",
                        "--no-cpu-baseline", "--no-extras", "--dump-flags", one], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--reads",
                        "20000", "--dump-flags", two], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["config"]["reads_total"] == 40000 and j["scaling"] == "weak"
    assert j["config"]["launch_backend"] == "gloo" and "comm" in j["stage_ms_per_step"]
    if shim:
        assert j["config"]["rccl_ranks"] == 2 and "library-owned RCCL communicator" in j["config"]["parallelism"]
    assert j["verify"]["ok"] is True and j["verify"]["reads"] == 400000, j.get("verify")
    want = np.load(one + ".rank0.npy")
    got = np.concatenate([np.load(two + ".rank%d.npy" % k) for k in range(2)])
    assert want.shape == got.shape and (want == got).all() and 0 < int(want.sum()) < len(want)


def test_bench_eight_ranks_exactly_as_the_driver_types_it(tmp_path):
    """Round-4 review, item 7: the first real 8-GPU run must be cheap and judge itself.  `python3 bench.py --gpus 8 --steps K --warmup W`
    (here with --reads 100000 per rank and all eight ranks on the one GPU of the box, the library's communicator over the loopback
    stand-in): one JSON line, n_gpus 8, the self-check of the N-rank stage green, and the exchange time split collective by collective
    — one all-gather of the mean qualities, eight device-side all-reduces and two host-visible sums per step — so that a run on eight
    real GPUs shows where a sub-linear result comes from."""
    import json
    shim_dir = os.path.join(ROOT, "tests", "shim")
    subprocess.check_call(["make", "-s", "-C", shim_dir])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "FLX_RANK_SORT")}
    env["FLX_RCCL_LIB"] = os.path.join(shim_dir, "libloopback_rccl.so")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--reads", "100000",
                        "--verify-reads", "20000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["reads_total"] == 800000 and j["scaling"] == "weak" and j["config"]["rccl_ranks"] == 8
    assert j["verify"]["ok"] is True and j["verify"]["flags_differing"] == 0, j.get("verify")
    split = j["stage_ms_per_step"]["comm_split"]
    assert split["allgather_means"]["calls_per_step"] == 1 and split["allgather_means"]["ms_per_step"] > 0
    assert split["allreduce_histograms_device"]["calls_per_step"] == 8
    assert split["counts_sum_host_visible"]["calls_per_step"] == 1 and 1 <= split["band_sum_host_visible"]["calls_per_step"] <= 2
    assert split["allgather_records_fallback"]["calls_per_step"] == 0 and split["broadcast_outcome_fallback"]["calls_per_step"] == 0
    assert j["config"]["allgather_bytes_per_rank"] == {"sent": 800000, "received": 6400000}
