"""The C++ multi-rank path (filtlong_amd/csrc/comm.hip: flx_comm_*, flx_rank_and_cut_comm; the command line's --gpus N)
with MORE THAN ONE RANK on hardware.  RCCL refuses two ranks on one device and the test boxes have one GPU, so the
library's eight RCCL entry points are served by tests/shim/loopback_rccl.cpp (FLX_RCCL_LIB): several processes on the
same GPU, every collective staged through shared host memory.  What is under test is everything ABOVE those entry points —
shard bookkeeping, the all-gather with equal and unequal counts, the device-side histogram sums between the selection
passes, the band exchange and audit, the replicated fallback for NaN scores and ties, the part files of the command line —
against the single-GPU stage on the same data, bit for bit."""
import json
import os
import subprocess
import sys

import zlib

import numpy as np
import pytest

import _cases
from filtlong_amd import api

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "shim", "libloopback_rccl.so")
BIN = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")


@pytest.fixture(scope="module")
def shim():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "shim")])
    assert os.path.exists(SHIM)
    return SHIM


def run_ranks(work, case, world, shim):
    env = dict(os.environ, FLX_RCCL_LIB=shim)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_comm2_worker.py"), str(r), str(world), work, case],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d: %s" % (r, se.decode()[-2000:])
    return [np.load(os.path.join(work, "%s.out%d.npz" % (case, r))) for r in range(world)]


def make_case(rng, n, kind):
    mean = rng.uniform(40, 100, n)
    if kind == "equal-quality":       # stdev 0 -> NaN scores -> every rank must fall back to the replicated stage
        mean = np.full(n, 77.0)
    window = mean * rng.uniform(0.3, 1.0, n)
    length = rng.randint(200, 30000, n).astype(np.int32)
    passed = (rng.uniform(0, 1, n) < 0.9).astype(np.uint8)
    if kind == "ties":                # groups of four equal qualities with different lengths; with length_weight 0 they tie
        g = rng.permutation(n) // 4   # exactly, and the kept set depends on the reference's own (unstable) sort order
        mean = mean[g * 4 % n]
        window = window[g * 4 % n]
    return mean, window, length, passed


CASES = [
    # name, kind, n, world, block boundaries as fractions, kwargs
    ("even2", "random", 40000, 2, [0, 0.5, 1], {"target_bases": "half"}),
    ("uneven2", "random", 30011, 2, [0, 0.37, 1], {"target_bases": "half"}),
    ("uneven3", "random", 50021, 3, [0, 0.2, 0.21, 1], {"keep_percent": 35.0}),
    ("empty-rank", "random", 20000, 3, [0, 0.6, 0.6, 1], {"target_bases": "half"}),
    ("no-cut", "random", 20000, 2, [0, 0.5, 1], {}),
    ("not-enough", "random", 20000, 2, [0, 0.5, 1], {"target_bases": "huge"}),
    ("ties", "ties", 20000, 2, [0, 0.5, 1], {"target_bases": "half", "length_weight": 0.0}),
    ("equal-quality", "equal-quality", 6000, 2, [0, 0.5, 1], {"target_bases": "half"}),
    ("weights", "random", 30000, 2, [0, 0.45, 1], {"target_bases": "half", "length_weight": 2.0, "mean_q_weight": 0.5, "window_q_weight": 3.0}),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_multi_rank_stage_matches_single_gpu(tmp_path, shim, case):
    name, kind, n, world, fracs, kw = case
    rng = np.random.RandomState(zlib.crc32(name.encode()) % 2 ** 31)
    mean, window, length, passed = make_case(rng, n, kind)
    total = int(length.astype(np.int64).sum())
    kw = dict(kw)
    if kw.get("target_bases") == "half":
        kw["target_bases"] = total // 2
    if kw.get("target_bases") == "huge":
        kw["target_bases"] = total * 2
    bounds = np.array([int(round(f * n)) for f in fracs], dtype=np.int64)
    work = str(tmp_path)
    np.savez(os.path.join(work, name + ".npz"), mean=mean, window=window, length=length, passed=passed, bounds=bounds,
             kw=json.dumps(kw))
    outs = run_ranks(work, name, world, shim)
    ctx = api.Context(0)
    want = ctx.rank_and_cut(mean, window, length, passed, total_bases=total, want_scores=True, **kw)
    ctx.close()
    got_pass = np.concatenate([o["passed"] for o in outs])
    assert len(got_pass) == n and (got_pass == want["passed"]).all(), "%s: pass flags differ at %s" % (
        name, np.nonzero(got_pass != want["passed"])[0][:10])
    rep = want["report"]
    for r, o in enumerate(outs):
        assert int(o["total_bases"]) == total
        assert [int(x) for x in o["report"][:3]] == [rep.target_bases, rep.kept_bases, rep.outcome], (name, r)
        st = o["stats"]
        for a, b in zip(st, (rep.mean_quality, rep.stdev_quality, rep.min_z, rep.max_z)):
            assert a == b or (np.isnan(a) and np.isnan(b)), (name, r, st)
    if kind == "random" and kw:
        got_fs = np.concatenate([o["final_score"] for o in outs])
        assert np.allclose(got_fs, want["final_score"], rtol=1e-12, atol=0, equal_nan=True)
    assert all(int(o["report"][3]) == rep.exact_fallback for o in outs), "exact-fallback flag differs from the single-GPU stage"
    if kind in ("ties", "equal-quality"):
        assert rep.exact_fallback == 1, "this case is meant to need the reference's own sort order"


def test_cli_two_ranks_on_one_gpu(tmp_path, shim):
    """--gpus 2 (forked ranks, id file, part files stitched by rank 0) gives the bytes of the one-rank run: Phred mode with a
    target, and k-mer mode with --trim --split (children make the shards uneven)."""
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes(n=3000, length=3000))
    contigs = _cases.synth_reference()
    fa = tmp_path / "ref.fasta"
    fa.write_bytes(_cases.fasta_bytes(contigs))
    kq = tmp_path / "kmer.fastq"
    kq.write_bytes(_cases.long_fastq_bytes(_cases.kmer_reads(contigs)))
    env1 = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env1.pop(k, None)
    env2 = dict(env1, FLX_RCCL_LIB=shim, FLX_DEVICE="0")
    for args in (["--target_bases", "4000000", str(fq)],
                 ["--keep_percent", "60", "--min_length", "1000", str(fq)],
                 ["-a", str(fa), "--trim", "--split", "100", "--target_bases", "200000", str(kq)],
                 ["-a", str(fa), "--trim", "--split", "100", str(kq)]):
        one = subprocess.run([BIN] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env1, timeout=300)
        for gpus in ("2", "3") + (("8",) if args[0] == "--target_bases" else ()):
            two = subprocess.run([BIN, "--gpus", gpus] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env2, timeout=300)
            assert one.returncode == 0 and two.returncode == 0, (args, gpus, two.stderr.decode()[-1500:])
            assert len(one.stdout) > 0 and two.stdout == one.stdout, (args, gpus)
            keep = lambda e: [l.split("\r")[-1] for l in e.decode().split("\n") if any(t in l for t in ("target:", "keeping", "after ", "not enough", "already"))]
            assert keep(two.stderr) == keep(one.stderr), (args, gpus)


def test_cli_gpus_launch_fails_fast_and_clean(tmp_path):
    """--gpus N must not hang or leave anything behind when the job cannot start: more ranks than visible GPUs is refused
    before any fork; ranks that cannot form a communicator (here: real RCCL refuses three ranks on ONE device) end the whole job
    with a message, exit code 1, no orphan process and no temporary directory."""
    import glob
    import time
    fq = tmp_path / "c1.fastq"
    fq.write_bytes(_cases.c1_fastq_bytes(n=200, length=2000))
    env = dict(os.environ, LANG="C", LC_ALL="C", TMPDIR=str(tmp_path))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FLX_DEVICE"):
        env.pop(k, None)
    t0 = time.time()
    r = subprocess.run([BIN, "--gpus", "60", "--target_bases", "100000", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env, timeout=120)
    assert r.returncode == 1 and b"--gpus 60 but only" in r.stderr and r.stdout == b""
    env2 = dict(env, FLX_DEVICE="0")
    env2.pop("FLX_RCCL_LIB", None)
    r = subprocess.run([BIN, "--gpus", "3", "--target_bases", "100000", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=env2, timeout=120)
    assert r.returncode == 1 and b"Error" in r.stderr and r.stdout == b""
    assert time.time() - t0 < 90
    assert glob.glob(str(tmp_path / "flx_*")) == []  # the private directory of the output parts is gone
    # a WORLD_SIZE inherited from some launcher's shell does not turn a plain run into a rank waiting for peers
    env3 = dict(env, WORLD_SIZE="4", RANK="0")
    one = subprocess.run([BIN, "--target_bases", "100000", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    two = subprocess.run([BIN, "--target_bases", "100000", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env3, timeout=120)
    assert one.returncode == 0 and two.returncode == 0 and two.stdout == one.stdout and len(one.stdout) > 0
