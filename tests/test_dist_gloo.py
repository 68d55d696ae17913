"""The N > 1 path on CPU: world_size-2 `gloo` processes shard reads by count, exchange the per-read records with
ONE all-gather (filtlong_amd/dist.py) and replicate the global stage; every rank must reach the pass set of the
single-process run.  (On the GPU box the same code runs over nccl = RCCL; the scoring and the global stage are
then the HIP library instead of the oracle stand-ins used here.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle
from filtlong_amd import dist as fdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_reads2(n, seed):
    rng = np.random.RandomState(seed)
    mean = rng.uniform(60, 99, n)
    window = mean * rng.uniform(0.3, 1.05, n)
    length = np.clip(rng.gamma(4, 2500, n), 1, None).astype(np.int32)
    passed = (rng.random_sample(n) > 0.1).astype(np.uint8)
    return mean, window, length, passed


def _worker(rank, world, port, n_total, shard_sizes, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mean, window, length, passed = make_reads2(n_total, seed)
        if shard_sizes is None:
            lo, hi = fdist.shard_range(n_total, rank, world)
        else:  # unequal shards, as --trim/--split children make them
            lo = sum(shard_sizes[:rank]); hi = lo + shard_sizes[rank]
        # "local scoring": this rank only knows its own block
        buf = fdist.numpy_records_to_buf(mean[lo:hi], window[lo:hi], length[lo:hi], passed[lo:hi])
        g_mean, g_win, g_len, g_pass, counts = fdist.gather_records(buf, hi - lo)
        assert counts == ([hi_ - lo_ for lo_, hi_ in (fdist.shard_range(n_total, r, world) for r in range(world))]
                          if shard_sizes is None else list(shard_sizes))
        # gathered arrays are the global arrays, bit for bit, in file order
        assert (g_mean.numpy().view(np.uint64) == mean.view(np.uint64)).all()
        assert (g_win.numpy().view(np.uint64) == window.view(np.uint64)).all()
        assert (g_len.numpy() == length).all() and (g_pass.numpy() == passed).all()
        tot = int(length.astype(np.int64).sum())
        res = _oracle.rank_and_cut(g_mean.numpy(), g_win.numpy(), g_len.numpy(), g_pass.numpy(), target_bases=tot // 2,
                                   total_bases=tot)
        a, b = fdist.local_slice(counts, rank)
        np.save(os.path.join(out_dir, "rank%d.npy" % rank), res["passed"][a:b])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,shards", [(1001, None), (64, None), (777, (500, 277)), (10, (0, 10))])
def test_two_ranks_reach_the_single_process_pass_set(tmp_path, n_total, shards):
    world, seed = 2, 123
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, shards, seed, str(tmp_path)), nprocs=world, join=True)
    mean, window, length, passed = make_reads2(n_total, seed)
    tot = int(length.astype(np.int64).sum())
    want = _oracle.rank_and_cut(mean, window, length, passed, target_bases=tot // 2, total_bases=tot)["passed"]
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "rank%d.npy" % r)) for r in range(world)])
    assert (got == want).all()


def test_shard_range_partitions_file_order():
    for n, w in ((10, 3), (0, 2), (7, 8), (80_000_000, 8)):
        edges = [fdist.shard_range(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in edges) - min(h - l for l, h in edges) <= 1


def _means_worker(rank, world, port, sizes, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(5)
        allm = rng.uniform(50, 99, sum(sizes))
        lo = sum(sizes[:rank])
        g, counts = fdist.gather_means(torch.from_numpy(allm[lo:lo + sizes[rank]].copy()))
        assert counts == list(sizes)
        assert (g.numpy().view(np.uint64) == allm.view(np.uint64)).all()
        # the callback handed to the C ABI: an in-place sum over ranks of a host uint64 buffer (wrap-around included)
        red = fdist.make_reduce()
        buf = np.array([rank + 1, 2 ** 63 + 5, 0 if rank else 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        red(buf)
        want = np.array([3, (2 * (2 ** 63 + 5)) % 2 ** 64, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
        assert (buf == want).all(), buf
        np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(500, 500), (300, 701), (0, 64)])
def test_gather_means_and_reduce_callback(tmp_path, sizes):
    port = _free_port()
    mp.spawn(_means_worker, args=(2, port, sizes, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), "ok%d.npy" % r)) for r in range(2))
