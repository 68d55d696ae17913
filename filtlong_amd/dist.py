"""Multi-GPU plumbing for the scoring hot path: reads are sharded by COUNT across ranks (contiguous blocks of
file order) and every rank scores its own shard.  The global stage (exact statistics, normalise, final score, cut;
reference src/main.cpp:169-261) then runs SHARDED (`sharded_rank_and_cut`):

  * ONE all-gather of the mean qualities (8 bytes per reads2 entry) — the statistics of main.cpp:170-196 are
    order-dependent folds over all of them, so every rank needs the array in file order;
  * final scores, keys and pass flags are computed for the local entries only; the cut is a weighted radix selection
    whose per-byte histograms of summed read lengths are all-reduced (a dozen collectives of <= 2 KB);
  * the rare cases in which only the reference's own std::sort order over all reads decides (NaN scores, equal scores
    straddling the cut) fall back to the REPLICATED stage: one all-gather of the full per-read records
    (mean_q f64, window_q f64, length i32, passed u8 — 21 bytes per entry, `gather_records`) and the single-GPU
    flx_rank_and_cut_dev on every rank.

The final score cannot be computed before the exchange because it depends on the global mean / stdev / min / max
(SURVEY §8e), so qualities — not scores — travel.

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI on MI355X) with device tensors, "gloo" with
CPU tensors (used by the world_size-2 CPU tests).
"""
import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 21  # 8 + 8 + 4 + 1


def shard_range(n_total, rank, world):
    """Contiguous block of file order owned by `rank`: [lo, hi).  Blocks differ by at most one read."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def record_views(buf, n):
    """Views of a packed record buffer (uint8[21 * n], device or CPU): mean f64[n], window f64[n], length i32[n],
    passed u8[n].  The buffer layout is field-major so each view is contiguous and 8/4-byte aligned."""
    return (buf[0:8 * n].view(torch.float64), buf[8 * n:16 * n].view(torch.float64),
            buf[16 * n:20 * n].view(torch.int32), buf[20 * n:21 * n])


def alloc_records(n, device):
    # 8-byte aligned start; 21*n bytes rounded up so that padded all-gathers stay aligned
    return torch.zeros(REC_BYTES * n + (-(REC_BYTES * n)) % 8, dtype=torch.uint8, device=device)


def gather_records(local_buf, n_local, group=None):
    """All-gather the packed records of every rank.  Returns (mean, window, length, passed, counts) where the four
    arrays hold ALL ranks' reads2 entries concatenated in rank (= file) order.  Shards may differ in size
    (children make reads2 counts unequal): a tiny all-gather of counts comes first, shards are padded to the
    largest, and ONE all_gather_into_tensor moves the payload."""
    world = dist.get_world_size(group)
    device = local_buf.device
    counts_t = torch.zeros(world, dtype=torch.int64, device=device)
    mine = torch.tensor([n_local], dtype=torch.int64, device=device)
    _all_gather_flat(counts_t, mine, group)
    counts = [int(c) for c in counts_t.cpu()]
    n_max = max(counts) if counts else 0
    total = sum(counts)
    slot = REC_BYTES * n_max + (-(REC_BYTES * n_max)) % 8
    if n_local == n_max and local_buf.numel() == slot:
        send = local_buf  # equal shards (the weak-scaling bench): the local buffer IS the exchange layout
    else:
        send = torch.zeros(slot, dtype=torch.uint8, device=device)
        # re-pack the local fields at the padded strides of the exchange buffer
        lm, lw, ll, lp = record_views(local_buf, n_local)
        sm, sw, sl, sp = record_views(send, n_max)
        sm[:n_local].copy_(lm); sw[:n_local].copy_(lw); sl[:n_local].copy_(ll); sp[:n_local].copy_(lp)
    recv = torch.empty(world * slot, dtype=torch.uint8, device=device)
    _all_gather_flat(recv, send, group)
    g_mean = torch.empty(total, dtype=torch.float64, device=device)
    g_win = torch.empty(total, dtype=torch.float64, device=device)
    g_len = torch.empty(total, dtype=torch.int32, device=device)
    g_pass = torch.empty(total, dtype=torch.uint8, device=device)
    at = 0
    for r in range(world):
        rm, rw, rl, rp = record_views(recv[r * slot:(r + 1) * slot], n_max)
        c = counts[r]
        g_mean[at:at + c].copy_(rm[:c]); g_win[at:at + c].copy_(rw[:c]); g_len[at:at + c].copy_(rl[:c]); g_pass[at:at + c].copy_(rp[:c])
        at += c
    return g_mean, g_win, g_len, g_pass, counts


def local_slice(counts, rank):
    """[lo, hi) of this rank's entries inside the gathered arrays."""
    lo = sum(counts[:rank])
    return lo, lo + counts[rank]


def numpy_records_to_buf(mean_q, window_q, length, passed, device="cpu"):
    """Pack numpy arrays into a record buffer (tests / host paths)."""
    n = len(mean_q)
    buf = alloc_records(n, device)
    m, w, l, p = record_views(buf, n)
    m.copy_(torch.from_numpy(np.ascontiguousarray(mean_q, dtype=np.float64)))
    w.copy_(torch.from_numpy(np.ascontiguousarray(window_q, dtype=np.float64)))
    l.copy_(torch.from_numpy(np.ascontiguousarray(length, dtype=np.int32)))
    p.copy_(torch.from_numpy(np.ascontiguousarray(passed, dtype=np.uint8)))
    return buf


# --------------------------------------------------------------------------------------------------------------------
# sharded global stage
# --------------------------------------------------------------------------------------------------------------------
def _all_gather_flat(recv, send, group=None):
    """all_gather_into_tensor; device tensors on a CPU-only backend (gloo in the one-GPU tests) are staged."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send, group=group)


def gather_means(local_mean, group=None):
    """All ranks' mean qualities in rank (= file) order: (f64[total], counts).  `local_mean` is f64[n_local] (device or
    CPU).  Equal shards (the weak-scaling bench) need no copy at all: the receive buffer IS the global array."""
    world = dist.get_world_size(group)
    device = local_mean.device
    n_local = local_mean.numel()
    counts_t = torch.zeros(world, dtype=torch.int64, device=device)
    _all_gather_flat(counts_t, torch.tensor([n_local], dtype=torch.int64, device=device), group)
    counts = [int(c) for c in counts_t.cpu()]
    n_max = max(counts) if counts else 0
    if n_max == 0:
        return torch.empty(0, dtype=torch.float64, device=device), counts
    if all(c == n_max for c in counts):
        recv = torch.empty(world * n_max, dtype=torch.float64, device=device)
        _all_gather_flat(recv, local_mean.contiguous(), group)
        return recv, counts
    send = torch.zeros(n_max, dtype=torch.float64, device=device)
    send[:n_local].copy_(local_mean)
    recv = torch.empty(world * n_max, dtype=torch.float64, device=device)
    _all_gather_flat(recv, send, group)
    out = torch.empty(sum(counts), dtype=torch.float64, device=device)
    at = 0
    for r, c in enumerate(counts):
        out[at:at + c].copy_(recv[r * n_max:r * n_max + c])
        at += c
    return out, counts


def make_reduce(group=None, device=None):
    """The all-reduce the library calls back into (flx_allreduce_u64_fn): sums a small host buffer over all ranks in
    place.  nccl (= RCCL) reduces on the device, gloo on the host."""
    backend = dist.get_backend(group)

    def reduce(buf):
        t = torch.from_numpy(buf.view(np.int64))  # shares the library's host buffer; wrap-around sums are the same bits
        if backend == "nccl":
            d = t.to(device)
            dist.all_reduce(d, group=group)
            t.copy_(d)
        else:
            dist.all_reduce(t, group=group)
    return reduce


def sharded_rank_and_cut(ctx, mean, window, length, passed, group=None, final_score=None, **cut):
    """Global stage over the reads2 entries of all ranks; `mean`/`window`/`length`/`passed` are this rank's device
    tensors (f64, f64, i32, u8), `passed` is updated in place with the final flags.  `cut` = the keyword arguments of
    Context.rank_and_cut_dev (weights, target_bases, keep_percent, total_bases — the GLOBAL total, required with a
    threshold); `final_score` (optional f64 tensor of the local size) receives the local final scores.  Returns the
    (global) report."""
    if "d_final_score" in cut:
        raise ValueError("pass final_score=<local f64 tensor>, not a raw d_final_score pointer (the replicated fallback "
                         "needs a buffer of the global size)")
    if (cut.get("target_bases") is not None or cut.get("keep_percent") is not None) and not cut.get("total_bases"):
        raise ValueError("total_bases (global sum of the original read lengths, src/main.cpp:89) is required with a threshold")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = mean.numel()
    g_mean, counts = gather_means(mean, group)
    first = sum(counts[:rank])
    torch.cuda.synchronize(mean.device)  # the library works on its own stream
    rep, need_replicated = ctx.rank_and_cut_sharded_dev(
        sum(counts), g_mean.data_ptr(), first, n_local, window.data_ptr(), length.data_ptr(), passed.data_ptr(),
        rank, world, reduce=make_reduce(group, mean.device),
        d_final_score=final_score.data_ptr() if final_score is not None else None, **cut)
    if not need_replicated:
        return rep
    # exact tie / NaN fallback: everything to every rank, the single-GPU stage replicated
    buf = alloc_records(n_local, mean.device)
    m, w, l, p = record_views(buf, n_local)
    m.copy_(mean); w.copy_(window); l.copy_(length); p.copy_(passed)
    a_mean, a_win, a_len, a_pass, counts = gather_records(buf, n_local, group)
    torch.cuda.synchronize(mean.device)
    all_scores = torch.empty(sum(counts), dtype=torch.float64, device=mean.device) if final_score is not None else None
    rep = ctx.rank_and_cut_dev(sum(counts), a_mean.data_ptr(), a_win.data_ptr(), a_len.data_ptr(), a_pass.data_ptr(),
                               d_final_score=all_scores.data_ptr() if all_scores is not None else None, **cut)
    lo, hi = local_slice(counts, rank)
    passed.copy_(a_pass[lo:hi])
    if final_score is not None:
        final_score.copy_(all_scores[lo:hi])
    torch.cuda.synchronize(mean.device)
    return rep
