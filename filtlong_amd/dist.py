"""Multi-GPU plumbing for the scoring hot path: reads are sharded by COUNT across ranks (contiguous blocks of
file order), every rank scores its own shard, and ONE all-gather of the per-read records
(mean_q f64, window_q f64, length i32, passed u8 — 21 bytes per reads2 entry) gives every rank the global
arrays in file order, on which the identical global stage (exact statistics, normalise, sort, cut;
reference src/main.cpp:169-261) is then replicated.  The final score cannot be computed before the exchange
because it depends on the global mean / stdev / min / max, so the records — not scores — travel.

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
    dist.all_gather_into_tensor(counts_t, mine, group=group)
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
    dist.all_gather_into_tensor(recv, send, group=group)
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
