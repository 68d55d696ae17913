#!/usr/bin/env python3
"""K-mer-mode measurement (BASELINE.json configs[2] / configs[3]; parity-test configurations, not the bench line):
N synthetic long reads drawn from a 5 Mbp random reference, scored through flx_score_batch_dev with
the reference 16-mer set built on the device.  Reports per-kernel milliseconds, lookups/s and Mbases/s.

  python tools/bench_kmer.py [--reads 1000000] [--trim-split] [--short-reads]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--ref-len", type=int, default=5_000_000)
    ap.add_argument("--trim-split", action="store_true", help="--trim --split 500 (C4)")
    ap.add_argument("--short-reads", action="store_true", help="reference = 40x of error-free 100 bp pairs (C4) instead of the assembly")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--profile", type=int, default=0, help="read profile (filtlong_amd/synth.py: seq_read): 0 substitutions only, 1 indels, 2 30 %% unrelated reads")
    args = ap.parse_args()
    import torch
    from filtlong_amd import api, synth, _lib

    ctx = api.Context(0)
    dev = torch.device("cuda", 0)
    n = args.reads
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, args.ref_len)
    t0 = time.time()
    ks = api.Kmers(ctx)
    if args.short_reads:
        # 1e6 pairs per 5 Mbp at 100 bp = 40x; -1 forward substring, -2 reverse complement 350 bp downstream
        npairs = args.ref_len // 5
        starts = (synth.mix(synth.SEED, synth.STREAM_START, np.arange(npairs, dtype=np.uint64) + np.uint64(1 << 40), 0)
                  % np.uint64(args.ref_len - 450)).astype(np.int64)
        comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
        r1 = [ref[s:s + 100].tobytes() for s in starts]
        r2 = [comp[ref[s + 350:s + 450]][::-1].tobytes() for s in starts]
        ks.add_read_fastqs([r1, r2])
    else:
        ks.add_assembly_fasta([ref.tobytes()])
    ks.finalize()
    build_s = time.time() - t0

    lengths = synth.lengths(n)
    offsets = np.zeros(n, dtype=np.uint64)
    pb = C.c_uint64()
    ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    order = api.length_order(lengths)
    bases = int(lengths.astype(np.int64).sum())
    d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lengths).to(dev)
    d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
    d_ref = torch.from_numpy(ref).to(dev)
    d_ids = torch.arange(n, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n,
                      d_ref.data_ptr(), args.ref_len, profile=args.profile)
    cap = 4 * n
    t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (
        ("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8), ("first", n, torch.int32),
        ("last", n, torch.int32), ("coff", n + 1, torch.int64), ("crng", 2 * cap, torch.int32), ("cmean", cap, torch.float64),
        ("cwin", cap, torch.float64), ("cpass", cap, torch.uint8))}
    torch.cuda.synchronize()
    params = api.make_params(trim=args.trim_split, split=500 if args.trim_split else None)
    s = _lib.Scores()
    s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(),
                                                      t["first"].data_ptr(), t["last"].data_ptr())
    s.child_offsets, s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (
        t["coff"].data_ptr(), t["crng"].data_ptr(), t["cmean"].data_ptr(), t["cwin"].data_ptr(), t["cpass"].data_ptr())
    s.child_capacity = cap
    ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, params, s)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rc = ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, params, s)
    ctx.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    cover_ms, cn = ctx.timing_get("flx_score_kmer_cover")
    fold_ms, fn = ctx.timing_get("flx_score_kmer_fold")
    lookups = bases - 15 * n
    out = {"reads": n, "bases": bases, "set_size": len(ks), "set_build_s": round(build_s, 2), "ms_per_call": round(el * 1e3, 2),
           "cover_ms": round(cover_ms / args.steps, 2), "fold_ms": round(fold_ms / args.steps, 2),
           "Glookups_per_s": round(lookups / (cover_ms / args.steps * 1e-3) / 1e9, 2), "Mbases_per_s": round(bases / el / 1e6, 1),
           "children": int(s.n_children), "mean_q_avg": float(t["mean"].mean().item()), "rc": rc,
           "fold_on_integer_grid": ctx.last_kmer_fold_grid(), "locus": ctx.last_kmer_locus(), "cover_kernel": ctx.last_kmer_cover(), "profile": args.profile}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
