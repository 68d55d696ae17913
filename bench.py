#!/usr/bin/env python3
"""bench.py — throughput of the Filtlong scoring hot path on MI355X.

One "step" = one pass of the hot path over the whole synthetic batch resident in HBM:
    flx_score_batch_dev   (per-read mean / sliding-window quality, hard cut-offs)
  + [N > 1: one RCCL all-gather of the per-read (mean_q, window_q, length, passed) records]
  + flx_rank_and_cut_dev  (exact statistics, normalise, final score, radix sort, --target_bases cut)

Workload (BASELINE.json configs[1], C2): 10 M synthetic reads per GPU, gamma(k=4) lengths with mean
10 kbp (1e11 bases per GPU), Phred-only, --target_bases 50g per GPU.  Weak scaling: rank r owns reads
[r*10M, (r+1)*10M) (configs[4], C5, at 8 GPUs); the global stage is replicated on every rank after the
all-gather so the threshold is exact.

Prints ONE JSON line (rank 0).  `value` = total bases over all ranks / max-over-ranks wall time of the
timed steps.  The `roofline` object is for the dominant kernel (flx_score_phred_ring), timed with HIP
events on the stream it runs on; `cpu_baseline` is the reference's own compiled code (oracle/_ref/ref_bench,
kind "reference") on a bounded sample on this box's host cores (1 thread: the reference is single-threaded).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def cpu_baseline(sample_reads):
    """Time the CPU reference on a bounded sample of the same workload (rank 0, N == 1 only)."""
    ref_bench = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    env = dict(os.environ, LANG="C", LC_ALL="C")
    cores = 1
    if os.path.exists(ref_bench):
        # ~half of the sample's bases as target, like --target_bases 50g on 1e11 bases
        out = subprocess.run([ref_bench, str(sample_reads), "0", str(sample_reads * 5000)], env=env, check=True,
                             stdout=subprocess.PIPE).stdout.decode()
        r = json.loads(out)
        return {"value": round(r["mbases_per_s"], 3), "unit": "Mbases/s", "cores": cores, "kind": "reference",
                "sample": "%d reads / %d bases of the same synthetic Phred-only workload, reference objects "
                          "(Read::Read + set_final_score + std::sort) in memory, %.1f s score + %.2f s rank"
                          % (r["reads"], r["bases"], r["score_s"], r["rank_s"]),
                "host_cores_available": os.cpu_count()}
    # fall back to the oracle restatement ("port")
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.flo_bench_phred.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    ss, rs, tb, kb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib.flo_bench_phred(sample_reads, 20250919, 0, sample_reads * 5000, ss, rs, tb, kb)
    return {"value": round(tb.value / (ss.value + rs.value) / 1e6, 3), "unit": "Mbases/s", "cores": cores,
            "kind": "port", "sample": "%d reads / %d bases, oracle restatement in memory" % (sample_reads, tb.value),
            "host_cores_available": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (default: the C2 workload)")
    ap.add_argument("--fixed-len", type=int, default=0, help="fixed read length (C1 uses 5000); 0 = gamma lengths")
    ap.add_argument("--target-frac", type=float, default=0.5, help="--target_bases as a fraction of all bases")
    ap.add_argument("--cpu-sample-reads", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--window-size", type=int, default=250)
    ap.add_argument("--global-stage", choices=("sharded", "replicated"), default="sharded",
                    help="N > 1: all-gather the mean qualities and select with all-reduced histograms (default), or "
                         "all-gather the full per-read records and replicate the single-GPU stage")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                                                       "one-GPU functional test of the N > 1 path)")
    ap.add_argument("--force-dist", action="store_true", help="take the N > 1 code path (process group, collectives) even "
                                                              "with one rank: exercises the RCCL calls on a 1-GPU box")
    ap.add_argument("--dump-flags", default="", help="write this rank's final pass flags to <path>.rank<r>.npy (tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from filtlong_amd import api, synth
    from filtlong_amd import dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched through torch.distributed.run" % args.gpus)
    device_index = local_rank % max(torch.cuda.device_count(), 1)  # == local_rank except in the one-GPU gloo test
    torch.cuda.set_device(device_index)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(args.backend)

    ctx = api.Context(device_index)
    dev = torch.device("cuda", device_index)
    n = args.reads
    first = rank * n

    # ---- build the packed batch in HBM (not timed) -----------------------------------------------
    t_setup = time.time()
    lengths = synth.lengths(n, first=first, fixed=args.fixed_len or None)
    offsets = np.zeros(n, dtype=np.uint64)
    pb = C.c_uint64()
    ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    plane_bytes = pb.value
    order = api.length_order(lengths)
    local_bases = int(lengths.astype(np.int64).sum())

    d_plane = torch.empty(plane_bytes, dtype=torch.uint8, device=dev)
    d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
    d_len = torch.from_numpy(lengths).to(dev)
    d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
    d_ids = torch.arange(first, first + n, dtype=torch.int64, device=dev)
    # packed per-read record buffer [mean f64 | window f64 | length i32 | passed u8] -> one all-gather
    d_rec = fdist.alloc_records(n, dev)
    p_mean = d_rec.data_ptr()
    p_win = p_mean + 8 * n
    p_len = p_mean + 16 * n
    p_pass = p_mean + 20 * n
    t_mean, t_win, t_len, t_pass = fdist.record_views(d_rec, n)
    t_len.copy_(d_len)
    torch.cuda.synchronize()
    ctx.synth_qual_dev(synth.SEED, d_plane.data_ptr(), plane_bytes, d_off.data_ptr(), d_len.data_ptr(),
                       d_ids.data_ptr(), n)
    del d_ids

    total_n = n * world
    if multi:
        tb = torch.tensor([local_bases], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tb)
        total_bases = int(tb.item())
    else:
        total_bases = local_bases
    target = int(total_bases * args.target_frac)
    params = api.make_params(window_size=args.window_size)
    setup_s = time.time() - t_setup

    def step():
        ctx.score_reads_dev(d_plane.data_ptr(), plane_bytes, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n,
                            params, p_mean, p_win, p_pass)
        if multi and args.global_stage == "sharded":
            # ONE RCCL all-gather of the mean qualities (the statistics fold over all of them in file order); final
            # scores and the cut are computed on the local reads, the selection's histograms all-reduced
            # (filtlong_amd/dist.py, flx_rank_and_cut_sharded_dev).
            return fdist.sharded_rank_and_cut(ctx, t_mean, t_win, t_len, t_pass, target_bases=target,
                                              total_bases=total_bases)
        if multi:
            # ONE RCCL all-gather of the full per-read records; every rank then runs the identical single-GPU stage.
            g_mean, g_win, g_len, g_pass, _counts = fdist.gather_records(d_rec, n)
            torch.cuda.synchronize()
            rep = ctx.rank_and_cut_dev(total_n, g_mean.data_ptr(), g_win.data_ptr(), g_len.data_ptr(),
                                       g_pass.data_ptr(), target_bases=target, total_bases=total_bases)
            t_pass.copy_(g_pass[rank * n:(rank + 1) * n])
            return rep
        return ctx.rank_and_cut_dev(n, p_mean, p_win, p_len, p_pass, target_bases=target, total_bases=total_bases)

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        rep = step()
    ctx.timing_enable(True)
    ctx.timing_reset()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rep = step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if multi:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    kernel_name = ctx.last_phred_kernel()
    k_ms, k_n = ctx.timing_get(kernel_name)
    rank_ms, _ = ctx.timing_get("flx_rank")
    sort_ms, _ = ctx.timing_get("flx_sort")
    ctx.timing_enable(False)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bases * args.steps / elapsed / 1e6
        # algorithmic bytes per launch of the scoring kernel: L + 8 (offset) + 17 (outputs) per read
        algo_bytes = local_bases + 25 * n
        avg_kernel_ms = k_ms / max(k_n, 1)
        achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9 if k_n else 0.0
        # HBM traffic of one launch from the PMC passes of tools/prof_phred.sh (separate rocprofv3 --pmc runs of this same
        # C2 command; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md prescribes), only when the workload is the same.
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_c2.json")
        if os.path.exists(tpath) and n == 10_000_000 and not args.fixed_len and args.window_size == 250:
            traffic = int(json.load(open(tpath))["traffic_bytes"])
        info = ctx.device_info()
        out = {
            "metric": "Mbases/s scored+sorted",
            "value": round(value, 1),
            "unit": "Mbases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s synthetic reads per GPU x %s, Phred-only, --target_bases %d (%.0f%% of bases)%s" % (
                    "{:,}".format(n), ("fixed %d bp" % args.fixed_len) if args.fixed_len else "gamma(k=4) mean 10 kbp",
                    target, args.target_frac * 100, "; C2" if (n == 10_000_000 and not args.fixed_len) else ""),
                "reads_total": total_n, "bases_total": total_bases, "window_size": args.window_size,
                "parallelism": ("1 GPU" if not multi else
                                "reads sharded by count; 1 RCCL all-gather of mean qualities + all-reduced selection histograms"
                                if args.global_stage == "sharded" else
                                "reads sharded by count; 1 RCCL all-gather of per-read records, global stage replicated"),
                "device": info["name"],
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_kernel_ms": round(avg_kernel_ms, 3), "launches": int(k_n), "algorithmic_bytes": int(algo_bytes),
            },
            "stage_ms_per_step": {"score_kernel": round(k_ms / args.steps, 3), "sort": round(sort_ms / args.steps, 3),
                                  "rank_other_kernels": round(rank_ms / args.steps, 3)},
            "cut": {"target_bases": int(rep.target_bases), "kept_bases": int(rep.kept_bases),
                    "outcome": int(rep.outcome), "audited": int(rep.audited), "exact_fallback": int(rep.exact_fallback)},
            "setup_s": round(setup_s, 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_reads)
        print(json.dumps(out), flush=True)
    if args.dump_flags:
        torch.cuda.synchronize()
        np.save("%s.rank%d.npy" % (args.dump_flags, rank), t_pass.cpu().numpy())
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
