#!/usr/bin/env python3
"""Time the global stage (flx_rank_and_cut_dev) alone on N synthetic reads2 records resident in HBM — the part that is
replicated on every rank after the all-gather (N = 8e7 for the 8-GPU C5 configuration)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=80_000_000)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from filtlong_amd import api
    ctx = api.Context(0)
    n = args.reads
    g = torch.Generator(device="cuda").manual_seed(1)
    mean = torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 40 + 59
    win = mean * (torch.rand(n, dtype=torch.float64, device="cuda", generator=g) * 0.7 + 0.3)
    length = (torch.rand(n, device="cuda", generator=g) * 20000 + 200).to(torch.int32)
    passed0 = (torch.rand(n, device="cuda", generator=g) > 0.1).to(torch.uint8)
    total = int(length.to(torch.int64).sum().item())
    passed = passed0.clone()
    torch.cuda.synchronize()
    ctx.rank_and_cut_dev(n, mean.data_ptr(), win.data_ptr(), length.data_ptr(), passed.data_ptr(), target_bases=total // 2, total_bases=total)
    ctx.timing_enable(True)
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        passed.copy_(passed0)
        torch.cuda.synchronize()
        rep = ctx.rank_and_cut_dev(n, mean.data_ptr(), win.data_ptr(), length.data_ptr(), passed.data_ptr(),
                                   target_bases=total // 2, total_bases=total)
    el = (time.perf_counter() - t0) / args.steps
    parts = {k: round(ctx.timing_get(k)[0] / args.steps, 3) for k in ("flx_rank_stats", "flx_rank_final_score", "flx_rank_select",
                                                                        "flx_rank_passed_bases", "flx_sort", "flx_rank_cut")}
    print(json.dumps({"reads": n, "ms_per_call": round(el * 1e3, 3), "kept_bases": rep.kept_bases, "audited": rep.audited, "parts_ms": parts}))


if __name__ == "__main__":
    main()
