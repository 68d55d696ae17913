#!/usr/bin/env python3
"""Counters of the instrumented k_kmer_cover_q<.., INDELS = true> (tools/exp/cover_queue_stats.hip): usage cq_stats.py [reads] [profile]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from filtlong_amd import api, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = api.Context(0); dev = torch.device("cuda", 0); REF = 5_000_000
ref = synth.bases_read(synth.STREAM_REF, 0, 0, REF)
ks = api.Kmers(ctx); ks.add_assembly_fasta([ref.tobytes()]); ks.finalize()
lengths = synth.lengths(n); offsets = np.zeros(n, dtype=np.uint64); pb = C.c_uint64()
ctx.L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
order = api.length_order(lengths)
d_plane = torch.empty(pb.value, dtype=torch.uint8, device=dev)
d_off = torch.from_numpy(offsets.view(np.int64)).to(dev); d_len = torch.from_numpy(lengths).to(dev)
d_ord = torch.from_numpy(order.view(np.int32)).to(dev); d_ref = torch.from_numpy(ref).to(dev)
d_ids = torch.arange(n, dtype=torch.int64, device=dev)
ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n, d_ref.data_ptr(), REF, profile=prof)
t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8), ("first", n, torch.int32), ("last", n, torch.int32), ("coff", n + 1, torch.int64))}
s = _lib.Scores()
s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(), t["first"].data_ptr(), t["last"].data_ptr())
s.child_offsets = t["coff"].data_ptr()
ctx.L.flx_debug_cq_stats_reset()
ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, api.make_params(), s)
ctx.synchronize()
out = (C.c_ulonglong * 32)(); ctx.L.flx_debug_cq_stats.argtypes = [C.c_void_p]; ctx.L.flx_debug_cq_stats(out)
names = ["lane_diag_spans", "whole_lanes", "matched", "matched_shifted", "unmatched_and_nothing_known", "lanes_with_known", "matched_at_12_or_more", "-", "spans", "need_lanes", "valid_lanes", "exact15_loads", "entries_found_by_lookup", "batches", "entries"]
r = {nm: int(out[i]) for i, nm in enumerate(names)}
sp = max(r["spans"], 1)
r["per_span"] = {k: round(v / sp, 3) for k, v in r.items() if k != "spans"}
r["handed_over"] = ctx.last_kmer_handed_over()
print(json.dumps(r, indent=1))
