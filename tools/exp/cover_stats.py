#!/usr/bin/env python3
"""Counters of the instrumented cover kernel (tools/exp/score_kmer_stats.hip, FLX_LIB_PATH=filtlong_amd/lib/exp/libfiltlong_hip_stats.so):
how many lanes of a span carry work in the prefilter / search blocks.  usage: cover_stats.py [reads] [--short-reads]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from filtlong_amd import api, synth, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
short = "--short-reads" in sys.argv
ctx = api.Context(0)
dev = torch.device("cuda", 0)
REF = 5_000_000
ref = synth.bases_read(synth.STREAM_REF, 0, 0, REF)
ks = api.Kmers(ctx)
if short:
    npairs = REF // 5
    starts = (synth.mix(synth.SEED, synth.STREAM_START, np.arange(npairs, dtype=np.uint64) + np.uint64(1 << 40), 0) % np.uint64(REF - 450)).astype(np.int64)
    comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    ks.add_read_fastqs([[ref[s:s + 100].tobytes() for s in starts], [comp[ref[s + 350:s + 450]][::-1].tobytes() for s in starts]])
else:
    ks.add_assembly_fasta([ref.tobytes()])
ks.finalize()
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
ctx.synth_seq_dev(synth.SEED, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ids.data_ptr(), n, d_ref.data_ptr(), REF)
t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8),
     ("first", n, torch.int32), ("last", n, torch.int32), ("coff", n + 1, torch.int64))}
s = _lib.Scores()
s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(), t["first"].data_ptr(), t["last"].data_ptr())
s.child_offsets = t["coff"].data_ptr()
ctx.L.flx_debug_cover_stats_reset()
ctx.score_kmer_dev(ks, d_plane.data_ptr(), pb.value, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, api.make_params(), s)
ctx.synchronize()
out = (C.c_ulonglong * 32)()
ctx.L.flx_debug_cover_stats.argtypes = [C.c_void_p]
ctx.L.flx_debug_cover_stats(out)
names = ["spans", "lanes_valid", "lanes_settled", "lanes_open_windows", "lanes_w1", "spans_any_w1", "lanes_w2", "spans_any_w2", "spans_round2_block",
         "probe_calls", "probe_lanes", "seed_iters", "compare_calls", "lanes_needB", "spans_any_needB", "pre11_loads", "exact15_loads", "lanes_found_by_lookup",
         "seed_lanes", "lanes_open_not_settled", "needB_0", "needB_1_4", "needB_5_8", "needB_9_16", "needB_17_24", "needB_25_32", "needB_33_48", "needB_49_64"]
r = {nm: int(out[i]) for i, nm in enumerate(names)}
r["bases"] = bases
r["per_span"] = {k: round(v / max(r["spans"], 1), 3) for k, v in r.items() if k not in ("spans", "bases")}
print(json.dumps(r, indent=1))
