# which fold path leaves outputs unwritten?  Device outputs pre-filled with a sentinel, k-mer reads of tests/_cases.py (k0 has length 0)
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import _cases
from filtlong_amd import api, _lib
ctx = api.Context(0); dev = torch.device("cuda", 0)
contigs = _cases.synth_reference()
reads = _cases.kmer_reads(contigs)
ks = api.Kmers(ctx); ks.add_assembly_fasta(contigs); ks.finalize()
plane, offsets, lengths = api.pack_reads([s for _, s, _ in reads])
order = api.length_order(lengths)
n = len(lengths)
d_plane = torch.from_numpy(plane).to(dev); d_off = torch.from_numpy(offsets.view(np.int64)).to(dev); d_len = torch.from_numpy(lengths).to(dev); d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
cap = 64 * n
def run(env):
    for k in ("FLX_KMER_FOLD_STREAMS", "FLX_KMER_FOLD_GRID", "FLX_KMER_FOLD"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = {k: torch.full((sz,), fill, dtype=dt, device=dev) for k, sz, dt, fill in (
        ("mean", n, torch.float64, 12345.0), ("win", n, torch.float64, 12345.0), ("pass", n, torch.uint8, 77), ("first", n, torch.int32, -7),
        ("last", n, torch.int32, -7), ("coff", n + 1, torch.int64, -7), ("crng", 2 * cap, torch.int32, -7), ("cmean", cap, torch.float64, 12345.0),
        ("cwin", cap, torch.float64, 12345.0), ("cpass", cap, torch.uint8, 77))}
    torch.cuda.synchronize()
    s = _lib.Scores()
    s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(), t["first"].data_ptr(), t["last"].data_ptr())
    s.child_offsets, s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (t["coff"].data_ptr(), t["crng"].data_ptr(), t["cmean"].data_ptr(), t["cwin"].data_ptr(), t["cpass"].data_ptr())
    s.child_capacity = cap
    rc = ctx.score_kmer_dev(ks, d_plane.data_ptr(), plane.nbytes, d_off.data_ptr(), d_len.data_ptr(), d_ord.data_ptr(), n, api.make_params(trim=True, split=100), s)
    torch.cuda.synchronize()
    m, w, p = t["mean"].cpu().numpy(), t["win"].cpu().numpy(), t["pass"].cpu().numpy()
    unw = [i for i in range(n) if m[i] == 12345.0 or w[i] == 12345.0 or p[i] == 77]
    nc = int(s.n_children)
    cm, cw, cp = t["cmean"].cpu().numpy()[:nc], t["cwin"].cpu().numpy()[:nc], t["cpass"].cpu().numpy()[:nc]
    unc = int(((cm == 12345.0) | (cw == 12345.0) | (cp == 77)).sum())
    print(env, "rc", rc, "grid", ctx.last_kmer_fold_grid(), "reads unwritten:", [(i, int(lengths[i])) for i in unw], "children", nc, "unwritten children", unc, "k0:", m[0], w[0], p[0])
    return m, w, p
a = run({})
for env in ({"FLX_KMER_FOLD_GRID": "0"}, {"FLX_KMER_FOLD_STREAMS": "global"}, {"FLX_KMER_FOLD_STREAMS": "global", "FLX_KMER_FOLD_GRID": "0"}, {"FLX_KMER_FOLD": "words"}, {"FLX_KMER_FOLD": "bits"}):
    b = run(env)
    same = all((x.view(np.uint64) if x.dtype == np.float64 else x).tolist() == (y.view(np.uint64) if y.dtype == np.float64 else y).tolist() for x, y in zip(a, b))
    print("   same as default:", same)
