"""End-to-end composition of the hot path the way the reference's main() composes it
(src/main.cpp:63-313), parametrised by backend so the SAME driver checks the CPU oracle (here) and
the HIP library (on the GPU box) against the golden outputs of the real reference binary.

TEST INFRASTRUCTURE.  The product's host side is the C++ CLI; this file only exists so that parity
tests read like the reference's own black-box tests (ordered output names, child coordinates,
summary numbers).
"""
import numpy as np

import _oracle


class OracleBackend:
    name = "oracle"

    def kmers(self, assembly=None, short_files=None):
        if assembly is None and not short_files:
            return None
        ks = _oracle.KmerSet()
        if assembly is not None:
            ks.add_assembly(assembly)
        for f in (short_files or []):
            ks.add_short_reads(f)
        return ks

    def kmers_empty(self, ks):
        return ks is None or len(ks) == 0

    def score(self, reads, pkw, ks):
        p = _oracle.make_params(**pkw)
        kmer_mode = not self.kmers_empty(ks)
        res = []
        for name, seq, qual in reads:
            o = _oracle.score_read(seq, qual if qual is not None else b"\0" * len(seq), p, ks if kmer_mode else None)
            res.append(o)
        return res

    def reads2(self, lengths, sc):
        return _oracle.reads2_gather(lengths, sc)

    def rank(self, mean_q, window_q, length, passed, **kw):
        r = _oracle.rank_and_cut(mean_q, window_q, length, passed, **kw)
        return r["passed"], r["target_bases"], r["kept_bases"], r["outcome"]


class HipBackend:
    name = "hip"

    def __init__(self, ctx=None):
        from filtlong_amd import api
        self.api = api
        self.ctx = ctx or api.Context(0)

    def kmers(self, assembly=None, short_files=None):
        if assembly is None and not short_files:
            return None
        ks = self.api.Kmers(self.ctx)
        if assembly is not None:
            ks.add_assembly_fasta(assembly)
        if short_files:
            ks.add_read_fastqs(short_files)
        ks.finalize()
        return ks

    def kmers_empty(self, ks):
        return ks is None or ks.empty()

    def score(self, reads, pkw, ks):
        api = self.api
        p = api.make_params(**pkw)
        kmer_mode = not self.kmers_empty(ks)
        strings = [(seq if kmer_mode else qual) for _, seq, qual in reads]
        plane, offsets, lengths = api.pack_reads(strings)
        order = api.length_order(lengths)
        o = self.ctx.score_reads(plane, offsets, lengths, p, kmers=ks if kmer_mode else None, order=order)
        res = []
        co = o["child_offsets"]
        for i in range(len(reads)):
            a, b = int(co[i]), int(co[i + 1])
            res.append({
                "length": int(lengths[i]), "mean_q": float(o["mean_q"][i]), "window_q": float(o["window_q"][i]),
                "passed": int(o["passed"][i]), "first": int(o["first"][i]), "last": int(o["last"][i]),
                "child_ranges": [tuple(int(x) for x in r) for r in o["child_ranges"][a:b]],
                "children": [{"length": int(o["child_ranges"][j][1] - o["child_ranges"][j][0]),
                              "mean_q": float(o["child_mean_q"][j]), "window_q": float(o["child_window_q"][j]),
                              "passed": int(o["child_passed"][j])} for j in range(a, b)],
            })
        return res

    def reads2(self, lengths, sc):
        return self.ctx.reads2_gather(lengths, sc)

    def rank(self, mean_q, window_q, length, passed, lw=1.0, mw=1.0, ww=1.0, **kw):
        r = self.ctx.rank_and_cut(mean_q, window_q, length, passed, length_weight=lw, mean_q_weight=mw,
                                  window_q_weight=ww, **kw)
        rep = r["report"]
        return r["passed"], rep.target_bases, rep.kept_bases, rep.outcome


def run_filter(backend, reads, ks, pkw=None, target_bases=None, keep_percent=None, lw=1.0, mw=1.0, ww=1.0):
    """Returns (ordered output names, after_count, after_bases, target, kept, outcome) like the reference CLI."""
    pkw = pkw or {}
    scored = backend.score(reads, pkw, ks)
    # reads2 gather, main.cpp:138-147, through the backend's seam (oracle: flo_reads2_gather, HIP: flx_reads2_gather);
    # child names read.cpp:135-136
    lengths = np.array([len(seq) for _, seq, _ in reads], dtype=np.int32)
    kids = [c for r in scored for c in r["children"]]
    rngs = [x for r in scored for x in r["child_ranges"]]
    sc = {"mean_q": np.array([r["mean_q"] for r in scored], dtype=np.float64),
          "window_q": np.array([r["window_q"] for r in scored], dtype=np.float64),
          "passed": np.array([r["passed"] for r in scored], dtype=np.uint8),
          "child_offsets": np.concatenate([[0], np.cumsum([len(r["children"]) for r in scored])]).astype(np.uint64),
          "child_ranges": np.array(rngs, dtype=np.int32).reshape(-1, 2),
          "child_mean_q": np.array([c["mean_q"] for c in kids], dtype=np.float64),
          "child_window_q": np.array([c["window_q"] for c in kids], dtype=np.float64),
          "child_passed": np.array([c["passed"] for c in kids], dtype=np.uint8)}
    r2 = backend.reads2(lengths, sc)
    names = []
    for par, ch in zip(r2["parent"], r2["child"]):
        if ch < 0:
            names.append(reads[par][0])
        else:
            names.append("%s_%d-%d" % (reads[par][0], rngs[ch][0] + 1, rngs[ch][1]))
    mean_q, window_q, length, passed = r2["mean_q"], r2["window_q"], r2["length"], r2["passed"]
    total_bases = sum(len(seq) for _, seq, _ in reads)  # original reads, main.cpp:89
    out_passed, target, kept, outcome = backend.rank(
        np.array(mean_q), np.array(window_q), np.array(length, dtype=np.int32), np.array(passed, dtype=np.uint8),
        lw=lw, mw=mw, ww=ww, target_bases=target_bases, keep_percent=keep_percent, total_bases=total_bases)
    out_names = [n for n, p in zip(names, out_passed) if p]
    return out_names, len(names), int(np.asarray(length, dtype=np.int64).sum()), int(target), int(kept), int(outcome)


def golden_args_to_kwargs(args):
    """Parse the filtlong argv stored in tests/golden/e2e.json into run_filter keyword arguments."""
    pkw, kw, ref = {}, {}, {}
    i = 0
    while i < len(args):
        a = args[i]
        v = args[i + 1] if i + 1 < len(args) else None
        if a == "--target_bases": kw["target_bases"] = int(v); i += 2
        elif a == "--keep_percent": kw["keep_percent"] = float(v); i += 2
        elif a == "--min_length": pkw["min_length"] = int(v); i += 2
        elif a == "--max_length": pkw["max_length"] = int(v); i += 2
        elif a == "--min_mean_q": pkw["min_mean_q"] = float(v); i += 2
        elif a == "--min_window_q": pkw["min_window_q"] = float(v); i += 2
        elif a == "--window_size": pkw["window_size"] = int(v); i += 2
        elif a == "--split": pkw["split"] = int(v); i += 2
        elif a == "--trim": pkw["trim"] = True; i += 1
        elif a == "--length_weight": kw["lw"] = float(v); i += 2
        elif a == "--mean_q_weight": kw["mw"] = float(v); i += 2
        elif a == "--window_q_weight": kw["ww"] = float(v); i += 2
        elif a == "-a": ref["a"] = v; i += 2
        elif a == "-1": ref["1"] = v; i += 2
        elif a == "-2": ref["2"] = v; i += 2
        else: raise ValueError(a)
    return pkw, kw, ref
