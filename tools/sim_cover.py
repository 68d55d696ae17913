#!/usr/bin/env python3
"""Host-side model of the wave-level cover kernel's lookups (csrc/score_kmer.hip, k_kmer_cover_w) on the synthetic C3 reads:
how many far requests (exact-membership lookups) and rounds a lane needs when ONE request answers G consecutive positions
(G = 1: a bit per 16-mer; 2: the exact15 pair table; 4 / 5: wider groups), with groups at fixed alignment or floating (the
request is placed so that its range ends at the asked position); "+ locus": the members along the read's own locus in the assembly
are known beforehand (not built) and only the candidates outside what they cover are asked.  Pure numpy + a python loop over lanes: a design tool, not
part of the product or the tests.   usage: sim_cover.py [n_reads] [ref_len]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from filtlong_amd import synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ref_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
CODE = np.zeros(256, dtype=np.uint8)
for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    CODE[ch] = v


def kmers(codes, k):
    """rolling k-mers (uint64) ending at positions k-1 .. n-1"""
    n = len(codes)
    out = np.zeros(n - k + 1, dtype=np.uint64)
    for j in range(k):
        out = (out << np.uint64(2)) | codes[j:n - k + 1 + j].astype(np.uint64)
    return out


ref = synth.bases_read(synth.STREAM_REF, 0, 0, ref_len)
rc = CODE[ref]
rr = (3 - rc)[::-1]
members = np.unique(np.concatenate([kmers(rc, 16), kmers(rr, 16)]))
p12 = np.zeros(1 << 24, dtype=bool)
p12[np.concatenate([kmers(rc, 12), kmers(rr, 12)]).astype(np.int64)] = True
print("set %d 16-mers, 12-mer table %.1f %% full" % (len(members), 100 * p12.mean()))


def simulate(cand, memb, G, floating, lcand_bet=True, per_side=1, locus=False):
    """cand / memb: bool arrays over the positions of one read (position j = 16-mer ending at j).  Returns (requests, sum of
    per-wave rounds, lanes)."""
    n = len(cand)
    n_lanes = (n + 15) // 16
    req = 0
    rounds_lane = np.zeros(n_lanes, dtype=np.int32)
    prev_hit15 = False
    prev_cand15 = False
    for ln in range(n_lanes):
        lo = ln * 16
        c = cand[lo:lo + 16]
        m = memb[lo:lo + 16]
        w = len(c)
        probed = ~c
        hits = np.zeros(w, dtype=bool)
        if locus:  # members confirmed by comparing the read with the assembly along its locus: known before any request
            hits = c & m
            probed = probed | hits
        lcand, lhit = prev_cand15, prev_hit15

        def ask(pos, top):
            # the request answering `pos`: positions [a, a + G)
            if floating:
                a = pos - G + 1 if top else pos
            else:
                a = (pos // G) * G
            nonlocal req
            req += 1
            for q in range(max(a, 0), min(a + G, w)):
                probed[q] = True
                hits[q] = c[q] and m[q]

        r = 0
        first = True
        while True:
            H = np.nonzero(hits)[0]
            opn = np.nonzero(c & ~probed)[0]
            have_left = lcand and lhit and (locus or not first)
            if len(H) or have_left:
                hi = H[-1] if len(H) else -1
                lo_h = -1 if have_left else H[0]
                above = opn[opn > hi]
                below = opn[opn < lo_h]
            else:
                above = below = opn
                if first and lcand and lcand_bet:
                    below = opn[:0]  # bet on the left neighbour's last position being a member
            asks = []
            if len(above):
                asks += [(int(p), True) for p in above[::-1][:per_side]]
            if len(below):
                asks += [(int(p), False) for p in below[:per_side] if (int(p), True) not in asks]
            first = False
            if not asks:
                break
            for p, t in asks:
                if not probed[p]:
                    ask(p, t)
            r += 1
        rounds_lane[ln] = r
        prev_cand15 = bool(c[15]) if w == 16 else False
        prev_hit15 = bool(hits[15]) if w == 16 else False
    wave_rounds = sum(int(rounds_lane[i:i + 64].max()) for i in range(0, n_lanes, 64))
    return req, wave_rounds, n_lanes


def simulate_known(cand, memb, known, G, floating):
    """like simulate(..., locus=True) with only the members in `known` given beforehand: the rest of the members are found by requests"""
    # members that are not known must still be asked for: model by running the plain policy on the read with the known members'
    # positions answered for free — i.e. requests are only counted for lanes where some candidate is not a known member
    req, wr, nl = 0, 0, 0
    n = len(cand)
    free = np.zeros(n, dtype=bool)
    for lo in range(0, n, 1024):
        hi = min(lo + 1024, n)
        seg = slice(lo, hi)
        if known[seg].any():
            r, w, l = simulate(cand[seg], memb[seg], G, floating, locus=True)
        else:
            r, w, l = simulate(cand[seg], memb[seg], G, floating, locus=False)
        req += r
        wr += w
        nl += l
    return req, wr, nl


def locus_mask(memb, K, span=1024):
    """members known beforehand when every span of `span` positions looks for its own seed: the 16-mers ending at span_start + 15 +
    16 j, j < K, are tried in turn (one far request each) and the first member among them places the span on its diagonal; a span
    without a seed gets nothing.  (The synthetic reads are substitution-only, so a member IS on the read's diagonal.)  Returns the
    mask and the number of seed requests."""
    known = np.zeros(len(memb), dtype=bool)
    tries = 0
    for lo in range(0, len(memb), span):
        hi = min(lo + span, len(memb))
        for j in range(K):
            p = lo + 15 + 16 * j
            if p >= hi:
                break
            tries += 1
            if memb[p]:
                known[lo:hi] = memb[lo:hi]
                break
    return known, tries


tot = {}
seed_tries = {}
positions = 0
lens = synth.lengths(n_reads)
for i in range(n_reads):
    L = int(lens[i])
    seq = CODE[synth.seq_read(i, L, ref)]
    k16 = kmers(seq, 16)
    memb = np.zeros(L, dtype=bool)
    idx = np.searchsorted(members, k16)
    idx[idx == len(members)] = 0
    memb[15:] = members[idx] == k16
    pres = np.zeros(L, dtype=bool)
    pres[11:] = p12[kmers(seq, 12).astype(np.int64)]
    cand = np.zeros(L, dtype=bool)
    cand[15:] = pres[15:] & pres[14:-1] & pres[13:-2] & pres[12:-3] & pres[11:-4]
    assert not (memb & ~cand).any()
    positions += L
    for name, G, fl, ps in (("G1", 1, False, 1), ("G2 fixed (shipped)", 2, False, 1), ("G2 floating", 2, True, 1),
                            ("G4 fixed", 4, False, 1), ("G4 floating", 4, True, 1), ("G5 floating", 5, True, 1),
                            ("G8 floating", 8, True, 1), ("G2 fixed, 2 per side", 2, False, 2),
                            ("G2 floating + locus", 2, True, 1), ("G2 floating + locus, seeds per span K=4", 2, True, 1),
                            ("G2 floating + locus, seeds per span K=8", 2, True, 1)):
        if "K=" in name:
            known, tries = locus_mask(memb, int(name.split("K=")[1]))
            seed_tries[name] = seed_tries.get(name, 0) + tries
            # what the kernel sees: candidates as before, but the known members are hits before any request
            r, wr, nl = simulate(cand, memb, G, fl, per_side=ps, locus=True) if known.all() else simulate_known(cand, memb, known, G, fl)
        else:
            r, wr, nl = simulate(cand, memb, G, fl, per_side=ps, locus=name.endswith("locus"))
        t = tot.setdefault(name, [0, 0, 0])
        t[0] += r
        t[1] += wr
        t[2] += nl
    if i == 0:
        print("read 0: L %d, members %.3f, candidates %.3f" % (L, memb.mean(), cand.mean()))
print("%d reads, %d positions" % (n_reads, positions))
for name, (r, wr, nl) in tot.items():
    extra = seed_tries.get(name, 0)
    print("%-42s far requests / position %.4f%s   rounds per wave-span %.2f" % (
        name, (r + extra) / positions, " (%.4f of them seeds)" % (extra / positions) if extra else "", wr / max(1, (nl + 63) // 64)))
