#!/usr/bin/env python3
"""Turns the text summaries one call of tools/r03_final.sh leaves under gpurun_out/ into the committed evidence under profiles/:
copies them under their round-3 names and writes profiles/r03_kmer_requests.json and profiles/r03_traffic_c2.json, each stamped with
the hash of the kernel source it was measured on (bench.py quotes them only while that hash still matches).
usage: python tools/make_profile_json.py [round prefix = r03]"""
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PFX = sys.argv[1] if len(sys.argv) > 1 else "r05"
KMER_READS = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000  # reads of the k-mer PMC passes
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def sha16(*names):
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(ROOT, "filtlong_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


KMER_SOURCES = ("cover_queue.hip", "cover_common.h", "score_kmer.hip", "kmerset.hip", "kmerset.h", "pathtext.hip")  # as in bench.py


def counters(path, kernel):
    """{counter: sum} and avg ms of `kernel` in a tools/rocprof_summary.py text"""
    out, ms = {}, None
    for line in open(path):
        t = line.split()
        if kernel not in line or len(t) < 3:
            continue
        if t[-1] == "1" and not t[-2].replace(".", "").isdigit():
            continue
        try:
            if len(t) >= 6 and all(x.replace(".", "").isdigit() for x in t[-5:]):
                ms = float(t[-3])  # calls total avg min max
            else:
                out[t[-3]] = float(t[-2])
        except ValueError:
            pass
    return out, ms


def main():
    # ---- k-mer cover kernel: requests by class
    sys.path.insert(0, ROOT)
    from filtlong_amd import synth
    bases = int(synth.lengths(KMER_READS).astype("int64").sum())  # the synthetic set
    req = {}
    for cfg in ("c3", "c4", "c3_indels", "c3_unrelated"):
        if not os.path.exists(os.path.join(G, "prof_kmer", "t_%s.txt" % cfg)):
            continue
        # the cover stage of round 6 is two launches (cover_queue.hip): every read, then the reads handed over to the kernel with a
        # diagonal per lane — their counters and times are summed
        t, f, ms = {}, {}, 0.0
        for kern in ("k_kmer_cover_q<true, false>", "k_kmer_cover_q<true, true>"):
            t1, _ = counters(os.path.join(G, "prof_kmer", "t_%s.txt" % cfg), kern)
            f1, _ = counters(os.path.join(G, "prof_kmer", "f_%s.txt" % cfg), kern)
            _, ms1 = counters(os.path.join(G, "prof_kmer", "k_%s.txt" % cfg), kern)
            for k, v in t1.items():
                t[k] = t.get(k, 0.0) + v
            for k, v in f1.items():
                f[k] = f.get(k, 0.0) + v
            ms += ms1 or 0.0
        req[cfg] = {
            "measured_at_reads": KMER_READS, "bases": bases, "kernel": "k_kmer_cover_q", "kernel_source_sha16": sha16(*KMER_SOURCES),
            "kernel_ms": ms, "far_requests": t["TCC_MISS_sum"], "far_requests_per_base": t["TCC_MISS_sum"] / bases,
            "l2_hit_requests": t["TCC_HIT_sum"], "l2_hit_requests_per_base": t["TCC_HIT_sum"] / bases,
            "fetch_size_kib": f["FETCH_SIZE"], "traffic_bytes": f["FETCH_SIZE"] * 1024, "traffic_bytes_per_base": f["FETCH_SIZE"] * 1024 / bases,
            "source": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum / FETCH_SIZE (separate passes) of `bench.py --config %s --reads %d --steps 1 "
                      "--warmup 0` (tools/prof_kmer.sh), both launches of the cover stage summed; FETCH_SIZE = 64 B x fabric read requests, NOT doubled: these are single 64-byte requests "
                      "(the x2 of the guide applies to 128-byte streaming requests)" % (cfg, KMER_READS)}
        for d in "kftsu":
            if not os.path.exists(os.path.join(G, "prof_kmer", "%s_%s.txt" % (d, cfg))):
                continue
            shutil.copy(os.path.join(G, "prof_kmer", "%s_%s.txt" % (d, cfg)), os.path.join(P, "%s_kmer_%s_%s.txt" % (PFX, d, cfg)))
    json.dump(req, open(os.path.join(P, PFX + "_kmer_requests.json"), "w"), indent=1)
    if len(sys.argv) > 3 and sys.argv[3] == "kmer-only":  # (on the GPU box, between the PMC passes and the bench line that quotes them)
        print(json.dumps({k: {"kernel_ms": v["kernel_ms"], "far/base": round(v["far_requests_per_base"], 4), "l2/base": round(v["l2_hit_requests_per_base"], 4)}
                          for k, v in req.items()}))
        return
    # ---- C2 Phred kernel: HBM traffic
    fe, _ = counters(os.path.join(G, "final", "pmc_fetch.txt"), "flx_score_phred_regs")
    wr, _ = counters(os.path.join(G, "final", "pmc_write.txt"), "flx_score_phred_regs")
    prev = os.path.join(P, PFX + "_traffic_c2.json")
    old = json.load(open(prev if os.path.exists(prev) else os.path.join(P, "r05_traffic_c2.json")))
    traffic = 2 * fe["FETCH_SIZE"] * 1024 + wr["WRITE_SIZE"] * 1024
    old.update({"kernel_source_sha16": sha16("score_phred_regs.hip"), "FETCH_SIZE_KiB": fe["FETCH_SIZE"], "WRITE_SIZE_KiB": wr["WRITE_SIZE"],
                "traffic_bytes": traffic, "ratio": traffic / old["algorithmic_bytes"]})
    json.dump(old, open(os.path.join(P, PFX + "_traffic_c2.json"), "w"), indent=1)
    for src, dst in (("final/c2_kernel_stats.txt", "_c2_kernel_stats_rocprofv3.txt"), ("final/pmc_lds.txt", "_pmc_lds_phred_c2.txt"),
                     ("final/pmc_fetch.txt", "_pmc_fetch_phred_c2.txt"), ("final/pmc_write.txt", "_pmc_write_phred_c2.txt"),
                     ("final/bench_default.json", "_bench_default.json"), ("final/bench_under_rocprof.json", "_bench_under_rocprof.json")):
        shutil.copy(os.path.join(G, src), os.path.join(P, PFX + dst))
    print(json.dumps({k: {"kernel_ms": v["kernel_ms"], "far/base": round(v["far_requests_per_base"], 4), "l2/base": round(v["l2_hit_requests_per_base"], 4)}
                      for k, v in req.items()}), "traffic ratio", round(old["ratio"], 4))


if __name__ == "__main__":
    main()
