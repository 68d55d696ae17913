#!/usr/bin/env python3
"""bench.py — throughput of the Filtlong scoring hot path on MI355X.

One "step" = one pass of the hot path over the whole synthetic batch resident in HBM:
    flx_score_batch_dev        per-read scoring (Phred: mean / sliding-window quality; k-mer: coverage, first/last,
                               trim/split children), hard cut-offs
  + the global stage           exact statistics, normalise, final score, --target_bases cut
                               (N = 1: flx_rank_and_cut_dev;  N > 1: flx_rank_and_cut_comm_dev = ONE RCCL all-gather of
                               the mean qualities over xGMI + the selection's histograms all-reduced on the device)

Workloads (BASELINE.json `configs`), `--config`:
    c2      10 M reads per GPU x gamma(k=4) mean 10 kbp (1e11 bases), Phred-only, --target_bases 50 %   (default: the
            configuration BASELINE.json's metric is quoted on; weak scaling, C5 at 8 GPUs)
    c2wide  the same with the wide quality profile (per-read centre Q3..Q44, q <= 50 instead of Q8..Q25 +-8)
    c3      10 M reads drawn from a 5 Mbp reference, -a assembly (k-mer mode), --target_bases 50 %
    c4      the same reads, -1/-2 short-read reference (1e6 error-free 100 bp pairs), --trim --split 500
The default run times c2 and then — outside the timed region, N = 1 only, `--no-extras` skips it — measures c2wide, c3
and c4 (one or two steps each) and both implementations of the cut (radix select / radix sort), reported under
`extras` on the same JSON line.

Prints ONE JSON line (rank 0).  `value` = total bases over all ranks / max-over-ranks wall time of the timed steps.
`roofline` is for the dominant kernel, timed with HIP events on the stream it runs on; `cpu_baseline` is the reference's
own compiled code (oracle/_ref/ref_bench, kind "reference") on a bounded sample on this box's host cores (1 thread: the
reference is single-threaded).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
RANDOM_REQ_PEAK_G = 55.0     # tools/tabench: random requests beyond the L2 (Infinity Cache or HBM alike), G/s, 64 B each
L2_LOOKUP_PEAK_G = 261.0     # tools/tabench: cache LINES per second from an L2-resident table (<= 4 MiB), however many lanes share a line
REF_LEN = 5_000_000


def source_sha16(*names):
    """sha256[:16] of kernel source files: figures quoted from profiles/ carry the hash of the sources they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(ROOT, "filtlong_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


KMER_SOURCES = ("cover_queue.hip", "cover_common.h", "score_kmer.hip", "kmerset.hip", "kmerset.h", "pathtext.hip")  # what the cover kernel's requests depend on (kernel + the set's tables and text)


def _run_ref_bench(argv):
    ref_bench = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    env = dict(os.environ, LANG="C", LC_ALL="C")
    out = subprocess.run([ref_bench] + [str(a) for a in argv], env=env, check=True, stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL).stdout.decode()
    return json.loads(out.strip().splitlines()[-1])


def _with_upper_bound(cb):
    """beside the honest 1-core number of the single-threaded reference, the optimistic figure "every host core scores
    its own shard" — NOT a valid Filtlong run (the normalisation and the cut are global, src/main.cpp:170-261), an upper bound only."""
    if cb and cb.get("host_cores_available"):
        cb["upper_bound_all_cores"] = {"value": round(cb["value"] * cb["host_cores_available"], 1), "unit": cb["unit"],
                                       "cores": cb["host_cores_available"],
                                       "note": "value x host cores, as if independent shards scaled perfectly: not a valid Filtlong run "
                                               "(global normalisation and cut), an upper bound on what the host could do"}
    return cb


def cpu_baseline_phred(sample_reads):
    """Time the CPU reference on a bounded sample of the same workload (rank 0, N == 1 only)."""
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_bench")):
        # ~half of the sample's bases as target, like --target_bases 50g on 1e11 bases
        r = _run_ref_bench([sample_reads, 0, sample_reads * 5000])
        return _with_upper_bound({"value": round(r["mbases_per_s"], 3), "unit": "Mbases/s", "cores": 1, "kind": "reference",
                "sample": "%d reads / %d bases of the same synthetic Phred-only workload, reference objects "
                          "(Read::Read + set_final_score + std::sort) in memory, %.1f s score + %.2f s rank"
                          % (r["reads"], r["bases"], r["score_s"], r["rank_s"]),
                "host_cores_available": os.cpu_count()})
    # fall back to the oracle restatement ("port")
    lib = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.flo_bench_phred.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int64, C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    ss, rs, tb, kb = C.c_double(), C.c_double(), C.c_int64(), C.c_int64()
    lib.flo_bench_phred(sample_reads, 20250919, 0, sample_reads * 5000, ss, rs, tb, kb)
    return _with_upper_bound({"value": round(tb.value / (ss.value + rs.value) / 1e6, 3), "unit": "Mbases/s", "cores": 1,
            "kind": "port", "sample": "%d reads / %d bases, oracle restatement in memory" % (sample_reads, tb.value),
            "host_cores_available": os.cpu_count()})


def cpu_baseline_kmer(cfg, sample_reads, full_set):
    """K-mer mode on the CPU reference: sample reads + the 5 Mbp reference written to a temp dir, scored by the reference's
    own objects in memory (oracle/_ref/ref_bench kmer).  C4's short-read set takes the reference ~80 s to hash
    (profiles/r02_cpu_kmer.txt); unless `full_set`, the C4 sample is scored against the set built from the assembly,
    which holds the same 16-mers (9,988,379 vs 9,988,279) — set-build time is reported separately either way."""
    from filtlong_amd import synth
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_bench")):
        return None
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, REF_LEN)
    lens = synth.lengths(sample_reads)
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "ref.fasta"), "wb") as f:
            f.write(b">ref\n" + ref.tobytes() + b"\n")
        with open(os.path.join(d, "reads.fastq"), "wb") as f:
            for i, L in enumerate(lens):
                f.write(b"@r%d\n" % i + synth.seq_read(i, int(L), ref).tobytes() + b"\n+\n" + b"I" * int(L) + b"\n")
        opts = ["-a", os.path.join(d, "ref.fasta")]
        note = "set from the assembly"
        if cfg == "c4":
            if full_set:
                r1, r2 = short_read_pairs(ref)
                for name, arr in (("r1.fq", r1), ("r2.fq", r2)):
                    with open(os.path.join(d, name), "wb") as f:
                        q = b"I" * 100
                        for i in range(arr.shape[0]):
                            f.write(b"@p%d\n" % i + arr[i].tobytes() + b"\n+\n" + q + b"\n")
                opts = ["-1", os.path.join(d, "r1.fq"), "-2", os.path.join(d, "r2.fq")]
                note = "set from the 1e6 short-read pairs"
            else:
                note = "set from the assembly (same 16-mers as the short-read set, which takes the reference ~80 s to hash)"
            opts += ["--trim", "--split", "500"]
        r = _run_ref_bench(["kmer", os.path.join(d, "reads.fastq"), int(lens.astype(np.int64).sum()) // 2] + opts)
    return _with_upper_bound({"value": round(r["mbases_per_s"], 3), "unit": "Mbases/s", "cores": 1, "kind": "reference",
            "sample": "%d reads / %d bases of the same synthetic k-mer workload, reference objects in memory, %s: "
                      "%.1f s score + %.3f s rank (set build %.1f s, not counted)"
                      % (r["reads"], r["bases"], note, r["score_s"], r["rank_s"], r["set_build_s"]),
            "host_cores_available": os.cpu_count()})


def short_read_pairs(ref):
    """C4 reference: 1e6 error-free pairs of 100 bp per 5 Mbp (40x): -1 forward substring, -2 reverse complement of the
    substring 350 bp downstream."""
    from filtlong_amd import synth
    npairs = len(ref) // 5
    starts = (synth.mix(synth.SEED, synth.STREAM_START, np.arange(npairs, dtype=np.uint64) + np.uint64(1 << 40), 0)
              % np.uint64(len(ref) - 450)).astype(np.int64)
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    idx = starts[:, None] + np.arange(100)[None, :]
    return ref[idx], comp[ref[idx + 350]][:, ::-1]


def end_to_end_cli(n_reads):
    """File -> stdout through filtlong_amd/bin/filtlong vs oracle/_ref/filtlong (the reference binary) on the same FASTQ
    (tools/gen_fastq: the synthetic Phred workload, ~20 kB per read), --target_bases = half the bases; timed here."""
    import hashlib
    gen = os.path.join(ROOT, "tools", "gen_fastq")
    if not os.path.exists(gen):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "oracle"), "-o", gen,
                               os.path.join(ROOT, "tools", "gen_fastq.cpp")])
    ours = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
    ref = os.path.join(ROOT, "oracle", "_ref", "filtlong")
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        fq = os.path.join(d, "reads.fastq")
        bases = int(subprocess.run([gen, str(n_reads), fq], check=True, stdout=subprocess.PIPE).stdout.decode().split()[0])
        size = os.path.getsize(fq)
        target = str(bases // 2)

        def run(binary, outp):
            if os.path.exists(outp):
                os.unlink(outp)  # truncating the previous run's output (1 GB of page cache) must not count as this run's time
            t = time.perf_counter()
            with open(outp, "wb") as fo:
                rc = subprocess.run([binary, "--target_bases", target, fq], stdout=fo, stderr=subprocess.DEVNULL, env=env).returncode
            return time.perf_counter() - t, rc

        def sha(path):
            h = hashlib.sha256()
            with open(path, "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            return h.hexdigest()

        run(ours, os.path.join(d, "a.out"))  # first run: device runtime start-up and page cache
        t_ours, rc_ours = min(run(ours, os.path.join(d, "a.out")) for _ in range(2))
        out = {"measured_in_this_run": True, "fastq_bytes": size, "reads": n_reads, "bases": bases, "target_bases": int(target),
               "seconds": round(t_ours, 3), "gbases_per_s": round(bases / t_ours / 1e9, 3), "exit_code": rc_ours,
               "stdout_bytes": os.path.getsize(os.path.join(d, "a.out")), "stdout_sha256": sha(os.path.join(d, "a.out"))[:16],
               "pcie": "the streamed ingest moves 1 byte per base host -> device in pinned chunks (two slots)"}
        if os.path.exists(ref):
            t_ref, rc_ref = run(ref, os.path.join(d, "r.out"))
            out.update({"reference_seconds": round(t_ref, 2), "reference_exit_code": rc_ref, "speedup": round(t_ref / t_ours, 1),
                        "stdout_identical_to_reference": sha(os.path.join(d, "r.out"))[:16] == out["stdout_sha256"]})
        return out


def end_to_end_cli_kmer(n_reads):
    """The same for k-mer mode: `-a ref.fasta --trim --split 500 --target_bases <half>` on a FASTQ of the synthetic k-mer reads (reads
    drawn from the 5 Mbp reference with substitutions and junk, filtlong_amd/synth.py), the drop-in binary and the reference binary on
    the same files, stdout compared by digest.  A small sample: the reference scores 13 Mbases/s and hashes the assembly for seconds."""
    import hashlib
    from filtlong_amd import synth
    ours = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "filtlong")
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, REF_LEN)
    lens = synth.lengths(n_reads)
    bases = int(lens.astype(np.int64).sum())
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        fa, fq = os.path.join(d, "ref.fasta"), os.path.join(d, "reads.fastq")
        with open(fa, "wb") as f:
            f.write(b">ref\n" + ref.tobytes() + b"\n")
        with open(fq, "wb") as f:
            for i, L in enumerate(lens):
                f.write(b"@r%d\n" % i + synth.seq_read(i, int(L), ref).tobytes() + b"\n+\n" + b"I" * int(L) + b"\n")
        argv = ["-a", fa, "--trim", "--split", "500", "--target_bases", str(bases // 2), fq]

        def run(binary, outp):
            t = time.perf_counter()
            with open(outp, "wb") as fo:
                rc = subprocess.run([binary] + argv, stdout=fo, stderr=subprocess.DEVNULL, env=env).returncode
            return time.perf_counter() - t, rc

        def sha(path):
            return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]

        run(ours, os.path.join(d, "a.out"))
        t_ours, rc_ours = min(run(ours, os.path.join(d, "a.out")) for _ in range(2))
        out = {"measured_in_this_run": True, "flags": "-a ref.fasta (5 Mbp) --trim --split 500 --target_bases <half>", "reads": n_reads, "bases": bases,
               "seconds": round(t_ours, 3), "exit_code": rc_ours, "stdout_bytes": os.path.getsize(os.path.join(d, "a.out")),
               "stdout_sha256": sha(os.path.join(d, "a.out")),
               "note": "the whole command: reading and hashing the assembly (the set, its text and seed table), scoring, trimming / splitting, output"}
        if os.path.exists(ref_bin):
            t_ref, rc_ref = run(ref_bin, os.path.join(d, "r.out"))
            out.update({"reference_seconds": round(t_ref, 2), "reference_exit_code": rc_ref, "speedup": round(t_ref / t_ours, 1),
                        "stdout_identical_to_reference": sha(os.path.join(d, "r.out")) == out["stdout_sha256"]})
        return out


def _timed_child(argv, stdout_path, env):
    """one command: wall clock, exit code and the command's own peak resident set.  Through tools/timed.py, a fresh small interpreter
    that does the wait4: a child forked from THIS process would carry the pages of its 3 GB (torch) in its ru_maxrss until it execs."""
    rep = stdout_path + ".time"
    with open(stdout_path, "wb") as fo:
        rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timed.py"), rep] + list(argv), stdout=fo, stderr=subprocess.DEVNULL,
                            env=env).returncode
    t = open(rep).read().split()  # "<seconds> s wall, <KiB> KiB peak RSS"
    return float(t[0]), rc, int(t[3]) // 1024


def end_to_end_cli_kmer_short(n_reads, n_pairs):
    """C4 through the command line: `-1 sr_1.fastq -2 sr_2.fastq --trim --split 500 --target_bases <half>` — the short-read
    reference (error-free 100 bp pairs from the 5 Mbp genome, tools/gen_kmer_inputs = the C4 definition of SURVEY §8d) is STREAMED
    into the device set in batches (cli/reference.h, round 5: host memory O(batch)), the set's text is built from the 24-mers of the
    pairs, long reads drawn from the genome are scored, trimmed and split.  The drop-in binary and the reference binary on the same
    files, stdout compared by digest; peak resident memory of both processes.  (The default run bounds the pairs so that the
    reference's hashing — ~80 s per 10^6 pairs — stays within the bench's minutes; profiles/r05_e2e_kmer.json holds the run with
    the full 10^6 pairs and one with a 10 GB short-read set.)"""
    import hashlib
    gen = os.path.join(ROOT, "tools", "gen_kmer_inputs")
    ours = os.path.join(ROOT, "filtlong_amd", "bin", "filtlong")
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "filtlong")
    env = dict(os.environ, LANG="C", LC_ALL="C")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        long_bases, short_bases = (int(x) for x in subprocess.run([gen, d, str(n_reads), str(n_pairs)], check=True, stdout=subprocess.PIPE).stdout.split())
        argv = ["-1", os.path.join(d, "sr_1.fastq"), "-2", os.path.join(d, "sr_2.fastq"), "--trim", "--split", "500",
                "--target_bases", str(long_bases // 2), os.path.join(d, "reads.fastq")]

        def sha(path):
            return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]

        _timed_child([ours] + argv, os.path.join(d, "a.out"), env)
        t_ours, rc_ours, rss_ours = min(_timed_child([ours] + argv, os.path.join(d, "a.out"), env) for _ in range(2))
        out = {"measured_in_this_run": True, "flags": "-1 sr_1.fastq -2 sr_2.fastq --trim --split 500 --target_bases <half>",
               "short_read_pairs": n_pairs, "short_read_bases": short_bases, "reads": n_reads, "bases": long_bases,
               "seconds": round(t_ours, 3), "exit_code": rc_ours, "peak_rss_mib": rss_ours,
               "stdout_bytes": os.path.getsize(os.path.join(d, "a.out")), "stdout_sha256": sha(os.path.join(d, "a.out")),
               "note": "the whole command: both short-read files parsed and streamed into the device set, multi-copy rule + Bloom replay, "
                       "the set's text from the pairs' 24-mers, scoring, trimming / splitting, output"}
        if os.path.exists(ref_bin):
            t_ref, rc_ref, rss_ref = _timed_child([ref_bin] + argv, os.path.join(d, "r.out"), env)
            out.update({"reference_seconds": round(t_ref, 2), "reference_exit_code": rc_ref, "reference_peak_rss_mib": rss_ref,
                        "speedup": round(t_ref / t_ours, 1),
                        "stdout_identical_to_reference": sha(os.path.join(d, "r.out")) == out["stdout_sha256"]})
        return out


class Batch:
    """The packed batch of one rank in HBM (layout, processing order, id range)."""

    def __init__(self, ctx, torch, dev, n, first, fixed_len):
        from filtlong_amd import api, synth
        self.n = n
        self.lengths = synth.lengths(n, first=first, fixed=fixed_len or None)
        offsets = np.zeros(n, dtype=np.uint64)
        pb = C.c_uint64()
        ctx.L.flx_plane_layout(self.lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
        self.plane_bytes = pb.value
        order = api.length_order(self.lengths)
        self.bases = int(self.lengths.astype(np.int64).sum())
        self.d_plane = torch.empty(self.plane_bytes, dtype=torch.uint8, device=dev)
        self.d_off = torch.from_numpy(offsets.view(np.int64)).to(dev)
        self.d_len = torch.from_numpy(self.lengths).to(dev)
        self.d_ord = torch.from_numpy(order.view(np.int32)).to(dev)
        self.d_ids = torch.arange(first, first + n, dtype=torch.int64, device=dev)


def run_kmer(ctx, torch, dev, cfg, n, steps, warmup, target_frac, fixed_len=0, profile=0):
    """C3 / C4 on one GPU: k-mer scoring + reads2 gather + global stage; returns the result dictionary.  profile: the reads (filtlong_amd/synth.py:
    seq_read) — 0 = SURVEY §8(d), 1 = a third of the errors insertions and a third deletions, 2 = 30 % of the reads unrelated to the reference."""
    from filtlong_amd import api, synth, _lib
    ref = synth.bases_read(synth.STREAM_REF, 0, 0, REF_LEN)
    t0 = time.time()
    ks = api.Kmers(ctx)
    if cfg == "c4":
        r1, r2 = short_read_pairs(ref)
        ks.add_read_fastqs([[r.tobytes() for r in r1], [r.tobytes() for r in r2]])
    else:
        ks.add_assembly_fasta([ref.tobytes()])
    ks.finalize()
    build_s = time.time() - t0
    b = Batch(ctx, torch, dev, n, 0, fixed_len)
    d_ref = torch.from_numpy(ref).to(dev)
    torch.cuda.synchronize()
    ctx.synth_seq_dev(synth.SEED, b.d_plane.data_ptr(), b.plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(),
                      b.d_ids.data_ptr(), n, d_ref.data_ptr(), REF_LEN, profile=profile)
    trim_split = cfg == "c4"
    cap = 4 * n if trim_split else 16
    t = {k: torch.zeros(sz, dtype=dt, device=dev) for k, sz, dt in (
        ("mean", n, torch.float64), ("win", n, torch.float64), ("pass", n, torch.uint8), ("first", n, torch.int32),
        ("last", n, torch.int32), ("coff", n + 1, torch.int64), ("crng", 2 * cap, torch.int32), ("cmean", cap, torch.float64),
        ("cwin", cap, torch.float64), ("cpass", cap, torch.uint8))}
    params = api.make_params(trim=trim_split, split=500 if trim_split else None)
    s = _lib.Scores()
    s.mean_q, s.window_q, s.passed, s.first, s.last = (t["mean"].data_ptr(), t["win"].data_ptr(), t["pass"].data_ptr(),
                                                      t["first"].data_ptr(), t["last"].data_ptr())
    s.child_offsets, s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (
        t["coff"].data_ptr(), t["crng"].data_ptr(), t["cmean"].data_ptr(), t["cwin"].data_ptr(), t["cpass"].data_ptr())
    s.child_capacity = cap
    target = int(b.bases * target_frac)
    # reads2 arrays (src/main.cpp:138-147): every read, or its children in its place
    cap2 = n + (cap if trim_split else 0)
    r2 = {k: torch.zeros(cap2, dtype=dt, device=dev) for k, dt in (("mean", torch.float64), ("win", torch.float64),
                                                                   ("len", torch.int32), ("pass", torch.uint8))}

    def step():
        ctx.score_kmer_dev(ks, b.d_plane.data_ptr(), b.plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(),
                           b.d_ord.data_ptr(), n, params, s)
        nc = int(s.n_children)
        n2 = ctx.reads2_gather_dev(n, b.d_len.data_ptr(), s, cap2, r2["mean"].data_ptr(), r2["win"].data_ptr(),
                                   r2["len"].data_ptr(), r2["pass"].data_ptr())
        rep = ctx.rank_and_cut_dev(n2, r2["mean"].data_ptr(), r2["win"].data_ptr(), r2["len"].data_ptr(), r2["pass"].data_ptr(),
                                   target_bases=target, total_bases=b.bases)
        return rep, nc, n2

    for _ in range(warmup):
        step()
    ctx.timing_enable(True)
    ctx.timing_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rep, nc, n2 = step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    cover_kernel = {"q": "k_kmer_cover_q", "w": "k_kmer_cover_w", "v2": "k_kmer_cover"}[ctx.last_kmer_cover()]
    cover_ms, cn = ctx.timing_get("flx_score_kmer_cover")
    fold_ms, _ = ctx.timing_get("flx_score_kmer_fold")
    rank_ms, _ = ctx.timing_get("flx_rank")
    gather_ms, _ = ctx.timing_get("flx_reads2")
    csort_ms, _ = ctx.timing_get("flx_sort")  # k-mer mode: the children ordered by length for the one-lane-per-child fold
    ctx.timing_enable(False)
    cover = cover_ms / max(cn, 1)
    lookups = b.bases - 15 * n
    # requests of one cover launch by class: PMC passes of this same command (tools/prof_kmer.sh), recorded in profiles/ (since
    # round 4 at the full 1e7 reads; per-position figures scale to other batch sizes, the mix of reads is the same) — NOT measured
    # in this run, and only quoted while the kernel's source file still has the hash recorded with the pass
    req, req_src = None, None
    fpath = next((q for q in (os.path.join(ROOT, "profiles", r + "_kmer_requests.json") for r in ("r06", "r05", "r04")) if os.path.exists(q)), "")
    if fpath:
        rec = json.load(open(fpath)).get(cfg + {0: "", 1: "_indels", 2: "_unrelated"}[profile])
        if rec and rec.get("kernel") == cover_kernel and rec.get("kernel_source_sha16") == source_sha16(*KMER_SOURCES):
            req = {"far_requests": rec["far_requests_per_base"] * b.bases, "traffic_bytes": rec["traffic_bytes_per_base"] * b.bases,
                   "l2_hits": rec["l2_hit_requests_per_base"] * b.bases}
            req_src = "profiles/" + os.path.basename(fpath) + " (PMC passes of this command at %s reads%s; not this run)" % (
                "{:,}".format(rec["measured_at_reads"]), "" if rec["measured_at_reads"] == n else ", scaled per position")
    algo_bytes = b.bases + 33 * n + 25 * nc  # L + 8 + 25 per read, 8 + 17 per child
    achieved = algo_bytes / (cover * 1e-3) / 1e9
    out = {
        "workload": "%s%s: %s reads x gamma(k=4) mean 10 kbp from a 5 Mbp reference, %s, --target_bases %d" % (
            cfg.upper(), {0: "", 1: " with indels (a third of the errors insertions, a third deletions of 1-3 bases: synth profile 1)",
                          2: " with 30 % of the reads unrelated to the reference (synth profile 2)"}[profile], "{:,}".format(n),
            "-1/-2 short-read reference, --trim --split 500" if trim_split else "-a assembly", target),
        "value": round(b.bases / el / 1e6, 1), "unit": "Mbases/s", "ms_per_step": round(el * 1e3, 2),
        "bases": b.bases, "set_size": len(ks), "set_build_s_device": round(build_s, 2), "children": nc, "reads2": n2,
        "stage_ms_per_step": {"cover_kernel": round(cover, 2), "fold_kernels": round(fold_ms / steps, 2),
                              "child_sort_kernels": round(csort_ms / steps, 3), "reads2_gather_kernels": round(gather_ms / steps, 3), "rank_kernels": round(rank_ms / steps, 2)},
        "lookups_per_s_G": round(lookups / (cover * 1e-3) / 1e9, 2),
        "roofline": {
            # the contract's fraction: SURVEY §8(d) algorithmic bytes of the launch / kernel time / HBM peak
            "bound": "hbm", "kernel": cover_kernel, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": int(req["traffic_bytes"]) if req else None,
            "traffic_source": req_src,
            "members_confirmed_along_loci": bool(ctx.last_kmer_locus()),
            "avg_kernel_ms": round(cover, 3), "algorithmic_bytes": int(algo_bytes),
            # what actually bounds the kernel: random lookups, priced per cache line (tools/tabench, profiles/r03_microbench.txt):
            # 261 G lines/s from an L2-resident table, 55 G/s beyond the L2, and the two classes add up
            "request_model": None if not req else {
                "l2_lines": int(req["l2_hits"]), "far_requests": int(req["far_requests"]),
                "l2_lookup_frac": round(req["l2_hits"] / (cover * 1e-3) / 1e9 / L2_LOOKUP_PEAK_G, 4),
                "far_request_frac": round(req["far_requests"] / (cover * 1e-3) / 1e9 / RANDOM_REQ_PEAK_G, 4),
                "model_ms": round((req["l2_hits"] / L2_LOOKUP_PEAK_G + req["far_requests"] / RANDOM_REQ_PEAK_G) / 1e6, 2),
                "note": "model_ms = l2_lines / 261 G/s + far_requests / 55 G/s, the floor its two request classes set (the HBM-streaming "
                        "fraction above is small because the kernel asks, it does not stream).  Since the text, U13 and S1 cut the "
                        "requests the kernel runs ABOVE that floor, on its vector instructions (SQ_INSTS_VALU per position: "
                        "profiles/r05_kmer_s_c3.txt, a PMC pass of this command, not this run; round 4: 0.72)"}},
        "folds_on_integer_grid": bool(ctx.last_kmer_fold_grid()),
        "cut": {"target_bases": int(rep.target_bases), "kept_bases": int(rep.kept_bases), "outcome": int(rep.outcome)},
    }
    ks.close()
    return out


def verify_ranks(ctx, torch, dist, dev, rank, world, args, backend_is_nccl):
    """Self-check of the N-rank path, behind the timed steps: every rank scores `--verify-reads` reads of
    its own id range and takes part in the same global stage as the timed steps; rank 0 then scores ALL of those reads alone and
    runs the single-GPU stage — pass flags, kept bases and the target must be identical.  A few seconds; outside every timing."""
    from filtlong_amd import api, synth
    from filtlong_amd import dist as fdist
    nv = args.verify_reads
    params = api.make_params(window_size=args.window_size)

    def scored(n, first):
        b = Batch(ctx, torch, dev, n, first, args.fixed_len)
        rec = fdist.alloc_records(n, dev)
        base = rec.data_ptr()
        views = fdist.record_views(rec, n)
        views[2].copy_(b.d_len)
        torch.cuda.synchronize()
        ctx.synth_qual_dev(synth.SEED, b.d_plane.data_ptr(), b.plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(), b.d_ids.data_ptr(), n, profile=0)
        ctx.score_reads_dev(b.d_plane.data_ptr(), b.plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(), b.d_ord.data_ptr(), n, params,
                            base, base + 8 * n, base + 20 * n)
        return b, rec, base, views

    b, rec, base, (t_mean, t_win, t_len, t_pass) = scored(nv, rank * nv)
    cdev = dev if backend_is_nccl else "cpu"
    tb = torch.tensor([b.bases], dtype=torch.int64, device=cdev)
    dist.all_reduce(tb)
    total = int(tb.item())
    target = int(total * args.target_frac)
    if args.global_stage == "rccl":
        rep = ctx.rank_and_cut_comm_dev(nv, base, base + 8 * nv, base + 16 * nv, base + 20 * nv, target_bases=target, total_bases=total)
    elif args.global_stage == "sharded":
        rep = fdist.sharded_rank_and_cut(ctx, t_mean, t_win, t_len, t_pass, target_bases=target, total_bases=total)
    else:
        g_mean, g_win, g_len, g_pass, _ = fdist.gather_records(rec, nv)
        torch.cuda.synchronize()
        rep = ctx.rank_and_cut_dev(nv * world, g_mean.data_ptr(), g_win.data_ptr(), g_len.data_ptr(), g_pass.data_ptr(), target_bases=target,
                                   total_bases=total)
        t_pass.copy_(g_pass[rank * nv:(rank + 1) * nv])
    torch.cuda.synchronize()
    mine = t_pass.to(cdev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = None
    if rank == 0:
        flags_ranks = torch.cat([p.cpu() for p in parts]).numpy()
        del b, rec
        torch.cuda.empty_cache()
        b1, rec1, base1, v1 = scored(nv * world, 0)
        n1 = nv * world
        rep1 = ctx.rank_and_cut_dev(n1, base1, base1 + 8 * n1, base1 + 16 * n1, base1 + 20 * n1, target_bases=target, total_bases=total)
        torch.cuda.synchronize()
        flags_one = v1[3].cpu().numpy()
        bad = int((flags_ranks != flags_one).sum())
        ok = bad == 0 and b1.bases == total and int(rep.kept_bases) == int(rep1.kept_bases) and int(rep.target_bases) == int(rep1.target_bases)
        out = {"ok": bool(ok), "reads": n1, "ranks": world, "bases": total, "flags_differing": bad, "kept_bases_ranks": int(rep.kept_bases),
               "kept_bases_one_gpu": int(rep1.kept_bases), "passed_reads": int(flags_one.sum()),
               "what": "pass flags of %d reads scored and cut by %d ranks == the same reads on rank 0 alone" % (n1, world)}
    dist.barrier()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c2wide", "c3", "c4"), default="c2")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (default: the C2 workload)")
    ap.add_argument("--read-profile", type=int, default=0, choices=(0, 1, 2), help="--config c3 / c4: 0 = the SURVEY reads, 1 = with indels, 2 = 30 %% unrelated reads")
    ap.add_argument("--fixed-len", type=int, default=0, help="fixed read length (C1 uses 5000); 0 = gamma lengths")
    ap.add_argument("--target-frac", type=float, default=0.5, help="--target_bases as a fraction of all bases")
    ap.add_argument("--cpu-sample-reads", type=int, default=100_000)
    ap.add_argument("--cpu-sample-reads-kmer", type=int, default=2000)
    ap.add_argument("--full-cpu-baseline", action="store_true", help="c4: let the CPU reference hash the short reads itself (~80 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e-kmer-pairs", type=int, default=250_000, help="short-read pairs of extras.end_to_end_cli_kmer_short (C4 has "
                    "10^6; the reference hashes ~12 000 pairs a second, so the default run takes a quarter)")
    ap.add_argument("--no-extras", action="store_true", help="default config only: skip the c2wide / c3 / c4 / cut-path extras")
    ap.add_argument("--window-size", type=int, default=250)
    ap.add_argument("--global-stage", choices=("rccl", "sharded", "replicated"), default="rccl",
                    help="N > 1: rccl = the library's own communicator (flx_rank_and_cut_comm_dev: one all-gather of the mean "
                         "qualities, histograms all-reduced on the device, no host round trips); sharded / replicated = the same "
                         "exchange driven from torch.distributed (host callback / full records)")
    ap.add_argument("--backend", default="auto", help="torch.distributed backend that launches the ranks and carries the "
                                                       "communicator id: auto = nccl (RCCL) with one GPU per rank, gloo when "
                                                       "ranks share a GPU (one-GPU functional tests of the N > 1 path)")
    ap.add_argument("--force-dist", action="store_true", help="take the N > 1 code path (process group, collectives) even "
                                                              "with one rank: exercises the RCCL calls on a 1-GPU box")
    ap.add_argument("--dump-flags", default="", help="write this rank's final pass flags to <path>.rank<r>.npy (tests)")
    ap.add_argument("--verify", action="store_true", help="N = 1: run the self-check of the N > 1 path anyway (needs --force-dist)")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the self-check behind the timed steps")
    ap.add_argument("--verify-reads", type=int, default=200_000, help="reads per rank of the self-check")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` as the driver types it: start the N ranks ourselves — one process per GPU under
        # torch.distributed.run on a free local port; rank 0 of that job prints the JSON line on our stdout
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])

    import torch
    import torch.distributed as dist
    from filtlong_amd import api, synth
    from filtlong_amd import dist as fdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and args.gpus != world:
        sys.exit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    n_dev = max(torch.cuda.device_count(), 1)
    device_index = local_rank % n_dev  # == local_rank unless several ranks share a GPU (the one-GPU functional tests)
    torch.cuda.set_device(device_index)
    multi = world > 1 or args.force_dist
    if args.backend == "auto":  # RCCL wants one device per rank; ranks that share a GPU rendezvous over gloo
        args.backend = "nccl" if n_dev >= world else "gloo"
    if args.backend != "nccl" and args.global_stage == "rccl" and not os.environ.get("FLX_RCCL_LIB"):
        args.global_stage = "sharded"  # the library's communicator is RCCL only (FLX_RCCL_LIB: the tests' loopback stand-in)
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

    if args.config in ("c3", "c4"):
        if multi:
            sys.exit("bench.py --config %s is a single-GPU configuration" % args.config)
        r = run_kmer(ctx, torch, dev, args.config, n, args.steps, args.warmup, args.target_frac, args.fixed_len, profile=args.read_profile)
        info = ctx.device_info()
        out = {"metric": "Mbases/s scored+sorted", "value": r["value"], "unit": "Mbases/s", "n_gpus": 1, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u32 k-mers / f64 scores", "data": "synthetic",
               "config": {"workload": r["workload"], "reads_total": n, "bases_total": r["bases"], "parallelism": "1 GPU",
                          "device": info["name"]},
               "roofline": r["roofline"], "stage_ms_per_step": r["stage_ms_per_step"], "lookups_per_s_G": r["lookups_per_s_G"],
               "children": r["children"], "set_size": r["set_size"], "cut": r["cut"]}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_kmer(args.config, args.cpu_sample_reads_kmer, args.full_cpu_baseline)
        print(json.dumps(out), flush=True)
        ctx.close()
        return

    profile = 1 if args.config == "c2wide" else 0
    # ---- build the packed batch in HBM (not timed) -----------------------------------------------
    t_setup = time.time()
    b = Batch(ctx, torch, dev, n, first, args.fixed_len)
    plane_bytes, local_bases = b.plane_bytes, b.bases
    # packed per-read record buffer [mean f64 | window f64 | length i32 | passed u8] -> one all-gather
    d_rec = fdist.alloc_records(n, dev)
    p_mean = d_rec.data_ptr()
    p_win = p_mean + 8 * n
    p_len = p_mean + 16 * n
    p_pass = p_mean + 20 * n
    t_mean, t_win, t_len, t_pass = fdist.record_views(d_rec, n)
    t_len.copy_(b.d_len)
    torch.cuda.synchronize()
    ctx.synth_qual_dev(synth.SEED, b.d_plane.data_ptr(), plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(),
                       b.d_ids.data_ptr(), n, profile=profile)

    total_n = n * world
    if multi:
        tb = torch.tensor([local_bases], dtype=torch.int64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tb)
        total_bases = int(tb.item())
        if args.global_stage == "rccl":
            # the library's own RCCL communicator: rank 0 draws the id, torch.distributed only carries its 128 bytes
            idt = torch.zeros(128, dtype=torch.uint8, device=dev if args.backend == "nccl" else "cpu")
            if rank == 0:
                idt.copy_(torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    else:
        total_bases = local_bases
    target = int(total_bases * args.target_frac)
    params = api.make_params(window_size=args.window_size)
    setup_s = time.time() - t_setup

    def score():
        ctx.score_reads_dev(b.d_plane.data_ptr(), plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(), b.d_ord.data_ptr(), n,
                            params, p_mean, p_win, p_pass)

    def step():
        score()
        if multi and args.global_stage == "rccl":
            return ctx.rank_and_cut_comm_dev(n, p_mean, p_win, p_len, p_pass, target_bases=target, total_bases=total_bases)
        if multi and args.global_stage == "sharded":
            # ONE all-gather of the mean qualities through torch.distributed; the selection's histograms go through the
            # host callback (filtlong_amd/dist.py, flx_rank_and_cut_sharded_dev).
            return fdist.sharded_rank_and_cut(ctx, t_mean, t_win, t_len, t_pass, target_bases=target,
                                              total_bases=total_bases)
        if multi:
            # ONE all-gather of the full per-read records; every rank then runs the identical single-GPU stage.
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
    comm_ms, _ = ctx.timing_get("flx_comm")
    # where a step's exchange time goes, collective by collective (rank 0's stream clocks; DESIGN §5: per step 1 all-gather of the mean
    # qualities, 8 device-side all-reduces of selection histograms — inside flx_rank_select's own bracket —, 2 host-visible sums, and
    # the three extra gathers + broadcasts only when the reference's own std::sort has to decide)
    comm_split = {}
    for key, prefix in (("counts_sum_host_visible", "flx_comm_counts"), ("allgather_means", "flx_comm_allgather_means"),
                        ("allreduce_histograms_device", "flx_comm_allreduce_dev"), ("band_sum_host_visible", "flx_comm_sum_host"),
                        ("allgather_records_fallback", "flx_comm_allgather_records"), ("broadcast_outcome_fallback", "flx_comm_broadcast_outcome")):
        ms, cnt = ctx.timing_get(prefix)
        comm_split[key] = {"ms_per_step": round(ms / args.steps, 4), "calls_per_step": round(cnt / args.steps, 2)}
    # (flx_comm_allreduce_dev is bracketed INSIDE flx_rank_select: taken out of the rank stage's time, so that `comm` and
    # `rank_other_kernels` do not overlap — advisor, round 5)
    nested_ms, _ = ctx.timing_get("flx_comm_allreduce_dev")
    rank_ms = max(rank_ms - nested_ms, 0.0)
    ctx.timing_enable(False)
    if args.dump_flags:
        torch.cuda.synchronize()
        np.save("%s.rank%d.npy" % (args.dump_flags, rank), t_pass.cpu().numpy())
    verify = None
    if multi and not args.no_verify and (world > 1 or args.verify):
        try:
            verify = verify_ranks(ctx, torch, dist, dev, rank, world, args, args.backend == "nccl")
        except Exception as e:  # (the check must not cost the line; a failure is reported in it)
            verify = {"ok": False, "error": repr(e)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bases * args.steps / elapsed / 1e6
        # algorithmic bytes per launch of the scoring kernel: L + 8 (offset) + 17 (outputs) per read
        algo_bytes = local_bases + 25 * n
        avg_kernel_ms = k_ms / max(k_n, 1)
        achieved = algo_bytes / (avg_kernel_ms * 1e-3) / 1e9 if k_n else 0.0
        # HBM traffic of one launch: PMC passes of this same command (tools/prof_phred.sh: separate rocprofv3 --pmc runs;
        # FETCH_SIZE x2 on gfx950 — calibrated for this kernel's 64-byte-per-read pattern on a known byte count,
        # profiles/r02_microbench.txt), recorded in profiles/ — NOT measured in this run; only quoted for the same workload.
        traffic, traffic_src = None, None
        tpath = next((q for q in (os.path.join(ROOT, "profiles", r + "_traffic_c2.json") for r in ("r06", "r05", "r04")) if os.path.exists(q)), "")
        if tpath and n == 10_000_000 and not args.fixed_len and args.window_size == 250 and profile == 0:
            rec = json.load(open(tpath))
            # only quoted while it still describes this kernel: same kernel name AND the kernel's source file unchanged since the pass
            if rec.get("kernel") == kernel_name and rec.get("kernel_source_sha16") == source_sha16("score_phred_regs.hip"):
                traffic, traffic_src = int(rec["traffic_bytes"]), "profiles/" + os.path.basename(tpath) + " (PMC pass of this command, not this run)"
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
                "workload": "%s synthetic reads per GPU x %s, Phred-only%s, --target_bases %d (%.0f%% of bases)%s" % (
                    "{:,}".format(n), ("fixed %d bp" % args.fixed_len) if args.fixed_len else "gamma(k=4) mean 10 kbp",
                    " (wide quality profile)" if profile else "", target, args.target_frac * 100,
                    "; C5 (80 M reads over 8 GPUs)" if (n == 10_000_000 and world == 8 and not args.fixed_len and not profile) else
                    "; C2" if (n == 10_000_000 and not args.fixed_len and not profile) else ""),
                "reads_total": total_n, "bases_total": total_bases, "window_size": args.window_size,
                "parallelism": ("1 GPU" if not multi else
                                "reads sharded by count; library-owned RCCL communicator: 1 all-gather of mean qualities + "
                                "selection histograms all-reduced on the device" if args.global_stage == "rccl" else
                                "reads sharded by count; 1 all-gather of mean qualities + all-reduced selection histograms "
                                "(torch.distributed, host callback)" if args.global_stage == "sharded" else
                                "reads sharded by count; 1 all-gather of per-read records, global stage replicated"),
                "device": info["name"], "gpus_visible": n_dev, "launch_backend": args.backend if multi else None,
                "rccl_ranks": ctx.L.flx_comm_world(ctx.h) if multi and args.global_stage == "rccl" else None,
                # the one array every rank needs from every other: the mean qualities, 8 bytes per read
                "allgather_bytes_per_rank": {"sent": 8 * n, "received": 8 * n * world} if multi else None,
                # which implementation of the cut (src/main.cpp:247-257) the timed steps ran: the weighted radix SELECT (default) or
                # the radix SORT + scan north_star names (FLX_RANK_SORT=1); both are timed in this run, stage_ms_per_step.cut_*
                "cut": "sort" if os.environ.get("FLX_RANK_SORT") == "1" else "select",
            },
            "roofline": {
                "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_kernel_ms": round(avg_kernel_ms, 3), "launches": int(k_n), "algorithmic_bytes": int(algo_bytes),
            },
            "stage_ms_per_step": {"score_kernel": round(k_ms / args.steps, 3), "sort": round(sort_ms / args.steps, 3),
                                  "rank_other_kernels": round(rank_ms / args.steps, 3),
                                  "comm": round(comm_ms / args.steps, 3),
                                  "comm_split": comm_split if multi else None},
            "cut": {"target_bases": int(rep.target_bases), "kept_bases": int(rep.kept_bases),
                    "outcome": int(rep.outcome), "audited": int(rep.audited), "exact_fallback": int(rep.exact_fallback)},
            "setup_s": round(setup_s, 1),
        }
        if verify is not None:
            out["verify"] = verify
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_phred(args.cpu_sample_reads)

        if world == 1 and not multi and args.config == "c2" and not args.no_extras:
            extras = {}
            # (1) both implementations of the cut on this batch's records: weighted radix SELECT (what the steps above ran)
            #     and radix SORT + scan (north_star's "device radix sort over final_score"); wall time of the whole stage
            score()  # pre-cut flags (hard cut-offs only) for both variants
            torch.cuda.synchronize()
            flags0 = t_pass.clone()
            for name, env in (("select", None), ("sort", "1")):
                if env:
                    os.environ["FLX_RANK_SORT"] = env
                else:
                    os.environ.pop("FLX_RANK_SORT", None)
                ts = []
                for _ in range(3):
                    score_flags = flags0.clone()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    ctx.rank_and_cut_dev(n, p_mean, p_win, p_len, score_flags.data_ptr(), target_bases=target,
                                         total_bases=total_bases)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
                extras.setdefault("global_stage_ms", {})[name] = round(min(ts), 3)
            os.environ.pop("FLX_RANK_SORT", None)
            # (round-5 review, item 4) both cuts in the headline record itself, and the fraction of the HBM peak the WHOLE timed
            # window reaches with either (algorithmic bytes of scoring + 46 B per read of the rank + cut stage)
            gs = extras["global_stage_ms"]
            out["stage_ms_per_step"]["cut_select"] = gs["select"]
            out["stage_ms_per_step"]["cut_sort"] = gs["sort"]
            ran = gs[out["config"]["cut"]]
            window_bytes = algo_bytes + 46 * n
            out["roofline"]["whole_window"] = {
                "algorithmic_bytes": int(window_bytes),
                "frac_as_timed": round(window_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_with_select": round(window_bytes / ((ms_per_step - ran + gs["select"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_with_sort": round(window_bytes / ((ms_per_step - ran + gs["sort"]) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "ms_per_step of the timed steps with the global stage's wall time exchanged for the other cut's (both measured in this run)"}
            # (1b) the exact-tie fallback (the reference's own std::sort over every entry on the host, csrc/rank.hip
            #      exact_host_cut), forced on the same records: its cost when an order-dependent tie group straddles the cut
            sel_flags = flags0.clone()
            ctx.rank_and_cut_dev(n, p_mean, p_win, p_len, sel_flags.data_ptr(), target_bases=target, total_bases=total_bases)
            os.environ["FLX_RANK_EXACT"] = "1"
            try:
                fb_flags = flags0.clone()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fb = ctx.rank_and_cut_dev(n, p_mean, p_win, p_len, fb_flags.data_ptr(), target_bases=target, total_bases=total_bases)
                torch.cuda.synchronize()
                extras["exact_fallback_ms"] = {"ms": round((time.perf_counter() - t1) * 1e3, 1), "reads": n,
                                               "taken": int(fb.exact_fallback), "host_threads": os.cpu_count(),
                                               "same_flags_as_select": bool(torch.equal(fb_flags, sel_flags)),
                                               "kept_bases": int(fb.kept_bases)}
            finally:
                os.environ.pop("FLX_RANK_EXACT", None)
            del sel_flags, fb_flags
            # (2) the same Phred workload with a wide quality spread
            ctx.synth_qual_dev(synth.SEED, b.d_plane.data_ptr(), plane_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(),
                               b.d_ids.data_ptr(), n, profile=1)
            score()
            ctx.timing_enable(True)
            ctx.timing_reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                score()
                ctx.rank_and_cut_dev(n, p_mean, p_win, p_len, p_pass, target_bases=target, total_bases=total_bases)
            torch.cuda.synchronize()
            elw = (time.perf_counter() - t1) / 2
            kname = ctx.last_phred_kernel()
            kw_ms, kw_n = ctx.timing_get(kname)
            ctx.timing_enable(False)
            extras["c2_wide_quality"] = {
                "workload": "C2 with per-read centre Q3..Q44 and q <= 50 (synth profile 1)", "value": round(local_bases / elw / 1e6, 1),
                "unit": "Mbases/s", "ms_per_step": round(elw * 1e3, 3), "kernel": kname,
                "avg_kernel_ms": round(kw_ms / max(kw_n, 1), 3),
                "roofline_frac": round(algo_bytes / (kw_ms / max(kw_n, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # (3) the k-mer configurations at BASELINE size (free the Phred batch first: 100 GB per plane)
            del b, d_rec, t_mean, t_win, t_len, t_pass, flags0, score_flags
            torch.cuda.empty_cache()
            for cfg in ("c3", "c4"):
                try:
                    r = run_kmer(ctx, torch, dev, cfg, n, 2, 1, args.target_frac)
                    if not args.no_cpu_baseline:
                        r["cpu_baseline"] = cpu_baseline_kmer(cfg, args.cpu_sample_reads_kmer, False)
                    extras[cfg] = r
                except Exception as e:  # an extra must not cost the headline line
                    extras[cfg] = {"error": repr(e)}
                torch.cuda.empty_cache()
            # (3b) C3 on reads the SURVEY §8(d) generator does not make (round-5 review, item 2) — beside C3, never instead of it:
            #      insertions / deletions (every one moves the read's diagonal in the set's text), and reads unrelated to the reference
            for key, prof in (("c3_indels", 1), ("c3_unrelated", 2)):
                try:
                    r = run_kmer(ctx, torch, dev, "c3", n, 2, 1, args.target_frac, profile=prof)
                    if isinstance(extras.get("c3"), dict) and "ms_per_step" in extras["c3"]:
                        r["ms_per_step_over_c3"] = round(r["ms_per_step"] / extras["c3"]["ms_per_step"], 3)
                    extras[key] = r
                except Exception as e:
                    extras[key] = {"error": repr(e)}
                torch.cuda.empty_cache()
            # (4) end to end (file -> stdout) through the C++ CLI, MEASURED IN THIS RUN: a 2 GB FASTQ of the same synthetic
            #     Phred workload, the drop-in binary and the reference binary on the same file, stdout compared byte for byte
            try:
                extras["end_to_end_cli"] = end_to_end_cli(100_000)
            except Exception as e:  # an extra must not cost the headline line
                extras["end_to_end_cli"] = {"measured_in_this_run": False, "error": repr(e)}
            try:
                extras["end_to_end_cli_kmer_short"] = end_to_end_cli_kmer_short(3000, args.e2e_kmer_pairs)
            except Exception as e:
                extras["end_to_end_cli_kmer_short"] = {"measured_in_this_run": False, "error": repr(e)}
            try:
                extras["end_to_end_cli_kmer"] = end_to_end_cli_kmer(3000)
            except Exception as e:
                extras["end_to_end_cli_kmer"] = {"measured_in_this_run": False, "error": repr(e)}
            # the 20 GB and gzip runs are too long for the default bench: recorded by tools/bench_e2e_big.sh / bench_e2e_gz.sh
            # (the newest recording there is; a figure of an earlier round says so — the ingest code has changed since)
            for key, name in (("end_to_end_cli_20GB_recorded", "e2e_big.json"), ("end_to_end_cli_gzip_recorded", "e2e_gz.json")):
                for rnd in ("r05", "r04", "r03"):
                    fp = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
                    if os.path.exists(fp):
                        e = json.load(open(fp))
                        extras[key] = {"source": "profiles/%s_%s (not measured in this run; recorded in round %s%s)" % (
                                           rnd, name, rnd[2:], "" if rnd == "r05" else ", on that round's ingest code"),
                                       "measured_in_this_run": False, **{k: e[k] for k in e if k not in ("note",)}}
                        break
            out["extras"] = extras
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
