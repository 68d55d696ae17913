"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and helpers to drive the real
reference binaries under oracle/_ref/.  TEST INFRASTRUCTURE ONLY — the product package
(filtlong_amd/) never imports this module.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_PROBE = os.path.join(REF_DIR, "ref_probe")
REF_FILTLONG = os.path.join(REF_DIR, "filtlong")
REF_BENCH = os.path.join(REF_DIR, "ref_bench")
REFERENCE_TESTS = "/root/reference/test"


class Params(C.Structure):
    _fields_ = [
        ("window_size", C.c_int32),
        ("min_length_set", C.c_int32), ("min_length", C.c_int32),
        ("max_length_set", C.c_int32), ("max_length", C.c_int32),
        ("min_mean_q_set", C.c_int32),
        ("min_window_q_set", C.c_int32),
        ("min_mean_q", C.c_double),
        ("min_window_q", C.c_double),
        ("trim", C.c_int32),
        ("split_set", C.c_int32), ("split", C.c_int32),
        ("_pad", C.c_int32),
    ]


class ReadResult(C.Structure):
    _fields_ = [
        ("mean_q", C.c_double), ("window_q", C.c_double), ("length_score", C.c_double),
        ("length", C.c_int32), ("passed", C.c_int32), ("first", C.c_int32), ("last", C.c_int32),
        ("n_bad", C.c_int32), ("n_child", C.c_int32),
    ]


class CutReport(C.Structure):
    _fields_ = [
        ("target_bases", C.c_int64), ("kept_bases", C.c_int64), ("outcome", C.c_int32), ("_pad", C.c_int32),
        ("mean_quality", C.c_double), ("stdev_quality", C.c_double), ("min_z", C.c_double), ("max_z", C.c_double),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.flo_qscore_to_quality.restype = C.c_double
    L.flo_qscore_to_quality.argtypes = [C.c_int]
    L.flo_phred_lut.argtypes = [C.c_void_p]
    L.flo_mean_quality.restype = C.c_double
    L.flo_mean_quality.argtypes = [C.c_void_p, C.c_uint64]
    L.flo_window_quality.restype = C.c_double
    L.flo_window_quality.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.flo_length_score.restype = C.c_double
    L.flo_length_score.argtypes = [C.c_int]
    for f in (L.flo_base_fwd, L.flo_base_rev):
        f.restype = C.c_uint32
        f.argtypes = [C.c_int]
    for f in (L.flo_start_kmer_fwd, L.flo_start_kmer_rev):
        f.restype = C.c_uint32
        f.argtypes = [C.c_char_p]
    L.flo_bloom_parameters.argtypes = [C.c_uint64, C.c_double, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.flo_bloom_salts.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    L.flo_bloom_hash.restype = C.c_uint32
    L.flo_bloom_hash.argtypes = [C.c_uint32, C.c_uint32]
    L.flo_kmerset_new.restype = C.c_void_p
    L.flo_kmerset_free.argtypes = [C.c_void_p]
    L.flo_kmerset_size.restype = C.c_uint64
    L.flo_kmerset_size.argtypes = [C.c_void_p]
    L.flo_kmerset_contains.restype = C.c_int
    L.flo_kmerset_contains.argtypes = [C.c_void_p, C.c_uint32]
    L.flo_kmerset_dump.restype = C.c_uint64
    L.flo_kmerset_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.flo_kmerset_add_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_int]
    L.flo_score_read.restype = C.c_int
    L.flo_score_read.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(Params),
                                 C.POINTER(ReadResult), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.flo_final_score.restype = C.c_double
    L.flo_final_score.argtypes = [C.c_double] * 6
    L.flo_rank_and_cut.restype = C.c_int
    L.flo_rank_and_cut.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                   C.c_double, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_int64,
                                   C.c_void_p, C.POINTER(CutReport)]
    L.flo_synth_mix.restype = C.c_uint64
    L.flo_synth_mix.argtypes = [C.c_uint64] * 4
    L.flo_synth_qual.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.flo_synth_bases.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
    L.flo_synth_seq.argtypes = [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
    _lib = L
    return L


def make_params(window_size=250, min_length=None, max_length=None, min_mean_q=None, min_window_q=None, trim=False,
                split=None):
    p = Params()
    p.window_size = window_size
    p.min_length_set, p.min_length = (1, min_length) if min_length is not None else (0, 0)
    p.max_length_set, p.max_length = (1, max_length) if max_length is not None else (0, 0)
    p.min_mean_q_set, p.min_mean_q = (1, min_mean_q) if min_mean_q is not None else (0, 0.0)
    p.min_window_q_set, p.min_window_q = (1, min_window_q) if min_window_q is not None else (0, 0.0)
    p.trim = 1 if trim else 0
    p.split_set, p.split = (1, split) if split is not None else (0, 0)
    return p


def params_to_argv(p):
    """The filtlong argv that sets the same hot-path parameters (src/arguments.cpp:152-215)."""
    a = ["--window_size", str(p.window_size)]
    if p.min_length_set:
        a += ["--min_length", str(p.min_length)]
    if p.max_length_set:
        a += ["--max_length", str(p.max_length)]
    if p.min_mean_q_set:
        a += ["--min_mean_q", repr(float(p.min_mean_q))]
    if p.min_window_q_set:
        a += ["--min_window_q", repr(float(p.min_window_q))]
    if p.trim:
        a += ["--trim"]
    if p.split_set:
        a += ["--split", str(p.split)]
    return a


class KmerSet:
    """Oracle reference 16-mer set (reference src/kmers.cpp)."""

    def __init__(self):
        self.h = lib().flo_kmerset_new()

    def add_assembly(self, seqs):
        for s in seqs:
            lib().flo_kmerset_add_sequence(self.h, bytes(s), len(s), 0)

    def add_short_reads(self, seqs):
        for s in seqs:
            lib().flo_kmerset_add_sequence(self.h, bytes(s), len(s), 1)

    def __len__(self):
        return int(lib().flo_kmerset_size(self.h))

    def contains(self, k):
        return bool(lib().flo_kmerset_contains(self.h, int(k)))

    def dump(self):
        n = len(self)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        lib().flo_kmerset_dump(self.h, out.ctypes.data, n)
        return out[:n]

    def __del__(self):
        try:
            lib().flo_kmerset_free(self.h)
        except Exception:
            pass


def score_read(seq, qual, params, kmerset=None, cap=4096):
    """Returns dict with the parent fields, bad ranges, child ranges and child results."""
    L = len(seq) if seq is not None else len(qual)
    res = ReadResult()
    bad = np.zeros(2 * cap, dtype=np.int32)
    kid = np.zeros(2 * cap, dtype=np.int32)
    kids = (ReadResult * cap)()
    seq_b = bytes(seq) if seq is not None else b"\0" * L
    qual_b = bytes(qual) if qual is not None else b"\0" * L
    rc = lib().flo_score_read(kmerset.h if kmerset is not None else None, seq_b, qual_b, L, C.byref(params),
                              C.byref(res), bad.ctypes.data, kid.ctypes.data, C.addressof(kids), cap)
    assert rc == 0
    return {
        "length": res.length, "length_score": res.length_score, "mean_q": res.mean_q, "window_q": res.window_q,
        "passed": res.passed, "first": res.first, "last": res.last,
        "bad": [(int(bad[2 * i]), int(bad[2 * i + 1])) for i in range(res.n_bad)],
        "child_ranges": [(int(kid[2 * i]), int(kid[2 * i + 1])) for i in range(res.n_child)],
        "children": [
            {"length": k.length, "length_score": k.length_score, "mean_q": k.mean_q, "window_q": k.window_q,
             "passed": k.passed, "first": k.first, "last": k.last}
            for k in kids[: res.n_child]
        ],
    }


def score_plane_mt(plane, offsets, lengths, params, kmerset=None, threads=None, child_cap=None):
    """flo_score_plane_mt: every read of a packed batch through flo_score_read on `threads` host threads (default: all the
    cores this process may use).  plane: uint8 array (quality strings, or sequences when kmerset is given); returns the per-read
    arrays and the children as a CSR (keys as filtlong_amd.api.Context.score_reads returns them)."""
    n = len(lengths)
    if threads is None:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    pl = np.ascontiguousarray(plane, dtype=np.uint8)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    cap = int(child_cap if child_cap is not None else 4 * n + 16)
    out = {"mean_q": np.zeros(n, np.float64), "window_q": np.zeros(n, np.float64), "passed": np.zeros(n, np.uint8),
           "first": np.zeros(n, np.int32), "last": np.zeros(n, np.int32), "child_offsets": np.zeros(n + 1, np.uint64),
           "child_ranges": np.zeros((cap, 2), np.int32), "child_mean_q": np.zeros(cap, np.float64),
           "child_window_q": np.zeros(cap, np.float64), "child_passed": np.zeros(cap, np.uint8)}
    f = lib().flo_score_plane_mt
    f.restype = C.c_int64
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(Params), C.c_int] + [C.c_void_p] * 6 \
        + [C.c_uint64] + [C.c_void_p] * 4
    nc = f(kmerset.h if kmerset is not None else None, 1 if kmerset is not None else 0, pl.ctypes.data, off.ctypes.data, ln.ctypes.data, n,
           C.byref(params), int(threads), out["mean_q"].ctypes.data, out["window_q"].ctypes.data, out["passed"].ctypes.data,
           out["first"].ctypes.data, out["last"].ctypes.data, out["child_offsets"].ctypes.data, cap, out["child_ranges"].ctypes.data,
           out["child_mean_q"].ctypes.data, out["child_window_q"].ctypes.data, out["child_passed"].ctypes.data)
    assert nc >= 0, "flo_score_plane_mt failed (child capacity?)"
    for k in ("child_ranges", "child_mean_q", "child_window_q", "child_passed"):
        out[k] = out[k][:nc]
    out["n_children"] = int(nc)
    return out


def reads2_gather(lengths, sc):
    """flo_reads2_gather on a score dictionary (keys as filtlong_amd.api.Context.score_reads returns them)."""
    n = len(lengths)
    ln = np.ascontiguousarray(lengths, dtype=np.int32)
    dt = {"mean_q": np.float64, "window_q": np.float64, "passed": np.uint8, "child_offsets": np.uint64,
          "child_ranges": np.int32, "child_mean_q": np.float64, "child_window_q": np.float64, "child_passed": np.uint8}
    a = {k: np.ascontiguousarray(sc[k], dtype=t) for k, t in dt.items()}
    cap = n + len(a["child_mean_q"])
    out = {"mean_q": np.zeros(cap, np.float64), "window_q": np.zeros(cap, np.float64), "length": np.zeros(cap, np.int32),
           "passed": np.zeros(cap, np.uint8), "parent": np.zeros(cap, np.uint32), "child": np.zeros(cap, np.int64)}
    f = lib().flo_reads2_gather
    f.restype = C.c_uint64
    f.argtypes = [C.c_uint64] + [C.c_void_p] * 15
    n2 = f(n, ln.ctypes.data, a["mean_q"].ctypes.data, a["window_q"].ctypes.data, a["passed"].ctypes.data,
           a["child_offsets"].ctypes.data, a["child_ranges"].ctypes.data, a["child_mean_q"].ctypes.data,
           a["child_window_q"].ctypes.data, a["child_passed"].ctypes.data, out["mean_q"].ctypes.data,
           out["window_q"].ctypes.data, out["length"].ctypes.data, out["passed"].ctypes.data, out["parent"].ctypes.data,
           out["child"].ctypes.data)
    return {k: v[:n2] for k, v in out.items()}


def rank_and_cut(mean_q, window_q, length, passed, lw=1.0, mw=1.0, ww=1.0, target_bases=None, keep_percent=None,
                 total_bases=None):
    n = len(mean_q)
    mq = np.array(mean_q, dtype=np.float64)
    wq = np.array(window_q, dtype=np.float64)
    ln = np.ascontiguousarray(length, dtype=np.int32)
    ps = np.array(passed, dtype=np.uint8)
    fs = np.zeros(n, dtype=np.float64)
    rep = CutReport()
    if total_bases is None:
        total_bases = int(ln.astype(np.int64).sum())
    lib().flo_rank_and_cut(n, mq.ctypes.data, wq.ctypes.data, ln.ctypes.data, ps.ctypes.data, lw, mw, ww,
                           1 if target_bases is not None else 0, int(target_bases or 0),
                           1 if keep_percent is not None else 0, float(keep_percent or 0.0), int(total_bases),
                           fs.ctypes.data, C.byref(rep))
    return {"mean_q": mq, "window_q": wq, "passed": ps, "final_score": fs, "target_bases": rep.target_bases,
            "kept_bases": rep.kept_bases, "outcome": rep.outcome, "mean_quality": rep.mean_quality,
            "stdev_quality": rep.stdev_quality, "min_z": rep.min_z, "max_z": rep.max_z}


# --------------------------------------------------------------------------------------------
# the real reference (oracle/_ref), when built
# --------------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(REF_PROBE) and os.path.exists(REF_FILTLONG)


def write_reads_bin(path, reads):
    """reads: list of (name, seq bytes, qual bytes or None)"""
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(reads)))
        for name, seq, qual in reads:
            nb = name.encode() if isinstance(name, str) else name
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", len(seq)))
            f.write(bytes(seq))
            if qual is None:
                f.write(b"\0")
            else:
                f.write(b"\1")
                f.write(bytes(qual))


def _parse_probe_read(tok):
    d = {"name": tok[1], "length": int(tok[2]), "length_score": float.fromhex(tok[3]),
         "mean_q": float.fromhex(tok[4]), "window_q": float.fromhex(tok[5]), "passed": int(tok[6]),
         "first": int(tok[7]), "last": int(tok[8])}
    i = 9
    nb = int(tok[i][1:])
    d["bad"] = [tuple(int(x) for x in t.split("-")) for t in tok[i + 1: i + 1 + nb]]
    i += 1 + nb
    nc = int(tok[i][1:])
    d["child_ranges"] = [tuple(int(x) for x in t.split("-")) for t in tok[i + 1: i + 1 + nc]]
    d["children"] = []
    return d


def ref_probe(reads, filtlong_args, kmer_queries=None):
    """Run the reference's Read constructor over `reads`; returns (kmers_empty, {kmer: present}, [read dicts])."""
    env = dict(os.environ, LANG="C", LC_ALL="C")
    with tempfile.TemporaryDirectory() as td:
        rp = os.path.join(td, "reads.bin")
        write_reads_bin(rp, reads)
        qp = "-"
        if kmer_queries is not None:
            qp = os.path.join(td, "q.bin")
            q = np.ascontiguousarray(kmer_queries, dtype=np.uint32)
            with open(qp, "wb") as f:
                f.write(struct.pack("<Q", len(q)))
                f.write(q.tobytes())
        out = subprocess.run([REF_PROBE, rp, qp, "--"] + list(filtlong_args) + [rp], env=env, check=True,
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    empty = None
    present = {}
    res = []
    for line in out.splitlines():
        tok = line.split(" ")
        if tok[0] == "E":
            empty = bool(int(tok[1]))
        elif tok[0] == "K":
            present[int(tok[1], 16)] = bool(int(tok[2]))
        elif tok[0] == "R":
            res.append(_parse_probe_read(tok))
        elif tok[0] == "c":
            res[-1]["children"].append(_parse_probe_read(tok))
    return empty, present, res


def run_ref_filtlong(args, cwd=None):
    """Run the real reference binary; returns (returncode, stdout bytes, stderr str)."""
    env = dict(os.environ, LANG="C", LC_ALL="C")
    p = subprocess.run([REF_FILTLONG] + list(args), env=env, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


# --------------------------------------------------------------------------------------------
# minimal FASTA/FASTQ reader for the (well-formed, 4-line / 2-line) fixtures used in tests
# --------------------------------------------------------------------------------------------
def read_fastx(path):
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    with op(path, "rb") as f:
        lines = [l.rstrip(b"\r\n") for l in f]
    i = 0
    while i < len(lines):
        if not lines[i]:
            i += 1
            continue
        hdr = lines[i]
        name = hdr[1:].split()[0].decode() if len(hdr) > 1 else ""
        if hdr[:1] == b"@":
            recs.append((name, lines[i + 1], lines[i + 3]))
            i += 4
        elif hdr[:1] == b">":
            j = i + 1
            seq = b""
            while j < len(lines) and lines[j][:1] != b">":
                seq += lines[j]
                j += 1
            recs.append((name, seq, None))
            i = j
        else:
            raise ValueError("bad record at line %d of %s" % (i, path))
    return recs
