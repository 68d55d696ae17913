"""Host-side Python mirror of the reference's interface for the scoring hot path.

Thin layer over the C ABI (include/filtlong_hip.h) used by the tests and bench.py.  Names follow the
reference: ``Kmers`` (src/kmers.h:28-56), per-read scoring = the batched ``Read::Read``
(src/read.cpp:25-144), ``rank_and_cut`` = the global stage inlined in ``main`` (src/main.cpp:169-261).
All compute happens in the HIP library; there is no CPU fallback here.
"""
import ctypes as C
import traceback
import weakref

import numpy as np

from . import _lib
from ._lib import CutReport, FlxError, Params, Scores

CUT_NONE, CUT_NOT_ENOUGH, CUT_ALREADY_BELOW, CUT_SORTED = 0, 1, 2, 3


def make_params(window_size=250, min_length=None, max_length=None, min_mean_q=None, min_window_q=None, trim=False,
                split=None):
    """flx_params from keyword arguments named like the reference's CLI flags (src/arguments.cpp:152-205)."""
    p = Params()
    p.window_size = int(window_size)
    p.min_length_set, p.min_length = (1, int(min_length)) if min_length is not None else (0, 0)
    p.max_length_set, p.max_length = (1, int(max_length)) if max_length is not None else (0, 0)
    p.min_mean_q_set, p.min_mean_q = (1, float(min_mean_q)) if min_mean_q is not None else (0, 0.0)
    p.min_window_q_set, p.min_window_q = (1, float(min_window_q)) if min_window_q is not None else (0, 0.0)
    p.trim = 1 if trim else 0
    p.split_set, p.split = (1, int(split)) if split is not None else (0, 0)
    return p


def pack_reads(strings):
    """Pack byte strings into the 16-byte-aligned read plane of flx_score_batch.

    Returns (plane uint8[plane_bytes], offsets uint64[n], lengths int32[n])."""
    L = _lib.load()
    n = len(strings)
    lengths = np.array([len(s) for s in strings], dtype=np.int32)
    offsets = np.zeros(max(n, 1), dtype=np.uint64)
    pb = C.c_uint64()
    rc = L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb))
    if rc:
        raise FlxError(rc, "flx_plane_layout")
    plane = np.zeros(pb.value, dtype=np.uint8)
    for s, o in zip(strings, offsets):
        if len(s):
            plane[int(o):int(o) + len(s)] = np.frombuffer(bytes(s), dtype=np.uint8)
    return plane, offsets[:n], lengths


def length_order(lengths):
    """Processing order: read indices by descending length (flx_length_order)."""
    L = _lib.load()
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    order = np.zeros(max(len(lengths), 1), dtype=np.uint32)
    rc = L.flx_length_order(lengths.ctypes.data, len(lengths), order.ctypes.data)
    if rc:
        raise FlxError(rc, "flx_length_order")
    return order[:len(lengths)]


class Context:
    """One flx_ctx (one GPU).  Fails loudly without a gfx950 device."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.flx_ctx_create(int(device), C.byref(h))
        if rc:
            raise FlxError(rc, self.L.flx_last_error(None).decode())
        self.h = h
        self._sets = []  # weak references to the Kmers created on this context (destroyed before the context)

    def close(self):
        if self.h:
            for ref in self._sets:
                ks = ref()
                if ks is not None:
                    ks.close()
            self._sets = []
            self.L.flx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            raise FlxError(rc, self.L.flx_last_error(self.h).decode())

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu, mem = C.c_int(), C.c_uint64()
        self._check(self.L.flx_ctx_device_info(self.h, name, 256, C.byref(ncu), C.byref(mem)))
        return {"name": name.value.decode(), "n_cu": ncu.value, "hbm_bytes": mem.value}

    def set_stream(self, stream_ptr):
        self._check(self.L.flx_ctx_set_stream(self.h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self._check(self.L.flx_ctx_synchronize(self.h))

    # ---- timing ------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self.L.flx_timing_enable(self.h, 1 if on else 0))

    def timing_reset(self):
        self._check(self.L.flx_timing_reset(self.h))

    def timing_get(self, prefix=""):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.flx_timing_get(self.h, prefix.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # ---- seam 2: per-read scoring (host buffers) -----------------------------------------------
    def score_reads(self, plane, offsets, lengths, params, kmers=None, order=None, child_capacity=None):
        """Batched Read::Read.  Returns a dict of numpy arrays in input order."""
        n = len(lengths)
        plane = np.ascontiguousarray(plane, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        out = {
            "mean_q": np.zeros(n, dtype=np.float64), "window_q": np.zeros(n, dtype=np.float64),
            "passed": np.zeros(n, dtype=np.uint8), "first": np.full(n, -1, dtype=np.int32),
            "last": np.full(n, -1, dtype=np.int32), "child_offsets": np.zeros(n + 1, dtype=np.uint64),
        }
        if child_capacity is None:
            child_capacity = max(16, 4 * n)
        ordp = None
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.uint32)
            ordp = order.ctypes.data
        while True:
            cr = np.zeros(2 * child_capacity, dtype=np.int32)
            cm = np.zeros(child_capacity, dtype=np.float64)
            cw = np.zeros(child_capacity, dtype=np.float64)
            cp = np.zeros(child_capacity, dtype=np.uint8)
            s = Scores()
            s.mean_q, s.window_q, s.passed = out["mean_q"].ctypes.data, out["window_q"].ctypes.data, out["passed"].ctypes.data
            s.first, s.last = out["first"].ctypes.data, out["last"].ctypes.data
            s.child_offsets = out["child_offsets"].ctypes.data
            s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (cr.ctypes.data, cm.ctypes.data,
                                                                              cw.ctypes.data, cp.ctypes.data)
            s.child_capacity = child_capacity
            rc = self.L.flx_score_batch(self.h, kmers.h if kmers is not None else None, plane.ctypes.data,
                                        plane.nbytes, offsets.ctypes.data, lengths.ctypes.data, ordp, n,
                                        C.byref(params), C.byref(s))
            if rc == 5 and s.n_children > child_capacity:  # FLX_ERR_CAPACITY: retry with the reported size
                child_capacity = int(s.n_children)
                continue
            self._check(rc)
            break
        nc = int(s.n_children)
        out["child_ranges"] = cr[:2 * nc].reshape(-1, 2)
        out["child_mean_q"], out["child_window_q"], out["child_passed"] = cm[:nc], cw[:nc], cp[:nc]
        return out

    # ---- seam 2, streaming ----------------------------------------------------------------------
    def score_stream(self, chunks, params, kmers=None, chunk_bytes=1 << 20, chunk_reads=1 << 12, grow=False):
        """flx_pipeline_*: `chunks` is an iterable of lists of byte strings (Phred mode: qualities, k-mer mode: sequences).
        Every chunk is packed into the pipeline's pinned buffer and submitted; returns the same dict as score_reads for
        all reads in submission order.  grow: enlarge the slots (flx_pipeline_reserve) for a chunk that does not fit."""
        L = self.L
        pipe = C.c_void_p()
        self._check(L.flx_pipeline_create(self.h, kmers.h if kmers is not None else None, C.byref(params), chunk_bytes,
                                          chunk_reads, C.byref(pipe)))
        try:
            for strings in chunks:
                n = len(strings)
                lengths = np.array([len(x) for x in strings], dtype=np.int32)
                offsets = np.zeros(max(n, 1), dtype=np.uint64)
                pb = C.c_uint64()
                self._check(L.flx_plane_layout(lengths.ctypes.data, n, offsets.ctypes.data, C.byref(pb)))
                if grow:
                    self._check(L.flx_pipeline_reserve(pipe, pb.value, n))
                buf, cap_b, cap_r = C.c_void_p(), C.c_uint64(), C.c_uint64()
                self._check(L.flx_pipeline_next_buffer(pipe, C.byref(buf), C.byref(cap_b), C.byref(cap_r)))
                if pb.value <= cap_b.value:  # an oversized chunk is submitted as is: the library must refuse it
                    view = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(max(int(pb.value), 1),))
                    view[:pb.value] = 0
                    for x, o in zip(strings, offsets):
                        if len(x):
                            view[int(o):int(o) + len(x)] = np.frombuffer(bytes(x), dtype=np.uint8)
                self._check(L.flx_pipeline_submit(pipe, pb.value, offsets.ctypes.data, lengths.ctypes.data, n))
            s, n_all = Scores(), C.c_uint64()
            self._check(L.flx_pipeline_finish(pipe, C.byref(s), C.byref(n_all)))
            n, nc = int(n_all.value), int(s.n_children)

            def take(ptr, dtype, count):
                if count == 0:
                    return np.zeros(0, dtype=dtype)
                ct = np.ctypeslib.as_ctypes_type(dtype)
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(count,)).copy()

            out = {"mean_q": take(s.mean_q, np.float64, n), "window_q": take(s.window_q, np.float64, n),
                   "passed": take(s.passed, np.uint8, n), "first": take(s.first, np.int32, n),
                   "last": take(s.last, np.int32, n), "child_offsets": take(s.child_offsets, np.uint64, n + 1),
                   "child_ranges": take(s.child_ranges, np.int32, 2 * nc).reshape(-1, 2),
                   "child_mean_q": take(s.child_mean_q, np.float64, nc),
                   "child_window_q": take(s.child_window_q, np.float64, nc),
                   "child_passed": take(s.child_passed, np.uint8, nc)}
            return out
        finally:
            L.flx_pipeline_destroy(pipe)

    # ---- seam 2, device-resident ---------------------------------------------------------------
    def score_reads_dev(self, d_plane, plane_bytes, d_offsets, d_lengths, d_order, n, params, d_mean_q, d_window_q,
                        d_passed, kmers=None):
        s = Scores()
        s.mean_q, s.window_q, s.passed = d_mean_q, d_window_q, d_passed
        self._check(self.L.flx_score_batch_dev(self.h, kmers.h if kmers is not None else None, d_plane, plane_bytes,
                                               d_offsets, d_lengths, d_order, n, C.byref(params), C.byref(s)))

    # ---- reads2 gather (src/main.cpp:138-147) ------------------------------------------------------
    def reads2_gather(self, lengths, scores):
        """flx_reads2_gather on the dictionary score_reads returns: file order, parents replaced in place by their children.
        Returns a dict of numpy arrays mean_q / window_q / length / passed / parent / child in reads2 order."""
        n = len(lengths)
        ln = np.ascontiguousarray(lengths, dtype=np.int32)
        nc = len(scores["child_mean_q"])
        keep = [np.ascontiguousarray(scores[k]) for k in ("mean_q", "window_q", "passed", "child_offsets", "child_ranges",
                                                          "child_mean_q", "child_window_q", "child_passed")]
        s = Scores()
        s.mean_q, s.window_q, s.passed, s.child_offsets = (keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data,
                                                           keep[3].ctypes.data)
        s.child_ranges, s.child_mean_q, s.child_window_q, s.child_passed = (keep[4].ctypes.data, keep[5].ctypes.data,
                                                                          keep[6].ctypes.data, keep[7].ctypes.data)
        s.child_capacity = s.n_children = nc
        cap = n + nc
        out = {"mean_q": np.zeros(cap, np.float64), "window_q": np.zeros(cap, np.float64), "length": np.zeros(cap, np.int32),
               "passed": np.zeros(cap, np.uint8), "parent": np.zeros(cap, np.uint32), "child": np.zeros(cap, np.int64)}
        n2 = C.c_uint64()
        self._check(self.L.flx_reads2_gather(self.h, n, ln.ctypes.data, C.byref(s), cap, out["mean_q"].ctypes.data,
                                             out["window_q"].ctypes.data, out["length"].ctypes.data, out["passed"].ctypes.data,
                                             out["parent"].ctypes.data, out["child"].ctypes.data, C.byref(n2)))
        return {k: v[:n2.value] for k, v in out.items()}

    def reads2_gather_dev(self, n, d_lengths, scores, capacity, d_mean2, d_window2, d_length2, d_passed2, d_parent2=None,
                          d_child2=None):
        """flx_reads2_gather_dev: `scores` is a _lib.Scores with device pointers (as filled by score_kmer_dev); returns n2."""
        n2 = C.c_uint64()
        self._check(self.L.flx_reads2_gather_dev(self.h, n, d_lengths, C.byref(scores), capacity, d_mean2, d_window2,
                                                 d_length2, d_passed2, d_parent2, d_child2, C.byref(n2)))
        return int(n2.value)

    # ---- seam 3: rank + cut --------------------------------------------------------------------
    def rank_and_cut(self, mean_q, window_q, length, passed, length_weight=1.0, mean_q_weight=1.0,
                     window_q_weight=1.0, target_bases=None, keep_percent=None, total_bases=None,
                     want_scores=True):
        n = len(mean_q)
        mq = np.ascontiguousarray(mean_q, dtype=np.float64)
        wq = np.ascontiguousarray(window_q, dtype=np.float64)
        ln = np.ascontiguousarray(length, dtype=np.int32)
        ps = np.array(passed, dtype=np.uint8)
        fs = np.zeros(n, dtype=np.float64) if want_scores else None
        if total_bases is None:
            total_bases = int(ln.astype(np.int64).sum())
        rep = CutReport()
        self._check(self.L.flx_rank_and_cut(self.h, n, mq.ctypes.data, wq.ctypes.data, ln.ctypes.data, ps.ctypes.data,
                                            length_weight, mean_q_weight, window_q_weight,
                                            1 if target_bases is not None else 0, int(target_bases or 0),
                                            1 if keep_percent is not None else 0, float(keep_percent or 0.0),
                                            int(total_bases), fs.ctypes.data if want_scores else None, C.byref(rep)))
        return {"passed": ps, "final_score": fs, "report": rep}

    def rank_and_cut_dev(self, n, d_mean_q, d_window_q, d_length, d_passed, length_weight=1.0, mean_q_weight=1.0,
                         window_q_weight=1.0, target_bases=None, keep_percent=None, total_bases=0,
                         d_final_score=None):
        rep = CutReport()
        self._check(self.L.flx_rank_and_cut_dev(self.h, n, d_mean_q, d_window_q, d_length, d_passed, length_weight,
                                                mean_q_weight, window_q_weight,
                                                1 if target_bases is not None else 0, int(target_bases or 0),
                                                1 if keep_percent is not None else 0, float(keep_percent or 0.0),
                                                int(total_bases), d_final_score, C.byref(rep)))
        return rep

    def rank_and_cut_sharded_dev(self, n_total, d_mean_q_all, first, n_local, d_window_q, d_length, d_passed, rank, world,
                                 reduce=None, length_weight=1.0, mean_q_weight=1.0, window_q_weight=1.0,
                                 target_bases=None, keep_percent=None, total_bases=0, d_final_score=None):
        """flx_rank_and_cut_sharded_dev.  `reduce(buf)` receives a writable numpy uint64 view of the library's host
        buffer and must replace it by its sum over all ranks (filtlong_amd.dist.make_reduce builds one over
        torch.distributed).  Returns (report, need_replicated)."""
        rep = CutReport()

        def _cb(_user, buf, count):
            try:
                reduce(np.ctypeslib.as_array(buf, shape=(int(count),)))
                return 0
            except Exception:  # an exception must not unwind through the C frames
                traceback.print_exc()
                return 1
        cb = _lib.ALLREDUCE_FN(_cb) if reduce is not None else C.cast(None, _lib.ALLREDUCE_FN)
        rc = self.L.flx_rank_and_cut_sharded_dev(self.h, n_total, d_mean_q_all, first, n_local, d_window_q, d_length, d_passed,
                                                 length_weight, mean_q_weight, window_q_weight,
                                                 1 if target_bases is not None else 0, int(target_bases or 0),
                                                 1 if keep_percent is not None else 0, float(keep_percent or 0.0),
                                                 int(total_bases), d_final_score, rank, world, cb, None, C.byref(rep))
        if rc == _lib.NEED_REPLICATED:
            return rep, True
        self._check(rc)
        return rep, False

    # ---- multi-GPU with the library's own RCCL communicator (include/filtlong_hip.h, flx_comm_*) -------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.L.flx_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._check(self.L.flx_comm_init(self.h, C.c_char_p(bytes(unique_id)), rank, world))

    def comm_destroy(self):
        self._check(self.L.flx_comm_destroy(self.h))

    def comm_sum_u64(self, values):
        a = np.ascontiguousarray(values, dtype=np.uint64).copy()
        self._check(self.L.flx_comm_sum_u64(self.h, a.ctypes.data, a.size))
        return a

    def rank_and_cut_comm(self, mean_q, window_q, length, passed, length_weight=1.0, mean_q_weight=1.0, window_q_weight=1.0,
                          target_bases=None, keep_percent=None, total_bases=0, want_scores=False):
        """flx_rank_and_cut_comm: this rank's reads2 scalars as host arrays (what the CLI holds); `total_bases` is the global
        sum.  Returns the same dictionary as rank_and_cut for the local block."""
        n = len(mean_q)
        mq = np.ascontiguousarray(mean_q, dtype=np.float64)
        wq = np.ascontiguousarray(window_q, dtype=np.float64)
        ln = np.ascontiguousarray(length, dtype=np.int32)
        ps = np.array(passed, dtype=np.uint8)
        fs = np.zeros(n, dtype=np.float64) if want_scores else None
        rep = CutReport()
        self._check(self.L.flx_rank_and_cut_comm(self.h, n, mq.ctypes.data, wq.ctypes.data, ln.ctypes.data, ps.ctypes.data,
                                                 length_weight, mean_q_weight, window_q_weight,
                                                 1 if target_bases is not None else 0, int(target_bases or 0),
                                                 1 if keep_percent is not None else 0, float(keep_percent or 0.0),
                                                 int(total_bases), fs.ctypes.data if want_scores else None, C.byref(rep)))
        return {"passed": ps, "final_score": fs, "report": rep}

    def rank_and_cut_comm_dev(self, n_local, d_mean_q, d_window_q, d_length, d_passed, length_weight=1.0, mean_q_weight=1.0,
                              window_q_weight=1.0, target_bases=None, keep_percent=None, total_bases=0, d_final_score=None):
        """flx_rank_and_cut_comm_dev: the global stage over all ranks of the context's RCCL communicator (one all-gather of
        the mean qualities + device-side all-reduces, no host round trips per pass); `total_bases` is the global sum."""
        rep = CutReport()
        self._check(self.L.flx_rank_and_cut_comm_dev(self.h, n_local, d_mean_q, d_window_q, d_length, d_passed, length_weight,
                                                     mean_q_weight, window_q_weight, 1 if target_bases is not None else 0,
                                                     int(target_bases or 0), 1 if keep_percent is not None else 0,
                                                     float(keep_percent or 0.0), int(total_bases), d_final_score, C.byref(rep)))
        return rep

    def last_phred_kernel(self):
        return self.L.flx_last_phred_kernel(self.h).decode()

    def last_kmer_cover(self):
        """"q", "w" or "v2": the coverage kernel the last k-mer scoring call launched."""
        return self.L.flx_last_kmer_cover(self.h).decode()

    def last_kmer_handed_over(self):
        """Reads of the last k-mer scoring call that went to the kernel with a diagonal per lane (insertions / deletions)."""
        return int(self.L.flx_last_kmer_handed_over(self.h))

    def last_kmer_locus(self):
        return bool(self.L.flx_last_kmer_locus(self.h))

    def last_kmer_fold_grid(self):
        return bool(self.L.flx_last_kmer_fold_grid(self.h))

    def synth_qual_dev(self, seed, d_plane, plane_bytes, d_offsets, d_lengths, d_read_ids, n, profile=0):
        self._check(self.L.flx_synth_qual_profile_dev(self.h, seed, profile, d_plane, plane_bytes, d_offsets, d_lengths,
                                                      d_read_ids, n))

    def synth_seq_dev(self, seed, d_plane, plane_bytes, d_offsets, d_lengths, d_read_ids, n, d_ref, ref_len, profile=0):
        self._check(self.L.flx_synth_seq_profile_dev(self.h, seed, int(profile), d_plane, plane_bytes, d_offsets, d_lengths, d_read_ids, n, d_ref,
                                                     ref_len))

    def score_kmer_dev(self, kmers, d_plane, plane_bytes, d_offsets, d_lengths, d_order, n, params, scores):
        """flx_score_batch_dev in k-mer mode; `scores` is a filled _lib.Scores with device pointers."""
        rc = self.L.flx_score_batch_dev(self.h, kmers.h, d_plane, plane_bytes, d_offsets, d_lengths, d_order, n,
                                        C.byref(params), C.byref(scores))
        if rc != 5:
            self._check(rc)
        return rc


class Kmers:
    """Reference 16-mer set on the device — mirrors the reference's Kmers (src/kmers.h:28-56)."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx.L.flx_kmerset_create(ctx.h, C.byref(h)))
        self.h = h
        self._final = False
        ctx._sets.append(weakref.ref(self))

    def _add(self, fn, seqs):
        seqs = [bytes(s) for s in seqs]
        n = len(seqs)
        lengths = np.array([len(s) for s in seqs], dtype=np.int64)
        offsets = np.zeros(max(n, 1), dtype=np.uint64)
        if n:
            offsets[1:n] = np.cumsum(lengths[:-1])
        bases = np.frombuffer(b"".join(seqs) or b"\0", dtype=np.uint8)
        self.ctx._check(fn(self.h, bases.ctypes.data, offsets.ctypes.data, lengths.ctypes.data, n))

    def add_assembly_fasta(self, seqs):
        """Kmers::add_assembly_fasta (src/kmers.cpp:61-72) on already-parsed contig sequences."""
        self._add(self.ctx.L.flx_kmerset_add_assembly, seqs)

    def add_read_fastqs(self, files_of_seqs):
        """Kmers::add_read_fastqs (src/kmers.cpp:50-58): one list of read sequences per file, -1 then -2."""
        for seqs in files_of_seqs:
            self._add(self.ctx.L.flx_kmerset_add_short_reads, seqs)

    def finalize(self):
        self.ctx._check(self.ctx.L.flx_kmerset_finalize(self.h))
        self._final = True

    def __len__(self):
        return int(self.ctx.L.flx_kmerset_size(self.h))

    def empty(self):
        return len(self) == 0

    def is_kmer_present(self, kmers):
        k = np.ascontiguousarray(kmers, dtype=np.uint32)
        out = np.zeros(len(k), dtype=np.uint8)
        self.ctx._check(self.ctx.L.flx_kmerset_contains(self.h, k.ctypes.data, len(k), out.ctypes.data))
        return out.astype(bool)

    def close(self):
        if self.h:
            if self.ctx.h:  # a set never outlives its context (Context.close() closes its sets first)
                self.ctx.L.flx_kmerset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
