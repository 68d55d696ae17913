"""ctypes loader for the C-ABI library (include/filtlong_hip.h).

There is no CPU fallback: if libfiltlong_hip.so is missing, or no gfx950 device is present when a
context is created, this fails loudly.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FLX_LIB_PATH") or os.path.join(HERE, "lib", "libfiltlong_hip.so")  # FLX_LIB_PATH: experiment builds
CSRC = os.path.join(HERE, "csrc")

FLX_OK = 0
STATUS_NAMES = {0: "FLX_OK", 1: "FLX_ERR_INVALID", 2: "FLX_ERR_HIP", 3: "FLX_ERR_NOMEM", 4: "FLX_ERR_STATE",
                5: "FLX_ERR_CAPACITY", 6: "FLX_ERR_NO_DEVICE"}

# every symbol include/filtlong_hip.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "flx_abi_version", "flx_version", "flx_device_count", "flx_ctx_create", "flx_ctx_destroy", "flx_last_error", "flx_ctx_set_stream",
    "flx_ctx_synchronize", "flx_ctx_device_info", "flx_timing_enable", "flx_timing_reset", "flx_timing_get",
    "flx_plane_layout", "flx_length_order", "flx_score_batch", "flx_score_batch_dev", "flx_reads2_gather", "flx_reads2_gather_dev",
    "flx_rank_and_cut",
    "flx_rank_and_cut_dev", "flx_rank_and_cut_sharded_dev", "flx_rank_and_cut_comm_dev", "flx_rank_and_cut_comm", "flx_comm_unique_id",
    "flx_pipeline_create", "flx_pipeline_reserve", "flx_pipeline_next_buffer", "flx_pipeline_submit", "flx_pipeline_finish", "flx_pipeline_destroy",
    "flx_comm_init", "flx_comm_destroy", "flx_comm_rank", "flx_comm_world", "flx_comm_sum_u64", "flx_kmerset_create", "flx_kmerset_destroy", "flx_kmerset_add_assembly",
    "flx_kmerset_add_short_reads", "flx_kmerset_finalize", "flx_kmerset_size", "flx_kmerset_contains",
    "flx_last_phred_kernel", "flx_last_kmer_locus", "flx_last_kmer_fold_grid", "flx_last_kmer_cover", "flx_last_kmer_handed_over", "flx_synth_qual_dev", "flx_synth_qual_profile_dev", "flx_synth_seq_dev", "flx_synth_seq_profile_dev",
]


# flx_allreduce_u64_fn: int (*)(void *user, uint64_t *buf, uint64_t count)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_uint64)
NEED_REPLICATED = 100  # flx_status FLX_NEED_REPLICATED


class Params(C.Structure):
    """flx_params — hot-path fields of the reference's Arguments (src/arguments.h:59-91)."""
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


class Scores(C.Structure):
    """flx_scores"""
    _fields_ = [
        ("mean_q", C.c_void_p), ("window_q", C.c_void_p), ("passed", C.c_void_p),
        ("first", C.c_void_p), ("last", C.c_void_p),
        ("child_offsets", C.c_void_p), ("child_ranges", C.c_void_p), ("child_mean_q", C.c_void_p),
        ("child_window_q", C.c_void_p), ("child_passed", C.c_void_p),
        ("child_capacity", C.c_uint64), ("n_children", C.c_uint64),
    ]


class CutReport(C.Structure):
    """flx_cut_report"""
    _fields_ = [
        ("target_bases", C.c_int64), ("kept_bases", C.c_int64), ("outcome", C.c_int32),
        ("exact_fallback", C.c_int32),
        ("mean_quality", C.c_double), ("stdev_quality", C.c_double), ("min_z", C.c_double), ("max_z", C.c_double),
        ("audited", C.c_uint64),
    ]


class FlxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(code, code), msg))
        self.code = code


def build(verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP runtime.  When torch shares the process (tests, bench.py use it for
    # device memory / torch.distributed), it must initialise first so that both sides talk to ONE runtime;
    # loading ours first leaves torch unable to see the GPU.  Pure C/C++ callers are unaffected.
    if os.environ.get("FLX_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u64, i64, i32, dbl = C.c_void_p, C.c_uint64, C.c_int64, C.c_int, C.c_double
    L.flx_abi_version.restype = i32
    L.flx_version.restype = C.c_char_p
    L.flx_ctx_create.argtypes = [i32, C.POINTER(vp)]
    L.flx_ctx_destroy.argtypes = [vp]
    L.flx_ctx_destroy.restype = None
    L.flx_last_error.argtypes = [vp]
    L.flx_last_error.restype = C.c_char_p
    L.flx_ctx_set_stream.argtypes = [vp, vp]
    L.flx_ctx_synchronize.argtypes = [vp]
    L.flx_ctx_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(i32), C.POINTER(u64)]
    L.flx_timing_enable.argtypes = [vp, i32]
    L.flx_timing_reset.argtypes = [vp]
    L.flx_timing_get.argtypes = [vp, C.c_char_p, C.POINTER(dbl), C.POINTER(u64)]
    L.flx_plane_layout.argtypes = [vp, u64, vp, C.POINTER(u64)]
    L.flx_length_order.argtypes = [vp, u64, vp]
    L.flx_score_batch.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, C.POINTER(Params), C.POINTER(Scores)]
    L.flx_score_batch_dev.argtypes = [vp, vp, vp, u64, vp, vp, vp, u64, C.POINTER(Params), C.POINTER(Scores)]
    L.flx_reads2_gather_dev.argtypes = [vp, u64, vp, C.POINTER(Scores), u64, vp, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.flx_reads2_gather.argtypes = [vp, u64, vp, C.POINTER(Scores), u64, vp, vp, vp, vp, vp, vp, C.POINTER(u64)]
    rank_args = [vp, u64, vp, vp, vp, vp, dbl, dbl, dbl, i32, i64, i32, dbl, i64, vp, C.POINTER(CutReport)]
    L.flx_rank_and_cut.argtypes = rank_args
    L.flx_rank_and_cut_dev.argtypes = rank_args
    L.flx_rank_and_cut_sharded_dev.argtypes = [vp, u64, vp, u64, u64, vp, vp, vp, dbl, dbl, dbl, i32, C.c_int64, i32, dbl,
                                               C.c_int64, vp, i32, i32, ALLREDUCE_FN, vp, C.POINTER(CutReport)]
    L.flx_rank_and_cut_comm_dev.argtypes = rank_args
    L.flx_rank_and_cut_comm.argtypes = rank_args
    L.flx_pipeline_create.argtypes = [vp, vp, C.POINTER(Params), u64, u64, C.POINTER(vp)]
    L.flx_pipeline_reserve.argtypes = [vp, u64, u64]
    L.flx_pipeline_next_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), C.POINTER(u64)]
    L.flx_pipeline_submit.argtypes = [vp, u64, vp, vp, u64]
    L.flx_pipeline_finish.argtypes = [vp, C.POINTER(Scores), C.POINTER(u64)]
    L.flx_pipeline_destroy.argtypes = [vp]
    L.flx_pipeline_destroy.restype = None
    L.flx_comm_unique_id.argtypes = [vp, vp]
    L.flx_comm_init.argtypes = [vp, vp, i32, i32]
    L.flx_comm_destroy.argtypes = [vp]
    L.flx_comm_rank.argtypes = [vp]
    L.flx_comm_world.argtypes = [vp]
    L.flx_comm_sum_u64.argtypes = [vp, vp, u64]
    L.flx_kmerset_create.argtypes = [vp, C.POINTER(vp)]
    L.flx_kmerset_destroy.argtypes = [vp]
    L.flx_kmerset_destroy.restype = None
    L.flx_kmerset_add_assembly.argtypes = [vp, vp, vp, vp, u64]
    L.flx_kmerset_add_short_reads.argtypes = [vp, vp, vp, vp, u64]
    L.flx_kmerset_finalize.argtypes = [vp]
    L.flx_kmerset_size.argtypes = [vp]
    L.flx_kmerset_size.restype = u64
    L.flx_kmerset_contains.argtypes = [vp, vp, u64, vp]
    L.flx_last_phred_kernel.argtypes = [vp]
    L.flx_last_phred_kernel.restype = C.c_char_p
    L.flx_last_kmer_locus.argtypes = [vp]
    L.flx_last_kmer_locus.restype = i32
    L.flx_last_kmer_fold_grid.argtypes = [vp]
    L.flx_last_kmer_fold_grid.restype = i32
    L.flx_last_kmer_cover.argtypes = [vp]
    L.flx_last_kmer_cover.restype = C.c_char_p
    L.flx_last_kmer_handed_over.argtypes = [vp]
    L.flx_last_kmer_handed_over.restype = i64
    L.flx_synth_qual_dev.argtypes = [vp, u64, vp, u64, vp, vp, vp, u64]
    L.flx_synth_qual_profile_dev.argtypes = [vp, u64, C.c_int, vp, u64, vp, vp, vp, u64]
    L.flx_synth_seq_dev.argtypes = [vp, u64, vp, u64, vp, vp, vp, u64, vp, u64]
    L.flx_synth_seq_profile_dev.argtypes = [vp, u64, C.c_int, vp, u64, vp, vp, vp, u64, vp, u64]
    _lib = L
    return L
