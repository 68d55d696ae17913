/* filtlong_hip.h — C ABI of the MI355X-native Filtlong scoring hot path.
 *
 * This is the drop-in boundary: a C-ABI shared library (libfiltlong_hip.so, gfx950) whose entry
 * points replace, in batched form, the in-process seams of the reference (rrwick/Filtlong v0.3.1;
 * paths below are relative to the reference root):
 *
 *   seam 1  reference 16-mer set build      Kmers::Kmers / add_assembly_fasta / add_read_fastqs /
 *                                           is_kmer_present / empty        src/kmers.h:31-44
 *   seam 2  per-read scoring                Read::Read(name, seq, qscores, length, Kmers*, Arguments*)
 *                                           and its public result fields   src/read.h:32-56
 *   seam 3  global rank + cut               the statistics / normalise / set_final_score /
 *                                           std::sort / cut walk inlined in main()
 *                                                                          src/main.cpp:169-261
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross the boundary;
 *   - every function returns an int status (FLX_OK == 0); no exceptions cross the boundary;
 *     flx_last_error(ctx) gives the message (the CLI prints it as "Error: ..." and exits 1,
 *     mirroring src/main.cpp:80-116);
 *   - one flx_ctx per process / rank / GPU; not thread-safe; calls are synchronous on return
 *     unless the name ends in _async;
 *   - `*_dev` variants take DEVICE pointers (HBM-resident data, e.g. a torch tensor's data_ptr())
 *     and run on the context's stream; the plain variants take HOST pointers and stage through
 *     the context's own device buffers;
 *   - there is NO CPU fallback: if the HIP runtime or a gfx950 device is missing,
 *     flx_ctx_create fails.
 */
#ifndef FILTLONG_HIP_H
#define FILTLONG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): + flx_device_count, flx_reads2_gather(_dev) (added in round 3 under version 1), flx_last_kmer_locus; finalize of a
 * set built from an assembly also keeps the assembly as a text + seed table (0.5 B per base + 10 B per distinct 16-mer of device
 * memory beside the 512 MiB bitmap, 1 GiB pair table and 2 + 2 MiB prefilters); when that memory cannot be had the set works without */
/* 3 (round 5): + flx_last_kmer_fold_grid */
/* 4 (round 6): + flx_last_kmer_cover, flx_last_kmer_handed_over, flx_synth_seq_profile_dev */
#define FLX_ABI_VERSION 4

enum flx_status {
    FLX_OK = 0,
    FLX_ERR_INVALID = 1,     /* bad argument                                  */
    FLX_ERR_HIP = 2,         /* HIP runtime error (message has the hipError)   */
    FLX_ERR_NOMEM = 3,       /* host or device allocation failed              */
    FLX_ERR_STATE = 4,       /* call order violated (e.g. set not finalized)  */
    FLX_ERR_CAPACITY = 5,    /* caller-provided output capacity too small     */
    FLX_ERR_NO_DEVICE = 6,   /* no usable gfx950 device                       */
    FLX_NEED_REPLICATED = 100 /* flx_rank_and_cut_sharded_dev only: not an error — nothing was modified; gather all
                                records and call flx_rank_and_cut_dev instead (NaN scores / ties straddling the cut) */
};

typedef struct flx_ctx flx_ctx;
typedef struct flx_kmerset flx_kmerset;

/* Hot-path parameters: the fields of the reference's Arguments that Read::Read consults
 * (src/arguments.h:59-91; read in src/read.cpp:61-73,86-117).  *_set mirrors the reference's
 * paired bools. */
typedef struct flx_params {
    int32_t window_size;                  /* --window_size, default 250 (src/arguments.cpp:205)   */
    int32_t min_length_set, min_length;   /* --min_length                                          */
    int32_t max_length_set, max_length;   /* --max_length                                          */
    int32_t min_mean_q_set;               /* --min_mean_q                                          */
    int32_t min_window_q_set;             /* --min_window_q                                        */
    double min_mean_q;
    double min_window_q;
    int32_t trim;                         /* --trim                                                */
    int32_t split_set, split;             /* --split                                               */
    int32_t _pad;
} flx_params;

/* ------------------------------------------------------------------------------------------
 * context
 * ---------------------------------------------------------------------------------------- */
int flx_abi_version(void);
const char *flx_version(void);
/* Number of HIP devices visible to this process (-1: no usable runtime).  Initialises the HIP runtime in the calling process:
 * a host that forks its ranks (the CLI's --gpus N) asks from a short-lived probe child. */
int flx_device_count(void);
int flx_ctx_create(int device_ordinal, flx_ctx **out);
void flx_ctx_destroy(flx_ctx *ctx);
const char *flx_last_error(const flx_ctx *ctx); /* ctx may be NULL: last create error */
/* Run all subsequent work of this context on the caller's hipStream_t (NULL = context's own). */
int flx_ctx_set_stream(flx_ctx *ctx, void *hip_stream);
int flx_ctx_synchronize(flx_ctx *ctx);
/* Device properties the host side reports (name is copied, NUL-terminated). */
int flx_ctx_device_info(const flx_ctx *ctx, char *name, size_t name_cap, int *n_cu, uint64_t *hbm_bytes);

/* Per-kernel timing with HIP events on the context's stream (for bench.py's roofline line).
 * While enabled, every launch of a hot kernel is bracketed by events; flx_timing_get drains them
 * (synchronising) and returns total milliseconds and launch count for kernels whose name starts
 * with `prefix` ("" = all). */
/* Name of the Phred scoring kernel the last Phred-mode scoring call launched (the library picks by window size and data:
 * "flx_score_phred_regs", "flx_score_phred_regs_private", "flx_score_phred_ring" or "flx_score_phred_direct"). */
const char *flx_last_phred_kernel(const flx_ctx *ctx);
/* 1 when the last k-mer-mode scoring call confirmed members along the reads' loci in the assembly text (sets built from an
 * assembly; FLX_KMER_LOCUS=0 switches it off), 0 otherwise.  Results are identical either way. */
int flx_last_kmer_locus(const flx_ctx *ctx);
/* 1 when the window recurrence of the last k-mer-mode scoring call ran on the integer grid (exact; window sizes whose step 1 / ws
 * rounds alike on the binades from 2^-3 to 2 — the default 250 among them; FLX_KMER_FOLD_GRID=0 switches it off), 0 when every step
 * was taken in floating point.  Results are identical either way. */
int flx_last_kmer_fold_grid(const flx_ctx *ctx);
/* Which coverage kernel the last k-mer-mode scoring call launched: "q" (phases with a queue in LDS between them, the default for a
 * set with a text), "q2" (FLX_KMER_COVER=q2: the same with every read in its second launch, tests), "w" (the wave-level kernel of rounds
 * 3-5: sets without a text, FLX_KMER_COVER=w) or "v2" (FLX_KMER_COVER=v2, sets without the pair table).  Results are identical whichever runs. */
const char *flx_last_kmer_cover(const flx_ctx *ctx);
/* How many reads of the last k-mer-mode scoring call the first coverage kernel handed to the one in which every lane follows a
 * diagonal of its own (reads with insertions / deletions; kernel "q" only, 0 otherwise; valid until the next scoring call of the
 * context; a device-to-host copy of one byte per read: for tests and diagnostics).  -1 on error. */
int64_t flx_last_kmer_handed_over(flx_ctx *ctx);
int flx_timing_enable(flx_ctx *ctx, int on);
int flx_timing_reset(flx_ctx *ctx);
int flx_timing_get(flx_ctx *ctx, const char *prefix, double *total_ms, uint64_t *launches);

/* ------------------------------------------------------------------------------------------
 * seam 2 — per-read scoring      (replaces Read::Read, src/read.cpp:25-144)
 *
 * Packed input ("read plane"): one byte plane + per-read start offsets + lengths.
 *   - Phred mode (set == NULL or empty): plane holds the QUALITY strings (the reference never
 *     touches seq in this mode, src/read.cpp:35-39);
 *   - k-mer mode: plane holds the SEQUENCE strings (qual is never touched, src/read.cpp:43-58).
 *   - offsets[i] must be a multiple of 16 and plane_bytes a multiple of 16 covering
 *     offsets[i] + round_up(lengths[i], 16) for every read (flx_plane_layout computes both).
 *   - `order` (optional, may be NULL) is a permutation of [0, n): the processing order; pass the
 *     reads sorted by DESCENDING length (flx_length_order) so that the 64 reads sharing a wavefront
 *     finish together.  Results are always written in input order (index i = read i).
 *
 * Outputs per read i: mean_q, window_q (the 0-100 pre-normalisation values of m_mean_quality /
 * m_window_quality), passed (m_passed after the hard cut-offs, src/read.cpp:64-73), and in k-mer
 * mode first/last (m_first_base_in_kmer / m_last_base_in_kmer, src/read.cpp:75-84) plus the child
 * reads produced by --trim/--split (src/read.cpp:86-141) in CSR form: children of read i are
 * child_offsets[i] .. child_offsets[i+1]-1, each with its half-open 0-based (start,end) range and
 * its own mean_q / window_q / passed.
 * ---------------------------------------------------------------------------------------- */
typedef struct flx_scores {
    double *mean_q;          /* [n]                                   */
    double *window_q;        /* [n]                                   */
    uint8_t *passed;         /* [n]                                   */
    int32_t *first;          /* [n] or NULL (k-mer mode only)         */
    int32_t *last;           /* [n] or NULL                           */
    uint64_t *child_offsets; /* [n+1] or NULL (trim/split only)       */
    int32_t *child_ranges;   /* [2*child_capacity] (start,end) pairs  */
    double *child_mean_q;    /* [child_capacity]                      */
    double *child_window_q;  /* [child_capacity]                      */
    uint8_t *child_passed;   /* [child_capacity]                      */
    uint64_t child_capacity; /* in: capacity of the child arrays      */
    uint64_t n_children;     /* out: total children written           */
} flx_scores;

/* Layout helper: offsets[i] (16-byte aligned starts) and the padded plane size for given lengths. */
int flx_plane_layout(const int32_t *lengths, uint64_t n_reads, uint64_t *offsets, uint64_t *plane_bytes);
/* Processing order: indices sorted by descending length (stable). Host arrays. */
int flx_length_order(const int32_t *lengths, uint64_t n_reads, uint32_t *order);

int flx_score_batch(flx_ctx *ctx, const flx_kmerset *set, const uint8_t *plane, uint64_t plane_bytes,
                    const uint64_t *offsets, const int32_t *lengths, const uint32_t *order, uint64_t n_reads,
                    const flx_params *params, flx_scores *out);

/* Device-resident variant: every pointer (including those inside `out`) is a device pointer. */
int flx_score_batch_dev(flx_ctx *ctx, const flx_kmerset *set, const void *d_plane, uint64_t plane_bytes,
                        const void *d_offsets, const void *d_lengths, const void *d_order, uint64_t n_reads,
                        const flx_params *params, flx_scores *out_dev);

/* ------------------------------------------------------------------------------------------
 * between seam 2 and seam 3 — the reads2 gather     (replaces src/main.cpp:138-147)
 *
 * reads2 = file order with every trimmed / split parent replaced IN PLACE by its children.  Gathers what the global
 * stage reads (mean quality, window quality, length, pass flag) from the per-read and per-child outputs of
 * flx_score_batch* into reads2 order: entry j is either read parent2[j] itself (child2[j] == -1) or its child number
 * child2[j] (index into the child arrays; length = end - start of its range, src/read.cpp:131-137).  Without children
 * (Phred mode, or no --trim / --split: scores->child_offsets NULL or n_children 0) reads2 is the reads themselves.
 * `capacity` = room in the output arrays (n_reads + n_children always suffices); *n2 = number of entries
 * (FLX_ERR_CAPACITY: *n2 is the capacity needed).  parent2 (uint32) and child2 (int64) may be NULL.
 * _dev: every pointer (also those inside `scores`) is a device pointer; one scan + one scatter kernel on the
 * context's stream.
 * ---------------------------------------------------------------------------------------- */
int flx_reads2_gather_dev(flx_ctx *ctx, uint64_t n_reads, const void *d_lengths, const flx_scores *scores_dev,
                          uint64_t capacity, void *d_mean_q2, void *d_window_q2, void *d_length2, void *d_passed2,
                          void *d_parent2, void *d_child2, uint64_t *n2);
int flx_reads2_gather(flx_ctx *ctx, uint64_t n_reads, const int32_t *lengths, const flx_scores *scores, uint64_t capacity,
                      double *mean_q2, double *window_q2, int32_t *length2, uint8_t *passed2, uint32_t *parent2,
                      int64_t *child2, uint64_t *n2);

/* ------------------------------------------------------------------------------------------
 * seam 3 — global rank + cut     (replaces src/main.cpp:169-261 + Read::set_final_score,
 *                                 src/read.cpp:249-267)
 *
 * Arrays are in "reads2" order: file order with each trimmed/split parent replaced in place by
 * its children (src/main.cpp:138-147).  `passed` is in/out: on return it holds the final pass
 * flags after the --target_bases / --keep_percent cut.  total_bases counts ORIGINAL read lengths
 * (src/main.cpp:89).  final_score (optional) receives the device-computed scores (the reference
 * prints them only under --verbose, 2 decimals); the pass set itself is exact (see DESIGN.md,
 * "boundary audit").
 * ---------------------------------------------------------------------------------------- */
enum flx_cut_outcome {
    FLX_CUT_NONE = 0,          /* neither --target_bases nor --keep_percent given             */
    FLX_CUT_NOT_ENOUGH = 1,    /* "not enough reads to reach target"          main.cpp:239-240 */
    FLX_CUT_ALREADY_BELOW = 2, /* "reads already fall below target after filtering"   242-243 */
    FLX_CUT_SORTED = 3         /* sorted and cut; kept_bases is "keeping N bp"        247-258 */
};

typedef struct flx_cut_report {
    int64_t target_bases;
    int64_t kept_bases;
    int32_t outcome;
    int32_t exact_fallback; /* 1 if a tie group straddled the cut and the host std::sort path decided */
    double mean_quality, stdev_quality, min_z, max_z; /* main.cpp:170-196 */
    uint64_t audited;       /* reads whose score was re-derived with the host libm at the boundary */
} flx_cut_report;

int flx_rank_and_cut(flx_ctx *ctx, uint64_t n, const double *mean_q, const double *window_q, const int32_t *length,
                     uint8_t *passed, double length_weight, double mean_q_weight, double window_q_weight,
                     int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                     int64_t total_bases, double *final_score, flx_cut_report *report);

int flx_rank_and_cut_dev(flx_ctx *ctx, uint64_t n, const void *d_mean_q, const void *d_window_q,
                         const void *d_length, void *d_passed, double length_weight, double mean_q_weight,
                         double window_q_weight, int target_bases_set, int64_t target_bases, int keep_percent_set,
                         double keep_percent, int64_t total_bases, void *d_final_score, flx_cut_report *report);

/* The same stage with the reads2 entries sharded over `world` ranks (one process per GPU).  The
 * statistics of src/main.cpp:170-196 are order-dependent folds over ALL mean qualities, so d_mean_q_all holds every
 * rank's mean qualities in global reads2 order (the one array that has to be all-gathered: 8 bytes per entry).
 * Everything else stays local: d_window_q / d_length / d_passed / d_final_score describe this rank's entries
 * [first, first + n_local) only.  The cut (src/main.cpp:247-257) is a weighted selection: per key byte a 256-bin
 * histogram of summed read lengths, whose global value is the SUM over ranks — the only exchange the selection
 * needs.  `reduce(user, buf, count)` must replace buf[0..count) (host memory) by its element-wise sum over all
 * ranks and return 0; it is called the same number of times with the same counts on every rank (about a dozen
 * calls of <= 2 KB + one of 40 bytes per boundary-audit candidate).  NULL is allowed when world == 1.
 * Returns FLX_OK with this rank's final pass flags in d_passed (report fields are global), or
 * FLX_NEED_REPLICATED — on every rank alike, before anything was modified — when the exact outcome needs the
 * reference's own std::sort over all records (NaN scores, equal scores straddling the cut, or FLX_RANK_SORT=1):
 * gather the records and call flx_rank_and_cut_dev. */
typedef int (*flx_allreduce_u64_fn)(void *user, uint64_t *buf, uint64_t count);

int flx_rank_and_cut_sharded_dev(flx_ctx *ctx, uint64_t n_total, const void *d_mean_q_all, uint64_t first,
                                 uint64_t n_local, const void *d_window_q, const void *d_length, void *d_passed,
                                 double length_weight, double mean_q_weight, double window_q_weight,
                                 int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                                 int64_t total_bases, void *d_final_score, int rank, int world,
                                 flx_allreduce_u64_fn reduce, void *user, flx_cut_report *report);

/* ------------------------------------------------------------------------------------------
 * seam 2, streaming — replaces the pass-1 loop of src/main.cpp:70-127 for inputs larger than memory
 *
 * The caller parses its input chunk by chunk.  For every chunk:
 *     flx_pipeline_next_buffer -> a PINNED staging buffer of `capacity_bytes` (blocks while both slots are in flight)
 *     pack the chunk's reads into it (flx_plane_layout gives offsets and the byte count; Phred mode: quality strings,
 *     k-mer mode: sequences — as for flx_score_batch)
 *     flx_pipeline_submit      -> starts the H2D copy and returns; a worker thread scores the chunk when it has landed
 * so chunk k is copied and scored on the GPU while the host packs chunk k+1 (two slots).  flx_pipeline_finish waits for
 * everything and returns the per-read results of ALL submitted reads in submission order as host arrays owned by the
 * pipeline (children in one global CSR); only these scalars survive between the chunks, like the reference, which keeps
 * one Read object per record and drops the record's text.  The context must not be used for anything else between
 * create and finish (its stream belongs to the worker).
 * ---------------------------------------------------------------------------------------- */
typedef struct flx_pipeline flx_pipeline;
int flx_pipeline_create(flx_ctx *ctx, const flx_kmerset *set /* NULL or empty: Phred mode */, const flx_params *params,
                        uint64_t chunk_plane_bytes, uint64_t chunk_reads, flx_pipeline **out);
int flx_pipeline_next_buffer(flx_pipeline *p, uint8_t **plane, uint64_t *capacity_bytes, uint64_t *capacity_reads);
/* Grow both slots to at least this capacity (never shrinks; waits for the chunks in flight first).  For callers that learn
 * the longest read only while streaming.  Not between next_buffer and submit. */
int flx_pipeline_reserve(flx_pipeline *p, uint64_t chunk_plane_bytes, uint64_t chunk_reads);
int flx_pipeline_submit(flx_pipeline *p, uint64_t plane_bytes, const uint64_t *offsets, const int32_t *lengths, uint64_t n_reads);
int flx_pipeline_finish(flx_pipeline *p, flx_scores *all, uint64_t *n_reads);
void flx_pipeline_destroy(flx_pipeline *p);

/* Multi-GPU without a host framework: one process per GPU, each with one context; the library owns the RCCL communicator
 * (loaded with dlopen on first use) and does the exchange of the global stage itself, on device buffers on the context's
 * stream (no host synchronisation inside the 8 selection passes):
 *     rank 0: flx_comm_unique_id(ctx, id) -> hand the 128 bytes to every rank (file, socket, launcher, ...)
 *     all:    flx_comm_init(ctx, id, rank, world)
 *     all:    flx_score_batch_dev(...) on the rank's own reads (contiguous block of file order, sharded by count)
 *     all:    flx_rank_and_cut_comm_dev(...)  = ONE all-gather of the mean qualities over xGMI + the sharded stage above;
 *             the rare NaN / tie cases that need the reference's std::sort over all reads gather the remaining fields and
 *             run the replicated stage internally.  `total_bases` is the GLOBAL sum (flx_comm_sum_u64 helps).
 * Without a communicator flx_rank_and_cut_comm_dev is flx_rank_and_cut_dev.  The reference has no counterpart (single
 * process, src/main.cpp:37-321). */
#define FLX_COMM_ID_BYTES 128
int flx_comm_unique_id(flx_ctx *ctx, void *id_out /* FLX_COMM_ID_BYTES */);
int flx_comm_init(flx_ctx *ctx, const void *id, int rank, int world);
int flx_comm_destroy(flx_ctx *ctx);
int flx_comm_rank(const flx_ctx *ctx);
int flx_comm_world(const flx_ctx *ctx);
int flx_comm_sum_u64(flx_ctx *ctx, uint64_t *host_buf, uint64_t count); /* element-wise sum over all ranks, in place */
/* host-array variant (the arrays are staged through the device, like flx_rank_and_cut) */
int flx_rank_and_cut_comm(flx_ctx *ctx, uint64_t n_local, const double *mean_q, const double *window_q, const int32_t *length,
                          uint8_t *passed, double length_weight, double mean_q_weight, double window_q_weight,
                          int target_bases_set, int64_t target_bases, int keep_percent_set, double keep_percent,
                          int64_t total_bases, double *final_score, flx_cut_report *report);
int flx_rank_and_cut_comm_dev(flx_ctx *ctx, uint64_t n_local, const void *d_mean_q, const void *d_window_q,
                              const void *d_length, void *d_passed, double length_weight, double mean_q_weight,
                              double window_q_weight, int target_bases_set, int64_t target_bases, int keep_percent_set,
                              double keep_percent, int64_t total_bases, void *d_final_score, flx_cut_report *report);

/* ------------------------------------------------------------------------------------------
 * seam 1 — reference 16-mer set   (replaces Kmers, src/kmers.cpp:28-172 + src/bloom_filter.h)
 *
 * Sequences are handed over packed: bases[offsets[i] .. offsets[i]+lengths[i]).  Call order for
 * short reads must be the reference's: all of -1, then all of -2 (src/kmers.cpp:54-55); the
 * ">= 4 copies, or 3 with a Bloom false positive" rule (src/kmers.cpp:142-166) is order
 * dependent only through the Bloom filter, which is restated bit-exactly.
 * After flx_kmerset_finalize the set is immutable.  An empty finalized set selects Phred mode,
 * like Kmers::empty() (src/kmers.h:34, src/read.cpp:35).
 * ---------------------------------------------------------------------------------------- */
int flx_kmerset_create(flx_ctx *ctx, flx_kmerset **out);
void flx_kmerset_destroy(flx_kmerset *set);
int flx_kmerset_add_assembly(flx_kmerset *set, const uint8_t *bases, const uint64_t *offsets,
                             const int64_t *lengths, uint64_t n_seqs);
int flx_kmerset_add_short_reads(flx_kmerset *set, const uint8_t *bases, const uint64_t *offsets,
                                const int64_t *lengths, uint64_t n_seqs);
int flx_kmerset_finalize(flx_kmerset *set);
uint64_t flx_kmerset_size(const flx_kmerset *set);
int flx_kmerset_contains(const flx_kmerset *set, const uint32_t *kmers, uint64_t n, uint8_t *present);

/* ------------------------------------------------------------------------------------------
 * bench / test support: deterministic synthetic Phred planes generated directly in HBM
 * (same integer hash as filtlong_amd/synth.py and oracle/synth.h).
 * d_read_ids[i] is the global index of read i in the synthetic population.
 * ---------------------------------------------------------------------------------------- */
int flx_synth_qual_dev(flx_ctx *ctx, uint64_t seed, void *d_plane, uint64_t plane_bytes, const void *d_offsets,
                       const void *d_lengths, const void *d_read_ids, uint64_t n_reads);
/* the same with a quality profile: 0 = the SURVEY §8(d) definition (what flx_synth_qual_dev writes), 1 = "wide": per-read
 * centre Q3..Q44, per-base jitter +-10, q <= 50 (a realistic spread; bench.py's second Phred line) */
int flx_synth_qual_profile_dev(flx_ctx *ctx, uint64_t seed, int profile, void *d_plane, uint64_t plane_bytes,
                               const void *d_offsets, const void *d_lengths, const void *d_read_ids, uint64_t n_reads);

/* k-mer configurations: reads drawn from a device-resident reference genome (d_ref, ASCII ACGT) with per-read
 * substitution rates and junk blocks. */
int flx_synth_seq_dev(flx_ctx *ctx, uint64_t seed, void *d_plane, uint64_t plane_bytes, const void *d_offsets,
                      const void *d_lengths, const void *d_read_ids, uint64_t n_reads, const void *d_ref,
                      uint64_t ref_len);
/* the same with a profile: 0 = the SURVEY §8(d) definition (what flx_synth_seq_dev writes: substitutions only), 1 = "indels": the same
 * per-read error rate, a third of the errors insertions and a third deletions of 1-3 bases, 2 = "unrelated": 30 % of the reads are
 * random bases with no relation to the reference (oracle/synth.h: flx_synth_seq_read; bench.py's extras.c3_indels / .c3_unrelated) */
int flx_synth_seq_profile_dev(flx_ctx *ctx, uint64_t seed, int profile, void *d_plane, uint64_t plane_bytes, const void *d_offsets,
                              const void *d_lengths, const void *d_read_ids, uint64_t n_reads, const void *d_ref,
                              uint64_t ref_len);

#ifdef __cplusplus
}
#endif
#endif /* FILTLONG_HIP_H */
