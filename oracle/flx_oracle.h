/* flx_oracle.h — C interface of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 * See flx_oracle.cpp for the parity status and the reference file:line of each function.
 * Loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
 */
#ifndef FLX_ORACLE_H
#define FLX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirrors the hot-path fields of the reference's Arguments (src/arguments.h:59-91) */
typedef struct flo_params {
    int32_t window_size;
    int32_t min_length_set, min_length;
    int32_t max_length_set, max_length;
    int32_t min_mean_q_set;
    int32_t min_window_q_set;
    double min_mean_q;
    double min_window_q;
    int32_t trim;
    int32_t split_set, split;
    int32_t _pad;
} flo_params;

/* public result fields of the reference's Read (src/read.h:40-56) */
typedef struct flo_read_result {
    double mean_q;
    double window_q;
    double length_score;
    int32_t length;
    int32_t passed;
    int32_t first; /* m_first_base_in_kmer */
    int32_t last;  /* m_last_base_in_kmer  */
    int32_t n_bad;
    int32_t n_child;
} flo_read_result;

enum { FLO_CUT_NONE = 0, FLO_CUT_NOT_ENOUGH = 1, FLO_CUT_ALREADY_BELOW = 2, FLO_CUT_SORTED = 3 };

typedef struct flo_cut_report {
    int64_t target_bases;
    int64_t kept_bases;
    int32_t outcome;
    int32_t _pad;
    double mean_quality, stdev_quality, min_z, max_z;
} flo_cut_report;

typedef struct flo_kmerset flo_kmerset;

double flo_qscore_to_quality(int c_signed_char);
void flo_phred_lut(double *lut256);
double flo_mean_quality(const double *q, uint64_t n);
double flo_window_quality(const double *q, uint64_t n, uint64_t ws);
double flo_length_score(int length);

uint32_t flo_base_fwd(int base);
uint32_t flo_base_rev(int base);
uint32_t flo_start_kmer_fwd(const char *s);
uint32_t flo_start_kmer_rev(const char *s);

void flo_bloom_parameters(uint64_t n_projected, double fp_prob, uint32_t *n_hashes, uint64_t *table_bits);
void flo_bloom_salts(uint64_t user_seed, uint32_t n_hashes, uint32_t *salts);
uint32_t flo_bloom_hash(uint32_t key, uint32_t salt);

flo_kmerset *flo_kmerset_new(void);
void flo_kmerset_free(flo_kmerset *s);
uint64_t flo_kmerset_size(const flo_kmerset *s);
int flo_kmerset_contains(const flo_kmerset *s, uint32_t kmer);
uint64_t flo_kmerset_dump(const flo_kmerset *s, uint32_t *out, uint64_t cap);
void flo_kmerset_add_sequence(flo_kmerset *s, const char *seq, uint64_t len, int multi_copy);

int flo_score_read(const flo_kmerset *set, const char *seq, const char *qual, int length, const flo_params *p,
                   flo_read_result *out, int32_t *bad_ranges, int32_t *child_ranges, flo_read_result *children,
                   int cap);

/* flo_score_read over a packed batch on n_threads host threads (kmer_plane: the plane holds sequences, else quality strings);
 * children as a CSR in read order; returns the number of children or -1 */
int64_t flo_score_plane_mt(const flo_kmerset *set, int kmer_plane, const uint8_t *plane, const uint64_t *offsets,
                           const int32_t *lengths, uint64_t n, const flo_params *p, int n_threads, double *mean_q,
                           double *window_q, uint8_t *passed, int32_t *first, int32_t *last, uint64_t *child_offsets,
                           uint64_t child_cap, int32_t *child_ranges, double *child_mean_q, double *child_window_q,
                           uint8_t *child_passed);

/* a19b: reads2 = file order with every parent replaced in place by its children (src/main.cpp:138-147).
 * child_offsets[n+1] is the CSR of the children (children of read i: child_offsets[i] .. child_offsets[i+1]-1, each with
 * its (start,end) range).  Outputs hold n - #parents-with-children + #children entries; returns that count. */
uint64_t flo_reads2_gather(uint64_t n, const int32_t *length, const double *mean_q, const double *window_q,
                           const uint8_t *passed, const uint64_t *child_offsets, const int32_t *child_ranges,
                           const double *child_mean_q, const double *child_window_q, const uint8_t *child_passed,
                           double *mean_q2, double *window_q2, int32_t *length2, uint8_t *passed2, uint32_t *parent2,
                           int64_t *child2);

double flo_final_score(double length_score, double mean_q, double window_q, double lw, double mw, double ww);

int flo_rank_and_cut(uint64_t n, double *mean_q, double *window_q, const int32_t *length, uint8_t *passed, double lw,
                     double mw, double ww, int target_bases_set, int64_t target_bases, int keep_percent_set,
                     double keep_percent, int64_t total_bases, double *final_score, flo_cut_report *rep);

uint64_t flo_synth_mix(uint64_t seed, uint64_t stream, uint64_t read, uint64_t pos);
void flo_synth_qual(uint64_t seed, uint64_t read, uint64_t length, uint8_t *out);
void flo_synth_bases(uint64_t seed, uint64_t stream, uint64_t read, uint64_t start, uint64_t length, uint8_t *out);
/* k-mer-mode long read of profile 0 / 1 / 2 (synth.h: flx_synth_seq_read) */
void flo_synth_seq(uint64_t seed, int profile, uint64_t read, int length, const uint8_t *ref, uint64_t ref_len, uint8_t *out);

int flo_bench_phred(uint64_t n, uint64_t seed, uint64_t first_read, int64_t target_bases, double *score_s,
                    double *rank_s, int64_t *total_bases_out, int64_t *kept_out);

#ifdef __cplusplus
}
#endif
#endif
