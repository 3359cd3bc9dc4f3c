/* oracle/fq_oracle.h -- TEST INFRASTRUCTURE ONLY.  Never include from the product.
 *
 * Plain-C, single-threaded restatement of the FASTQuick `align` hot path
 * used as the parity checker for the HIP implementation.  Pinned against the real reference
 * (oracle/_ref/fq_ref_driver, built from /root/reference) through tests/golden fixtures.
 */
#ifndef FQ_ORACLE_H
#define FQ_ORACLE_H
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fqo_index fqo_index;
typedef struct fqo_ctx fqo_ctx;

/* option block: gap_opt_t defaults libbwa/bwtaln.c:24-48, pe_opt_t defaults libbwa/bwape.c:7-20 */
typedef struct {
  int s_mm, s_gapo, s_gape;
  int mode;
  int indel_end_skip, max_del_occ, max_entries;
  double fnr;
  int max_diff, max_gapo, max_gape;
  int max_seed_diff, seed_len;
  int max_top2;
  int trim_qual;
  int filter_thresh;      /* RollParam.thresh, src/BwtIndexer.cpp:558 */
  /* pe */
  int max_isize, force_isize;
  uint32_t max_occ;
  int n_multi, N_multi;
  int is_sw;
  double ap_prior;
} fqo_opts;

void fqo_default_opts(fqo_opts *o);

/* prefix = path of the reduced reference FASTA (<x>.FASTQuick.fa); bitmaps come from
 * <prefix>.rollhash.sparse, else <prefix>.rollhash, else are rebuilt from the FASTA. */
fqo_index *fqo_index_load(const char *prefix);
void fqo_index_free(fqo_index *ix);
int64_t fqo_index_lpac(const fqo_index *ix);
/* digest helpers for index-builder parity tests */
uint64_t fqo_bitmap_popcount(const fqo_index *ix, int table);
uint64_t fqo_bitmap_fnv(const fqo_index *ix, int table);
/* rebuild bitmaps from the FASTA into a fresh set and return per-table fnv/popcount (src/BwtIndexer.cpp:611-713) */
int fqo_bitmaps_from_fasta(const char *fasta, uint64_t popc[6], uint64_t fnv[6]);

fqo_ctx *fqo_ctx_create(const fqo_index *ix, const fqo_opts *o);
void fqo_ctx_free(fqo_ctx *c);

/* One batch of n pairs.  seq/qual: ASCII, row i of end e at base + (e*n + i)*stride; lens[e*n+i];
 * names: n rows of name_stride bytes, NUL terminated; names_mate: the second mates' names when they differ, else NULL.  Writes the canonical stage dump to
 * `stages` and reduced-coordinate SAM text (bwa_print_sam1 dialect) to `sam`; either may be NULL.
 * Returns number of pairs that produced SAM records, <0 on error. */
int fqo_align_batch(fqo_ctx *c, int n, const char *names, const char *names_mate, int name_stride, const uint8_t *seq, const uint8_t *qual,
                    const int32_t *lens, int stride, FILE *stages, FILE *sam);
/* single-end reads (BwtMapper::SingleEndMapper): rows [n][stride], one file */
int fqo_align_batch_se(fqo_ctx *c, int n, const char *names, int name_stride, const uint8_t *seq, const uint8_t *qual, const int32_t *lens, int stride,
                       FILE *st, FILE *sam);
void fqo_print_sam_header(const fqo_index *ix, FILE *sam);
/* --t of the reference: stage A (cal_width + match_gap) of every following batch runs on n_threads workers sliced as
 * src/BwtMapper.cpp:1490-1513 does (rounded up to even; half per end); <=1 = serial.  Results do not depend on it. */
void fqo_ctx_set_threads(fqo_ctx *c, int n_threads);
void fqo_ctx_set_rng(fqo_ctx *c, uint64_t state);   /* the drand48 state (X of X' = 0x5DEECE66D X + 0xB mod 2^48) */

/* counters for the algorithmic-byte model: accumulated over the ctx lifetime */
typedef struct {
  uint64_t occ_block_touches, filter_probes, stack_pops, sa_calls, sa_steps, reads_aligned, pairs;
  uint64_t occ_gap_touches;   /* the part of occ_block_touches made inside bwt_match_gap (incl. bwt_match_exact_alt) */
} fqo_counters;
void fqo_get_counters(const fqo_ctx *c, fqo_counters *out);

/* small exported primitives for unit tests */
double fqo_drand48_selftest(int n_calls, uint64_t *state_out);
int fqo_cal_maxdiff(int l, double err, double thres);
uint32_t fqo_occ(const fqo_index *ix, int which_bwt, uint32_t k, int c);
uint32_t fqo_sa(const fqo_index *ix, int which_bwt, uint32_t k);
/* returns score, fills cigar (op<<14|len) */
int fqo_global_align(const uint8_t *ref, int len1, const uint8_t *qry, int len2, int band, int gap_end, uint16_t *cigar,
                     int *n_cigar);

#ifdef __cplusplus
}
#endif
#endif
