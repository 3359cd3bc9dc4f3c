/* fastquick_amd.h -- C ABI of the MI355X-native FASTQuick `align` hot path.
 *
 * Drop-in boundary: the reference has no FFI; its seam is the batch of
 * bwa_seq_t records that BwtMapper::PairEndMapper hands from the producer stages to its two
 * consumers.  Each entry point below names the reference interface it replaces
 * (paths under the Griffan/FASTQuick tree).  Plain pointers and sizes only; no C++ or torch
 * types cross this boundary.  All functions return 0 (or a handle) on success and a negative
 * FQ_E* code on failure -- never exit(), never a silent CPU fallback: without a usable HIP
 * device fq_index_load()/fq_ctx_create() fail with FQ_ENODEV.
 */
#ifndef FASTQUICK_AMD_H
#define FASTQUICK_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQ_OK 0
#define FQ_EINVAL (-1)   /* bad argument / option out of the supported range */
#define FQ_EIO (-2)      /* index file missing or malformed */
#define FQ_ENODEV (-3)   /* no HIP device / HIP runtime error */
#define FQ_ENOMEM (-4)
#define FQ_ELIMIT (-5)   /* input exceeds a documented limit (read length, batch size) */

/* record vocabulary: libbwa/bwtaln.h:6-22 */
#define FQ_TYPE_NO_MATCH 0
#define FQ_TYPE_UNIQUE 1
#define FQ_TYPE_REPEAT 2
#define FQ_TYPE_MATESW 3
#define FQ_MAX_READ_LEN 500

typedef struct fq_index fq_index_t;
typedef struct fq_ctx fq_ctx_t;

/* gap_opt_t (libbwa/bwtaln.h:98-119) + pe_opt_t (:124-130): the fields the hot path reads.
 * fq_default_opts() fills gap_init_opt() (bwtaln.c:24-48) / bwa_init_pe_opt() (bwape.c:7-20). */
typedef struct {
  int32_t s_mm, s_gapo, s_gape;
  int32_t mode;               /* BWA_MODE_GAPE(1) | BWA_MODE_COMPREAD(2) [| LOGGAP(4) | NONSTOP(0x10) | IL13(0x200): Phred+64 qualities] */
  int32_t indel_end_skip, max_del_occ, max_entries;
  double fnr;                 /* >0: max_diff from bwa_cal_maxdiff(len, 0.02, fnr) */
  int32_t max_diff, max_gapo, max_gape;
  int32_t max_seed_diff, seed_len;
  int32_t max_top2;
  int32_t trim_qual;          /* --q */
  int32_t filter_thresh;      /* --kmer_thresh; BwtIndexer RollParam.thresh (src/BwtIndexer.cpp:558) */
  int32_t max_isize, force_isize;
  uint32_t max_occ;
  int32_t n_multi, N_multi;
  int32_t is_sw;
  double ap_prior;
  int32_t host_threads;       /* threads for the per-pair host phases of large batches (0 = up to 8, 16 on hosts of 32 cores or more, divided among the calls in flight) */
  int32_t batch_pairs;        /* reference batch size, READ_BUFFER_SIZE = 262144 (src/BwtMapper.h:36): insert-size
                                 inference and the last_ii chain work on consecutive groups of this many pairs, so one
                                 call may carry many reference batches and still reproduce the reference's output */
  int32_t single_end;         /* 1: BwtMapper::SingleEndMapper (src/BwtMapper.cpp:1266-1407) -- reads of one file.  A batch then
                                 holds n_pairs READS (rows [n_pairs][stride], len [n_pairs]; names_mate unused); a result keeps
                                 the pair layout with an empty record at 2*s+1; main hit and up to three alternative hits per read
                                 (N_OCC, :33), no pairing, no mate rescue.  Packed batches: fq_pack_single_reads_into. */
  int32_t pad_opts;
} fq_opts_t;

void fq_default_opts(fq_opts_t *o);

/* One batch of read pairs as produced by the FASTQ tokenizer (replaces the output side of
 * bwa_read_seq_with_hash_dev, src/BwtMapper.cpp:476-613, before encoding/trim/filter which now
 * run on the GPU).  Row r of end e starts at seq + ((size_t)e*n_pairs + r)*stride; ASCII bases and
 * Sanger qualities.  The read filter looks at the first 96 bases of a row whatever the read's length, as the reference does
 * (BwtIndexer.cpp:524-543): for a read shorter than 96 bp the bytes [len, min(96, stride)) of its row count -- leave them 0
 * (a fresh read slot of the reference), or put there what the reference's reused slot would still hold of earlier, longer
 * reads (the CLI does, fq_cli.cpp ReadSlots).  Names are NUL-terminated rows of name_stride bytes.  The reference keeps one name per read
 * (bwa_seq_t::name, libbwa/bwaseqio.c:210) and prints each record under its own: when the second mates' names differ from
 * the first mates' pass them in names_mate (same stride), otherwise leave it NULL and both mates print `names`. */
typedef struct {
  int32_t n_pairs;
  int32_t stride;
  const uint8_t *seq;
  const uint8_t *qual;
  const int32_t *len;         /* [2*n_pairs] */
  const char *names;          /* may be NULL when only records (no SAM text) are wanted */
  int32_t name_stride;
  const char *names_mate;     /* NULL: the second mates carry the same names */
} fq_read_batch_t;

/* XA hit: bwt_multi1_t (libbwa/bwtaln.h:51-55) */
typedef struct {
  uint32_t pos;
  uint32_t cigar_off;         /* into fq_result_batch_t.cigar */
  uint16_t n_cigar;
  uint8_t gap, mm;
  uint8_t strand, pad[3];
} fq_multi_t;

/* Alignment record: the bwa_seq_t fields consumers read (libbwa/bwtaln.h:57-86; list in
 * SURVEY.md 8b).  pos is the 0-based offset in the concatenated reduced reference. */
typedef struct {
  uint32_t pos, sa;
  uint32_t c1, c2;
  int32_t score;
  int32_t len, full_len, clip_len;
  uint8_t type, strand, filtered, extra_flag;
  uint8_t n_mm, n_gapo, n_gape, mapQ;
  uint8_t seQ;
  uint8_t revived;            /* 1: filtered on input, brought back for the mate rescue because its mate passed (expand_seq, libbwa/bwape.c:447-462):
                                 the record prints under its mate's name */
  uint16_t nm;
  uint16_t n_cigar, n_multi;
  uint32_t cigar_off;         /* bwa_cigar_t (op<<14|len) entries in .cigar */
  uint32_t md_off;            /* NUL-terminated MD string in .md ; 0xffffffff = none */
  uint32_t multi_off;         /* into .multi */
} fq_result_t;

/* isize_info_t, libbwa/bwape.h:92-95 */
typedef struct {
  double avg, std, ap_prior;
  uint32_t low, high, high_bayesian;
  uint32_t pad;
} fq_isize_t;

/* Results of one batch.  Only pairs with at least one unfiltered mate ("survivors") get
 * records: rec[2*s + e] for survivor s, whose index in the input batch is pair_idx[s].
 * Storage is owned by the context and valid until the next fq_align_* call on it. */
typedef struct {
  int32_t n_pairs;
  int32_t n_survivors;
  int32_t n_both_filtered;    /* FileStatCollector.TotalFiltered increment (src/BwtMapper.cpp:2034) */
  int32_t n_both_unmapped;    /* BwaUnmapped increment (:2038) */
  const int32_t *pair_idx;
  const fq_result_t *rec;
  const uint16_t *cigar;
  const char *md;
  const fq_multi_t *multi;
  fq_isize_t isize;
  int64_t n_bases;            /* NumBase increment */
  int32_t n_sub;              /* reference batches in this call */
  const fq_isize_t *isize_sub; /* their insert-size estimates; .isize is the last one */
} fq_result_batch_t;

/* ---- index ------------------------------------------------------------------------------ */
/* Builds <fa>.pac .rpac .ann .amb .bwt .rbwt .sa .rsa (and <fa>.rollhash when write_rollhash)
 * from a reduced reference FASTA written by `FASTQuick index` (RefBuilder::PrepareRefSeq).
 * Replaces BwtIndexer::BuildIndex, src/BwtIndexer.cpp:716-762.  Host-only. */
int fq_index_build(const char *fasta_path, int write_rollhash);

/* Parses the index files and stages them into HBM (Occ blocks re-laid out for 32-byte lookups,
 * sampled SA, 2-bit pac, six 2^32-bit filter bitmaps).  Replaces BwtIndexer::LoadIndex,
 * src/BwtIndexer.cpp:803-837.  Bitmaps come from <prefix>.rollhash if present, else
 * <prefix>.rollhash.sparse, else they are rebuilt from the FASTA itself. */
int fq_index_load(const char *prefix, int device_ordinal, fq_index_t **out);
void fq_index_destroy(fq_index_t *ix);
int64_t fq_index_l_pac(const fq_index_t *ix);
int32_t fq_index_n_contigs(const fq_index_t *ix);
/* contig of a concatenated-reference position: bns_coor_pac2real, libbwa/bntseq.c:268 */
int fq_index_contig(const fq_index_t *ix, int32_t id, const char **name, int64_t *offset, int32_t *len);

/* ---- alignment context (one per FASTQ pair; carries drand48 state, last_ii and the (k,l)
 * position cache exactly like BwtMapper::PairEndMapper's locals, src/BwtMapper.cpp:1811-1871) */
int fq_ctx_create(const fq_index_t *ix, const fq_opts_t *opts, int32_t max_pairs_per_batch, fq_ctx_t **out);
void fq_ctx_destroy(fq_ctx_t *c);
const char *fq_ctx_last_error(const fq_ctx_t *c);
/* keep per-stage snapshots of the records so that fq_stage_dump_last() can print them (tests only) */
int fq_ctx_set_debug(fq_ctx_t *c, int keep_stage_snapshots);

/* The whole per-batch hot path: encode+trim+filter (bwa_read_seq_with_hash_dev), gap search
 * (bwa_cal_sa_reg_gap :63), pairing (bwa_cal_pac_pos_pe :721), mate rescue (bwa_paired_sw
 * bwape.c:463) and refinement (bwa_refine_gapped bwase.c:339). */
int fq_align_batch(fq_ctx_t *c, const fq_read_batch_t *in, fq_result_batch_t *out);

/* Same, split so that a caller can stage inputs into HBM ahead of the run:
 * fq_batch_upload copies the batch to the device; fq_align_resident runs the path on it.
 * Lifetime: the records and the fq_*_last formatters refer to the caller's arrays (lengths, names, bases, qualities) -- they
 * must stay valid and unchanged until the next fq_batch_upload / fq_align_* call on the context, or its destruction. */
int fq_batch_upload(fq_ctx_t *c, const fq_read_batch_t *in);
int fq_align_resident(fq_ctx_t *c, fq_result_batch_t *out);

/* ---- packed batches: the measured boundary of SURVEY.md 8(d) -------------------------------------------------------------
 * "Decoded read batch, packed, in pinned host memory" -> "result records in host memory".  A packed batch is what a FASTQ
 * front end hands over instead of the bwa_seq_t::seq byte arrays of bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:526-547):
 *   head  the three 32-mers of every read's first 96 bases exactly as the read filter forms them
 *         (IsReadInHashByCountMoreChunck, src/BwtIndexer.cpp:441-456: kmer = kmer << 2 | code, a non-ACGT code OR-ed in unmasked;
 *         for a read shorter than 96 bp the row bytes behind it count as in fq_read_batch_t): 24 bytes per read, stored as
 *         three arrays head[ch][row], ch = 0..2 -- all the filter needs, and the only part of a filtered read that crosses PCIe;
 *   body  every read 2-bit packed (A0 C1 G2 T3; base i in byte i >> 2 at bit 2 * (i & 3)); a non-ACGT base holds 0 and is
 *         listed in exc as row << 32 | position << 8 | code (nst_nt4_table: 4 = N / any other character, 5 = '-'), ascending;
 *   qual  ASCII quality rows; len the read lengths (NULL when uniform_len > 0);
 *   qual_last  the quality byte of every read's last base (optional).  bwa_trim_read (libbwa/bwaseqio.c:75-88) leaves a read
 *         whole when that base is above the --q threshold, and infer_isize needs the longest trimmed read of a reference batch,
 *         filtered reads included (libbwa/bwape.c:60-61): with this byte per read the library can tell that from the device
 *         without the quality rows of the filtered reads; without it (or when no read of a batch is left whole) it uploads them.
 * Rows are numbered end * n_pairs + pair.  The library uploads head for all reads and body (+ qual when --q trimming is on)
 * for the reads of surviving pairs only.  Arrays should be pinned (fq_pinned_alloc) for the upload to overlap compute; the
 * lifetime rule of fq_batch_upload applies. */
typedef struct {
  int32_t n_pairs;
  int32_t uniform_len;        /* > 0: every read has this many bases (len may be NULL) */
  const uint64_t *head;       /* [3][2 * n_pairs] */
  const uint8_t *body;        /* [2 * n_pairs][body_stride] */
  int32_t body_stride;
  int32_t qual_stride;
  const uint16_t *len;        /* [2 * n_pairs] or NULL */
  const uint64_t *exc;        /* [n_exc], ascending */
  int64_t n_exc;
  const uint8_t *qual;        /* [2 * n_pairs][qual_stride] */
  const uint8_t *qual_last;   /* [2 * n_pairs] or NULL */
  const char *names;          /* as in fq_read_batch_t */
  int32_t name_stride;
  const char *names_mate;
  uint64_t serial;            /* identity of the CONTENT: the packer gives every packing a new value, so a batch object that is
                                 packed again is a new batch to fq_packed_prefetch / fq_align_packed.  A caller that fills a batch by
                                 hand sets a new non-zero value whenever it changes the arrays (0 = the object's address alone). */
  int32_t single_end;         /* 1: a batch of n_pairs READS of one file (BwtMapper::SingleEndMapper): every array has n_pairs rows */
  int32_t pad_packed;
} fq_packed_batch_t;

void *fq_pinned_alloc(size_t bytes);   /* page-locked host memory (hipHostMalloc); NULL on failure */
void fq_pinned_free(void *p);
/* Packs an ASCII batch (the tokenizer's output) on `threads` host threads into pinned storage.  qual / names of the result
 * alias the input's arrays (they are only read for surviving pairs).  Free with fq_packed_free. */
int fq_pack_reads(const fq_read_batch_t *in, int threads, fq_packed_batch_t **out);
void fq_packed_free(fq_packed_batch_t *b);
/* The same with the pinned storage kept from batch to batch (page-locking a few hundred MB per batch costs more than packing
 * them): fq_packed_create sizes a batch object for max_pairs pairs of max_len bases (it grows when a batch needs more),
 * fq_pack_reads_into packs into it.  The object must not be packed again, nor freed, while a prefetch of it is pending on a
 * context: align it (fq_align_packed) or fq_packed_cancel it first. */
int fq_packed_create(int32_t max_pairs, int32_t max_len, fq_packed_batch_t **out);
int fq_pack_reads_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst);
/* The same for a batch of single-end reads (in->n_pairs reads in in->n_pairs rows): what a single-end context aligns. */
int fq_pack_single_reads_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst);
/* Starts the upload of `next`'s head on the context's copy stream and returns: it runs under the kernels of the
 * fq_align_packed call that follows for the current batch -- the overlap the reference gets from its IO worker reading
 * batch k+1 while batch k is aligned (IOworkerAlt, src/BwtMapper.cpp:1973-1980, 2095-2104).  Optional. */
int fq_packed_prefetch(fq_ctx_t *c, const fq_packed_batch_t *next);
/* Forgets a prefetched batch that will not be aligned: waits for its upload to finish (the copy engine reads the batch's pinned
 * arrays until then) and releases the buffer.  FQ_OK also when the batch is not pending on the context. */
int fq_packed_cancel(fq_ctx_t *c, const fq_packed_batch_t *b);
/* The whole hot path on a packed batch, host memory in -> host memory out. */
int fq_align_packed(fq_ctx_t *c, const fq_packed_batch_t *in, fq_result_batch_t *out);
/* n_streams streams run to their ends inside the library -- the driver loop of a stream (prefetch the next batch, align this one, hand the result
 * on), which the reference runs as a producer / consumer pair per line of --fq_list on a thread pool (src/BwtMapper.cpp:232-262, 1840-1845).
 * Stream s aligns batches[s][(first[s] + k) % n_batches[s]] for k = 0 .. n_calls - 1 on ctxs[s] (distinct contexts; first NULL: from 0), the next
 * batch's upload under the current call's kernels.  on_call (may be NULL) is called after each call on that stream's thread -- the result and the
 * context's records (fq_sam_format_last ..) are valid until it returns; a non-zero return ends the stream.  survivors_out (may be NULL): per stream,
 * the surviving pairs of its calls.  The library starts one thread per stream beyond the first (asleep while the device works) and joins them:
 * the caller needs none.  flags: FQ_STREAM_PREFETCH_BEYOND -- the batch behind a stream's last call is prefetched too (a following fq_stream_run that
 * continues the walk finds it uploaded; fq_packed_cancel it otherwise).  Returns FQ_OK or the first failing stream's code (fq_ctx_last_error). */
#define FQ_STREAM_PREFETCH_BEYOND 1
typedef int (*fq_stream_call_fn)(void *user, int32_t stream, int32_t call, const fq_result_batch_t *result);
int fq_stream_run(fq_ctx_t *const *ctxs, int32_t n_streams, const fq_packed_batch_t *const *const *batches, const int32_t *n_batches, const int32_t *first,
                  int32_t n_calls, int32_t flags, fq_stream_call_fn on_call, void *user, int64_t *survivors_out);

/* ---- FASTQ front end ---------------------------------------------------------------------------------------------------------
 * One FASTQ file -> rows of a fq_read_batch_t, with the tokens of kseq_read3_fpc (libbwa/kseq.h:327-371) as
 * bwa_read_seq_with_hash_dev consumes them (src/BwtMapper.cpp:476-613) -- on `threads` threads: BGZF files are inflated
 * member-parallel, any other gzip stream / plain text on one thread beside the tokenising; four-line records are emitted by all
 * threads, anything else byte-wise exactly as the reference's reader takes it (or refuses it: fq_fastq_last_error carries its
 * message).  Replaces the IO worker of a file (IOworkerAlt, src/BwtMapper.cpp:1973-1980).
 * fq_fastq_configure (before the first read): batch_pairs = READ_BUFFER_SIZE of the run; slot_mode = how the reference's reused
 * read slots are modelled -- REUSED: names keep the tails of longer earlier names of their slot and a read
 * shorter than 96 bp gets the slot's earlier bases behind it in its row (PairEndMapper); CLEAN_NAMES: bases only; FRESH: nothing
 * lingers (SingleEndMapper's reader hands out zeroed buffers).  block_bytes: inflated text per block (0: 16 MiB). */
typedef struct fq_fastq fq_fastq_t;
typedef struct {
  int32_t stride, name_stride;
  uint8_t *seq, *qual;        /* [max_reads][stride], cleared behind each read */
  int32_t *len;               /* [max_reads] */
  char *names;                /* [max_reads][name_stride], NUL padded */
} fq_fastq_rows_t;
#define FQ_FASTQ_SLOTS_REUSED 0
#define FQ_FASTQ_SLOTS_CLEAN_NAMES 1
#define FQ_FASTQ_SLOTS_FRESH 2
int fq_fastq_open(const char *path, int threads, fq_fastq_t **out);
int fq_fastq_configure(fq_fastq_t *r, int32_t batch_pairs, int32_t slot_mode, int64_t block_bytes);
/* --frac_samp (src/BwtMapper.cpp:483, 500-507): every reference batch draws from the reference's Random(seed = number of the batch),
 * one number per record it meets, and a record whose number is above `frac` is read and dropped; both files of a pair drop the same
 * records.  1.0 (the default): every record is kept. */
int fq_fastq_set_sampling(fq_fastq_t *r, double frac);
/* up to max_reads records into rows 0..; returns their number (0: end of file) or a negative FQ_E* code */
int64_t fq_fastq_read(fq_fastq_t *r, int64_t max_reads, const fq_fastq_rows_t *rows);
const char *fq_fastq_last_error(const fq_fastq_t *r);
/* name of a last record without line end behind its quality string, which the reference's reader does not return (NULL: none) */
const char *fq_fastq_dropped_record(const fq_fastq_t *r);
/* 1 once a record has come to a read slot that held a longer read before (slot modes REUSED / CLEAN_NAMES).  The reference prints
 * QUAL as a C string out of the slot's unterminated buffer (src/BwtMapper.cpp:549-558, libbwa/bwase.c:401): such a record's QUAL column
 * carries the tail of the longer read (longer than SEQ: not valid SAM) -- the one column of its output this library does not
 * reproduce (QUAL always has the read's length); a caller that promises the reference's bytes warns or stops on it. */
int fq_fastq_unequal_lengths(const fq_fastq_t *r);
int fq_fastq_is_bgzf(const fq_fastq_t *r);
void fq_fastq_close(fq_fastq_t *r);
/* The front end's decoder of one BGZF member's payload and the checksum of its trailer, on their own (the reference reads gzip through
 * zlib's gzread: libbwa/bwaseqio.c:41-52; kseq.h:327-371).  fq_inflate_raw: one complete raw DEFLATE stream (RFC 1951) of n bytes -> exactly
 * out_len bytes at dst; FQ_OK, or FQ_EIO for a stream it does not accept (the reader then lets zlib decide).  fq_crc32: CRC-32 as gzip's. */
int fq_inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t out_len);
uint32_t fq_crc32(const uint8_t *p, size_t n);

/* ---- FASTQ front end on the device ---------------------------------------------------------------------------------------------
 * The member decoder of the device front end on its own (the reference reads gzip through zlib's gzread: libbwa/bwaseqio.c:41-52): n_streams
 * raw DEFLATE streams, one wavefront each, into dst[k] (out_len[k] bytes promised, CRC-32 crc[k] expected).  status[k]: 0 inflated and
 * checked; 1 a stream the device's decoder does not take (the reader gives such a member to fq_inflate_raw, then zlib: their verdict
 * stands); 2 CRC mismatch.  repeats > 1 launches the kernel that often (measurement); *kernel_ms = one launch.
 * fq_bgzf_inflate_device: the same for a run of whole BGZF members (a file image): their text back to back in `out`. */
int fq_inflate_device(int device, int n_streams, const uint8_t *const *src, const size_t *n, uint8_t *const *dst, const uint32_t *out_len, const uint32_t *crc,
                      uint32_t *status, int repeats, double *kernel_ms);
int fq_bgzf_inflate_device(int device, const uint8_t *file, size_t n, uint8_t *out, size_t out_cap, int64_t *n_members, int64_t *text_len, uint32_t *status, int64_t status_cap,
                           int repeats, double *kernel_ms);

/* The front end itself: one FASTQ pair (fq2 NULL or "": one single-end file) of BGZF files -> batches whose reads are resident in HBM, with
 * the tokens of kseq_read3_fpc (libbwa/kseq.h:327-371) as bwa_read_seq_with_hash_dev consumes them (src/BwtMapper.cpp:476-613) and the
 * reference's read-slot history (slot_mode as in fq_fastq_configure; a single-end file always reads into fresh slots).
 * Replaces the two IO workers of a pair (IOworkerAlt, src/BwtMapper.cpp:1973-1980) and fq_fastq_read + fq_pack_reads_into of the host path.
 * The host reads compressed bytes and walks member headers; inflating (k_inflate_bgzf), cutting lines and records, the filter's keys, slot
 * history and names happen on the device.  Chunks hold whole reference batches (batch_pairs) -- up to chunk_pairs pairs -- but for the
 * stream's last.  max_read_len: the longest read taken (the reference's read_len: 151; a longer read ends the device's part).
 * fq_frontend_open: FQ_EIO when a file is not a regular BGZF file (the host reader's case: fq_fastq_open).
 * fq_frontend_next: the next batch in *out and its number of pairs; 0 at the end of the stream; FQ_EFALLBACK when the rest of the stream is
 *   the host reader's -- a record that is not four plain lines, a file that ends inside a record, a read longer than max_read_len: the
 *   device's part ends at a reference-batch boundary and fq_frontend_handover gives readers (fq_fastq_*) standing exactly there, read
 *   slots included, whose verdicts (tokens, refusals, messages) are the host path's; another negative code on failure.
 * fq_align_text: the whole hot path on such a batch.  fq_frontend_release: the batch's buffers may be reused -- after the consumers of the
 *   call's records (fq_sam_format_last, fq_qc_add_last, fq_bam_*) have run; at most three batches are out at a time. */
#define FQ_EFALLBACK (-6)
typedef struct fq_frontend fq_frontend_t;
typedef struct fq_text_batch fq_text_batch_t;
typedef struct {
  double ms_inflate, ms_tokenise;        /* device time (HIP events on the front end's stream): k_inflate_bgzf; everything behind it */
  int64_t members, refused, text_bytes, comp_bytes, pairs;
  double ms_lines, ms_records, ms_slots; /* ms_tokenise by kernel group: line index; record checks + filter keys; read slots + names */
  int64_t inflate_launches, chunks;
  double ms_wait_reader, ms_wait_slot;   /* the producer thread's waits (wall): for a chunk's compressed bytes in HBM; for a batch the caller still holds */
  double ms_read, ms_upload;             /* the reader threads' time (wall, summed over the files): pread into pinned memory; copies to HBM */
} fq_frontend_stats_t;
int fq_frontend_open(int device, const char *fq1, const char *fq2, int32_t batch_pairs, int64_t chunk_pairs, int32_t slot_mode, int32_t max_read_len, fq_frontend_t **out);
int64_t fq_frontend_next(fq_frontend_t *fe, fq_text_batch_t **out);
void fq_frontend_release(fq_frontend_t *fe, fq_text_batch_t *b);
int fq_frontend_handover(fq_frontend_t *fe, int threads, fq_fastq_t **out /* [2] */);
int fq_frontend_unequal_lengths(const fq_frontend_t *fe);      /* as fq_fastq_unequal_lengths */
void fq_frontend_stats(const fq_frontend_t *fe, fq_frontend_stats_t *s);
const char *fq_frontend_last_error(const fq_frontend_t *fe);
void fq_frontend_close(fq_frontend_t *fe);
int32_t fq_text_batch_pairs(const fq_text_batch_t *b);
/* the name the first pair of the batch's sub_batch-th reference batch prints under, end 0 / 1 (src/BwtMapper.cpp:2087-2092 compares them) */
const char *fq_text_batch_first_name(const fq_text_batch_t *b, int32_t sub_batch, int32_t end);
int fq_align_text(fq_ctx_t *c, const fq_text_batch_t *b, fq_result_batch_t *out);
/* the batch's per-read arrays copied to the host (tests): head [3][rows] as fq_packed_batch_t::head, len [rows], names [rows][stride];
 * rows = 2 * pairs (pairs for a single-end batch).  Returns the name stride, or a negative code. */
int fq_text_batch_fetch(fq_frontend_t *fe, const fq_text_batch_t *b, uint64_t *head, uint16_t *len, char *names, int64_t names_cap);

/* ---- one FASTQ stream over several ranks -------------------------------------------------------------------------------
 * A stream shards by reference batch: everything per-read of a batch is independent, and three pieces of state
 * are handed on in batch order -- the drand48 stream (bwa_aln2seq_core, srand48 once per FASTQ pair, src/BwtMapper.cpp:1817), the
 * last_ii fallback (:780-781) and the (k,l) position cache (:815-843).  Every rank aligns its own batches on its own context;
 * the hooks run inside a call around its short order-dependent part: `before` -- receive the state from the owner of the
 * previous batch and fq_ctx_state_import it; `after` -- fq_ctx_state_export and send it to the owner of the next.  Filter, gap
 * search and SA walks of a rank's batch run before `before`, pairing, mate rescue and refinement after `after`, so the ranks
 * overlap everywhere else.  (fastquick_amd/dist.py StreamShard does this over torch.distributed: RCCL or gloo.) */
typedef void (*fq_serial_hook)(void *user);
int fq_ctx_set_serial_hooks(fq_ctx_t *c, fq_serial_hook before, fq_serial_hook after, void *user);
/* Called from inside a hook that failed (a receive timed out, the host language raised): the call returns FQ_EIO right after the
 * hook instead of computing from a stale state, and every state exported from now on carries the broken mark, so the ranks behind
 * fail too. */
int fq_ctx_mark_stream_broken(fq_ctx_t *c);
int64_t fq_ctx_state_export(const fq_ctx_t *c, void *buf, int64_t cap);   /* bytes written, or needed when buf is NULL / too small */
int fq_ctx_state_import(fq_ctx_t *c, const void *buf, int64_t len);
/* the same hand-over between two contexts of one process, without serialising: `to` takes the stream's state, `from` keeps an empty (k,l) cache */
int fq_ctx_state_move(fq_ctx_t *to, fq_ctx_t *from);

/* Experiment / test knobs by name (defaults are what DESIGN.md measures): gap_long_pops, gap_long_always, gap_pool,
 * gap_nogap_min, gap_pipeline_min, gap_pipeline_segs, gap_long_pops2, gap_split_hard, gap_no_order, gap_order_asc, gap_waves_per_cu, gap_coop_waves, gap_refill_min, sw_wave_max, host_threads, host_par_min, filter_no_turns,
 * refine_lanes, packed_bulk_min, trace.  FQ_EINVAL for an unknown key. */
int fq_ctx_set_tuning(fq_ctx_t *c, const char *key, int64_t value);

/* ---- consumers ---------------------------------------------------------------------------
 * SAM text in the --sam_out dialect of bwa_print_sam1 (libbwa/bwase.c:455-581): header, then the
 * records of the last batch.  Each returns the number of bytes written (excluding NUL) or the
 * required size if buf is too small / NULL. */
int64_t fq_sam_header(const fq_index_t *ix, char *buf, int64_t cap);
int64_t fq_sam_format_last(fq_ctx_t *c, char *buf, int64_t cap);
/* canonical per-stage text dump of the last batch (tests): same format as oracle/ref_driver.cpp */
int64_t fq_stage_dump_last(fq_ctx_t *c, char *buf, int64_t cap);

/* ---- QC consumer --------------------------------------------------------------------------------------------------------
 * StatCollector's side of the boundary (src/StatCollector.h:151, src/BwtMapper.cpp:2047-2050): every surviving pair of a batch
 * goes through AddAlignment in input order, and ProcessCore writes <out>.InsertSizeTable .DepthDist .GCDist .EmpRepDist
 * .EmpCycleDist .RawInsertSizeDist .AdjustedInsertSizeDist .SexChromInfo .Pileup .FASTQ.csv .Sequence.csv .Summary .vcf, byte for byte
 * the reference's (the .vcf but for its ##fileDate line).
 * ref_prefix is the reduced reference (<index_prefix>.FASTQuick.fa): <ref_prefix>.SelectedSite.vcf, .dbSNP.subset.vcf and .gc as
 * `FASTQuick index` writes them.  One fq_qc_begin_file / fq_qc_end_file bracket per FASTQ pair (FileStatCollector). */
typedef struct fq_qc fq_qc_t;
typedef struct {
  int32_t flank_len, flank_long_len;   /* SHORT_FLANK_LENGTH / LONG_FLANK_LENGTH of the index's .param (250 / 1000) */
  int32_t read_len;                    /* gap_opt_t::read_len (151): the flank edge that is not counted is 0.65 x this */
  int32_t cal_dup;                     /* gap_opt_t::cal_dup (1) */
  int64_t genome_size, genome_n_size;  /* BwtIndexer::LoadContigSize: sums over the original reference's .fai / .amb */
  int32_t mode;                        /* taken from the alignment context (Phred+64 input) */
  int32_t pad;
} fq_qc_opts_t;
void fq_qc_default_opts(fq_qc_opts_t *o);
int fq_qc_create(const fq_index_t *ix, const char *ref_prefix, const char *out_prefix, const fq_qc_opts_t *o, fq_qc_t **out);
void fq_qc_destroy(fq_qc_t *q);
const char *fq_qc_last_error(const fq_qc_t *q);
int fq_qc_begin_file(fq_qc_t *q, const char *fastq_1, const char *fastq_2);
int fq_qc_add_last(fq_qc_t *q, fq_ctx_t *c);   /* the records of the context's last batch; call before the next fq_align_* on it */
int fq_qc_end_file(fq_qc_t *q);
int fq_qc_write(fq_qc_t *q);                   /* ProcessCore: writes the files (once, at the end) */
/* The consumer over several ranks (StatCollector is one object per run in the reference: FileStatCollector sums per FASTQ pair,
 * src/StatCollector.h:46-62, AddFSC; tables filled by AddSingleAlignment / ProcessPairStatus, src/StatCollector.cpp:424-921).
 * Each rank feeds its shard of the input -- whole FASTQ pairs of a --fq_list, or reference batches of one pair -- to a consumer of
 * its own (any out_prefix of its own: the .InsertSizeTable lines are kept there).  fq_qc_state_reset makes a consumer a shard
 * consumer and starts a segment (call it before the first batch, and after every export); fq_qc_state_export serialises what the
 * segment gathered (returns the bytes written, or needed when buf is NULL / too small); fq_qc_merge adds a segment to a consumer AS
 * IF its records had been fed after everything that consumer holds: sums add, .InsertSizeTable lines, pileup strings and
 * first-seen sex-chromosome contigs append, duplicate keys are looked up in the union.  Merging the segments in input order into
 * one consumer and writing there gives the files of the single-process run. A segment that ends inside a FASTQ pair carries that
 * pair's partial counters: the merging consumer must have the pair open (fq_qc_begin_file) or get it from an earlier segment. */
int fq_qc_state_reset(fq_qc_t *q);
int64_t fq_qc_state_export(fq_qc_t *q, void *buf, int64_t cap);
int fq_qc_merge(fq_qc_t *q, const void *buf, int64_t len);
/* StatCollector on the device (fq_emit.h).  A context with a consumer attached counts every call inside the call, on the result arrays where
 * they lie: a thread per pair decides what AddAlignment decides (src/StatCollector.cpp:950-1101) and measures the pair's .InsertSizeTable line and
 * its pileup entries; prefix sums place them in input order; a wavefront per added read adds its bases to the depth / Q20 / Q30 tables of the
 * flank regions and to the quality / cycle histograms (AddSingleAlignment, :424-620), which stay in HBM until fq_qc_write / fq_qc_state_export
 * fetch them; proper pairs' duplicate keys go into a hash set in HBM (ProcessPairStatus, :623-921).  fq_qc_add_last(q, c) then only appends what
 * came back in input order (lines, pileup entries) and adds the call's counters.  The files are the host path's byte for byte.  A consumer
 * counts on one side only: every context that feeds it must have it attached (or none); calls of contexts that share a consumer must not
 * overlap.  q = NULL detaches. */
int fq_ctx_attach_qc(fq_ctx_t *c, fq_qc_t *q);

/* ---- BAM consumer ---------------------------------------------------------------------------------------------------------
 * BwtMapper::SetSamRecord / SetSamFileHeader (src/BwtMapper.cpp:947-1264) as a BAM file (own BGZF layer): genome coordinates
 * (contig CHR:POS@REF/ALT[|L] -> RNAME CHR, POS - flank + offset - 1), @SQ from the original reference's .fai, RG:Z on every
 * record.  flank_len / flank_long_len of `o` are the index's; rg_line is --RG ("@RG\tID:foo\tSM:bar" is runAlign's default). */
typedef struct fq_bam fq_bam_t;
int fq_bam_create(const fq_index_t *ix, const char *fai_path, const char *bam_path, const char *rg_line, const fq_qc_opts_t *o, fq_bam_t **out);
int fq_bam_add_last(fq_bam_t *b, fq_ctx_t *c);   /* the records of the context's last batch, in input order */
/* For several producers and one file: a writer created with bam_path = NULL has no file and only formats -- fq_bam_format_last hands
 * out the last batch's records as bytes (owned by the writer, valid until its next call); fq_bam_write_records appends such bytes to
 * a writer that has a file.  The BGZF stream does not depend on how the bytes were cut: the file is the one fq_bam_add_last writes. */
int fq_bam_format_last(fq_bam_t *b, fq_ctx_t *c, const void **data, int64_t *len);
int fq_bam_write_records(fq_bam_t *b, const void *data, int64_t len);
int fq_bam_close(fq_bam_t *b);                   /* writes the BGZF end-of-file block, closes and frees */
/* BAM records on the device (fq_emit.h): a context with a writer attached formats the records of every call in kernels inside the call
 * (SetSamRecord's fields, the tags in the order of SamRecord's hash) and leaves them in HBM; fq_bam_add_last / fq_bam_format_last of THAT writer
 * then fetch the bytes instead of formatting on the host -- the same bytes.  b = NULL detaches. */
int fq_ctx_attach_bam(fq_ctx_t *c, fq_bam_t *b);
/* (tests, tools) n bytes as BGZF members written by the device's compressor (one wavefront per block of 53,248 bytes: greedy LZ77, one fixed-Huffman
 * block, CRC-32 -- csrc/fq_deflate.h), the members behind each other in out; replaces the zlib deflate of the reference's BGZF layer
 * (VerifyBamID/statgen/BgzfFileType.h over htslib's bgzf_write) for the records a context with a writer attached formats on the device. */
int fq_bgzf_deflate_device(int device, const uint8_t *in, int64_t n, uint8_t *out, int64_t cap, int64_t *out_len, double *kernel_ms);

/* ---- the consumers on the device ---------------------------------------------------------------
 * The reference hands every record to its consumers on its main thread, one after the other (src/BwtMapper.cpp:2030-2085): bwa_print_sam1
 * (libbwa/bwase.c:455-581) formats one line at a time.  Here the result arrays of a call are resident in HBM when the call ends, and the
 * consumers may run there: with FQ_EMIT_SAM the SAM text of a call (the --sam_out dialect, byte for byte what fq_sam_format_last returns) is
 * formatted by kernels inside the call -- a thread per record measures its line, a prefix sum places the lines, a thread per record writes --
 * and stays on the device until the next call on the context.  fq_sam_device_last streams it to `sink` in slices, in order, over streams of
 * its own (it may run on another thread beside the next call on ANOTHER context); returns the bytes handed over or a negative code.
 * fq_sam_device_bytes: the size of that text. */
#define FQ_EMIT_SAM 1
#define FQ_EMIT_DEVICE_ONLY 2   /* every consumer of the context's records runs on the device: a call leaves its result arrays and the surviving reads' rows
                                   in HBM (fq_result_batch_t carries the counts and NULL arrays; the host-side consumers refuse such a batch) */
typedef int (*fq_sink_fn)(void *user, const void *data, int64_t bytes);   /* 0: go on; anything else ends the stream with FQ_EIO */
int fq_ctx_set_emit(fq_ctx_t *c, int32_t flags);
int64_t fq_sam_device_last(fq_ctx_t *c, fq_sink_fn sink, void *user);
int64_t fq_sam_device_bytes(const fq_ctx_t *c);

/* ---- measurement -------------------------------------------------------------------------
 * Per-kernel device time (HIP events on the context's stream) and algorithmic work counters
 * accumulated since the last reset.  Kernel ids: see FQ_K_*. */
#define FQ_K_PREP 0      /* encode + trim + filter + compaction */
#define FQ_K_WIDTH 1     /* bwt_cal_width x4 */
#define FQ_K_GAP 2       /* bwt_match_gap (the Occ-lookup kernel) */
#define FQ_K_SA 3        /* bwt_sa over enumerated rows */
#define FQ_K_SW 4        /* mate-rescue Smith-Waterman */
#define FQ_K_REFINE 5    /* banded global DP + MD/NM */
#define FQ_K_PREP_KERNEL 6   /* k_prep alone, kernel begin/end timestamps (hipExtLaunchKernelGGL events) */
#define FQ_K_GAP_KERNEL 7    /* the gap-search kernels alone, same: the full search (one read per lane / per wavefront) */
#define FQ_K_GAP_NOGAP 8     /* ... the first round of a device-filling launch (the search without gap children) */
#define FQ_K_WIDTH_KERNEL 9  /* k_width alone (kernel begin/end timestamps, like the three above) */
#define FQ_K_SA_KERNEL 10    /* k_sa */
#define FQ_K_SW_KERNEL 11    /* the mate-rescue kernels */
#define FQ_K_REFINE_KERNEL 12 /* the banded global DP kernels */
#define FQ_K_MD_KERNEL 13    /* MD / NM of every mapped read */
#define FQ_K_REC_KERNEL 14   /* the record stages (fq_records.h): set-up, main hit, pairing, XA, task lists, flattening */
#define FQ_K_EMIT 15         /* the consumers' kernels (fq_emit.h): SAM text, StatCollector's sums */
#define FQ_K_COUNT 16
typedef struct {
  double kernel_ms[FQ_K_COUNT];
  uint64_t kernel_launches[FQ_K_COUNT];
  uint64_t occ_block_touches;   /* 32-byte Occ blocks fetched by FQ_K_WIDTH+FQ_K_GAP+FQ_K_SA */
  uint64_t gap_occ_touches;     /* ... by FQ_K_GAP alone */
  uint64_t gap_nogap_touches;   /* ... of which by searches the first round (FQ_K_GAP_NOGAP) completed */
  uint64_t filter_probes;       /* bitmap probes issued by FQ_K_PREP */
  uint64_t stack_pops, stack_pushes;
  uint64_t sa_rows;
  uint64_t reads_searched, pairs, sw_tasks, refine_tasks;
  uint64_t tier_retries;        /* reads re-run with a larger search pool */
  uint64_t max_pops_per_read, reads_over_4k_pops;   /* tail of the search-length distribution */
  uint64_t max_wave_trips;      /* loop iterations of the busiest wavefront of the gap kernel (per launch, max) */
  double host_ms_serial, host_ms_pair, host_ms_total, wall_ms_total;   /* host_ms_*: the host's own time in the order-dependent part / behind it / both, waits for the device excluded */
  uint64_t wave_trips;          /* loop iterations summed over the gap kernel's wavefronts */
  uint64_t lane_trips;          /* ... summed over lanes that held a read in that iteration (wave_trips x 64 = all slots) */
  uint64_t h2d_bytes, d2h_bytes; /* bytes the calls moved over PCIe (inputs, task lists; results) */
  uint64_t pairs_on_device;     /* both-mapped pairs whose pairing (libbwa/bwape.c:119-213) ran in k_pair; the rest ran the same routine on the host */
  uint64_t dbg[16];             /* experiment counters of instrumented builds (-DFQ_GAP_INSTR), zero otherwise */
  uint64_t width_occ_touches;   /* 32-byte Occ blocks fetched by FQ_K_WIDTH alone */
  uint64_t md_reads;            /* mapped reads FQ_K_MD_KERNEL wrote an MD string for */
  uint64_t host_pairs;          /* both-mapped pairs the host paired (Q6 intervals, many rows): pairs_on_device counts the others */
  double device_wait_ms;        /* time the calls' threads slept waiting for the device (host_ms_* exclude the waits inside their sections) */
  double host_cpu_ms;           /* CPU time the calls' own threads used, set-up to result arrays (the pooled workers of the few parallel passes not included) */
} fq_stats_t;
void fq_stats_get(const fq_ctx_t *c, fq_stats_t *out);
void fq_stats_reset(fq_ctx_t *c);

const char *fq_version(void);
/* CPUs this process may use: the hardware threads it sees, cut to its cgroup's CPU quota (cpu.max) when there is one, and divided by
 * LOCAL_WORLD_SIZE when the process is one of several ranks of a node (FASTQUICK_HOST_CPUS states the share outright) -- what the
 * library sizes a call's host threads by (opts.host_threads = 0) and what a caller should size its packer / reader threads by.
 * (No reference counterpart: bwa's --t is the caller's number.) */
int fq_host_cpus(void);
/* Process-level settings, made only when the host program asks -- before its first HIP call (loading the library changes nothing):
 *   hw_queues > 0       GPU_MAX_HW_QUEUES = hw_queues for the HIP runtime, unless the environment already holds a value.  Every
 *                       alignment context drives a stream of its own; contexts that share a hardware queue (the runtime's default
 *                       is 4) wait for each other's long kernels.  contexts + 4 is the measured optimum (16 streams: 20).
 *   blocking_waits != 0 devices the library opens get hipDeviceScheduleBlockingSync: every wait of the runtime on them sleeps instead
 *                       of spinning (the library's own waits sleep on blocking events either way).
 * The library warns once on stderr when more contexts are created on a device than the queues in effect keep apart. */
int fq_runtime_configure(int hw_queues, int blocking_waits);
/* HIP devices the process sees (0: none, or no usable runtime): what a device ordinal of fq_index_load is checked against before
 * anything is started on it.  (No reference counterpart.) */
int fq_device_count(void);
/* (tests) table t (0..5) of the read filter's bitmaps as fq_index_load left it on the device: 2^29 bytes into out -- what BwtIndexer keeps in
 * roll_hash_table[t] (src/BwtIndexer.cpp:96-160, 803-837), whether it came from .rollhash, from a list of bits or from the reference's 32-mers. */
int fq_index_bitmap_fetch(const fq_index_t *ix, int32_t t, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif /* FASTQUICK_AMD_H */
