// fq_pipeline.h -- host-side per-batch state shared by fq_align.cpp (producer) and fq_sam.cpp (consumers)
#pragma once
#include <algorithm>
#include <cstdlib>
#include <new>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_kernels.h"
#include "fq_records.h"
#include "fq_emit.h"

struct FqMulti {             // bwt_multi1_t as the consumers read it
  uint32_t pos = 0;
  int gap = 0, mm = 0, strand = 0;
  std::vector<uint16_t> cigar;
};

// One record in the host's vocabulary: the bwa_seq_t fields the hot path writes (libbwa/bwtaln.h:57-86).  The consumers (SAM / BAM
// text, StatCollector, the stage dumps) materialise one from the C-ABI arrays when they look at a record (FqBatchState::read); the
// hot path itself never holds them -- its records are FqDRec structs on the device (fq_records.h).
struct FqRead {
  int r = 0;                 // row in the input batch: end*n_pairs + pair
  int score = 0;
  uint32_t sa = 0, pos = 0, c1 = 0, c2 = 0;
  int16_t len = 0, full_len = 0, clip_len = 0;
  int16_t nm = 0;
  uint8_t filtered = 0, type = 0, strand = 0, extra_flag = 0;
  uint8_t n_mm = 0, n_gapo = 0, n_gape = 0, mapQ = 0, seQ = 0;
  bool has_md = false;
  bool revived = false;        // filtered on input, brought back for the mate SW because its mate passed (expand_seq, bwape.c:447-462)
  std::vector<FqMulti> multi;
  std::vector<uint16_t> cigar;
  std::string md;
};

struct FqBatchState {
  int n_pairs = 0, n_surv = 0;
  const int32_t *pair_idx = nullptr;    // survivor pair -> pair of the batch (pinned, where the copy engine landed it)
  const FqSurvInfo *surv = nullptr;     // [2 n_surv]: survivor read -> search index (-1: filtered)
  std::vector<FqRead> stage_P, stage_S; // snapshots after pairing / after mate SW (debug only)
  struct AlnView {                      // concatenated hit lists: a view of the context's pinned buffer the device's lists land in
    const FqAln *p = nullptr;
    size_t n = 0;
    const FqAln *data() const { return p; }
    size_t size() const { return n; }
    const FqAln &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
  } aln;
  std::vector<uint64_t> aln_off;        // by search index
  std::vector<uint32_t> aln_n;
  fq_isize_t isize{};
  std::vector<fq_isize_t> isize_sub;
  std::vector<int> sub_lo;              // first survivor of each reference batch (+ sentinel)
  int batch_pairs = 262144;
  // the C-ABI arrays of the last call: written by the device (fq_flat_fill_thread), landed in the context's pinned buffers
  const fq_result_t *rec = nullptr;
  const uint16_t *cigar = nullptr;
  const char *md = nullptr;
  const fq_multi_t *multi = nullptr;
  int n_both_unmapped = 0;              // pairs with both ends FQ_TYPE_NO_MATCH

  int s_of(size_t i) const { return surv[i].sidx; }
  void clear() {
    n_pairs = n_surv = 0; n_both_unmapped = 0;
    pair_idx = nullptr; surv = nullptr; rec = nullptr; cigar = nullptr; md = nullptr; multi = nullptr;
    stage_P.clear(); stage_S.clear(); aln.clear(); aln_off.clear(); aln_n.clear();
  }
  // record i in the consumers' vocabulary
  FqRead read(size_t i) const {
    const fq_result_t &o = rec[i];
    FqRead s;
    s.r = (int)(i & 1) * n_pairs + pair_idx[i >> 1];
    s.score = o.score; s.sa = o.sa; s.pos = o.pos; s.c1 = o.c1; s.c2 = o.c2;
    s.len = (int16_t)o.len; s.full_len = (int16_t)o.full_len; s.clip_len = (int16_t)o.clip_len; s.nm = (int16_t)o.nm;
    s.filtered = o.filtered; s.type = o.type; s.strand = o.strand; s.extra_flag = o.extra_flag;
    s.n_mm = o.n_mm; s.n_gapo = o.n_gapo; s.n_gape = o.n_gape; s.mapQ = o.mapQ; s.seQ = o.seQ; s.revived = o.revived != 0;
    if (o.n_cigar) s.cigar.assign(cigar + o.cigar_off, cigar + o.cigar_off + o.n_cigar);
    if (o.md_off != 0xffffffffu) { s.has_md = true; s.md = md + o.md_off; }
    s.multi.resize(o.n_multi);
    for (size_t j = 0; j < o.n_multi; ++j) {
      const fq_multi_t &m = multi[o.multi_off + j];
      FqMulti &q = s.multi[j];
      q.pos = m.pos; q.gap = m.gap; q.mm = m.mm; q.strand = m.strand;
      if (m.n_cigar) q.cigar.assign(cigar + m.cigar_off, cigar + m.cigar_off + m.n_cigar);
    }
    return s;
  }
};

struct fq_index;
const FqBatchState *fq_ctx_state(const fq_ctx_t *c);
const fq_index *fq_ctx_index(const fq_ctx_t *c);
// The caller's batch as the host-side consumers read it: ASCII rows (fq_read_batch_t) or a packed batch (fq_packed_batch_t).
struct FqHostReads {
  const fq_read_batch_t *a = nullptr;
  const fq_packed_batch_t *p = nullptr;
  int n_pairs = 0;
  const char *names = nullptr, *names_mate = nullptr;
  int name_stride = 0;
  // a text-resident batch (fq_align_text): only the reads of surviving pairs have come to the host, as compact rows t = 2 * survivor + end
  // (bases as the device's rows hold them: "ACGT", 'N', '-'), found through the ascending list of the survivors' pairs
  bool compact = false;
  const uint8_t *c_seq = nullptr, *c_qual = nullptr;
  const int32_t *c_len = nullptr, *c_pair_idx = nullptr;
  const char *c_names = nullptr;
  const uint16_t *c_len_all = nullptr;   // every row's length (debug dumps), or NULL: c_uniform_len
  int c_stride = 0, c_n_surv = 0, c_uniform_len = 0;
  long crow(size_t r) const {            // compact row of batch row r, -1: not a surviving pair's
    const int pair = (int)(r % (size_t)n_pairs), e = (int)(r / (size_t)n_pairs);
    const int32_t *it = std::lower_bound(c_pair_idx, c_pair_idx + c_n_surv, pair);
    return it != c_pair_idx + c_n_surv && *it == pair ? 2 * (long)(it - c_pair_idx) + e : -1;
  }
  int len(size_t r) const {
    if (compact) { if (c_len_all) return c_len_all[r]; const long t = crow(r); return t >= 0 ? c_len[t] : c_uniform_len; }
    return a ? a->len[r] : (p->uniform_len > 0 ? p->uniform_len : (int)p->len[r]);
  }
  // nst_nt4_table codes of the first n bases of row r
  void codes(size_t r, int n, uint8_t *out) const {
    if (compact) { const long t = crow(r); const uint8_t *row = c_seq + (size_t)t * (size_t)c_stride; for (int j = 0; j < n; ++j) out[j] = (uint8_t)fq_nt4(row[j]); return; }
    if (a) { const uint8_t *row = a->seq + r * (size_t)a->stride; for (int j = 0; j < n; ++j) out[j] = (uint8_t)fq_nt4(row[j]); return; }
    const uint8_t *b = p->body + r * (size_t)p->body_stride;
    for (int j = 0; j < n; ++j) out[j] = (uint8_t)((b[j >> 2] >> (2 * (j & 3))) & 3);
    if (p->n_exc) {
      const uint64_t *lo = std::lower_bound(p->exc, p->exc + p->n_exc, (uint64_t)r << 32);
      for (; lo < p->exc + p->n_exc && (*lo >> 32) == (uint64_t)r; ++lo) { const int pos = (int)((*lo >> 8) & 0xffff); if (pos < n) out[pos] = (uint8_t)(*lo & 0xff); }
    }
  }
  const uint8_t *qual(size_t r) const {
    if (compact) return c_qual + (size_t)crow(r) * (size_t)c_stride;
    return a ? a->qual + r * (size_t)a->stride : p->qual + r * (size_t)p->qual_stride;
  }
  bool has_qual() const { return compact ? c_qual != nullptr : a ? a->qual != nullptr : p->qual != nullptr; }
  bool has_names() const { return compact ? c_names != nullptr : names != nullptr; }
  bool mates_named() const { return compact || names_mate != nullptr; }      // the second mates carry names of their own
  const char *name_of(int pair, int end) const {
    if (compact) { const long t = crow((size_t)end * (size_t)n_pairs + (size_t)pair); return t >= 0 ? c_names + (size_t)t * (size_t)name_stride : ""; }
    return (end && names_mate ? names_mate : names) + (size_t)pair * (size_t)name_stride;
  }
};
FqHostReads fq_ctx_host_reads(const fq_ctx_t *c);
// What a call that counted on the device (fq_ctx_attach_qc) leaves for the consumer's host side: the order-dependent outputs in input order,
// the call's counters.  NULL: the context has no consumer attached.
struct fq_qc;
struct FqQcCallOut {
  const fq_qc *owner = nullptr;
  bool ready = false;
  uint64_t ist_bytes = 0;                                      // .InsertSizeTable lines   } in HBM until the consumer's host side
  uint64_t n_pile = 0;                                         // pileup entries of the markers } fetches them: fq_ctx_qc_stream
  const uint64_t *dup_key = nullptr;                           // [n_surv] duplicate keys of a shard consumer's pairs (~0: none), else NULL
  int n_surv = 0;
  uint64_t cnt[FQ_QC_C_COUNT] = {};
};
const FqQcCallOut *fq_ctx_qc_out(const fq_ctx_t *c);
// ... and a call that formatted its BAM records on the device (fq_ctx_attach_bam): bytes in HBM, streamed off by the writer
struct fq_bam;
struct FqBamCallOut { const fq_bam *owner = nullptr; bool ready = false; uint64_t bytes = 0, z_bytes = 0; };   // z_bytes: the same records as finished BGZF members (0: not made)
const FqBamCallOut *fq_ctx_bam_out(const fq_ctx_t *c);
int fq_bam_device_prepare(fq_bam *b, FqBamArgs *a);
int64_t fq_ctx_bam_stream(fq_ctx_t *c, fq_sink_fn sink, void *user, int members);   // members != 0: the BGZF members instead of the raw records
bool fq_bam_wants_members(const fq_bam *b);
int fq_qc_device_prepare(fq_qc *q, FqQcArgs *a, int n_surv);
// the counting steps of calls that share a consumer run one at a time, in the order in which the calls passed their order-dependent part
uint64_t fq_qc_gate_ticket(fq_qc *q);
void fq_qc_gate_enter(fq_qc *q, uint64_t ticket);
void fq_qc_gate_leave(fq_qc *q);
int64_t fq_ctx_qc_stream(fq_ctx_t *c, int which, fq_sink_fn sink, void *user);
int fq_ctx_emit_wait(fq_ctx_t *c);       // the last call's consumer kernels were only enqueued: wait for them (before its counts / sizes are read)
int64_t fq_ctx_last_bases(const fq_ctx_t *c);   // sum of the read lengths of the last batch (NumBase increment)
// the name a record prints under (fq_sam.cpp): `/1` `/2` stripped, a revived mate under its partner's name
std::string fq_read_name(const FqHostReads *hb, int pair, int end, bool revived);
const fq_opts_t *fq_ctx_opts(const fq_ctx_t *c);
