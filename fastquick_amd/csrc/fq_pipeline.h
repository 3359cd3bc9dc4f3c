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

struct FqMulti {             // bwt_multi1_t while it is being built
  uint32_t pos = 0;
  int gap = 0, mm = 0, strand = 0;
  int aln = 0;               // hit it came from, and which row of that hit
  uint32_t row_in_aln = 0;
  std::vector<uint16_t> cigar;
};

struct alignas(64) FqRead {  // the bwa_seq_t fields the hot path writes (libbwa/bwtaln.h:57-86)
  // Everything the order-dependent host phases read and write sits in the record's first cache line (records are 64-byte aligned):
  // they walk two million records serially, and a record used to span three lines.
  int r = 0;                 // row in the input batch: end*n_pairs + pair
  int dr = 0;                // row of the read's ASCII copy on the device (= r for ASCII input; 2*survivor + end for packed input)
  int score = 0;
  uint32_t sa = 0, pos = 0, c1 = 0, c2 = 0;
  int16_t len = 0, full_len = 0, clip_len = 0;
  int16_t main_aln = 0;      // (a hit list holds at most 8,192 hits: the exact tier's cap)
  int16_t nm = 0;
  uint8_t filtered = 0, type = 0, strand = 0, extra_flag = 0;
  uint8_t n_mm = 0, n_gapo = 0, n_gape = 0, mapQ = 0, seQ = 0;
  bool has_md = false;
  bool revived = false;        // filtered on input, brought back for the mate SW because its mate passed (expand_seq, bwape.c:447-462)
  std::vector<FqMulti> multi;
  std::vector<uint16_t> cigar;
  std::string md;
  void reset() {             // back to a fresh record, keeping the containers' storage (records are reused from call to call)
    r = dr = 0; len = full_len = clip_len = 0; filtered = type = strand = extra_flag = 0;
    n_mm = n_gapo = n_gape = mapQ = seQ = score = 0; sa = pos = c1 = c2 = 0; main_aln = 0; nm = 0; has_md = false; revived = false;
    multi.clear(); cigar.clear(); md.clear();
  }
};

struct FqBatchState {
  int n_pairs = 0, n_surv = 0;
  std::vector<int32_t> pair_idx;
  std::vector<FqRead> reads;            // 2 per survivor pair, final state
  std::vector<FqRead> stage_P, stage_S; // snapshots after pairing / after mate SW (debug only)
  struct AlnView {                      // concatenated hit lists: a view of the context's pinned buffer the device's lists land in
    const FqAln *p = nullptr;
    size_t n = 0;
    const FqAln *data() const { return p; }
    size_t size() const { return n; }
    const FqAln &operator[](size_t i) const { return p[i]; }
    void clear() { n = 0; }
  } aln;
  std::vector<int> s_of;                // survivor read -> search index or -1
  std::vector<uint64_t> aln_off;
  std::vector<uint32_t> aln_n;
  fq_isize_t isize{};
  std::vector<fq_isize_t> isize_sub;
  std::vector<int> sub_lo;              // first survivor of each reference batch (+ sentinel)
  int batch_pairs = 262144;
  // flattened C-ABI view (grow-only buffers, never value-initialised: flatten() writes every element it publishes)
  template <class T> struct Flat {
    T *p = nullptr;
    size_t cap = 0;
    Flat() = default;
    Flat(const Flat &) = delete;
    Flat &operator=(const Flat &) = delete;
    ~Flat() { std::free(p); }
    T *data() const { return p; }
    void reserve(size_t n) {
      if (n <= cap) return;
      std::free(p);
      cap = n + n / 8 + 16;
      p = (T *)std::malloc(cap * sizeof(T));
      if (!p) { cap = 0; throw std::bad_alloc(); }
    }
  };
  Flat<fq_result_t> rec;
  Flat<uint16_t> cigar;
  Flat<char> md;
  Flat<fq_multi_t> multi;
  int n_both_unmapped = 0;              // pairs with both ends FQ_TYPE_NO_MATCH (counted by flatten)

  void clear() {
    n_pairs = n_surv = 0;
    pair_idx.clear(); stage_P.clear(); stage_S.clear(); aln.clear(); s_of.clear(); aln_off.clear(); aln_n.clear();
  }
  // C-ABI arrays from the per-read records.  Offsets into the side arenas are prefix sums of per-record sizes: every thread sums
  // the sizes of its range of pairs, the ranges' first offsets follow from those sums, then every thread copies its range.
  void flatten(int threads, size_t par_min) {
    const size_t nrec = reads.size(), npair = nrec / 2;
    const int T = (threads <= 1 || nrec < par_min) ? 1 : threads;
    struct Sum { uint64_t cc = 0, mm = 0, xx = 0; int unm = 0; char pad[36]; };
    std::vector<Sum> sums((size_t)T + 1);
    auto range = [&](int t, size_t &lo, size_t &hi) {   // whole pairs; an odd last record (there is none today) goes to the last range
      const size_t per = (npair + (size_t)T - 1) / (size_t)T;
      lo = std::min(npair, (size_t)t * per) * 2; hi = t == T - 1 ? nrec : std::min(npair, ((size_t)t + 1) * per) * 2;
    };
    auto run = [&](auto fn) {
      if (T == 1) { fn(0); return; }
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(fn, t);
      fn(0);
      for (auto &x : th) x.join();
    };
    run([&](int t) {
      size_t lo, hi;
      range(t, lo, hi);
      Sum a;
      for (size_t i = lo; i < hi; ++i) {
        const FqRead &s = reads[i];
        a.cc += s.cigar.size();
        for (const FqMulti &q : s.multi) a.cc += q.cigar.size();
        if (s.has_md) a.mm += s.md.size() + 1;
        a.xx += s.multi.size();
        if ((i & 1) && s.type == FQ_TYPE_NO_MATCH && reads[i - 1].type == FQ_TYPE_NO_MATCH) ++a.unm;
      }
      sums[(size_t)t + 1] = a;
    });
    n_both_unmapped = 0;
    for (int t = 1; t <= T; ++t) {   // sums[t] becomes the first offsets of range t
      n_both_unmapped += sums[t].unm;
      sums[t].cc += sums[t - 1].cc; sums[t].mm += sums[t - 1].mm; sums[t].xx += sums[t - 1].xx;
    }
    const uint64_t cc = sums[T].cc, mm = sums[T].mm, xx = sums[T].xx;
    if (cc > 0xffffffffull || mm > 0xfffffffeull || xx > 0xffffffffull) throw std::length_error("fq: result arenas exceed 32-bit offsets");
    rec.reserve(nrec ? nrec : 1); cigar.reserve(cc ? cc : 1); md.reserve(mm ? mm : 1); multi.reserve(xx ? xx : 1);
    if (!cc) cigar.p[0] = 0;
    if (!mm) md.p[0] = 0;
    if (!xx) multi.p[0] = fq_multi_t{};
    run([&](int t) {
      size_t lo, hi;
      range(t, lo, hi);
      uint32_t ca = (uint32_t)sums[t].cc, ma = (uint32_t)sums[t].mm, xa = (uint32_t)sums[t].xx;
      for (size_t i = lo; i < hi; ++i) {
        const FqRead &s = reads[i];
        fq_result_t o{};
        o.pos = s.pos; o.sa = s.sa; o.c1 = s.c1; o.c2 = s.c2; o.score = s.score;
        o.len = s.len; o.full_len = s.full_len; o.clip_len = s.clip_len;
        o.type = (uint8_t)s.type; o.strand = (uint8_t)s.strand; o.filtered = (uint8_t)s.filtered; o.extra_flag = (uint8_t)s.extra_flag;
        o.n_mm = (uint8_t)s.n_mm; o.n_gapo = (uint8_t)s.n_gapo; o.n_gape = (uint8_t)s.n_gape; o.mapQ = (uint8_t)s.mapQ;
        o.seQ = (uint8_t)s.seQ; o.pad0 = 0; o.nm = (uint16_t)s.nm;
        o.n_cigar = (uint16_t)s.cigar.size(); o.n_multi = (uint16_t)s.multi.size();
        o.cigar_off = ca;
        std::copy(s.cigar.begin(), s.cigar.end(), cigar.p + ca);
        ca += (uint32_t)s.cigar.size();
        if (s.has_md) { o.md_off = ma; std::copy(s.md.begin(), s.md.end(), md.p + ma); md.p[ma + s.md.size()] = 0; ma += (uint32_t)s.md.size() + 1; }
        else o.md_off = 0xffffffffu;
        o.multi_off = xa;
        for (const FqMulti &q : s.multi) {
          fq_multi_t m{};
          m.pos = q.pos; m.cigar_off = ca; m.n_cigar = (uint16_t)q.cigar.size(); m.gap = (uint8_t)q.gap; m.mm = (uint8_t)q.mm;
          m.strand = (uint8_t)q.strand;
          std::copy(q.cigar.begin(), q.cigar.end(), cigar.p + ca);
          ca += (uint32_t)q.cigar.size();
          multi.p[xa++] = m;
        }
        rec.p[i] = o;
      }
    });
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
  int len(size_t r) const { return a ? a->len[r] : (p->uniform_len > 0 ? p->uniform_len : (int)p->len[r]); }
  // nst_nt4_table codes of the first n bases of row r
  void codes(size_t r, int n, uint8_t *out) const {
    if (a) { const uint8_t *row = a->seq + r * (size_t)a->stride; for (int j = 0; j < n; ++j) out[j] = (uint8_t)fq_nt4(row[j]); return; }
    const uint8_t *b = p->body + r * (size_t)p->body_stride;
    for (int j = 0; j < n; ++j) out[j] = (uint8_t)((b[j >> 2] >> (2 * (j & 3))) & 3);
    if (p->n_exc) {
      const uint64_t *lo = std::lower_bound(p->exc, p->exc + p->n_exc, (uint64_t)r << 32);
      for (; lo < p->exc + p->n_exc && (*lo >> 32) == (uint64_t)r; ++lo) { const int pos = (int)((*lo >> 8) & 0xffff); if (pos < n) out[pos] = (uint8_t)(*lo & 0xff); }
    }
  }
  const uint8_t *qual(size_t r) const { return a ? a->qual + r * (size_t)a->stride : p->qual + r * (size_t)p->qual_stride; }
  bool has_qual() const { return a ? a->qual != nullptr : p->qual != nullptr; }
};
FqHostReads fq_ctx_host_reads(const fq_ctx_t *c);
int64_t fq_ctx_last_bases(const fq_ctx_t *c);   // sum of the read lengths of the last batch (NumBase increment)
// the name a record prints under (fq_sam.cpp): `/1` `/2` stripped, a revived mate under its partner's name
std::string fq_read_name(const FqHostReads *hb, int pair, int end, bool revived);
const fq_opts_t *fq_ctx_opts(const fq_ctx_t *c);
