// fq_pack.cpp -- host side of the packed-batch boundary (include/fastquick_amd.h, fq_packed_batch_t): what a FASTQ front end
// does to a tokenised record before it hands it over -- nst_nt4_table codes (libbwa/bntseq.c:38-55) packed 2 bits per base, the
// three 32-mers of the first 96 bases as the read filter forms them (src/BwtIndexer.cpp:441-456), and the list of non-ACGT bases.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"

namespace {
struct Owned {                 // a packed batch and the pinned storage behind it
  fq_packed_batch_t b{};
  void *head = nullptr, *body = nullptr, *len = nullptr, *exc = nullptr, *qlast = nullptr;
};
}  // namespace

extern "C" void *fq_pinned_alloc(size_t bytes) { return fqdev::hmalloc(bytes); }
extern "C" void fq_pinned_free(void *p) { fqdev::hfree(p); }

extern "C" void fq_packed_free(fq_packed_batch_t *b) {
  if (!b) return;
  Owned *o = reinterpret_cast<Owned *>(b);   // b is the first member
  fqdev::hfree(o->head); fqdev::hfree(o->body); fqdev::hfree(o->len); fqdev::hfree(o->exc); fqdev::hfree(o->qlast);
  delete o;
}

extern "C" int fq_pack_reads(const fq_read_batch_t *in, int threads, fq_packed_batch_t **out) {
  if (!in || !out || in->n_pairs < 0 || !in->seq || !in->len || in->stride < 1) return FQ_EINVAL;
  *out = nullptr;
  const size_t n2 = (size_t)in->n_pairs * 2;
  int max_len = 0, min_len = 1 << 30;
  for (size_t r = 0; r < n2; ++r) { max_len = std::max(max_len, (int)in->len[r]); min_len = std::min(min_len, (int)in->len[r]); }
  if (n2 == 0) { max_len = min_len = 0; }
  if (max_len > in->stride || max_len > 65535) return FQ_ELIMIT;
  Owned *o = new (std::nothrow) Owned;
  if (!o) return FQ_ENOMEM;
  fq_packed_batch_t &b = o->b;
  b.n_pairs = in->n_pairs;
  b.uniform_len = (n2 && max_len == min_len) ? max_len : 0;
  b.body_stride = ((max_len + 3) / 4 + 7) & ~7;
  if (b.body_stride == 0) b.body_stride = 8;
  b.qual = in->qual; b.qual_stride = in->stride;
  b.names = in->names; b.name_stride = in->name_stride; b.names_mate = in->names_mate;
  o->head = fqdev::hmalloc(n2 * 24 + 64);
  o->body = fqdev::hmalloc(n2 * (size_t)b.body_stride + 64);
  if (!b.uniform_len) o->len = fqdev::hmalloc(n2 * 2 + 64);
  if (in->qual) o->qlast = fqdev::hmalloc(n2 + 64);
  if (!o->head || !o->body || (!b.uniform_len && !o->len) || (in->qual && !o->qlast)) { fq_packed_free(&o->b); return FQ_ENOMEM; }
  uint8_t *qlast = (uint8_t *)o->qlast;
  uint64_t *head = (uint64_t *)o->head;
  uint8_t *body = (uint8_t *)o->body;
  uint16_t *len = (uint16_t *)o->len;
  if (threads < 1) threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (n2 < 65536) threads = 1;
  std::vector<std::vector<uint64_t>> exc_part((size_t)threads);
  const int stride = in->stride, bstride = b.body_stride;
  auto work = [&](int t) {
    const size_t per = (n2 + threads - 1) / threads, lo = (size_t)t * per, hi = std::min(n2, lo + per);
    std::vector<uint64_t> &exc = exc_part[t];
    for (size_t r = lo; r < hi; ++r) {
      const uint8_t *row = in->seq + r * (size_t)stride;
      const int L = in->len[r];
      if (len) len[r] = (uint16_t)L;
      if (qlast) qlast[r] = L > 0 ? in->qual[r * (size_t)stride + (size_t)(L - 1)] : 0;
      // the filter's view: the first 96 bytes of the row whatever the read's length; behind the read, what the row holds (the
      // reference's reused slot), 0 = never written = code 0
      for (int ch = 0; ch < 3; ++ch) {
        uint64_t k = 0;
        for (int j = 0; j < 32; ++j) {
          const int p = 32 * ch + j;
          const uint8_t c = (p < L || (p < stride && row[p])) ? row[p] : (uint8_t)'A';
          k = (k << 2) | (uint64_t)fq_nt4(c);
        }
        head[(size_t)ch * n2 + r] = k;
      }
      uint8_t *brow = body + r * (size_t)bstride;
      memset(brow, 0, (size_t)bstride);
      for (int i = 0; i < L; ++i) {
        const int c = fq_nt4(row[i]);
        if (c < 4) brow[i >> 2] |= (uint8_t)(c << (2 * (i & 3)));
        else exc.push_back((uint64_t)r << 32 | (uint64_t)i << 8 | (uint64_t)c);
      }
    }
  };
  if (threads == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  size_t ne = 0;
  for (auto &v : exc_part) ne += v.size();
  o->exc = fqdev::hmalloc(ne * 8 + 64);
  if (!o->exc) { fq_packed_free(&o->b); return FQ_ENOMEM; }
  uint64_t *e = (uint64_t *)o->exc;
  for (auto &v : exc_part) { memcpy(e, v.data(), v.size() * 8); e += v.size(); }   // thread ranges are ascending row ranges
  b.qual_last = qlast;
  b.head = head; b.body = body; b.len = len; b.exc = (const uint64_t *)o->exc; b.n_exc = (int64_t)ne;
  *out = &o->b;
  return FQ_OK;
}
