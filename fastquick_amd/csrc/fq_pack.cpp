// fq_pack.cpp -- host side of the packed-batch boundary (include/fastquick_amd.h, fq_packed_batch_t): what a FASTQ front end
// does to a tokenised record before it hands it over -- nst_nt4_table codes (libbwa/bntseq.c:38-55) packed 2 bits per base, the
// three 32-mers of the first 96 bases as the read filter forms them (src/BwtIndexer.cpp:441-456), and the list of non-ACGT bases.
//
// The packer has to feed a device that filters 10^9 pairs a second, so a row is packed 32 bases at a time (AVX2: two multiply-adds
// fold four 2-bit codes into a byte, once in the body's bit order and once in the k-mer's) and a batch's pinned storage is reused
// from call to call (fq_packed_create / fq_pack_reads_into): page-locking 300 MB per batch costs more than packing it.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"
#include "fq_pool.h"

namespace {
struct Owned {                 // a packed batch and the pinned storage behind it
  fq_packed_batch_t b{};
  void *head = nullptr, *body = nullptr, *len = nullptr, *exc = nullptr, *qlast = nullptr;
  size_t cap_reads = 0, cap_body = 0, cap_exc = 0;   // capacities: reads (head, len, qlast), body bytes, exception entries
  FqWorkPool pool;                                   // the packer's workers stay with the batch object it packs into
};
std::atomic<uint64_t> g_serial{1};   // every packing gets a new serial (fq_packed_batch_t::serial)

// ---- one row, scalar: the definition ------------------------------------------------------------------------------------------
// the filter's view is the first 96 bytes of the row whatever the read's length; behind the read, what the row holds (the
// reference's reused slot), 0 = never written = code 0
inline uint64_t kmer_scalar(const uint8_t *row, int ch, int L, int stride) {
  uint64_t k = 0;
  for (int j = 0; j < 32; ++j) {
    const int p = 32 * ch + j;
    const uint8_t c = (p < L || (p < stride && row[p])) ? row[p] : (uint8_t)'A';
    k = (k << 2) | (uint64_t)fq_nt4(c);
  }
  return k;
}
inline void body_scalar(const uint8_t *row, int i0, int i1, uint8_t *brow, uint64_t r, std::vector<uint64_t> &exc) {   // positions [i0, i1); brow bytes are zero
  for (int i = i0; i < i1; ++i) {
    const int c = fq_nt4(row[i]);
    if (c < 4) brow[i >> 2] |= (uint8_t)(c << (2 * (i & 3)));
    else exc.push_back(r << 32 | (uint64_t)i << 8 | (uint64_t)c);
  }
}
void pack_row_scalar(const uint8_t *row, int L, int stride, uint64_t r, size_t n2, uint64_t *head, uint8_t *brow, int bstride, std::vector<uint64_t> &exc) {
  for (int ch = 0; ch < 3; ++ch) head[(size_t)ch * n2 + r] = kmer_scalar(row, ch, L, stride);
  memset(brow, 0, (size_t)bstride);
  body_scalar(row, 0, L, brow, r, exc);
}

// ---- one row, 32 bases per step -------------------------------------------------------------------------------------------------
// code(ch) = ((ch >> 1) ^ (ch >> 2)) & 3 for the eight letters ACGTacgt (fq_nt4_fast); a group with any other byte takes the scalar
// path for what it touches (an N in a WGS read: one group in a few hundred).
__attribute__((target("avx2"))) void pack_row_avx2(const uint8_t *row, int L, int stride, uint64_t r, size_t n2, uint64_t *head, uint8_t *brow, int bstride, std::vector<uint64_t> &exc) {
  const __m256i m3 = _mm256_set1_epi8(3), mDF = _mm256_set1_epi8((char)0xDF);
  const __m256i cA = _mm256_set1_epi8(0x41), cC = _mm256_set1_epi8(0x43), cG = _mm256_set1_epi8(0x47), cT = _mm256_set1_epi8(0x54);
  const __m256i w_body8 = _mm256_set1_epi16(0x0401), w_body16 = _mm256_set1_epi32(0x00100001);     // c0 + 4 c1 ; n0 + 16 n1
  const __m256i w_head8 = _mm256_set1_epi16(0x0104), w_head16 = _mm256_set1_epi32(0x00010010);     // 4 c0 + c1 ; 16 n0 + n1
  const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i iota = _mm256_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31);
  const int n_groups = (L + 31) >> 5;
  const int body_groups = bstride >> 3;                     // 8 body bytes per group of 32 bases
  for (int g = 0; g < std::max(n_groups, 3); ++g) {
    const int p0 = 32 * g;
    const int nv = std::min(32, L - p0);                    // bases of the read in this group (<= 0: none)
    // the 32 bytes are read whenever the row holds them; bytes behind the read are masked out of the body (they count for the
    // k-mer -- a read shorter than 96 bp -- and that case goes the scalar way)
    bool fast = nv > 0 && p0 + 32 <= stride && (nv == 32 || g >= 3);
    uint64_t body8 = 0, kmer = 0;
    if (fast) {
      const __m256i x = _mm256_loadu_si256((const __m256i *)(row + p0));
      const __m256i u = _mm256_and_si256(x, mDF);
      const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)), _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
      const uint32_t lenmask = nv >= 32 ? 0xffffffffu : (1u << nv) - 1u;
      if ((~(uint32_t)_mm256_movemask_epi8(ok) & lenmask) != 0) fast = false;
      else {
        const __m256i x16 = _mm256_srli_epi16(x, 1);        // (bits that cross byte borders are masked off below)
        __m256i code = _mm256_and_si256(_mm256_xor_si256(x16, _mm256_srli_epi16(x16, 1)), m3);
        code = _mm256_and_si256(code, _mm256_cmpgt_epi8(_mm256_set1_epi8((char)nv), iota));
        const __m256i b = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(code, w_body8), w_body16), gather);
        body8 = (uint64_t)(uint32_t)_mm256_cvtsi256_si32(b) | (uint64_t)(uint32_t)_mm256_extract_epi32(b, 4) << 32;
        if (g < 3) {
          const __m256i h = _mm256_shuffle_epi8(_mm256_madd_epi16(_mm256_maddubs_epi16(code, w_head8), w_head16), gather);
          kmer = __builtin_bswap64((uint64_t)(uint32_t)_mm256_cvtsi256_si32(h) | (uint64_t)(uint32_t)_mm256_extract_epi32(h, 4) << 32);
        }
      }
    }
    if (fast) {
      if (g < body_groups) memcpy(brow + 8 * g, &body8, 8);
      if (g < 3) head[(size_t)g * n2 + r] = kmer;
    } else {
      if (g < body_groups) {
        memset(brow + 8 * g, 0, 8);
        if (nv > 0) body_scalar(row, p0, p0 + nv, brow, r, exc);
      }
      if (g < 3) head[(size_t)g * n2 + r] = kmer_scalar(row, g, L, stride);
    }
  }
  for (int g = std::max(n_groups, 3); g < body_groups; ++g) memset(brow + 8 * g, 0, 8);
  if (bstride & 7) memset(brow + 8 * body_groups, 0, (size_t)(bstride & 7));
}

bool have_avx2() {
  static const bool v = __builtin_cpu_supports("avx2");
  return v;
}
bool grow(void **p, size_t *cap, size_t need, size_t unit, size_t slack) {   // pinned storage only grows; contents are not kept
  if (need <= *cap && *p) return true;
  fqdev::hfree(*p);
  *p = fqdev::hmalloc(need * unit + slack);
  *cap = *p ? need : 0;
  return *p != nullptr;
}
}  // namespace

extern "C" void *fq_pinned_alloc(size_t bytes) { return fqdev::hmalloc(bytes); }
extern "C" void fq_pinned_free(void *p) { fqdev::hfree(p); }

extern "C" void fq_packed_free(fq_packed_batch_t *b) {
  if (!b) return;
  Owned *o = reinterpret_cast<Owned *>(b);   // b is the first member
  fqdev::hfree(o->head); fqdev::hfree(o->body); fqdev::hfree(o->len); fqdev::hfree(o->exc); fqdev::hfree(o->qlast);
  delete o;
}

extern "C" int fq_packed_create(int32_t max_pairs, int32_t max_len, fq_packed_batch_t **out) {
  if (!out || max_pairs < 0 || max_len < 0 || max_len > 65535) return FQ_EINVAL;
  *out = nullptr;
  Owned *o = new (std::nothrow) Owned;
  if (!o) return FQ_ENOMEM;
  const size_t n2 = (size_t)max_pairs * 2;
  const size_t bstride = (((size_t)max_len + 3) / 4 + 7) & ~(size_t)7;
  size_t cap_len = 0, cap_q = 0, cap_head = 0;
  void *len = nullptr, *ql = nullptr;
  bool ok = grow(&o->head, &cap_head, n2 * 3, 8, 64) && grow(&o->body, &o->cap_body, n2 * std::max<size_t>(bstride, 8), 1, 64) &&
            grow(&len, &cap_len, n2, 2, 64) && grow(&ql, &cap_q, n2, 1, 64) && grow(&o->exc, &o->cap_exc, std::max<size_t>(1024, n2 / 64), 8, 64);
  o->len = len; o->qlast = ql; o->cap_reads = n2;
  if (!ok) { fq_packed_free(&o->b); return FQ_ENOMEM; }
  *out = &o->b;
  return FQ_OK;
}

static int pack_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst, int rows_per_pair);
extern "C" int fq_pack_reads_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst) { return pack_into(in, threads, dst, 2); }
extern "C" int fq_pack_single_reads_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst) { return pack_into(in, threads, dst, 1); }
static int pack_into(const fq_read_batch_t *in, int threads, fq_packed_batch_t *dst, int rows_per_pair) {
  if (!in || !dst || in->n_pairs < 0 || !in->seq || !in->len || in->stride < 1) return FQ_EINVAL;
  Owned *o = reinterpret_cast<Owned *>(dst);
  const size_t n2 = (size_t)in->n_pairs * (size_t)rows_per_pair;   // rows of the batch
  if (threads < 1) threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (n2 < 65536) threads = 1;
  const int stride = in->stride;
  // lengths: min / max over the batch (parallel: at 10^8 pairs/s even this pass counts)
  std::vector<int> tmax((size_t)threads, 0), tmin((size_t)threads, 1 << 30);
  auto range = [&](int t, size_t *lo, size_t *hi) { const size_t per = (n2 + threads - 1) / threads; *lo = std::min(n2, (size_t)t * per); *hi = std::min(n2, *lo + per); };
  auto scan = [&](int t) { size_t lo, hi; range(t, &lo, &hi); int mx = 0, mn = 1 << 30; for (size_t r = lo; r < hi; ++r) { const int L = in->len[r]; mx = std::max(mx, L); mn = std::min(mn, L); } tmax[t] = mx; tmin[t] = mn; };
  auto run = [&](auto fn) { o->pool.run(threads, fn); };
  run(scan);
  int max_len = 0, min_len = 1 << 30;
  for (int t = 0; t < threads; ++t) { max_len = std::max(max_len, tmax[t]); min_len = std::min(min_len, tmin[t]); }
  if (n2 == 0) max_len = min_len = 0;
  if (min_len < 0 || max_len > stride || max_len > 65535) return FQ_ELIMIT;
  fq_packed_batch_t &b = o->b;
  const int bstride = std::max(8, (int)((((size_t)max_len + 3) / 4 + 7) & ~(size_t)7));
  if (n2 > o->cap_reads) {
    size_t c1 = 0, c2 = 0, c3 = 0;
    fqdev::hfree(o->head); fqdev::hfree(o->len); fqdev::hfree(o->qlast); o->head = o->len = o->qlast = nullptr; o->cap_reads = 0;
    if (!grow(&o->head, &c1, n2 * 3, 8, 64) || !grow(&o->len, &c2, n2, 2, 64) || !grow(&o->qlast, &c3, n2, 1, 64)) return FQ_ENOMEM;
    o->cap_reads = n2;
  }
  if (!grow(&o->body, &o->cap_body, std::max<size_t>(o->cap_body, n2 * (size_t)bstride), 1, 64)) return FQ_ENOMEM;
  b.n_pairs = in->n_pairs;
  b.uniform_len = (n2 && max_len == min_len) ? max_len : 0;
  b.body_stride = bstride;
  b.qual = in->qual; b.qual_stride = in->stride;
  b.names = in->names; b.name_stride = in->name_stride; b.names_mate = in->names_mate;
  uint8_t *qlast = in->qual ? (uint8_t *)o->qlast : nullptr;
  uint64_t *head = (uint64_t *)o->head;
  uint8_t *body = (uint8_t *)o->body;
  uint16_t *len = b.uniform_len ? nullptr : (uint16_t *)o->len;
  std::vector<std::vector<uint64_t>> exc_part((size_t)threads);
  const bool vec = have_avx2();
  auto work = [&](int t) {
    size_t lo, hi;
    range(t, &lo, &hi);
    std::vector<uint64_t> &exc = exc_part[t];
    for (size_t r = lo; r < hi; ++r) {
      const uint8_t *row = in->seq + r * (size_t)stride;
      const int L = in->len[r];
      if (len) len[r] = (uint16_t)L;
      if (qlast) qlast[r] = L > 0 ? in->qual[r * (size_t)stride + (size_t)(L - 1)] : 0;
      uint8_t *brow = body + r * (size_t)bstride;
      if (vec) pack_row_avx2(row, L, stride, (uint64_t)r, n2, head, brow, bstride, exc);
      else pack_row_scalar(row, L, stride, (uint64_t)r, n2, head, brow, bstride, exc);
    }
  };
  run(work);
  size_t ne = 0;
  for (auto &v : exc_part) ne += v.size();
  if (!grow(&o->exc, &o->cap_exc, std::max<size_t>(o->cap_exc, ne + 1), 8, 64)) return FQ_ENOMEM;
  uint64_t *e = (uint64_t *)o->exc;
  for (auto &v : exc_part) { if (!v.empty()) memcpy(e, v.data(), v.size() * 8); e += v.size(); }   // thread ranges are ascending row ranges
  b.qual_last = qlast;
  b.head = head; b.body = body; b.len = len; b.exc = (const uint64_t *)o->exc; b.n_exc = (int64_t)ne;
  b.serial = g_serial.fetch_add(1, std::memory_order_relaxed);
  b.single_end = rows_per_pair == 1 ? 1 : 0;
  return FQ_OK;
}

extern "C" int fq_pack_reads(const fq_read_batch_t *in, int threads, fq_packed_batch_t **out) {
  if (!in || !out || in->n_pairs < 0 || !in->seq || !in->len || in->stride < 1) return FQ_EINVAL;
  *out = nullptr;
  fq_packed_batch_t *b = nullptr;
  int rc = fq_packed_create(0, 0, &b);
  if (rc) return rc;
  rc = fq_pack_reads_into(in, threads, b);
  if (rc) { fq_packed_free(b); return rc; }
  *out = b;
  return FQ_OK;
}
