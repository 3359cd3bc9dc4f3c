// fq_index.cpp -- index files <-> HBM tables, and the index builder.
//
// Loader: replaces BwtIndexer::LoadIndex (src/BwtIndexer.cpp:803-837) + bwt_restore_bwt/sa
// (libbwa/bwtio.c:29-70) + bns_restore_core (libbwa/bntseq.c:87-140).  File formats: SURVEY.md
// section 10.  Builder: replaces BwtIndexer::BuildIndex (src/BwtIndexer.cpp:716-762): Fa2Pac (:839-975),
// Fa2RevPac (:1287), Pac2Bwt (:1317) + bwt_bwtupdate_core (:1369), bwt_cal_sa (:1394), bns_dump
// (libbwa/bntseq.c:57-85) and the k-mer bitmap fill AddSeq2HashCore (:611-713).
#include "fq_index.h"

#include <algorithm>
#include <chrono>
#include <atomic>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"

namespace fqdev {
int launch_bitmap_scatter(uint8_t *bitmap, const uint32_t *bits, uint64_t n);
int launch_bitmap_kmers(const FqBitmapArgs &a);
}

namespace {

bool slurp(const std::string &path, std::vector<uint8_t> &out) {
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) return false;
  fseek(fp, 0, SEEK_END);
  long sz = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  out.resize((size_t)sz);
  bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
  fclose(fp);
  return ok;
}
bool exists(const std::string &p) { FILE *fp = fopen(p.c_str(), "rb"); if (!fp) return false; fclose(fp); return true; }

inline int nt4(int ch) {
  switch (ch) {
    case 'A': case 'a': return 0; case 'C': case 'c': return 1;
    case 'G': case 'g': return 2; case 'T': case 't': return 3;
    case '-': return 5; default: return 4;
  }
}

// ---- k-mer bitmap contents from the reduced reference (AddSeq2HashCore) -----------------------------
inline uint64_t code_or_rand(int ch) { int c = nt4(ch); return c < 4 ? (uint64_t)c : (uint64_t)(rand() % 4); }  // NST_NT4_TABLE, :59-61

void add_seq_bits(std::vector<uint32_t> bits[6], const std::string &s, const char alleles[2]) {
  const size_t L = s.size();
  if (L < 64) return;
  for (int t = 0; t < 6; ++t) {
    uint64_t d = 0;
    size_t i = 0;
    for (; i != 32; ++i) d = (d << 2) | code_or_rand(s[i]);
    bits[t].push_back(fq_kmer_project(d, t));
    for (; i != L / 2; ++i) { d = (d << 2) | code_or_rand(s[i]); bits[t].push_back(fq_kmer_project(d, t)); }
    uint64_t tmp = d;
    for (int a = 0; a < 2; ++a) {
      tmp = d;
      for (size_t j = i; j != L / 2 + 32; ++j) {
        tmp = (tmp << 2) | code_or_rand(j == L / 2 ? alleles[a] : s[j]);
        bits[t].push_back(fq_kmer_project(tmp, t));
      }
    }
    d = tmp;   // continues from the last allele's register, like the reference
    for (i = L / 2 + 32; i != L; ++i) { d = (d << 2) | code_or_rand(s[i]); bits[t].push_back(fq_kmer_project(d, t)); }
  }
}

struct FastaRec { std::string name, seq; };
bool read_reduced_fasta(const std::string &path, std::vector<FastaRec> &recs) {
  FILE *fp = fopen(path.c_str(), "r");
  if (!fp) return false;
  char *line = nullptr;
  size_t cap = 0;
  ssize_t n;
  // the reduced reference has exactly one sequence line per record (src/RefBuilder.cpp:585-613)
  while ((n = getline(&line, &cap, fp)) > 0) {
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    if (n == 0) continue;
    FastaRec r;
    r.name = line + 1;   // CurrentSeqName.erase(begin)
    n = getline(&line, &cap, fp);
    if (n <= 0) break;
    while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    r.seq = line;
    recs.push_back(std::move(r));
  }
  free(line);
  fclose(fp);
  return !recs.empty();
}
std::string revcomp_ref(const std::string &s) {   // BwtIndexer::ReverseComplement: non-ACGT -> '\0'
  std::string o(s.size(), '\0');
  for (size_t i = 0; i < s.size(); ++i) {
    char ch = s[s.size() - 1 - i], c = 0;
    switch (ch) { case 'A': case 'a': c = 'T'; break; case 'C': case 'c': c = 'G'; break; case 'G': case 'g': c = 'C'; break; case 'T': case 't': c = 'A'; break; default: c = 0; }
    o[i] = c;
  }
  return o;
}
void bitmap_bits_of_records(const std::vector<FastaRec> &recs, size_t lo, size_t hi, std::vector<uint32_t> bits[6]) {
  for (size_t k = lo; k < hi; ++k) {
    const FastaRec &r = recs[k];
    char alleles[2] = {'N', 'N'};
    size_t at = r.name.find('@');
    if (at != std::string::npos && at + 3 < r.name.size() + 1) { alleles[0] = r.name[at + 1]; alleles[1] = r.name[at + 3]; }
    add_seq_bits(bits, r.seq, alleles);
    add_seq_bits(bits, revcomp_ref(r.seq), alleles);
  }
}
void bitmap_bits_from_fasta(const std::vector<FastaRec> &recs, std::vector<uint32_t> bits[6]) { bitmap_bits_of_records(recs, 0, recs.size(), bits); }
// The same set of bits, listed by several threads (one slice of the records each; the order of a list of bits to set does not matter).
// A base that is not ACGT draws from rand() (code_or_rand), whose stream is one per process and in record order: a reference that
// holds one -- or an allele pair that is not two of ACGT -- keeps the single-thread walk.
struct BitParts { std::vector<uint32_t> bits[6]; };
bool all_plain_bases(const std::vector<FastaRec> &recs) {
  bool lut[256] = {};
  for (char c : {'A', 'C', 'G', 'T', 'a', 'c', 'g', 't'}) lut[(unsigned char)c] = true;
  for (const auto &r : recs) {
    for (char c : r.seq) if (!lut[(unsigned char)c]) return false;
    const size_t at = r.name.find('@');
    if (at == std::string::npos || at + 3 >= r.name.size() + 1 || !lut[(unsigned char)r.name[at + 1]] || !lut[(unsigned char)r.name[at + 3]]) return false;
  }
  return true;
}
void bitmap_bits_from_fasta_threads(const std::vector<FastaRec> &recs, std::vector<BitParts> &parts) {
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char *e = getenv("FASTQUICK_HOST_CPUS")) { const int v = atoi(e); if (v > 0) nt = std::min(nt, (unsigned)v); }
  if (recs.size() < 64 || !all_plain_bases(recs)) nt = 1;
  parts.resize(nt);
  if (nt == 1) { bitmap_bits_of_records(recs, 0, recs.size(), parts[0].bits); return; }
  // slices of about equal numbers of BASES (the long flanks of the contamination markers sit together at the end of the file)
  size_t total = 0;
  for (const auto &r : recs) total += r.seq.size();
  std::vector<size_t> cut(nt + 1, recs.size());
  cut[0] = 0;
  size_t acc = 0, k = 1;
  for (size_t i = 0; i < recs.size() && k < nt; ++i) {
    acc += recs[i].seq.size();
    while (k < nt && acc >= total * k / nt) cut[k++] = i + 1;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] { for (int q = 0; q < 6; ++q) parts[t].bits[q].reserve((cut[t + 1] - cut[t]) ? 2 * (total / nt) + 4096 : 0); bitmap_bits_of_records(recs, cut[t], cut[t + 1], parts[t].bits); });
  for (auto &x : th) x.join();
}

// ---- suffix sorting for the builder ---------------------------------------------------------------------
// Plain comparison sort over 32-base packed keys; suffixes that run into the end of the text compare
// as if followed by a sentinel smaller than A.  Offline tool: clarity over speed.
struct SufCmp {
  const uint8_t *t;
  const uint64_t *key;
  uint32_t n;
  bool operator()(uint32_t a, uint32_t b) const {
    if (a == b) return false;
    for (;;) {
      const uint32_t ra = n - a, rb = n - b;
      if (ra >= 32 && rb >= 32) {
        const uint64_t ka = key[a], kb = key[b];
        if (ka != kb) return ka < kb;
        a += 32; b += 32;
        if (a == n) return true;    // a exhausted first -> smaller
        if (b == n) return false;
        continue;
      }
      const uint32_t m = ra < rb ? ra : rb;
      for (uint32_t i = 0; i < m; ++i) if (t[a + i] != t[b + i]) return t[a + i] < t[b + i];
      return ra < rb;
    }
  }
};

struct BuiltFM {
  uint32_t primary = 0, L2[5] = {0, 0, 0, 0, 0};
  std::vector<uint8_t> bwt;      // n symbols, $ row removed
  std::vector<uint32_t> sa;      // samples, sa[0] = 0xffffffff
};
void build_fm(const std::vector<uint8_t> &text, BuiltFM &out) {
  const uint32_t n = (uint32_t)text.size();
  std::vector<uint64_t> key((size_t)n + 1, 0);
  {
    uint64_t k = 0;
    // key[i] = bases i..i+31 (zero padded); build right to left
    for (int64_t i = (int64_t)n - 1; i >= 0; --i) {
      k = (k >> 2) | ((uint64_t)text[(size_t)i] << 62);
      key[(size_t)i] = k;
    }
  }
  std::vector<uint32_t> sa(n);
  // bucket by the first 3 bases (64 buckets) so buckets can be sorted on separate threads
  std::vector<uint32_t> bstart(66, 0);
  auto bucket_of = [&](uint32_t i) -> uint32_t {
    // suffixes shorter than 3 sort before longer ones with the same prefix; give them their own order by full compare later
    uint32_t b = 0;
    for (int d = 0; d < 3; ++d) b = b * 4 + (i + d < n ? text[i + d] : 0);
    return b;
  };
  for (uint32_t i = 0; i < n; ++i) ++bstart[bucket_of(i) + 1];
  for (int b = 0; b < 64; ++b) bstart[b + 1] += bstart[b];
  {
    std::vector<uint32_t> fill(bstart.begin(), bstart.begin() + 64);
    for (uint32_t i = 0; i < n; ++i) sa[fill[bucket_of(i)]++] = i;
  }
  SufCmp cmp{text.data(), key.data(), n};
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> th;
  std::atomic<int> next{0};
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (;;) {
        int b = next.fetch_add(1);
        if (b >= 64) break;
        std::sort(sa.begin() + bstart[b], sa.begin() + bstart[b + 1], cmp);
      }
    });
  for (auto &t : th) t.join();
  // rows of T$: row 0 is the sentinel suffix (position n); row i+1 = sa[i]
  out.bwt.assign(n, 0);
  out.sa.assign(((size_t)n + 32) / 32, 0);
  out.sa[0] = 0xffffffffu;
  uint32_t w = 0;
  out.bwt[w++] = text[n - 1];            // row 0: char preceding the sentinel
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t row = i + 1, p = sa[i];
    if (row % 32 == 0) out.sa[row / 32] = p;
    if (p == 0) out.primary = row; else out.bwt[w++] = text[p - 1];
  }
  for (uint32_t i = 0; i < n; ++i) ++out.L2[1 + text[i]];
  for (int c = 1; c <= 4; ++c) out.L2[c] += out.L2[c - 1];
}

bool write_bwt_file(const std::string &path, const BuiltFM &fm, uint32_t n) {
  // interleaved counts + 2-bit words, libbwa/bwt.h:56-63 / BwtIndexer::bwt_bwtupdate_core :1369-1392
  std::vector<uint32_t> buf;
  uint32_t c[4] = {0, 0, 0, 0}, word = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (i % 128 == 0) buf.insert(buf.end(), c, c + 4);
    word |= (uint32_t)fm.bwt[i] << ((15 - (i & 15)) << 1);
    ++c[fm.bwt[i]];
    if ((i & 15) == 15 || i == n - 1) { buf.push_back(word); word = 0; }
  }
  buf.insert(buf.end(), c, c + 4);
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) return false;
  fwrite(&fm.primary, 4, 1, fp);
  fwrite(fm.L2 + 1, 4, 4, fp);
  fwrite(buf.data(), 4, buf.size(), fp);
  fclose(fp);
  return true;
}
bool write_sa_file(const std::string &path, const BuiltFM &fm, uint32_t n) {
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) return false;
  const uint32_t intv = 32;
  fwrite(&fm.primary, 4, 1, fp);
  fwrite(fm.L2 + 1, 4, 4, fp);
  fwrite(&intv, 4, 1, fp);
  fwrite(&n, 4, 1, fp);
  fwrite(fm.sa.data() + 1, 4, fm.sa.size() - 1, fp);
  fclose(fp);
  return true;
}

// glibc lrand48 after srand48(seed): used by Fa2Pac for N bases (src/BwtIndexer.cpp:851, :932)
struct Rand48 {
  uint64_t x;
  explicit Rand48(uint32_t seed) : x(((uint64_t)seed << 16) | 0x330E) {}
  uint32_t lrand() { x = (0x5DEECE66DULL * x + 0xB) & 0xFFFFFFFFFFFFULL; return (uint32_t)(x >> 17); }
};

}  // namespace

extern "C" int fq_index_build(const char *fasta_path, int write_rollhash) {
  if (!fasta_path) return FQ_EINVAL;
  const std::string P = fasta_path;
  std::vector<FastaRec> recs;
  if (!read_reduced_fasta(P, recs)) return FQ_EIO;
  // ---- pac + annotations (Fa2Pac)
  std::vector<uint8_t> text;
  std::vector<FqContig> contigs;
  std::vector<FqHole> holes;
  std::vector<int> n_ambs;
  Rand48 rng(11);
  for (const auto &r : recs) {
    FqContig c;
    c.name = r.name;
    c.offset = contigs.empty() ? 0 : contigs.back().offset + contigs.back().len;
    c.len = (int32_t)r.seq.size();
    int lasts = 0, na = 0;
    for (size_t i = 0; i < r.seq.size(); ++i) {
      int code = nt4((unsigned char)r.seq[i]);
      if (code >= 4) {
        if (lasts == r.seq[i] && !holes.empty()) ++holes.back().len;
        else { holes.push_back({c.offset + (int64_t)i, 1, r.seq[i]}); ++na; }
        code = (int)(rng.lrand() & 3);
      }
      lasts = r.seq[i];
      text.push_back((uint8_t)code);
    }
    contigs.push_back(c);
    n_ambs.push_back(na);
  }
  const uint64_t n64 = text.size();
  if (n64 == 0 || n64 >= 0xfffffff0ull) return FQ_ELIMIT;
  const uint32_t n = (uint32_t)n64;
  {
    std::vector<uint8_t> pac(((size_t)n >> 2) + 1, 0);
    for (uint32_t i = 0; i < n; ++i) pac[i >> 2] |= (uint8_t)(text[i] << ((3 - (i & 3)) << 1));
    FILE *fp = fopen((P + ".pac").c_str(), "wb");
    if (!fp) return FQ_EIO;
    fwrite(pac.data(), 1, (n >> 2) + ((n & 3) == 0 ? 0 : 1), fp);
    uint8_t ct = 0;
    if (n % 4 == 0) fwrite(&ct, 1, 1, fp);
    ct = (uint8_t)(n % 4);
    fwrite(&ct, 1, 1, fp);
    fclose(fp);
    std::vector<uint8_t> rpac(((size_t)n >> 2) + 1, 0);
    for (uint32_t j = 0; j < n; ++j) rpac[j >> 2] |= (uint8_t)(text[n - 1 - j] << ((~j & 3) << 1));
    fp = fopen((P + ".rpac").c_str(), "wb");
    if (!fp) return FQ_EIO;
    fwrite(rpac.data(), 1, rpac.size(), fp);
    fwrite(&ct, 1, 1, fp);
    fclose(fp);
  }
  {
    FILE *fp = fopen((P + ".ann").c_str(), "w");
    if (!fp) return FQ_EIO;
    fprintf(fp, "%lld %d %u\n", (long long)n, (int)contigs.size(), 11u);
    for (size_t i = 0; i < contigs.size(); ++i) {
      fprintf(fp, "0 %s (null)\n", contigs[i].name.c_str());
      fprintf(fp, "%lld %d %d\n", (long long)contigs[i].offset, contigs[i].len, n_ambs[i]);
    }
    fclose(fp);
    fp = fopen((P + ".amb").c_str(), "w");
    if (!fp) return FQ_EIO;
    fprintf(fp, "%lld %d %u\n", (long long)n, (int)contigs.size(), (unsigned)holes.size());
    for (const auto &h : holes) fprintf(fp, "%lld %d %c\n", (long long)h.offset, h.len, h.amb);
    fclose(fp);
  }
  // ---- BWT + SA for the text and its reverse
  {
    BuiltFM fm;
    build_fm(text, fm);
    if (!write_bwt_file(P + ".bwt", fm, n) || !write_sa_file(P + ".sa", fm, n)) return FQ_EIO;
  }
  {
    std::vector<uint8_t> rev(text.rbegin(), text.rend());
    BuiltFM fm;
    build_fm(rev, fm);
    if (!write_bwt_file(P + ".rbwt", fm, n) || !write_sa_file(P + ".rsa", fm, n)) return FQ_EIO;
  }
  if (write_rollhash) {
    std::vector<uint32_t> bits[6];
    bitmap_bits_from_fasta(recs, bits);
    FILE *fp = fopen((P + ".rollhash").c_str(), "wb");
    if (!fp) return FQ_EIO;
    std::vector<uint8_t> tab((size_t)1 << 29);
    for (int t = 0; t < 6; ++t) {
      std::fill(tab.begin(), tab.end(), 0);
      for (uint32_t x : bits[t]) tab[x >> 3] |= (uint8_t)(1u << (x & 7));
      if (fwrite(tab.data(), 1, tab.size(), fp) != tab.size()) { fclose(fp); return FQ_EIO; }
    }
    fclose(fp);
  }
  return FQ_OK;
}

// ---- loader -------------------------------------------------------------------------------------------
namespace {
int load_fm(const std::string &prefix, const char *bext, const char *sext, fq_index *ix, int which) {
  std::vector<uint8_t> raw;
  if (!slurp(prefix + bext, raw) || raw.size() < 20) return FQ_EIO;
  const uint32_t *w = (const uint32_t *)raw.data();
  FqFM &f = ix->dev.fm[which];
  f.primary = w[0];
  f.L2[0] = 0;
  for (int c = 1; c <= 4; ++c) f.L2[c] = w[c];
  f.seq_len = f.L2[4];
  const uint32_t n = f.seq_len;
  const uint32_t *bw = w + 5;
  const size_t n_words = raw.size() / 4 - 5;
  f.n_blk = (n + 63) / 64 + 1;
  std::vector<FqOccBlk> blk(f.n_blk);
  uint32_t cnt[4] = {0, 0, 0, 0};
  for (uint32_t b = 0; b < f.n_blk; ++b) {
    FqOccBlk x;
    memcpy(x.cnt, cnt, 16);
    x.lo = x.hi = 0;
    for (uint32_t t = 0; t < 64; ++t) {
      const uint64_t i = (uint64_t)b * 64 + t;
      if (i >= n) break;
      const size_t wi = (size_t)(i >> 7) * 12 + 4 + ((i & 127) >> 4);
      if (wi >= n_words) return FQ_EIO;
      const uint32_t c = bw[wi] >> ((~i & 15) << 1) & 3;
      x.lo |= (uint64_t)(c & 1) << (63 - t);
      x.hi |= (uint64_t)(c >> 1) << (63 - t);
      ++cnt[c];
    }
    blk[b] = x;
  }
  for (int c = 0; c < 4; ++c) if (cnt[c] != f.L2[c + 1] - f.L2[c]) return FQ_EIO;   // counts must agree with C()
  ix->d_blk[which] = fqdev::dmalloc(blk.size() * sizeof(FqOccBlk));
  if (!ix->d_blk[which]) return FQ_ENOMEM;
  if (fqdev::h2d(ix->d_blk[which], blk.data(), blk.size() * sizeof(FqOccBlk)) || fqdev::sync()) return FQ_ENODEV;
  f.blk = (const FqOccBlk *)ix->d_blk[which];
  if (!slurp(prefix + sext, raw) || raw.size() < 28) return FQ_EIO;
  w = (const uint32_t *)raw.data();
  if (w[0] != f.primary || w[6] != f.seq_len) return FQ_EIO;
  f.sa_intv = w[5];
  if (f.sa_intv == 0) return FQ_EIO;
  f.n_sa = (f.seq_len + f.sa_intv) / f.sa_intv;
  if (raw.size() / 4 - 7 < f.n_sa - 1) return FQ_EIO;
  std::vector<uint32_t> sa(f.n_sa);
  sa[0] = 0xffffffffu;
  memcpy(sa.data() + 1, w + 7, (size_t)(f.n_sa - 1) * 4);
  ix->d_sa[which] = fqdev::dmalloc(sa.size() * 4);
  if (!ix->d_sa[which]) return FQ_ENOMEM;
  if (fqdev::h2d(ix->d_sa[which], sa.data(), sa.size() * 4) || fqdev::sync()) return FQ_ENODEV;
  f.sa = (const uint32_t *)ix->d_sa[which];
  return FQ_OK;
}
}  // namespace

extern "C" int fq_index_load(const char *prefix_c, int device_ordinal, fq_index_t **out) {
  if (!prefix_c || !out) return FQ_EINVAL;
  *out = nullptr;
  // FASTQUICK_TRACE=1: where the load's time goes, on stderr
  static const bool trace = [] { const char *e = getenv("FASTQUICK_TRACE"); return e && *e && *e != '0'; }();
  const auto t_load0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) { if (trace) fprintf(stderr, "TRACE -   index load %8.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load0).count(), what); };
  // the load has its own short-lived device state (stream for the staging copies); it ends with the load
  struct DevScope {
    fqdev::State *s;
    ~DevScope() { if (s) fqdev::state_destroy(s); }
  } scope{fqdev::state_create(device_ordinal)};
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  mark("device state (HIP runtime up)");
  std::unique_ptr<fq_index> ix(new fq_index);
  ix->prefix = prefix_c;
  ix->device = device_ordinal;
  const std::string P = prefix_c;
  int rc;
  auto fail = [&](int code) { fq_index_destroy(ix.release()); return code; };
  if ((rc = load_fm(P, ".bwt", ".sa", ix.get(), 0)) != FQ_OK) return fail(rc);
  if ((rc = load_fm(P, ".rbwt", ".rsa", ix.get(), 1)) != FQ_OK) return fail(rc);
  {  // .ann / .amb
    FILE *fp = fopen((P + ".ann").c_str(), "r");
    if (!fp) return fail(FQ_EIO);
    long long lp; int ns; unsigned seed;
    if (fscanf(fp, "%lld%d%u", &lp, &ns, &seed) != 3) { fclose(fp); return fail(FQ_EIO); }
    ix->l_pac = lp; ix->seed = seed;
    for (int i = 0; i < ns; ++i) {
      unsigned gi; char nm[2048]; long long off; int len, nambs, ch;
      if (fscanf(fp, "%u%2047s", &gi, nm) != 2) { fclose(fp); return fail(FQ_EIO); }
      while ((ch = fgetc(fp)) != '\n' && ch != EOF) {}
      if (fscanf(fp, "%lld%d%d", &off, &len, &nambs) != 3) { fclose(fp); return fail(FQ_EIO); }
      ix->contigs.push_back({nm, off, len});
    }
    fclose(fp);
    fp = fopen((P + ".amb").c_str(), "r");
    if (!fp) return fail(FQ_EIO);
    int nh;
    if (fscanf(fp, "%lld%d%d", &lp, &ns, &nh) != 3) { fclose(fp); return fail(FQ_EIO); }
    for (int i = 0; i < nh; ++i) {
      long long off; int len; char s[64];
      if (fscanf(fp, "%lld%d%63s", &off, &len, s) != 3) { fclose(fp); return fail(FQ_EIO); }
      ix->holes.push_back({off, len, s[0]});
    }
    fclose(fp);
  }
  if ((uint64_t)ix->l_pac != ix->dev.fm[0].seq_len || ix->dev.fm[0].seq_len != ix->dev.fm[1].seq_len) return fail(FQ_EIO);
  for (int c = 0; c < 5; ++c)   // a text and its reversal have the same base counts; the search kernel relies on one C() array
    if (ix->dev.fm[0].L2[c] != ix->dev.fm[1].L2[c]) return fail(FQ_EIO);
  if (!slurp(P + ".pac", ix->pac) || ix->pac.size() < (size_t)(ix->l_pac / 4)) return fail(FQ_EIO);
  ix->pac.resize((size_t)(ix->l_pac / 4) + 256, 0);
  ix->d_pac = fqdev::dmalloc(ix->pac.size());
  if (!ix->d_pac) return fail(FQ_ENOMEM);
  if (fqdev::h2d(ix->d_pac, ix->pac.data(), ix->pac.size()) || fqdev::sync()) return fail(FQ_ENODEV);
  ix->dev.pac = (const uint8_t *)ix->d_pac;
  ix->dev.l_pac = ix->l_pac;
  {   // the contig table and the N holes, for the consumers on the device (fq_emit.h)
    const size_t nc = ix->contigs.size(), nh = ix->holes.size();
    std::vector<int64_t> off(nc), hoff(nh + 1, 0);
    std::vector<int32_t> len(nc), hlen(nh + 1, 0);
    std::vector<uint32_t> noff(nc + 1, 0);
    std::string names;
    for (size_t i = 0; i < nc; ++i) { off[i] = ix->contigs[i].offset; len[i] = ix->contigs[i].len; noff[i] = (uint32_t)names.size(); names += ix->contigs[i].name; }
    noff[nc] = (uint32_t)names.size();
    names.push_back(0);
    for (size_t i = 0; i < nh; ++i) { hoff[i] = ix->holes[i].offset; hlen[i] = ix->holes[i].len; }
    auto up = [&](const void *src, size_t bytes) -> void * {
      void *d = fqdev::dmalloc(bytes ? bytes : 16);
      if (d) { ix->load_scratch.push_back(d); if (bytes && fqdev::h2d(d, src, bytes)) return nullptr; }
      return d;
    };
    FqDevContigs &C = ix->dev_contigs;
    C.n = (int32_t)nc; C.n_holes = (int32_t)nh;
    C.off = (const int64_t *)up(off.data(), nc * 8); C.len = (const int32_t *)up(len.data(), nc * 4);
    C.name_off = (const uint32_t *)up(noff.data(), (nc + 1) * 4); C.names = (const char *)up(names.data(), names.size());
    C.hole_off = (const int64_t *)up(hoff.data(), (nh + 1) * 8); C.hole_len = (const int32_t *)up(hlen.data(), (nh + 1) * 4);
    if (!C.off || !C.len || !C.name_off || !C.names || !C.hole_off || !C.hole_len || fqdev::sync()) return fail(FQ_ENODEV);
  }
  mark("FM index, SA, pac staged");
  // ---- six 2^32-bit filter tables, contiguous in HBM (3 GiB)
  const size_t TB = (size_t)1 << 29;
  ix->d_bitmap = fqdev::dmalloc(6 * TB);
  mark("3 GiB allocated");
  if (!ix->d_bitmap) return fail(FQ_ENOMEM);
  for (int t = 0; t < 6; ++t) ix->dev.bitmap[t] = (const uint8_t *)ix->d_bitmap + (size_t)t * TB;
  if (exists(P + ".rollhash")) {
    FILE *fp = fopen((P + ".rollhash").c_str(), "rb");
    const size_t CH = (size_t)64 << 20;
    uint8_t *stage = (uint8_t *)fqdev::hmalloc(CH);
    if (!fp || !stage) { if (fp) fclose(fp); fqdev::hfree(stage); return fail(FQ_EIO); }
    for (size_t off = 0; off < 6 * TB; off += CH) {
      if (fread(stage, 1, CH, fp) != CH) { fclose(fp); fqdev::hfree(stage); return fail(FQ_EIO); }
      if (fqdev::h2d((uint8_t *)ix->d_bitmap + off, stage, CH) || fqdev::sync()) { fclose(fp); fqdev::hfree(stage); return fail(FQ_ENODEV); }
    }
    fclose(fp);
    fqdev::hfree(stage);
  } else {
    std::vector<BitParts> parts;
    if (exists(P + ".rollhash.sparse")) {
      parts.resize(1);
      FILE *fp = fopen((P + ".rollhash.sparse").c_str(), "rb");
      for (int t = 0; t < 6; ++t) {
        uint64_t cnt;
        if (fread(&cnt, 8, 1, fp) != 1) { fclose(fp); return fail(FQ_EIO); }
        parts[0].bits[t].resize(cnt);
        if (cnt && fread(parts[0].bits[t].data(), 4, cnt, fp) != cnt) { fclose(fp); return fail(FQ_EIO); }
      }
      fclose(fp);
    } else {
      // The reference is in HBM already (pac) and every record's alleles are in its name: the device enters the 32-mers itself -- unless a base
      // or an allele is not one of ACGT (.amb lists the runs of other letters; the reference draws those from rand(), in record order: the host's walk)
      bool on_device = ix->holes.empty() && !ix->contigs.empty() && !getenv("FASTQUICK_HOST_BITMAPS");
      std::vector<int64_t> rec_off;
      std::vector<uint8_t> alle;
      if (on_device) {
        for (const auto &c : ix->contigs) {
          const size_t at = c.name.find('@');
          const int a0 = at != std::string::npos && at + 3 < c.name.size() + 1 ? nt4(c.name[at + 1]) : 4, a1 = a0 < 4 ? nt4(c.name[at + 3]) : 4;
          if (a0 > 3 || a1 > 3) { on_device = false; break; }
          rec_off.push_back(c.offset);
          alle.push_back((uint8_t)(a0 | a1 << 2));
        }
        rec_off.push_back(ix->l_pac);
        for (size_t k = 0; on_device && k + 1 < rec_off.size(); ++k) if (rec_off[k + 1] - rec_off[k] != ix->contigs[k].len) on_device = false;   // (records back to back)
      }
      if (on_device) {
        if (fqdev::dzero(ix->d_bitmap, 6 * TB)) return fail(FQ_ENODEV);
        int64_t *d_off = (int64_t *)fqdev::dmalloc(rec_off.size() * 8);
        uint8_t *d_al = (uint8_t *)fqdev::dmalloc(alle.size() + 8);
        if (!d_off || !d_al) { fqdev::dfree(d_off); fqdev::dfree(d_al); return fail(FQ_ENOMEM); }
        FqBitmapArgs ba{};
        ba.pac = (const uint8_t *)ix->d_pac; ba.l_pac = ix->l_pac; ba.rec_off = d_off; ba.alleles = d_al; ba.n_rec = (int)alle.size();
        for (int t = 0; t < 6; ++t) ba.bitmap[t] = (uint32_t *)((uint8_t *)ix->d_bitmap + (size_t)t * TB);
        int e = fqdev::h2d(d_off, rec_off.data(), rec_off.size() * 8);
        if (!e) e = fqdev::h2d(d_al, alle.data(), alle.size());
        if (!e) e = fqdev::launch_bitmap_kmers(ba);
        if (!e) e = fqdev::sync();
        ix->load_scratch.push_back(d_off); ix->load_scratch.push_back(d_al);
        if (e) return fail(FQ_ENODEV);
        mark("bitmaps filled by the device from the reference in HBM");
        ix->load_state = scope.s; scope.s = nullptr;
        *out = ix.release();
        return FQ_OK;
      }
      std::vector<FastaRec> recs;
      if (!read_reduced_fasta(P, recs)) return fail(FQ_EIO);
      bitmap_bits_from_fasta_threads(recs, parts);
    }
    mark("set bits listed (file, or the FASTA's k-mers)");
    if (fqdev::dzero(ix->d_bitmap, 6 * TB)) return fail(FQ_ENODEV);
    size_t most = 0;
    for (const auto &p : parts) for (int t = 0; t < 6; ++t) most = std::max(most, p.bits[t].size());
    uint32_t *d = most ? (uint32_t *)fqdev::dmalloc(most * 4) : nullptr;
    if (most && !d) return fail(FQ_ENOMEM);
    for (const auto &p : parts)
      for (int t = 0; t < 6; ++t) {
        if (p.bits[t].empty()) continue;
        int e = fqdev::h2d(d, p.bits[t].data(), p.bits[t].size() * 4);
        if (!e) e = fqdev::launch_bitmap_scatter((uint8_t *)ix->d_bitmap + (size_t)t * TB, d, p.bits[t].size());
        if (!e) e = fqdev::sync();
        if (e) { fqdev::dfree(d); return fail(FQ_ENODEV); }
      }
    fqdev::dfree(d);
  }
  mark("bitmaps filled");
  ix->load_state = scope.s; scope.s = nullptr;
  *out = ix.release();
  return FQ_OK;
}

// (tests) table t of the filter's six bitmaps as it stands in HBM: 2^29 bytes
extern "C" int fq_index_bitmap_fetch(const fq_index_t *ix, int32_t t, uint8_t *out) {
  if (!ix || !out || t < 0 || t > 5 || !ix->d_bitmap) return FQ_EINVAL;
  struct DevScope { fqdev::State *s; ~DevScope() { fqdev::state_destroy(s); } } scope{fqdev::state_create(ix->device)};
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  if (fqdev::d2h(out, (const uint8_t *)ix->d_bitmap + ((size_t)t << 29), (size_t)1 << 29) || fqdev::sync()) return FQ_ENODEV;
  return FQ_OK;
}
extern "C" void fq_index_destroy(fq_index_t *ix) {
  if (!ix) return;
  if (ix->load_state) { fqdev::state_destroy((fqdev::State *)ix->load_state); ix->load_state = nullptr; }
  for (void *p : ix->load_scratch) fqdev::dfree(p);
  ix->load_scratch.clear();
  for (int j = 0; j < 2; ++j) { fqdev::dfree(ix->d_blk[j]); fqdev::dfree(ix->d_sa[j]); }
  fqdev::dfree(ix->d_pac);
  fqdev::dfree(ix->d_bitmap);
  delete ix;
}
extern "C" int64_t fq_index_l_pac(const fq_index_t *ix) { return ix ? ix->l_pac : 0; }
extern "C" int32_t fq_index_n_contigs(const fq_index_t *ix) { return ix ? (int32_t)ix->contigs.size() : 0; }
extern "C" int fq_index_contig(const fq_index_t *ix, int32_t id, const char **name, int64_t *offset, int32_t *len) {
  if (!ix || id < 0 || id >= (int32_t)ix->contigs.size()) return FQ_EINVAL;
  if (name) *name = ix->contigs[id].name.c_str();
  if (offset) *offset = ix->contigs[id].offset;
  if (len) *len = ix->contigs[id].len;
  return FQ_OK;
}

int fq_coor_pac2real(const fq_index *ix, int64_t pos, int len, int *seqid) {
  int left = 0, mid = 0, right = (int)ix->contigs.size(), nn = 0;
  const int ns = right;
  while (left < right) {
    mid = (left + right) >> 1;
    if (pos >= ix->contigs[mid].offset) {
      if (mid == ns - 1) break;
      if (pos < ix->contigs[mid + 1].offset) break;
      left = mid + 1;
    } else right = mid;
  }
  *seqid = mid;
  left = 0; right = (int)ix->holes.size();
  while (left < right) {
    const int m = (left + right) >> 1;
    const FqHole &h = ix->holes[m];
    if (pos >= h.offset + h.len) left = m + 1;
    else if (pos + len <= h.offset) right = m;
    else {
      if (pos >= h.offset) nn += h.offset + h.len < pos + len ? (int)(h.offset + h.len - pos) : len;
      else nn += h.offset + h.len < pos + len ? h.len : (int)(len - (h.offset - pos));
      break;
    }
  }
  return nn;
}
