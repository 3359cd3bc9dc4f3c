// fq_frontend.cpp -- host side of the FASTQ front end on the device (kernels: fq_frontend.h).
//
// Replaces, for BGZF files of four-line records, the reader side of bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:476-613): gzread
// (libbwa/bwaseqio.c:41-52) under kseq_read3_fpc (libbwa/kseq.h:327-371) on one IO thread per file (IOworkerAlt, src/BwtMapper.cpp:1973-1980).
// The host reads compressed bytes and walks the members' headers (18 bytes per 64 KiB of text); everything per byte of text happens in HBM.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"
#include "fq_inflate.h"

namespace {
// ---- BGZF members of a compressed image (SAM spec 4.1: gzip members with a 'BC' extra subfield that states the member's size) ----
struct HostMember { size_t at, size, hdr; uint32_t isize, crc; };
// size of the member that starts at h (>= 18 readable bytes), 0 if it is not a BGZF member
size_t bgzf_member(const uint8_t *h, size_t n, size_t *hdr_len) {
  if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
  const size_t xlen = h[10] | (size_t)h[11] << 8;
  if (12 + xlen > n) return 0;
  size_t p = 12;
  while (p + 4 <= 12 + xlen) {
    const size_t sl = h[p + 2] | (size_t)h[p + 3] << 8;
    if (h[p] == 'B' && h[p + 1] == 'C' && sl == 2 && p + 6 <= 12 + xlen) { *hdr_len = 12 + xlen; return (size_t)(h[p + 4] | (size_t)h[p + 5] << 8) + 1; }
    p += 4 + sl;
  }
  return 0;
}
inline uint32_t le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
// the whole members of [p, p + n): false when something that is not a member stands at a member boundary
bool walk_members(const uint8_t *p, size_t n, std::vector<HostMember> &ms, size_t *consumed) {
  size_t at = 0;
  while (n - at >= 18) {
    size_t hdr = 0;
    const size_t sz = bgzf_member(p + at, n - at, &hdr);
    if (sz == 0) { *consumed = at; return false; }
    if (sz < hdr + 8) { *consumed = at; return false; }
    if (n - at < sz) break;
    ms.push_back({at, sz, hdr, le32(p + at + sz - 4), le32(p + at + sz - 8)});
    at += sz;
  }
  *consumed = at;
  return true;
}

struct DevScope {
  fqdev::State *s;
  explicit DevScope(int dev) : s(fqdev::state_create(dev)) {}
  ~DevScope() { fqdev::state_destroy(s); }
};
struct DevMem {
  void *p = nullptr;
  ~DevMem() { fqdev::dfree(p); }
  bool alloc(size_t n) { p = fqdev::dmalloc(n); return p != nullptr; }
};
}  // namespace

// ---- the member decoder on its own (tests, measurement) ----------------------------------------------------------------------------
// n_streams raw DEFLATE streams (RFC 1951) src[k] of n[k] bytes, each promising out_len[k] bytes with CRC-32 crc[k], inflated by the
// device's decoder -- one wavefront per stream -- into dst[k] (out_len[k] bytes each); status[k] = 0 inflated and checked, 1 refused
// (a stream the device does not take: the reader hands such a member to fq_inflate.h, then zlib), 2 CRC mismatch.
extern "C" int fq_inflate_device(int device, int n_streams, const uint8_t *const *src, const size_t *n, uint8_t *const *dst, const uint32_t *out_len, const uint32_t *crc,
                                 uint32_t *status, int repeats, double *kernel_ms) {
  if (n_streams < 0 || (n_streams && (!src || !n || !dst || !out_len || !crc || !status))) return FQ_EINVAL;
  DevScope scope(device);
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  std::vector<FqzMember> mem((size_t)n_streams);
  size_t c_total = 0, o_total = 0;
  for (int k = 0; k < n_streams; ++k) {
    if (n[k] > 0xffffffffull) return FQ_ELIMIT;
    if (o_total + out_len[k] > 0xfffffff0ull) return FQ_ELIMIT;      // (32-bit text positions inside one launch)
    mem[k].in_off = c_total; mem[k].in_len = (uint32_t)n[k]; mem[k].out_off = (uint32_t)o_total; mem[k].out_len = out_len[k]; mem[k].crc = crc[k]; mem[k].pad[0] = mem[k].pad[1] = 0;
    c_total += n[k] + (k % 3);       // (members of a file follow each other at any alignment)
    o_total += out_len[k];
  }
  std::vector<uint8_t> comp(c_total + 1024 + 16, 0);
  for (int k = 0; k < n_streams; ++k) if (n[k]) memcpy(comp.data() + mem[k].in_off, src[k], n[k]);
  DevMem d_comp, d_out, d_mem, d_status;
  if (!d_comp.alloc(comp.size()) || !d_out.alloc(o_total + 512) || !d_mem.alloc(sizeof(FqzMember) * (size_t)std::max(1, n_streams)) || !d_status.alloc(4 * (size_t)std::max(1, n_streams))) return FQ_ENOMEM;
  const FqzCrcConst *cc = fqdev::crc_const();
  if (!cc) return FQ_ENODEV;
  if (fqdev::h2d(d_comp.p, comp.data(), comp.size()) || fqdev::h2d(d_mem.p, mem.data(), sizeof(FqzMember) * mem.size()) || fqdev::dzero(d_out.p, o_total + 512) || fqdev::sync()) return FQ_ENODEV;
  FqInflateArgs a{};
  a.comp = (const uint8_t *)d_comp.p; a.mem = (const FqzMember *)d_mem.p; a.n_mem = n_streams; a.out = (uint8_t *)d_out.p; a.status = (uint32_t *)d_status.p; a.crc = cc;
  double ms[FQ_K_COUNT] = {0};
  uint64_t launches[FQ_K_COUNT] = {0};
  for (int r = 0; r < std::max(1, repeats); ++r) {
    fqdev::time_begin(0);
    if (fqdev::launch_inflate(a)) return FQ_ENODEV;
    fqdev::time_end(0);
  }
  if (fqdev::sync()) return FQ_ENODEV;
  fqdev::time_collect(ms, launches, FQ_K_COUNT);
  if (kernel_ms) *kernel_ms = launches[0] ? ms[0] / (double)launches[0] : 0.0;
  std::vector<uint8_t> out(o_total + 1);
  if (fqdev::d2h(out.data(), d_out.p, o_total) || fqdev::d2h(status, d_status.p, 4 * (size_t)n_streams) || fqdev::sync()) return FQ_ENODEV;
  for (int k = 0; k < n_streams; ++k) if (out_len[k]) memcpy(dst[k], out.data() + mem[k].out_off, out_len[k]);
  return FQ_OK;
}

// The members of a BGZF image (a whole file, or any run of whole members) inflated on the device into `out` (their text back to back).
// status[k] per member as above (NULL: not wanted); *n_members, *text_len: what was found.  FQ_EIO: not a run of BGZF members;
// FQ_ELIMIT: out_cap too small.
extern "C" int fq_bgzf_inflate_device(int device, const uint8_t *file, size_t n, uint8_t *out, size_t out_cap, int64_t *n_members, int64_t *text_len, uint32_t *status, int64_t status_cap,
                                      int repeats, double *kernel_ms) {
  if (!file || !out) return FQ_EINVAL;
  std::vector<HostMember> ms;
  size_t used = 0;
  if (!walk_members(file, n, ms, &used) || used != n) return FQ_EIO;
  std::vector<const uint8_t *> src;
  std::vector<size_t> len;
  std::vector<uint8_t *> dst;
  std::vector<uint32_t> olen, crc;
  size_t o = 0;
  for (const auto &m : ms) {
    if (m.isize == 0) continue;          // (the empty end-of-file member)
    src.push_back(file + m.at + m.hdr); len.push_back(m.size - m.hdr - 8); dst.push_back(out + o); olen.push_back(m.isize); crc.push_back(m.crc);
    o += m.isize;
    if (o > out_cap) return FQ_ELIMIT;
  }
  if (n_members) *n_members = (int64_t)src.size();
  if (text_len) *text_len = (int64_t)o;
  std::vector<uint32_t> st(src.size() + 1);
  const int rc = fq_inflate_device(device, (int)src.size(), src.data(), len.data(), dst.data(), olen.data(), crc.data(), st.data(), repeats, kernel_ms);
  if (rc) return rc;
  if (status) memcpy(status, st.data(), 4 * (size_t)std::min<int64_t>(status_cap, (int64_t)src.size()));
  return FQ_OK;
}

// =====================================================================================================================================
// The reader: one FASTQ pair (or one single-end file) -> batches whose text, records, filter keys and names are resident in HBM.
//
//   reader threads (one per file)   pread the compressed bytes of the next chunk into pinned staging, walk the members' headers, upload
//   the producer thread             inflates the chunk's members (k_inflate_bgzf; a member the device refuses: fq_inflate.h, then zlib),
//                                   indexes the line ends, checks and cuts the records, forms the filter's keys, walks the read slots,
//                                   writes the names -- and queues the batch
//   fq_frontend_next()              hands the next batch to the caller (who gives it to fq_align_text)
//
// A chunk holds whole reference batches (batch_pairs: insert sizes are inferred per reference batch, libbwa/bwape.c:49-117) unless it is
// the stream's last.  Text behind a chunk's last record stays in HBM and leads the next chunk.  Whatever the device does not take -- a
// record that is not four plain lines, a file that ends inside a record, a member neither decoder accepts -- ends the device's part of the
// stream at a reference-batch boundary: fq_frontend_next() returns FQ_EFALLBACK and fq_frontend_handover() gives the host readers
// (fq_fastq.cpp) standing exactly there, read slots included; their verdicts stand.
// =====================================================================================================================================
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <zlib.h>

#include "fq_fastq_internal.h"

#include "fq_text_batch.h"

namespace {
template <class T> struct DBuf {
  T *p = nullptr; size_t cap = 0;
  ~DBuf() { fqdev::dfree(p); }
  bool ensure(size_t n) {
    if (n <= cap) return true;
    fqdev::dfree(p);
    cap = n + n / 8 + 256;
    p = (T *)fqdev::dmalloc(cap * sizeof(T));
    if (!p) cap = 0;
    return p != nullptr;
  }
  bool ensure_keep(size_t n, size_t keep) {   // the first `keep` elements survive (copied on the bound state's stream, then waited for)
    if (n <= cap) return true;
    const size_t ncap = n + n / 8 + 256;
    T *q = (T *)fqdev::dmalloc(ncap * sizeof(T));
    if (!q) return false;
    if (keep && (fqdev::d2d(q, p, keep * sizeof(T)) || fqdev::sync())) { fqdev::dfree(q); return false; }
    fqdev::dfree(p); p = q; cap = ncap;
    return true;
  }
};
template <class T> struct HBuf {
  T *p = nullptr; size_t cap = 0;
  ~HBuf() { fqdev::hfree(p); }
  bool ensure(size_t n) {
    if (n <= cap) return true;
    fqdev::hfree(p);
    cap = n + n / 8 + 256;
    p = (T *)fqdev::hmalloc(cap * sizeof(T));
    if (!p) cap = 0;
    return p != nullptr;
  }
};
const size_t kPiece = (size_t)32 << 20;       // pinned staging pieces of the compressed stream
const uint64_t kMaxText = (uint64_t)3500 << 20;   // text positions are 32-bit inside a launch
const int kSlots = 4;                          // batches: two with the caller (one under its kernels, one under its consumers), one being tokenised, one being inflated

struct CompChunk {                             // one file's compressed bytes of one chunk, in HBM, with their member table
  DBuf<uint8_t> d_comp;
  DBuf<FqzMember> d_mem;
  DBuf<uint32_t> d_status;
  std::vector<FqzMember> mem;                  // out_off relative to the chunk's first new byte of text
  size_t comp_len = 0;
  uint64_t text_len = 0;
  int64_t file_off_end = 0;                    // file offset behind the chunk's last member
  bool eof = false;
  std::string err;
};
struct FileSide {
  std::string path;
  int fd = -1;
  int64_t file_off = 0;                        // next byte to read
  bool file_eof = false;
  std::vector<uint8_t> tail;                   // bytes of an incomplete member at the end of what was read
  fqdev::State *st = nullptr;                  // the reader thread's device state (its stream carries the uploads)
  HBuf<uint8_t> pin[2];
  CompChunk chunk[3];                          // the reader runs up to two chunks ahead of the inflation
  std::thread th;
  // hand-shake with the producer: chunk slots are filled by the reader in turn and released by the producer in the same order
  std::mutex mu;
  std::condition_variable cv;
  bool filled[3] = {false, false, false};
  uint64_t want_text = 0;                      // text a chunk should hold (the producer's latest estimate)
  bool stop = false;
  int64_t inflated_file_off = 0;               // the file offset behind the last chunk that has been inflated
  // text
  DBuf<uint8_t> d_text[kSlots];                // one per batch slot
  DBuf<uint32_t> d_nl;
  uint64_t carry_off = 0, carry_len = 0;       // the text behind the last chunk's records: in d_text[carry_slot]
  int carry_slot = -1;
  int64_t records_done = 0;                    // records of this file in batches so far
  double text_per_record = 0;
  bool all_launched = false;                   // every member of the file has been handed to the decoder
  // slots
  DBuf<uint8_t> d_slot_base, d_slot_name;
  DBuf<uint16_t> d_slot_len;
  DBuf<uint32_t> d_stat;
  int shorter_after_longer = 0;
  uint64_t file_size = 0;
  double comp_ratio = 0.4;                     // compressed bytes per byte of text (the first member's; then the last chunk's)
  int chunks_read = 0;                         // (by the reader thread)
  int read_threads = 1;                        // threads of one pread_threads call
  double ms_read = 0, ms_upload = 0;           // the reader thread's time in pread / in uploads (fq_frontend_stats)
};
inline double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
}  // namespace

struct fq_frontend {
  int device = 0, n_files = 0, slot_mode = FQ_FASTQ_SLOTS_REUSED, batch_pairs = 262144;
  int64_t chunk_pairs = 16 * 262144;
  int max_len = 160;                           // rows the aligner sizes: a longer read ends the device's part
  FileSide f[2];
  fqdev::State *st = nullptr;                  // the producer's
  std::thread producer;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<fq_text_batch *> ready;
  bool slot_free[kSlots] = {true, true, true, true};
  uint64_t headroom_min = (uint64_t)64 << 20;   // least room kept in front of a chunk's new text for the carried text (FASTQUICK_FE_HEADROOM: tests)
  bool overlap = true;                         // the next chunk's members are inflated beside this chunk's kernels (FASTQUICK_FE_OVERLAP=0: one after the other)
  bool done = false, fallback = false, stop = false;
  std::string err;
  int rc = FQ_OK;
  fq_text_batch batch[kSlots];
  DBuf<FqTextRec> d_rec[kSlots];
  DBuf<uint64_t> d_head[kSlots];
  DBuf<uint16_t> d_hlen[kSlots];
  DBuf<char> d_names[kSlots];
  HBuf<uint32_t> h_stat;
  int max_name_ever = 0, min_name_ever = INT32_MAX;
  int64_t pairs_done = 0;
  int next_slot = 0;
  // totals (fq_frontend_stats)
  double ms_inflate = 0, ms_tokenise = 0, ms_lines = 0, ms_records = 0, ms_slots = 0;
  int64_t n_members = 0, n_refused = 0, text_bytes = 0, comp_bytes = 0, n_launch_inflate = 0, n_chunks = 0;
  double ms_wait_reader = 0, ms_wait_slot = 0;  // the producer's waits: for a chunk's compressed bytes; for a batch slot the caller still holds
  ~fq_frontend();
};

namespace {

// A piece of the file into (pinned) memory on several threads: a read from the page cache is a memcpy by the kernel, 8-12 GB/s on one
// thread -- slower than the device inflates what it brings.  Returns the bytes read from `off` on (short only at the end of the file).
ssize_t pread_threads(int fd, uint8_t *dst, size_t len, off_t off, int threads) {
  auto whole = [fd](uint8_t *d, size_t n, off_t o) -> ssize_t {
    size_t done = 0;
    while (done < n) {
      const ssize_t g = pread(fd, d + done, n - done, o + (off_t)done);
      if (g < 0) { if (errno == EINTR) continue; return -1; }
      if (g == 0) break;
      done += (size_t)g;
    }
    return (ssize_t)done;
  };
  const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, threads), len >> 22));
  if (nt == 1) return whole(dst, len, off);
  const size_t per = ((len + nt - 1) / nt + 4095) & ~(size_t)4095;
  std::vector<ssize_t> got((size_t)nt, 0);
  std::vector<std::thread> th;
  for (int t = 1; t < nt; ++t)
    th.emplace_back([&, t] { const size_t lo = per * t; got[(size_t)t] = lo < len ? whole(dst + lo, std::min(per, len - lo), off + (off_t)lo) : 0; });
  got[0] = whole(dst, std::min(per, len), off);
  for (auto &x : th) x.join();
  ssize_t total = 0;
  for (int t = 0; t < nt; ++t) {
    if (got[(size_t)t] < 0) return -1;
    total += got[(size_t)t];
    if ((size_t)got[(size_t)t] < std::min(per, len - std::min(len, per * t))) break;      // the file ended inside this slice
  }
  return total;
}

// ---- reader thread: the next chunk's compressed bytes -> pinned staging -> HBM, with the member table -------------------------------
void reader_main(fq_frontend *fe, int e) {
  FileSide &F = fe->f[e];
  if (fqdev::bind(F.st)) { std::lock_guard<std::mutex> lk(F.mu); for (int k = 0; k < 3; ++k) { F.chunk[k].err = "no device"; F.filled[k] = true; } F.cv.notify_all(); return; }
  for (int k = 0;; k = (k + 1) % 3) {
    uint64_t want;
    {
      std::unique_lock<std::mutex> lk(F.mu);
      F.cv.wait(lk, [&] { return F.stop || !F.filled[k]; });
      if (F.stop) return;
      want = F.want_text;
    }
    // the stream's first chunks are short ones (an eighth, a quarter, a half of a chunk): the first batch is out after an eighth of the time,
    // and the kernels behind the reader have work while it reads
    static const int ramp = [] { const char *e = getenv("FASTQUICK_FE_RAMP"); const int v = e ? atoi(e) : 3; return v < 0 ? 0 : v > 5 ? 5 : v; }();      // (experiment knob: how many short chunks lead a stream)
    if (F.chunks_read < ramp) want = std::max<uint64_t>(want >> (ramp - F.chunks_read), (uint64_t)((double)fe->batch_pairs * F.text_per_record * 1.02) + (1u << 20));   // (a reference batch at least)
    ++F.chunks_read;
    CompChunk &C = F.chunk[k];
    C.mem.clear(); C.comp_len = 0; C.text_len = 0; C.err.clear(); C.eof = false;
    // (sized in one go for what the chunk is expected to hold: growing the buffer under the uploads means waiting for them, and a hipFree waits for the device)
    (void)C.d_comp.ensure((size_t)std::min<double>((double)F.file_size - (double)F.file_off + (double)F.tail.size(), (double)want * F.comp_ratio * 1.1) + 2 * kPiece);
    // pieces: [tail of the previous read | new bytes] -> whole members go to the device, the rest is the next tail
    uint64_t text = 0;
    size_t dev_at = 0;
    int pin_i = 0;
    bool pin_used[2] = {false, false};
    bool more = true;
    while (more && C.err.empty()) {
      HBuf<uint8_t> &P = F.pin[pin_i];
      size_t have = F.tail.size();
      if (pin_used[pin_i]) { if (fqdev::copy_wait(pin_i)) { C.err = "upload failed"; break; } pin_used[pin_i] = false; }   // (the upload that last read this buffer)
      if (!P.ensure(have + kPiece + (1 << 17))) { C.err = "out of pinned host memory"; break; }
      if (have) memcpy(P.p, F.tail.data(), have);
      F.tail.clear();
      if (!F.file_eof && have < kPiece) {                    // (a long tail -- the rest of a piece whose chunk was full -- is used up first)
        const auto t0 = std::chrono::steady_clock::now();
        const ssize_t got = pread_threads(F.fd, P.p + have, kPiece, (off_t)F.file_off, F.read_threads);
        F.ms_read += ms_since(t0);
        if (got < 0) { C.err = "read error"; break; }
        if (got == 0) F.file_eof = true;
        F.file_off += got;
        have += (size_t)got;
      }
      // whole members of the piece
      size_t at = 0;
      while (have - at >= 18) {
        size_t hdr = 0;
        const size_t sz = bgzf_member(P.p + at, have - at, &hdr);
        if (sz == 0 || sz < hdr + 8) { C.err = "not a BGZF member at a member boundary (a bgzip file followed by something else?)"; break; }
        if (have - at < sz) break;
        const uint32_t isize = le32(P.p + at + sz - 4);
        if (isize) {
          FqzMember m{};
          m.in_off = dev_at + at + hdr; m.in_len = (uint32_t)(sz - hdr - 8); m.out_off = (uint32_t)text; m.out_len = isize; m.crc = le32(P.p + at + sz - 8);
          C.mem.push_back(m);
          text += isize;
        }
        at += sz;
        if (text >= want || text >= kMaxText) { more = false; break; }
      }
      if (!C.err.empty()) break;
      // the members' bytes go up; what is behind them waits for the next piece
      if (at) {
        if (C.d_comp.cap < dev_at + at + 2048) {               // (the uploads so far land before the buffer moves)
          for (int i = 0; i < 2; ++i) if (pin_used[i] && fqdev::copy_wait(i)) C.err = "upload failed";
          if (!C.err.empty()) break;
        }
        if (!C.d_comp.ensure_keep(dev_at + at + 2048, dev_at)) { C.err = "out of device memory"; break; }
        // (on the copy stream, with an event per staging buffer: the next piece is read while this one goes up)
        const auto t0 = std::chrono::steady_clock::now();
        if (fqdev::h2d_copy(C.d_comp.p + dev_at, P.p, at) || fqdev::copy_record(pin_i)) { C.err = "upload failed"; break; }
        pin_used[pin_i] = true;
        F.ms_upload += ms_since(t0);
        dev_at += at;
      }
      F.tail.assign(P.p + at, P.p + have);
      if (more && F.file_eof) {                              // the file has ended: every whole member has been taken, nothing may be left
        if (!F.tail.empty()) C.err = "truncated BGZF member at the end of the file";
        more = false;
      }
      pin_i ^= 1;
    }
    {
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 2; ++i) if (pin_used[i] && fqdev::copy_wait(i) && C.err.empty()) C.err = "upload failed";
      F.ms_upload += ms_since(t0);
    }
    if (C.err.empty()) {
      C.file_off_end = F.file_off - (int64_t)F.tail.size();      // the first byte of the file that is not in this chunk
      C.comp_len = dev_at;
      C.text_len = text;
      if (text > (1u << 20)) F.comp_ratio = (double)dev_at / (double)text;
      C.eof = F.file_eof && F.tail.empty();
      if (!C.d_mem.ensure(C.mem.size() + 1) || !C.d_status.ensure(C.mem.size() + 1)) C.err = "out of device memory";
      if (C.err.empty() && dev_at && (fqdev::dzero(C.d_comp.p + dev_at, 1024) || fqdev::sync())) C.err = "upload failed";
    }
    const bool done = !C.err.empty() || C.eof;
    {
      std::lock_guard<std::mutex> lk(F.mu);
      F.filled[k] = true;
    }
    F.cv.notify_all();
    if (done) return;
  }
}

// One chunk's inflation, started one chunk ahead of the kernels that read its text (producer_main)
struct Ahead {
  bool open = false;                // a batch slot is taken and the members of the files that still have some are being inflated
  int slot = -1, comp_k = -1;
  bool has[2] = {false, false};     // the file had a compressed chunk for this one
  uint64_t H[2] = {0, 0};           // where its new text begins in d_text[slot] (a multiple of 256: the carried text goes in front of it)
};

// the producer: chunk after chunk until the end of the stream, a failure, or something the device does not take.
// Two streams: the members of chunk j + 1 are inflated (aux stream) while chunk j's text is indexed, checked and keyed (main stream) -- the
// decoder is bound by the latency of its own dependent steps and leaves the memory system to the kernels beside it.  A chunk's text begins with
// what the chunk before it left over behind its last whole reference batch, and how much that is is known only when that chunk has been
// tokenised: the new text is inflated at a fixed distance H from the buffer's start and the carried text copied in front of it afterwards.
void producer_main(fq_frontend *fe) {
  auto finish = [&](int rc, const std::string &err, bool fallback) {
    std::lock_guard<std::mutex> lk(fe->mu);
    if (rc) { fe->rc = rc; fe->err = err; }
    fe->fallback = fallback;
    fe->done = true;
    fe->cv.notify_all();
  };
  if (fqdev::bind(fe->st)) { finish(FQ_ENODEV, "no device", false); return; }
  if (fe->overlap && fqdev::stream_aux(1)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }   // (creates the second stream)
  fqdev::stream_aux(0);
  const int NF = fe->n_files;
  const int B = fe->batch_pairs;
  const int n_slots = 2 * B;
  int comp_next = 0;                 // the compressed slot the next chunk's members are in
  int64_t n_started = 0, n_taken = 0;   // chunks whose inflation has been started / whose text has been taken up (their marks alternate)
  Ahead next;
  // Starts the next chunk's inflation.  block: wait for a batch slot and for the readers; otherwise only when both are there already.
  // Returns 1 started (or nothing is left to inflate: the chunk is its carried text), 0 not now, -1 failed / stopped (finish() has been called).
  auto start_ahead = [&](bool block) -> int {
    if (next.open) return 1;
    {
      const auto t0 = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(fe->mu);
      if (!block && !(fe->stop || fe->slot_free[fe->next_slot])) return 0;
      fe->cv.wait(lk, [&] { return fe->stop || fe->slot_free[fe->next_slot]; });
      fe->ms_wait_slot += ms_since(t0);
      if (fe->stop) return -1;
    }
    CompChunk *Cs[2] = {nullptr, nullptr};
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      if (F.all_launched) continue;
      const auto t0 = std::chrono::steady_clock::now();
      std::unique_lock<std::mutex> lk(F.mu);
      if (!block && !F.filled[comp_next]) return 0;
      F.cv.wait(lk, [&] { return F.stop || F.filled[comp_next]; });
      fe->ms_wait_reader += ms_since(t0);
      if (!F.filled[comp_next]) return -1;                   // (the front end is being closed in the middle of the stream)
      Cs[e] = &F.chunk[comp_next];
    }
    int slot;
    {
      std::lock_guard<std::mutex> lk(fe->mu);
      slot = fe->next_slot;
      fe->slot_free[slot] = false;
      fe->next_slot = (slot + 1) % kSlots;
    }
    next = Ahead();
    next.open = true; next.slot = slot; next.comp_k = comp_next;
    FqInflateArgs ia[2] = {FqInflateArgs(), FqInflateArgs()};
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      CompChunk *C = Cs[e];
      if (!C) continue;
      next.has[e] = true;
      if (!C->err.empty()) { finish(FQ_EIO, F.path + ": " + C->err, false); return -1; }
      if (C->eof) F.all_launched = true;
      // room in front of the new text for what the chunk before leaves over: a reference batch's worth and a half, 64 MB at least
      const uint64_t H = ((uint64_t)std::max<double>((double)fe->headroom_min, fe->headroom_min ? 1.5 * (double)B * F.text_per_record : 0.0) + 255) & ~(uint64_t)255;
      next.H[e] = H;
      if (H + C->text_len > 0xfff00000ull) { finish(FQ_ELIMIT, "a chunk's text exceeds 4 GiB", false); return -1; }
      if (!F.d_text[slot].ensure((size_t)(H + C->text_len) + 4096)) { finish(FQ_ENOMEM, "out of device memory (text)", false); return -1; }
      if (C->mem.empty()) continue;
      for (auto &m : C->mem) m.out_off += (uint32_t)H;       // (256-byte alignment of the buffer's base is what the decoder's stores rely on: the shift is in the members' offsets)
      FqInflateArgs &a = ia[e];
      a.comp = C->d_comp.p; a.mem = C->d_mem.p; a.n_mem = (int)C->mem.size(); a.out = F.d_text[slot].p; a.status = C->d_status.p;
    }
    if (ia[0].n_mem > 0 || ia[1].n_mem > 0) {
      // one launch for the two files' members: the last of a launch's rounds of wavefronts is the fuller for it
      if (fe->overlap) fqdev::stream_aux(1);
      bool bad = false;
      for (int e = 0; e < NF; ++e) {
        if (ia[e].n_mem <= 0) continue;
        CompChunk *C = Cs[e];
        bad = bad || fqdev::h2d(C->d_mem.p, C->mem.data(), C->mem.size() * sizeof(FqzMember));
        ia[e].crc = fqdev::crc_const();
        bad = bad || !ia[e].crc;
      }
      if (!bad) {
        fqdev::time_begin(0);
        bad = fqdev::launch_inflate2(ia[0], ia[1]) != 0;
        fqdev::time_end(0);
      }
      fqdev::stream_aux(0);
      if (bad) { finish(FQ_ENODEV, fqdev::last_error(), false); return -1; }
    }
    if (fe->overlap) { fqdev::stream_aux(1); const int rc = fqdev::stream_mark((int)(n_started & 1)); fqdev::stream_aux(0); if (rc) { finish(FQ_ENODEV, fqdev::last_error(), false); return -1; } }
    ++n_started;
    comp_next = (comp_next + 1) % 3;
    return 1;
  };
  for (;;) {
    // ---- this chunk: its members are being inflated already, or are started now ----
    if (start_ahead(true) < 0) return;
    const Ahead cur = next;
    next = Ahead();
    const int slot = cur.slot;
    fq_text_batch &TB = fe->batch[slot];
    TB = fq_text_batch();
    TB.slot = slot; TB.device = fe->device; TB.single_end = NF == 1; TB.batch_pairs = B; TB.row_cap = fe->max_len;
    uint32_t n_lines[2] = {0, 0};
    uint64_t n_text[2] = {0, 0};      // bytes of the chunk's view: [0, skip) in front of the text for the alignment, then carried text, then new text
    uint32_t skip[2] = {0, 0};
    uint64_t view[2] = {0, 0};        // where the view begins in d_text[slot] (a multiple of 16)
    bool at_eof[2] = {true, true};
    double t_ms[FQ_K_COUNT] = {0};
    uint64_t t_n[FQ_K_COUNT] = {0};
    if (fe->overlap && fqdev::stream_wait_mark((int)(n_taken & 1))) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
    ++n_taken;
    // ---- the carried text in front of the new text ----
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      CompChunk *C = cur.has[e] ? &F.chunk[cur.comp_k] : nullptr;
      const uint64_t new_text = C ? C->text_len : 0;
      uint64_t H = C ? cur.H[e] : ((F.carry_len + 255) & ~(uint64_t)255);
      if (!C && !F.d_text[slot].ensure((size_t)(H + 4096))) { finish(FQ_ENOMEM, "out of device memory (text)", false); return; }
      if (C && F.carry_len > H) {
        // more was left over than there is room in front of the new text (reads of kilobytes; files whose records differ much in size): the new
        // text moves back, through a second buffer
        const uint64_t H2 = (F.carry_len + 255) & ~(uint64_t)255;
        DBuf<uint8_t> tmp;
        if (H2 + new_text > 0xfff00000ull) { finish(FQ_ELIMIT, "a chunk's text exceeds 4 GiB", false); return; }
        if (!tmp.ensure((size_t)new_text + 256) || fqdev::d2d(tmp.p, F.d_text[slot].p + H, (size_t)new_text) || fqdev::sync() ||
            !F.d_text[slot].ensure_keep((size_t)(H2 + new_text) + 4096, 0) || fqdev::d2d(F.d_text[slot].p + H2, tmp.p, (size_t)new_text) || fqdev::sync()) { finish(FQ_ENOMEM, "out of device memory (text)", false); return; }
        for (auto &m : C->mem) m.out_off += (uint32_t)(H2 - H);
        H = H2;
      }
      const uint64_t t0 = H - F.carry_len;
      view[e] = t0 & ~(uint64_t)15; skip[e] = (uint32_t)(t0 & 15);
      n_text[e] = skip[e] + F.carry_len + new_text;
      if (F.carry_len && fqdev::d2d(F.d_text[slot].p + t0, F.d_text[F.carry_slot].p + F.carry_off, (size_t)F.carry_len)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      if (C) F.inflated_file_off = C->file_off_end;
      at_eof[e] = C ? C->eof : true;
    }
    // ---- members the device refused: the host's decoder, then zlib, whose verdict stands ----
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      if (!cur.has[e]) continue;
      CompChunk &C = F.chunk[cur.comp_k];
      if (C.mem.empty()) continue;
      std::vector<uint32_t> status(C.mem.size());
      if (fqdev::d2h(status.data(), C.d_status.p, status.size() * 4) || fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      std::unique_ptr<fqz::Inflater> fz;
      std::vector<uint8_t> tmp, cbytes;
      for (size_t k = 0; k < status.size(); ++k) {
        if (status[k] == FQZ_OK) continue;
        ++TB.refused;
        const FqzMember &m = C.mem[k];
        cbytes.resize(m.in_len + 8);
        if (fqdev::d2h(cbytes.data(), C.d_comp.p + m.in_off, m.in_len) || fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        const uint8_t *src = cbytes.data();
        tmp.resize(m.out_len);
        bool ok = false;
        if (!fz) fz.reset(new fqz::Inflater);
        if (fqz::inflate_raw(*fz, src, m.in_len, tmp.data(), m.out_len)) ok = fqz::crc32(tmp.data(), m.out_len) == m.crc;
        else {
          z_stream zs;
          memset(&zs, 0, sizeof zs);
          if (inflateInit2(&zs, -15) == Z_OK) {
            zs.next_in = const_cast<Bytef *>(src); zs.avail_in = m.in_len; zs.next_out = tmp.data(); zs.avail_out = m.out_len;
            const int zr = inflate(&zs, Z_FINISH);
            ok = zr == Z_STREAM_END && zs.avail_out == 0 && (uint32_t)crc32(crc32(0L, Z_NULL, 0), tmp.data(), m.out_len) == m.crc;
            inflateEnd(&zs);
          }
        }
        if (!ok) { finish(FQ_EIO, F.path + ": corrupt BGZF member (inflate or CRC failed)", false); return; }
        if (fqdev::h2d(F.d_text[slot].p + m.out_off, tmp.data(), m.out_len) || fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      }
      TB.members += (int64_t)C.mem.size(); TB.comp_bytes += (int64_t)C.comp_len; TB.text_bytes += (int64_t)C.text_len;
    }
    // ---- the compressed slot is free again: the reader threads run up to two chunks ahead ----
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      if (!cur.has[e]) continue;
      std::lock_guard<std::mutex> lk(F.mu);
      F.filled[cur.comp_k] = false;
      F.cv.notify_all();
    }
    // ---- the next chunk's members, beside this chunk's kernels (when its compressed bytes and a batch slot are there already) ----
    if (fe->overlap && start_ahead(false) < 0) return;
    // ---- line ends ----
    fqdev::time_begin(1);
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      const size_t cap = (size_t)(4 * (fe->chunk_pairs + B) + 64);
      if (!F.d_nl.ensure(cap) || !F.d_stat.ensure(FQT_N_STAT + 8)) { finish(FQ_ENOMEM, "out of device memory (line index)", false); return; }
      uint8_t *T = F.d_text[slot].p + view[e];
      if (fqdev::dzero(T + n_text[e], 256)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      if (fqdev::launch_nl_index(T, (uint32_t)n_text[e], skip[e], F.d_nl.p, (uint32_t)cap, F.d_stat.p + FQT_N_STAT)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
    }
    fqdev::time_end(1);
    if (!fe->h_stat.ensure(64)) { finish(FQ_ENOMEM, "out of pinned host memory", false); return; }
    for (int e = 0; e < NF; ++e) if (fqdev::copy_pinned(fe->h_stat.p + 32 * e, fe->f[e].d_stat.p + FQT_N_STAT, 4, 0)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
    if (fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
    for (int e = 0; e < NF; ++e) n_lines[e] = fe->h_stat.p[32 * e];
    // ---- how many pairs this chunk holds ----
    int64_t n = fe->chunk_pairs;
    bool all_eof = true;
    for (int e = 0; e < NF; ++e) {
      const size_t cap_lines = (size_t)(4 * (fe->chunk_pairs + B));
      n = std::min<int64_t>(n, (int64_t)(std::min<size_t>(n_lines[e], cap_lines) / 4));
      all_eof = all_eof && at_eof[e];
    }
    bool last = false;
    if (all_eof) {
      // the stream's end is in sight: every whole record that is left goes into this chunk when they fit it
      int64_t rest = INT64_MAX;
      for (int e = 0; e < NF; ++e) rest = std::min<int64_t>(rest, (int64_t)(n_lines[e] / 4));
      if (rest <= fe->chunk_pairs) { n = rest; last = true; }
    }
    if (!last) n = n / B * B;
    bool fall = false;
    // ---- records, keys (again after a cut: the rows of the second file begin at n) ----
    int n_rows = (int)(NF * n);
    while (n > 0) {
      n_rows = (int)(NF * n);
      if (!fe->d_rec[slot].ensure((size_t)n_rows + 1) || !fe->d_head[slot].ensure((size_t)n_rows * 3 + 8) || !fe->d_hlen[slot].ensure((size_t)n_rows + 8)) { finish(FQ_ENOMEM, "out of device memory (records)", false); return; }
      for (int e = 0; e < NF; ++e) {
        FileSide &F = fe->f[e];
        const uint32_t init[FQT_N_STAT] = {0xffffffffu, 0, 0xffffffffu, 0, 0, 0xffffffffu, 0, 0};
        if (fqdev::h2d(F.d_stat.p, init, sizeof init)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        FqTokArgs a{};
        a.text = F.d_text[slot].p + view[e]; a.text0 = skip[e]; a.nl = F.d_nl.p; a.n_rec = (int)n; a.row0 = (int)(e * n); a.n_rows = n_rows; a.max_len = fe->max_len;
        a.rec = fe->d_rec[slot].p; a.head = fe->d_head[slot].p; a.hlen = fe->d_hlen[slot].p; a.stat = F.d_stat.p;
        fqdev::time_begin(2);
        if (fqdev::launch_tok_rec(a) || fqdev::launch_tok_pieces(a)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        fqdev::time_end(2);
        if (fqdev::copy_pinned(fe->h_stat.p + 32 * e, F.d_stat.p, 4 * FQT_N_STAT, 0)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      }
      if (fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      int64_t first_bad = INT64_MAX;
      for (int e = 0; e < NF; ++e) if (fe->h_stat.p[32 * e + FQT_FIRST_BAD] != 0xffffffffu) first_bad = std::min<int64_t>(first_bad, (int64_t)fe->h_stat.p[32 * e + FQT_FIRST_BAD]);
      if (first_bad >= n) break;
      // a record the device does not take: its reference batch, and everything behind it, goes the host's way; the batches in front of it
      // are good (the same launch checked them)
      n = first_bad / B * B;
      fall = true; last = false;
    }
    if (n > 0) {
      // (min / max over the records that are in the chunk: the statistics above cover records [0, n_checked) of which the chunk may hold
      //  fewer after a cut -- a longer maximum only sizes a buffer more generously; lengths are told apart per row by the aligner)
      int min_len = INT32_MAX, max_len = 0, max_name = 0, min_name = INT32_MAX;
      for (int e = 0; e < NF; ++e) {
        min_name = std::min<int>(min_name, (int)std::min<uint32_t>(fe->h_stat.p[32 * e + FQT_MIN_NAME], 0x7fffffffu));
        min_len = std::min<int>(min_len, (int)std::min<uint32_t>(fe->h_stat.p[32 * e + FQT_MIN_LEN], 0x7fffffffu));
        max_len = std::max<int>(max_len, (int)fe->h_stat.p[32 * e + FQT_MAX_LEN]);
        max_name = std::max<int>(max_name, (int)fe->h_stat.p[32 * e + FQT_MAX_NAME]);
      }
      fe->max_name_ever = std::max(fe->max_name_ever, max_name);
      fe->min_name_ever = std::min(fe->min_name_ever, min_name);
      // (a cut chunk's statistics cover the records behind the cut too: at worst the general kernels run where the plain ones would have done)
      const bool plain_names = fe->slot_mode != FQ_FASTQ_SLOTS_REUSED || fe->min_name_ever == fe->max_name_ever;
      const bool all_long = min_len >= 96;
      const int name_stride = std::min(304, (fe->max_name_ever + 1 + 15) & ~15);
      TB.n_pairs = (int)n; TB.uniform_len = (!fall && min_len == max_len) ? max_len : 0; TB.max_len = max_len; TB.name_stride = name_stride;
      if (!fe->d_names[slot].ensure((size_t)n_rows * (size_t)name_stride + 64)) { finish(FQ_ENOMEM, "out of device memory (names)", false); return; }
      // ---- the read slots: bases behind short reads, names ----
      for (int e = 0; e < NF; ++e) {
        FileSide &F = fe->f[e];
        FqSlotArgs s{};
        s.text = F.d_text[slot].p + view[e]; s.rec = fe->d_rec[slot].p; s.n_rec = (int)n; s.row0 = (int)(e * n); s.n_rows = n_rows; s.g0 = F.records_done;
        s.n_slots = n_slots; s.mode = fe->slot_mode; s.slot_base = F.d_slot_base.p; s.slot_len = F.d_slot_len.p; s.slot_name = F.d_slot_name.p;
        s.head = fe->d_head[slot].p; s.names = fe->d_names[slot].p; s.name_stride = name_stride; s.stat = F.d_stat.p;
        s.all_long = all_long ? 1 : 0; s.plain_names = plain_names ? 1 : 0;
        fqdev::time_begin(3);
        if (fe->slot_mode != FQ_FASTQ_SLOTS_FRESH && fqdev::launch_slot_bases(s)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        if (fqdev::launch_slot_names(s)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        fqdev::time_end(3);
        if (fqdev::copy_pinned(fe->h_stat.p + 32 * e, F.d_stat.p, 4 * FQT_N_STAT, 0)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      }
      // the names of every reference batch's first pair, for the caller's order check
      const int n_sub = (int)((n + B - 1) / B);
      TB.first_names.assign((size_t)n_sub * 2 * (size_t)name_stride, 0);
      std::vector<char> &fn = TB.first_names;
      for (int sb = 0; sb < n_sub; ++sb)
        for (int e = 0; e < NF; ++e)
          if (fqdev::d2h(&fn[((size_t)sb * 2 + e) * name_stride], fe->d_names[slot].p + ((size_t)e * n + (size_t)sb * B) * name_stride, (size_t)name_stride)) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      if (fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
      for (int e = 0; e < NF; ++e) if (fe->h_stat.p[32 * e + FQT_SHORTER_AFTER_LONGER]) fe->f[e].shorter_after_longer = 1;
      TB.d_text[0] = fe->f[0].d_text[slot].p + view[0]; TB.d_text[1] = NF > 1 ? fe->f[1].d_text[slot].p + view[1] : nullptr;
      TB.d_rec = fe->d_rec[slot].p; TB.d_head = fe->d_head[slot].p; TB.d_hlen = fe->d_hlen[slot].p; TB.d_names = fe->d_names[slot].p;
    }
    // ---- what stays for the next chunk: the text behind record n ----
    for (int e = 0; e < NF; ++e) {
      FileSide &F = fe->f[e];
      uint32_t cut = skip[e];
      if (n > 0) {
        if (fqdev::d2h(&cut, F.d_nl.p + 4 * (size_t)n - 1, 4) || fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        cut += 1;
      }
      F.carry_slot = slot; F.carry_off = view[e] + cut; F.carry_len = n_text[e] - cut;
      const int64_t recs = (int64_t)(n_lines[e] / 4);
      if (recs > 0 && n_text[e] > 0) {
        uint32_t last_nl = 0;
        const size_t cap_lines = (size_t)(4 * (fe->chunk_pairs + B));
        const size_t idx = std::min<size_t>((size_t)recs * 4, cap_lines) - 1;
        if (fqdev::d2h(&last_nl, F.d_nl.p + idx, 4) || fqdev::sync()) { finish(FQ_ENODEV, fqdev::last_error(), false); return; }
        F.text_per_record = (double)(last_nl + 1 - skip[e]) / (double)((idx + 1) / 4);
      }
      F.records_done += n;
      {
        std::lock_guard<std::mutex> lk(F.mu);
        F.want_text = (uint64_t)std::min<double>((double)kMaxText, (double)fe->chunk_pairs * F.text_per_record * 1.005 + (256 << 10));
      }
    }
    fe->pairs_done += n;
    fqdev::time_collect(t_ms, t_n, FQ_K_COUNT);
    fe->ms_inflate += t_ms[0]; fe->ms_lines += t_ms[1]; fe->ms_records += t_ms[2]; fe->ms_slots += t_ms[3]; fe->ms_tokenise += t_ms[1] + t_ms[2] + t_ms[3];
    fe->n_launch_inflate += (int64_t)t_n[0]; fe->n_chunks += 1;
    fe->n_members += TB.members; fe->n_refused += TB.refused; fe->text_bytes += TB.text_bytes; fe->comp_bytes += TB.comp_bytes;
    {   // how many pairs follow this batch, by the files' sizes and the bytes a pair has taken so far (a hint: fq_align_text sizes a short first call's buffers by it)
      double left = 0;
      for (int e = 0; e < NF; ++e) left += (double)fe->f[e].file_size;
      left -= (double)fe->comp_bytes;
      TB.pairs_behind = fe->comp_bytes > 0 && left > 0 ? (int64_t)(left * (double)fe->pairs_done / (double)fe->comp_bytes) : 0;
    }
    // ---- the end of the stream, or of the device's part of it ----
    bool stream_end = false;
    if (last) {
      // nothing but whole records may be left: otherwise the host's reader says what the rest is (a last record without line end, blank
      // lines, a partner file that goes on)
      bool clean = true;
      for (int e = 0; e < NF; ++e) if (fe->f[e].carry_len != 0) clean = false;
      if (clean) stream_end = true; else fall = true;
    }
    if (!last && !fall && n == 0) {
      // not one whole reference batch yet: the text is carried into the next chunk -- unless a whole chunk's worth of text is there already
      // (reads of many kilobytes): not the device's case
      for (int e = 0; e < NF; ++e) if ((double)n_text[e] >= (double)fe->chunk_pairs * fe->f[e].text_per_record && n_text[e] > ((uint64_t)4 << 20)) fall = true;
    }
    if (stream_end || fall) {
      // (a chunk whose members were started ahead is dropped: the hand-over stands behind this chunk, F.inflated_file_off)
      if (next.open) { if (fe->overlap) { fqdev::stream_aux(1); (void)fqdev::sync(); fqdev::stream_aux(0); } }
      (void)fqdev::sync();
      double t2[FQ_K_COUNT] = {0}; uint64_t n2[FQ_K_COUNT] = {0};
      fqdev::time_collect(t2, n2, FQ_K_COUNT);
      if (stream_end) { fe->ms_inflate += t2[0]; fe->n_launch_inflate += (int64_t)n2[0]; }
    }
    {
      std::lock_guard<std::mutex> lk(fe->mu);
      if (n > 0) fe->ready.push_back(&TB); else fe->slot_free[slot] = true;
      if (stream_end || fall) { fe->done = true; fe->fallback = fall; if (next.open) fe->slot_free[next.slot] = true; }
      fe->cv.notify_all();
    }
    if (stream_end || fall) return;
  }
}
}  // namespace

fq_frontend::~fq_frontend() {
  { std::lock_guard<std::mutex> lk(mu); stop = true; }
  cv.notify_all();
  for (int e = 0; e < 2; ++e) {
    { std::lock_guard<std::mutex> lk(f[e].mu); f[e].stop = true; }
    f[e].cv.notify_all();
  }
  if (producer.joinable()) producer.join();
  for (int e = 0; e < 2; ++e) {
    if (f[e].th.joinable()) f[e].th.join();
    if (f[e].fd >= 0) close(f[e].fd);
  }
  // (device buffers are freed by the members' destructors; the states last)
  if (st) { fqdev::bind(st); }
}


// fq2 NULL / "": a single-end file (BwtMapper::SingleEndMapper's reader hands out fresh buffers: slot_mode is FRESH whatever is asked).
extern "C" int fq_frontend_open(int device, const char *fq1, const char *fq2, int32_t batch_pairs, int64_t chunk_pairs, int32_t slot_mode, int32_t max_read_len, fq_frontend_t **out) {
  if (!fq1 || !out || batch_pairs < 1 || chunk_pairs < batch_pairs || slot_mode < 0 || slot_mode > 2 || max_read_len < 16 || max_read_len > 4096) return FQ_EINVAL;
  *out = nullptr;
  // (declared in front of `fe`: destroyed behind it -- the device states of an open that fails go when the members' buffers have been freed;
  //  ~fq_frontend leaves them to fq_frontend_close, which an object that was never handed out does not reach)
  struct StatesGuard { fqdev::State *s[3] = {nullptr, nullptr, nullptr}; bool armed = true; ~StatesGuard() { if (armed) for (fqdev::State *x : s) if (x) fqdev::state_destroy(x); } } guard;
  std::unique_ptr<fq_frontend> fe(new fq_frontend);
  fe->device = device; fe->batch_pairs = batch_pairs; fe->chunk_pairs = chunk_pairs / batch_pairs * batch_pairs; fe->max_len = max_read_len;
  fe->n_files = (fq2 && *fq2) ? 2 : 1;
  if (const char *ev = getenv("FASTQUICK_FE_HEADROOM")) fe->headroom_min = strtoull(ev, nullptr, 10);                 // (0: none at all -- the carried text always moves the new text back)
  if (const char *ev = getenv("FASTQUICK_FE_OVERLAP")) fe->overlap = atoi(ev) != 0;      // (0: measurements of the kernels on their own)
  fe->slot_mode = fe->n_files == 1 ? FQ_FASTQ_SLOTS_FRESH : slot_mode;
  const char *paths[2] = {fq1, fq2};
  for (int e = 0; e < fe->n_files; ++e) {
    FileSide &F = fe->f[e];
    F.path = paths[e];
    struct stat sb;
    if (stat(paths[e], &sb) != 0 || !S_ISREG(sb.st_mode)) return FQ_EIO;           // (a pipe cannot be looked at twice: the host reader's case)
    F.fd = open(paths[e], O_RDONLY);
    if (F.fd < 0) return FQ_EIO;
    uint8_t h[1 << 16];
    const ssize_t got = pread(F.fd, h, sizeof h, 0);
    size_t hdr = 0;
    const size_t sz = got >= 18 ? bgzf_member(h, (size_t)got, &hdr) : 0;
    if (sz == 0) return FQ_EIO;
    F.file_size = (uint64_t)sb.st_size;
    if (sz >= hdr + 8 && (size_t)got >= sz && le32(h + sz - 4)) F.comp_ratio = (double)sz / (double)le32(h + sz - 4);                                                     // not BGZF: the host reader's case
    // the text a record takes, from the file's first member (inflated here, on the host: 64 KiB)
    F.text_per_record = 320.0;
    if ((size_t)got >= sz && sz >= hdr + 8) {
      const uint32_t isize = le32(h + sz - 4);
      std::vector<uint8_t> t(isize + 1);
      std::unique_ptr<fqz::Inflater> fz(new fqz::Inflater);
      if (isize && fqz::inflate_raw(*fz, h + hdr, sz - hdr - 8, t.data(), isize)) {
        size_t lines = 0, last = 0;
        for (size_t i = 0; i < isize; ++i) if (t[i] == '\n') { ++lines; if (lines % 4 == 0) last = i + 1; }
        if (lines >= 4) F.text_per_record = (double)last / (double)(lines / 4);
      }
    }
  }
  // device side
  fe->st = guard.s[0] = fqdev::state_create(device);
  if (!fe->st || fqdev::bind(fe->st)) return FQ_ENODEV;
  const size_t n_slots = (size_t)2 * (size_t)batch_pairs;
  for (int e = 0; e < fe->n_files; ++e) {
    FileSide &F = fe->f[e];
    F.st = guard.s[1 + e] = fqdev::state_create(device);
    if (!F.st) return FQ_ENODEV;
    if (fqdev::bind(fe->st)) return FQ_ENODEV;
    if (!F.d_slot_base.ensure(n_slots * 96) || !F.d_slot_len.ensure(n_slots) || !F.d_slot_name.ensure(n_slots * 304) || !F.d_stat.ensure(FQT_N_STAT + 8)) return FQ_ENOMEM;
    if (fqdev::dzero(F.d_slot_base.p, n_slots * 96) || fqdev::dzero(F.d_slot_len.p, n_slots * 2) || fqdev::dzero(F.d_slot_name.p, n_slots * 304)) return FQ_ENODEV;
  }
  if (fqdev::sync()) return FQ_ENODEV;
  for (int e = 0; e < fe->n_files; ++e) {
    fe->f[e].want_text = (uint64_t)std::min<double>((double)kMaxText, (double)fe->chunk_pairs * fe->f[e].text_per_record * 1.005 + (256 << 10));
    fe->f[e].read_threads = std::max(1, std::min(4, fq_host_cpus() / (2 * fe->n_files)));     // (of the CPUs this process may use: fq_host_cpus)
    if (const char *ev = getenv("FASTQUICK_FE_READ_THREADS")) fe->f[e].read_threads = std::max(1, std::min(16, atoi(ev)));
    fe->f[e].th = std::thread(reader_main, fe.get(), e);
  }
  fe->producer = std::thread(producer_main, fe.get());
  guard.armed = false;
  *out = fe.release();
  return FQ_OK;
}
extern "C" void fq_frontend_close(fq_frontend_t *fe) {
  if (!fe) return;
  fqdev::State *states[3] = {fe->st, fe->f[0].st, fe->f[1].st};
  const int device = fe->device;
  (void)device;
  delete fe;            // joins the threads, frees the buffers
  for (fqdev::State *s : states) if (s) fqdev::state_destroy(s);
}
extern "C" const char *fq_frontend_last_error(const fq_frontend_t *fe) { return fe ? fe->err.c_str() : ""; }
// The next batch (*out; it stays valid until fq_frontend_release).  Returns its number of pairs; 0 at the end of the stream; FQ_EFALLBACK when
// the rest of the stream is the host reader's (fq_frontend_handover); another negative code on failure (fq_frontend_last_error).
extern "C" int64_t fq_frontend_next(fq_frontend_t *fe, fq_text_batch_t **out) {
  if (!fe || !out) return FQ_EINVAL;
  *out = nullptr;
  std::unique_lock<std::mutex> lk(fe->mu);
  fe->cv.wait(lk, [&] { return !fe->ready.empty() || fe->done; });
  if (!fe->ready.empty()) {
    fq_text_batch *b = fe->ready.front();
    fe->ready.pop_front();
    *out = b;
    return b->n_pairs;
  }
  if (fe->rc) return fe->rc;
  return fe->fallback ? FQ_EFALLBACK : 0;
}
extern "C" void fq_frontend_release(fq_frontend_t *fe, fq_text_batch_t *b) {
  if (!fe || !b || b->slot < 0) return;
  { std::lock_guard<std::mutex> lk(fe->mu); fe->slot_free[b->slot] = true; }
  fe->cv.notify_all();
}
// Host readers standing where the device's part of the stream ended (after fq_frontend_next returned FQ_EFALLBACK and every batch has been
// released): out[e] for each file, opened with `threads` threads each.
extern "C" int fq_frontend_handover(fq_frontend_t *fe, int threads, fq_fastq_t **out) {
  if (!fe || !out) return FQ_EINVAL;
  { std::lock_guard<std::mutex> lk(fe->mu); if (!fe->done || !fe->fallback) return FQ_EINVAL; }
  if (fe->producer.joinable()) fe->producer.join();
  if (fqdev::bind(fe->st)) return FQ_ENODEV;
  const size_t n_slots = (size_t)2 * (size_t)fe->batch_pairs;
  for (int e = 0; e < fe->n_files; ++e) {
    FileSide &F = fe->f[e];
    { std::lock_guard<std::mutex> lk(F.mu); F.stop = true; }
    F.cv.notify_all();
    if (F.th.joinable()) F.th.join();
    const int64_t file_off = F.inflated_file_off;          // behind the last chunk that was inflated (a chunk the reader thread filled ahead has not been used)
    std::vector<uint8_t> text((size_t)F.carry_len), names, bases;
    std::vector<uint16_t> lens;
    if (F.carry_len && fqdev::d2h(text.data(), F.d_text[F.carry_slot].p + F.carry_off, (size_t)F.carry_len)) return FQ_ENODEV;
    if (fe->slot_mode != FQ_FASTQ_SLOTS_FRESH) {
      names.resize(n_slots * 304); bases.resize(n_slots * 96); lens.resize(n_slots);
      if (fqdev::d2h(names.data(), F.d_slot_name.p, names.size()) || fqdev::d2h(bases.data(), F.d_slot_base.p, bases.size()) || fqdev::d2h(lens.data(), F.d_slot_len.p, lens.size() * 2)) return FQ_ENODEV;
    }
    if (fqdev::sync()) return FQ_ENODEV;
    fq_fastq_t *r = nullptr;
    int rc = fq_fastq_open(F.path.c_str(), threads, &r);
    if (rc) return rc;
    fq_fastq_configure(r, fe->batch_pairs, fe->slot_mode, 0);
    rc = fq_fastq_resume(r, file_off, text.data(), text.size(), F.records_done, names.empty() ? nullptr : names.data(), bases.empty() ? nullptr : bases.data(), lens.empty() ? nullptr : lens.data(), F.shorter_after_longer);
    if (rc) { fq_fastq_close(r); return rc; }
    out[e] = r;
  }
  return FQ_OK;
}
extern "C" int fq_frontend_unequal_lengths(const fq_frontend_t *fe) { return fe && (fe->f[0].shorter_after_longer || fe->f[1].shorter_after_longer) ? 1 : 0; }
extern "C" void fq_frontend_stats(const fq_frontend_t *fe, fq_frontend_stats_t *s) {
  if (!fe || !s) return;
  s->ms_inflate = fe->ms_inflate; s->ms_tokenise = fe->ms_tokenise; s->members = fe->n_members; s->refused = fe->n_refused;
  s->text_bytes = fe->text_bytes; s->comp_bytes = fe->comp_bytes; s->pairs = fe->pairs_done;
  s->ms_wait_reader = fe->ms_wait_reader; s->ms_wait_slot = fe->ms_wait_slot; s->ms_read = fe->f[0].ms_read + fe->f[1].ms_read; s->ms_upload = fe->f[0].ms_upload + fe->f[1].ms_upload;
  s->ms_lines = fe->ms_lines; s->ms_records = fe->ms_records; s->ms_slots = fe->ms_slots; s->inflate_launches = fe->n_launch_inflate; s->chunks = fe->n_chunks;
}
// batch accessors
extern "C" int32_t fq_text_batch_pairs(const fq_text_batch_t *b) { return b ? b->n_pairs : 0; }
extern "C" const char *fq_text_batch_first_name(const fq_text_batch_t *b, int32_t sub_batch, int32_t end) {
  if (!b || sub_batch < 0 || end < 0 || end > 1 || (size_t)(((size_t)sub_batch * 2 + end + 1) * b->name_stride) > b->first_names.size()) return nullptr;
  return &b->first_names[((size_t)sub_batch * 2 + end) * b->name_stride];
}

// The batch's per-read arrays copied to the host (tests): head [3][rows] as fq_packed_batch_t::head, len [rows], names [rows][name_stride];
// rows = 2 * pairs (pairs for a single-end batch).  Returns the name stride, or a negative code.
extern "C" int fq_text_batch_fetch(fq_frontend_t *fe, const fq_text_batch_t *b, uint64_t *head, uint16_t *len, char *names, int64_t names_cap) {
  if (!fe || !b) return FQ_EINVAL;
  const size_t rows = (size_t)b->n_pairs * (b->single_end ? 1 : 2);
  DevScope scope(fe->device);                                  // (a state of this call's own: the readers' and the producer's are their threads')
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  if (head && fqdev::d2h(head, b->d_head, rows * 24)) return FQ_ENODEV;
  if (len && fqdev::d2h(len, b->d_hlen, rows * 2)) return FQ_ENODEV;
  if (names) { if ((int64_t)(rows * (size_t)b->name_stride) > names_cap) return FQ_ELIMIT; if (fqdev::d2h(names, b->d_names, rows * (size_t)b->name_stride)) return FQ_ENODEV; }
  if (fqdev::sync()) return FQ_ENODEV;
  return b->name_stride;
}
