// fq_fastq.cpp -- the FASTQ front end: inflate + tokenise + read-slot history, on many threads.
//
// Replaces, for one FASTQ file, the reader side of bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:476-613): kseq_read3_fpc
// (libbwa/kseq.h:327-371) over a gzFile, one record at a time on one IO thread per file (IOworkerAlt, src/BwtMapper.cpp:1973-1980).
// Here:
//   * the inflated text arrives in blocks from a producer thread.  BGZF files (bgzip: gzip members with a 'BC' extra field that
//     states each member's size) are inflated member-parallel straight into place -- the sizes give every member its output offset
//     before a byte is inflated; any other gzip file (one member of any size, or many: what `gzip` and sequencers write) is decoded as a
//     stream by the same decoder on that thread, block after block with the window carried along (fill_gz_stream: twice gzread's speed,
//     and gzread's verdict on anything the decoder does not take); plain text and pipes go through zlib's gzread.  Either way the
//     inflating overlaps with the tokenising of the previous block;
//   * a block is cut into lines by all threads (memchr), and records are emitted by all threads under the assumption that the
//     file is what sequencers write -- four lines per record.  Every record is checked against exactly the conditions under which
//     kseq_read3_fpc would return the same tokens ('@' first, name up to the first white space, a base line of printable characters
//     without '+', '>' or '@', a '+' line, a quality line as long as the base line); the first record that fails them, and
//     everything behind it in the block, goes through the byte-wise restatement of kseq_read3_fpc instead, so that odd files
//     (wrapped bases, blank lines, a last record without line end, quality strings of the wrong length) come out as the reference
//     reads them or are refused with its message;
//   * the reference's reused read slots (names without terminator, bases behind a short read: SURVEY Q7/Q8) are modelled per
//     slot; the slots of 2 * batch_pairs consecutive records are distinct, so a window of that many records updates them in parallel.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_pool.h"
#include "fq_inflate.h"
#include "fq_fastq_internal.h"

namespace {
template <class F>
void par_for(FqWorkPool &pool, int threads, size_t n, size_t min_per_thread, F fn) {   // fn(lo, hi, t) over [0, n), on the pool's workers
  const int T = (int)std::min<size_t>((size_t)std::max(1, threads), std::max<size_t>(1, n / std::max<size_t>(1, min_per_thread)));
  if (T <= 1) { fn((size_t)0, n, 0); return; }
  const size_t per = (n + T - 1) / T;
  pool.run(T, [&](int t) { const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per); if (lo < hi) fn(lo, hi, t); });
}

struct Block {                        // inflated text [data + head, data + head + n); `head` bytes of room in front for the carry
  std::unique_ptr<uint8_t[]> data;
  size_t cap = 0, head = 0, n = 0;
  bool last = false;
  std::string err;                    // the source failed while filling this block
};
const size_t kHeadroom = 1 << 20;
}  // namespace

struct fq_fastq {
  std::string path, err;
  FqWorkPool pool_in, pool_tok;       // workers of the producer thread's inflate and of the caller's tokenising (a pool runs one pass at a time)
  int threads = 1;
  size_t block_bytes = (size_t)16 << 20;
  // ---- source ----
  int fd = -1;
  gzFile gz = nullptr;
  bool bgzf = false, src_eof = false;
  bool gz_regular = false;            // a regular file with a gzip header that is not BGZF (blocks of the stream decoder's size whichever reader reads it)
  size_t room = 0;                    // text a block takes (set per block by the producer)
  std::vector<uint8_t> cbuf;          // BGZF: compressed bytes not yet inflated
  size_t cpos = 0, cend = 0;
  bool file_eof = false;
  // ---- a gzip file that is not BGZF (one member of any size, or several): the stream decoded by fq_inflate.h's decoder on the producer thread,
  //      twice zlib's speed, into the blocks as they fill (fill_gz_stream); anything it does not take -- a damaged or truncated stream, a wrong
  //      check sum, bytes behind the last member -- is gzread's: the file is opened again, the text handed out so far skipped, and what gzread
  //      makes of the rest (text, end, error message) is the reader's
  struct GzStream {
    bool on = false;
    std::vector<uint8_t> in;            // compressed bytes [pos, end)
    size_t pos = 0, end = 0;
    bool eof = false;                   // the file has been read to its end
    fqz::Inflater Z;
    fqz::Dec d;
    bool in_member = false, in_block = false;
    uint32_t crc = 0;
    uint64_t member_out = 0;            // text of the current member so far (ISIZE is its low 32 bits)
    std::vector<uint8_t> win;           // what the next block begins with: the window its first matches may reach back into (up to 32 KiB of the
                                        // current member in front of the block's boundary) and the text decoded beyond the boundary (`excess`)
    size_t excess = 0, win_member = 0;  // ... of which the last `excess` bytes are the next block's first text; bytes of `win` that are the current member's
    uint64_t emitted = 0;               // text handed out in earlier blocks
  } gs;
  // ---- producer thread ----
  std::thread producer;
  std::mutex mu;
  std::condition_variable cv_full, cv_free;
  std::deque<Block> ready;            // filled blocks
  std::vector<Block> spare;           // storage to reuse
  bool stop = false, started = false;
  std::string src_err;                // (producer thread only)
  // ---- tokeniser ----
  Block cur;                          // block being consumed: [cur_pos, cur_end)
  size_t cur_pos = 0, cur_end = 0;
  bool have_cur = false, at_eof = false;
  std::vector<uint8_t> carry;         // bytes of an incomplete record at the end of the previous block
  bool notice_dropped = false;
  std::string dropped_name;
  // ---- slots ----
  int batch_pairs = 262144;
  int slot_mode = FQ_FASTQ_SLOTS_REUSED;
  std::vector<std::string> slot_name[2];
  std::vector<uint8_t> slot_base[2];
  std::vector<uint16_t> slot_len[2];      // longest read the slot has held
  std::atomic<int> shorter_after_longer{0};   // a read came to a slot that has held a longer one (fq_fastq_unequal_lengths)
  long long records_seen = 0;
  // ---- --frac_samp (src/BwtMapper.cpp:483, 500-507): every reference batch draws from Random(seed = the batch's number) -- the
  //      Mersenne twister of VerifyBamID/Random.cpp, Next() = (y + 0.5) / 2^32 -- one number per record met; a record whose number is
  //      above the fraction is read and dropped (kseq_read4_fpc), the batch ends with its batch_pairs-th kept record.  The batch's
  //      number is PairEndMapper's `round` at the time the batch is read: the first batch is read with 0, every later one by the IO
  //      worker that is started BEFORE round is incremented (:1973-1985) -- seeds 0, 0, 1, 2, ...  (SingleEndMapper seeds with
  //      clock(), :1286: its sampling is not reproducible; FRESH slots use the same seeds as pairs.)
  double frac = 1.0;
  struct Sampler {
    std::mt19937 gen{0};
    long long round = 0, in_batch = 0;
    bool keep(double frac, int batch_pairs) {     // the draw for the next record met
      if (in_batch == batch_pairs) { ++round; in_batch = 0; gen.seed((uint32_t)(round - 1)); }
      const double x = ((double)gen() + 0.5) * (1.0 / 4294967296.0);
      if (x > frac) return false;
      ++in_batch;
      return true;
    }
  } sampler;

  ~fq_fastq() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv_free.notify_all(); cv_full.notify_all();
    if (producer.joinable()) producer.join();
    if (gz) gzclose(gz);
    if (fd >= 0) close(fd);
  }
};

namespace {
// ---- source: BGZF member-parallel, anything else through gzread ---------------------------------------------------------------
bool looks_bgzf(const uint8_t *h, size_t n) {   // RFC 1952 header with FEXTRA and a 'B','C' subfield of length 2 (SAM spec 4.1)
  if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
  const size_t xlen = h[10] | (size_t)h[11] << 8;
  size_t p = 12;
  while (p + 4 <= 12 + xlen && p + 4 <= n) {
    const size_t sl = h[p + 2] | (size_t)h[p + 3] << 8;
    if (h[p] == 'B' && h[p + 1] == 'C' && sl == 2) return true;
    p += 4 + sl;
  }
  return false;
}
// size of the member that starts at h (needs >= 18 bytes), 0 if it is not a BGZF member
size_t bgzf_member_size(const uint8_t *h, size_t n, size_t *hdr_len) {
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
struct Member { size_t in_off, in_len, hdr, out_off, out_len; };

bool fill_bgzf(fq_fastq *r, Block &b) {
  b.n = 0;
  std::vector<Member> ms;
  size_t out = 0;
  for (;;) {
    // complete members in the compressed buffer
    while (r->cend - r->cpos >= 18) {
      size_t hdr = 0;
      const size_t sz = bgzf_member_size(r->cbuf.data() + r->cpos, r->cend - r->cpos, &hdr);
      if (sz == 0) {
        const uint8_t *h = r->cbuf.data() + r->cpos;
        const bool gzip_hdr = h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4);
        if (gzip_hdr && 12 + ((size_t)h[10] | (size_t)h[11] << 8) > r->cend - r->cpos && !r->file_eof) break;   // a long extra field: read on
        r->src_err = "not a BGZF member at a member boundary (a bgzip file followed by something else?)"; return false;
      }
      if (sz < hdr + 8) { r->src_err = "BGZF member too short"; return false; }
      if (r->cend - r->cpos < sz) break;
      const uint8_t *m = r->cbuf.data() + r->cpos;
      const size_t isize = m[sz - 4] | (size_t)m[sz - 3] << 8 | (size_t)m[sz - 2] << 16 | (size_t)m[sz - 1] << 24;
      if (out + isize > b.cap - b.head) {
        if (!ms.empty()) goto inflate;
        // (a member larger than a block: blocks of a few hundred bytes are a test setting) -- the block grows to hold it
        std::unique_ptr<uint8_t[]> bigger(new uint8_t[b.head + isize + 16]);
        b.data = std::move(bigger); b.cap = b.head + isize + 16;
      }
      ms.push_back({r->cpos, sz, hdr, out, isize});
      out += isize;
      r->cpos += sz;
    }
    if (r->file_eof) {
      if (r->cend != r->cpos) { r->src_err = "truncated BGZF member at the end of the file"; return false; }
      break;
    }
    {   // read on
      if (r->cpos > 0 && ms.empty()) { memmove(r->cbuf.data(), r->cbuf.data() + r->cpos, r->cend - r->cpos); r->cend -= r->cpos; r->cpos = 0; }
      if (r->cend == r->cbuf.size()) {
        if (!ms.empty()) goto inflate;     // buffer full of members waiting to be inflated
        r->cbuf.resize(r->cbuf.size() * 2);
      }
      const ssize_t got = read(r->fd, r->cbuf.data() + r->cend, r->cbuf.size() - r->cend);
      if (got < 0) { r->src_err = "read error"; return false; }
      if (got == 0) r->file_eof = true;
      r->cend += (size_t)got;
    }
  }
inflate:
  std::atomic<int> bad{0};
  uint8_t *dst = b.data.get() + b.head;
  const uint8_t *src = r->cbuf.data();
  // Members go through the front end's own decoder (fq_inflate.h); one it refuses is given to zlib, whose verdict stands -- what is
  // accepted and what is reported as corrupt is what gzread / inflate() decide (FASTQUICK_ZLIB_INFLATE=1: zlib for everything, to compare).
  static const bool zlib_only = [] { const char *e = getenv("FASTQUICK_ZLIB_INFLATE"); return e && *e && *e != '0'; }();
  par_for(r->pool_in, r->threads, ms.size(), 4, [&](size_t lo, size_t hi, int) {
    std::unique_ptr<fqz::Inflater> fz(zlib_only ? nullptr : new fqz::Inflater);
    z_stream zs;
    bool zs_ready = false;
    for (size_t i = lo; i < hi; ++i) {
      const Member &m = ms[i];
      if (m.out_len == 0) continue;      // (the empty end-of-file member)
      const uint8_t *t = src + m.in_off + m.in_len - 8;
      const uint32_t crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
      if (fz && fqz::inflate_raw(*fz, src + m.in_off + m.hdr, m.in_len - m.hdr - 8, dst + m.out_off, m.out_len)) {
        if (fqz::crc32(dst + m.out_off, m.out_len) != crc) { bad = 1; break; }
        continue;
      }
      if (!zs_ready) {
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) { bad = 1; break; }
        zs_ready = true;
      }
      inflateReset(&zs);
      zs.next_in = const_cast<Bytef *>(src + m.in_off + m.hdr); zs.avail_in = (uInt)(m.in_len - m.hdr - 8);
      zs.next_out = dst + m.out_off; zs.avail_out = (uInt)m.out_len;
      const int rc = inflate(&zs, Z_FINISH);
      if (rc != Z_STREAM_END || zs.avail_out != 0) { bad = 1; break; }
      if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst + m.out_off, (uInt)m.out_len) != crc) { bad = 1; break; }
    }
    if (zs_ready) inflateEnd(&zs);
  });
  if (bad) { r->src_err = "corrupt BGZF member (inflate or CRC failed)"; return false; }
  b.n = out;
  if (r->cpos == r->cend) { r->cpos = r->cend = 0; }
  b.last = r->file_eof && r->cpos == r->cend;
  return true;
}
bool fill_gz(fq_fastq *r, Block &b) {
  b.n = 0;
  const size_t room = r->room;
  while (b.n < room) {
    const int got = gzread(r->gz, b.data.get() + b.head + b.n, (unsigned)std::min<size_t>(room - b.n, (size_t)1 << 30));
    if (got < 0) { int en = 0; r->src_err = std::string("gzread: ") + gzerror(r->gz, &en); return false; }
    if (got == 0) { b.last = true; break; }
    b.n += (size_t)got;
  }
  return true;
}
// ---- source: a gzip stream through the fast decoder, block by block ------------------------------------------------------------------
// (the window a match reaches back into lies in front of the block's text, where the tokeniser later puts its carry: the same bytes)
// gzread takes over.  Blocks have the size gzread's would have (r->room), so what has been handed out so far is what the gzread reader would have
// handed out in front of the block the trouble is in; that block is read again by gzread, whose text, end or error message is the reader's.
bool gz_fallback(fq_fastq *r, Block &b, size_t n_valid) {
  fq_fastq::GzStream &G = r->gs;
  G.on = false;
  std::vector<uint8_t>().swap(G.in);
  const bool whole = n_valid >= r->room;            // the trouble lies behind this block's boundary: the block is good, gzread begins behind it
  if (whole) G.emitted += r->room;
  r->gz = gzopen(r->path.c_str(), "rb");
  if (!r->gz) { r->src_err = "cannot reopen " + r->path; return false; }
  gzbuffer(r->gz, 1 << 16);                         // (a block's gzread is then a direct one -- 192 KiB at least, twice this and more: what it inflates is the block,
                                                    //  and a failure loses exactly the block it happens in)
  uint64_t skip = G.emitted;
  std::vector<uint8_t> scratch((size_t)4 << 20);
  while (skip > 0) {
    const int got = gzread(r->gz, scratch.data(), (unsigned)std::min<uint64_t>(skip, scratch.size()));
    if (got <= 0) { int en = 0; r->src_err = std::string("gzread: ") + gzerror(r->gz, &en); return false; }   // (cannot happen: the decoder produced this text from the same bytes)
    skip -= (uint64_t)got;
  }
  if (whole) { b.n = r->room; return true; }
  return fill_gz(r, b);
}
bool fill_gz_stream(fq_fastq *r, Block &b) {
  fq_fastq::GzStream &G = r->gs;
  fqz::Dec &d = G.d;
  uint8_t *const text = b.data.get() + b.head;
  const size_t room = r->room;
  b.n = 0;
  // in front of the block: the window; at its start: what the block before decoded beyond its boundary
  const size_t pre = G.win.size() - G.excess;
  if (!G.win.empty()) memcpy(text - pre, G.win.data(), G.win.size());
  d.dst = text + G.excess - G.win_member; d.out = text + G.excess; d.out_end = b.data.get() + b.cap;
  const uint8_t *crc_from = d.out;
  auto more_input = [&]() -> bool {                  // the unread bytes to the front, the buffer filled from the file; false: nothing more came
    if (G.eof) return false;
    const size_t used = (size_t)(d.in - G.in.data()), left = G.end - used;
    memmove(G.in.data(), G.in.data() + used, left);
    G.end = left;
    bool any = false;
    while (G.end < G.in.size()) {
      const ssize_t got = read(r->fd, G.in.data() + G.end, G.in.size() - G.end);
      if (got < 0) { if (errno == EINTR) continue; break; }
      if (got == 0) { G.eof = true; break; }
      G.end += (size_t)got; any = true;
    }
    d.in = G.in.data(); d.in_end = G.in.data() + G.end;
    return any;
  };
  auto have = [&](size_t n) -> bool {                // n unread bytes in the buffer (or everything that is left of the file)
    while ((size_t)(d.in_end - d.in) < n) if (!more_input()) break;
    return (size_t)(d.in_end - d.in) >= n;
  };
  auto fold_crc = [&] { G.crc = fqz::crc32(crc_from, (size_t)(d.out - crc_from), G.crc); G.member_out += (uint64_t)(d.out - crc_from); crc_from = d.out; };
  for (;;) {
    if ((size_t)(d.out - text) >= room) break;       // the block is full (what lies beyond its boundary begins the next one)
    if (!G.in_member) {
      // ---- a member's header (RFC 1952), or the end of the file
      if (!have(18) && d.in_end == d.in) { b.last = true; break; }
      const uint8_t *h = d.in;
      const size_t avail = (size_t)(d.in_end - d.in);
      if (avail < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 0xe0)) return gz_fallback(r, b, (size_t)(d.out - text));
      (void)have((size_t)1 << 17);                    // (every field of a header a file is likely to have)
      h = d.in;
      const size_t av = (size_t)(d.in_end - d.in);
      size_t p = 10;
      bool ok = true;
      if (h[3] & 4) { if (p + 2 > av) ok = false; else { const size_t xl = h[p] | (size_t)h[p + 1] << 8; p += 2 + xl; } }
      for (int fl : {8, 16}) if (ok && (h[3] & fl)) { while (p < av && h[p]) ++p; ++p; }
      if (ok && (h[3] & 2)) p += 2;
      if (!ok || p + 8 > av) return gz_fallback(r, b, (size_t)(d.out - text));
      d.in += p;
      d.bb = 0; d.bc = 0; d.over = 0; d.last = false; d.LT = d.DT = nullptr;
      d.dst = d.out;                                  // (a member's matches do not reach in front of it)
      G.in_member = true; G.in_block = false; G.crc = 0; G.member_out = 0;
      crc_from = d.out;
    }
    if (!G.in_block) {
      // ---- the next block's header: a stored block copies up to 64 KiB, a dynamic block's code lengths are under 1 KiB (the room behind the
      //      block's boundary is there for that)
      (void)have(((size_t)1 << 16) + 4096);
      const int nb = fqz::dec_next_block(d, true);
      if (nb < 0 || d.over > ((size_t)d.bc >> 3)) return gz_fallback(r, b, (size_t)(d.out - text));      // (a stored block that was refused copied nothing)
      if (nb == 2) continue;                          // a stored block has been copied
      if (nb == 1) {
        // ---- the member's trailer: CRC-32 and size of its text, at the next byte boundary
        fold_crc();
        const size_t back = (size_t)d.bc >> 3;
        if (d.over > back) return gz_fallback(r, b, (size_t)(d.out - text));
        d.in -= back - d.over; d.over = 0; d.bb = 0; d.bc = 0;
        if (!have(8)) return gz_fallback(r, b, (size_t)(d.out - text));
        const uint32_t crc = d.in[0] | (uint32_t)d.in[1] << 8 | (uint32_t)d.in[2] << 16 | (uint32_t)d.in[3] << 24;
        const uint32_t isz = d.in[4] | (uint32_t)d.in[5] << 8 | (uint32_t)d.in[6] << 16 | (uint32_t)d.in[7] << 24;
        if (crc != G.crc || isz != (uint32_t)G.member_out) return gz_fallback(r, b, (size_t)(d.out - text));
        d.in += 8;
        G.in_member = false;
        continue;
      }
      G.in_block = true;
    }
    // ---- the block's symbols
    const int fr = fqz::dec_fast_loop(d);
    if (fr == 1) { G.in_block = false; continue; }
    if (fr < 0) return gz_fallback(r, b, (size_t)(d.out - text));
    if ((size_t)(d.out_end - d.out) < 258 + 72) continue;                    // (behind the block's boundary by now: the loop's first test ends it)
    if (more_input()) continue;
    if ((size_t)(d.in_end - d.in) >= 16) continue;
    // the file's last bytes: a symbol at a time, every bound checked; a stream that wants bytes behind the file's end is truncated
    uint8_t *const out0 = d.out;
    const int cr = fqz::dec_careful_turn(d);
    if (cr < 0 || d.over > ((size_t)d.bc >> 3)) return gz_fallback(r, b, (size_t)(out0 - text));
    if (cr == 1) G.in_block = false;
  }
  if (G.in_member) fold_crc();
  const size_t produced = (size_t)(d.out - text);
  b.n = produced < room ? produced : room;
  // what the next block begins with: the window in front of the boundary (of the current member) and the text beyond it
  {
    const uint8_t *const boundary = text + b.n;
    const uint8_t *member0 = G.in_member ? d.dst : d.out;                    // (nothing reaches back across a member's end)
    const uint8_t *c0 = member0 < boundary ? std::max<const uint8_t *>(member0, boundary - std::min<size_t>(32768, (size_t)(boundary - (text - pre)))) : boundary;
    std::vector<uint8_t> nw(c0, (const uint8_t *)d.out);
    G.excess = (size_t)(d.out - boundary);
    G.win_member = G.in_member ? (size_t)(d.out - std::max<const uint8_t *>(member0, c0)) : 0;
    G.win.swap(nw);
  }
  G.emitted += b.n;
  return true;
}
void producer_main(fq_fastq *r) {
  for (;;) {
    Block b;
    {
      std::unique_lock<std::mutex> lk(r->mu);
      r->cv_free.wait(lk, [&] { return r->stop || r->ready.size() < 2; });
      if (r->stop) return;
      if (!r->spare.empty()) { b = std::move(r->spare.back()); r->spare.pop_back(); }
    }
    // (a gzip file's blocks: 192 KiB at least, whichever reader reads it -- the stream decoder wants room; behind a block's text, for that decoder, room
    //  for a stored block and the fast loop's margin: it stops at the first look behind the boundary and the next block begins with what lies beyond)
    const size_t want_text = r->gz_regular ? std::max<size_t>(r->block_bytes, (size_t)192 << 10) : r->block_bytes;
    const size_t slack = r->gs.on ? ((size_t)1 << 16) + 4096 : 0;
    r->room = want_text;
    if (b.cap < want_text + kHeadroom + slack) { b.cap = want_text + kHeadroom + slack; b.data.reset(new uint8_t[b.cap]); }
    b.head = kHeadroom; b.n = 0; b.last = false;
    const bool ok = r->bgzf ? fill_bgzf(r, b) : r->gs.on ? fill_gz_stream(r, b) : fill_gz(r, b);
    if (!ok) { b.last = true; b.err = r->src_err; }
    static const bool reader_debug = [] { const char *e = getenv("FASTQUICK_READER_DEBUG"); return e && *e && *e != '0'; }();   // (read once: the host program may change its environment beside this thread)
    if (reader_debug) fprintf(stderr, "[reader] block: %zu bytes, last %d, err '%s' (%s)\n", b.n, (int)b.last, b.err.c_str(), r->bgzf ? "bgzf" : r->gs.on ? "stream" : "gzread");
    const bool done = b.last;
    {
      std::lock_guard<std::mutex> lk(r->mu);
      r->ready.push_back(std::move(b));
    }
    r->cv_full.notify_all();
    if (done) return;
  }
}

// ---- tokeniser ---------------------------------------------------------------------------------------------------------------------
struct Rows {
  const fq_fastq_rows_t *o;
  int64_t base;       // row index of the first record of this call's current block
};
inline bool is_space(int c) { return c == ' ' || (c >= 9 && c <= 13); }
inline bool is_graph(int c) { return c > 32 && c < 127; }

// what fill_chunk did per record: rows cleared, truncation rules of the reference's buffers
inline int emit(const fq_fastq_rows_t *o, int64_t row, const uint8_t *name, size_t nl, const uint8_t *seq, const uint8_t *qual, size_t L, std::string *err) {
  if ((int64_t)L > (int64_t)o->stride) {
    *err = "read " + std::string((const char *)name, std::min<size_t>(nl, 301)) + " is longer than the batch rows (" + std::to_string(L) + " > " + std::to_string(o->stride) + "): pass --read_len";
    return -1;
  }
  uint8_t *sr = o->seq + (size_t)row * (size_t)o->stride, *qr = o->qual + (size_t)row * (size_t)o->stride;
  memcpy(sr, seq, L); memset(sr + L, 0, (size_t)o->stride - L);
  memcpy(qr, qual, L); memset(qr + L, 0, (size_t)o->stride - L);
  o->len[row] = (int32_t)L;
  if (nl > 301) nl = 301;                                  // the reference's name buffer holds 2 * read_len = 302 bytes
  if ((int64_t)nl >= (int64_t)o->name_stride) nl = (size_t)o->name_stride - 1;
  char *nr = o->names + (size_t)row * (size_t)o->name_stride;
  memcpy(nr, name, nl); memset(nr + nl, 0, (size_t)o->name_stride - nl);
  return 0;
}

// ReadSlots::put of round 2's command line, per record g of the file (the name printed for a record and the bytes the filter sees
// behind a short read depend on what earlier records left in the record's slot)
void slot_apply(fq_fastq *r, const fq_fastq_rows_t *o, int64_t row, long long g) {
  char *nr = o->names + (size_t)row * (size_t)o->name_stride;
  size_t l = strlen(nr);
  const bool mate_suffix = l > 2 && nr[l - 2] == '/' && (nr[l - 1] == '1' || nr[l - 1] == '2');   // src/BwtMapper.cpp:565-570
  if (r->slot_mode == FQ_FASTQ_SLOTS_FRESH) { if (mate_suffix) memset(nr + l - 2, 0, 2); return; }
  const int set = (int)((g / r->batch_pairs) & 1);
  const size_t slot = (size_t)(g % r->batch_pairs);
  uint8_t *sr = o->seq + (size_t)row * (size_t)o->stride;
  const size_t n = (size_t)o->len[row];
  uint8_t *h = &r->slot_base[set][slot * 96];
  uint16_t &longest = r->slot_len[set][slot];
  if (n < longest) r->shorter_after_longer.store(1, std::memory_order_relaxed);
  else longest = (uint16_t)std::min<size_t>(n, 65535);
  for (size_t i = n; i < 96 && i < (size_t)o->stride; ++i) sr[i] = h[i];
  memcpy(h, sr, std::min<size_t>(n, 96));
  if (r->slot_mode == FQ_FASTQ_SLOTS_CLEAN_NAMES) { if (mate_suffix) memset(nr + l - 2, 0, 2); return; }
  std::string &b = r->slot_name[set][slot];
  if (b.size() < l) b.resize(l, '\0');
  b.replace(0, l, nr, l);
  if (mate_suffix) b[l - 2] = '\0';
  const size_t pl = strlen(b.c_str());
  const size_t keep = std::min<size_t>(pl, (size_t)o->name_stride - 1);
  memcpy(nr, b.data(), keep); memset(nr + keep, 0, (size_t)o->name_stride - keep);
}

// kseq_read3_fpc (libbwa/kseq.h:327-371) over memory.  Returns 1: a record (tokens set, *pp behind it); 0: end of input (final) or
// more bytes needed (!final; *pp unchanged -- the caller carries [*pp, end) over); -1: error (r->err).  -2: a last record that the
// reference's reader drops (no line end after its quality string).
int exact_next(fq_fastq *r, const uint8_t **pp, const uint8_t *end, bool final, std::string &name, std::string &seq, std::string &qual) {
  const uint8_t *p = *pp;
  while (p < end && *p != '@' && *p != '>') ++p;
  if (p == end) { *pp = end; return 0; }   // (what precedes a record is skipped for good)
  const uint8_t *rec = p;
  ++p;
  name.clear(); seq.clear();
  while (p < end && !is_space(*p)) name.push_back((char)*p++);
  if (p == end) { if (!final) { *pp = rec; return 0; } }
  if (p < end && *p != '\n') { while (p < end && *p != '\n') ++p; }
  if (p < end) ++p;                                             // the line end of the name line
  else if (!final) { *pp = rec; return 0; }
  int c = -1;
  while (p < end && (c = *p) != '+' && c != '>' && c != '@') { if (is_graph(c)) seq.push_back((char)c); ++p; }
  if (p == end) {
    if (!final) { *pp = rec; return 0; }
    c = -1;
  }
  if (c != '+') { r->err = "FASTA input is not supported by align (no quality line for " + name + ")"; return -1; }
  while (p < end && *p != '\n') ++p;
  if (p < end) ++p; else if (!final) { *pp = rec; return 0; }
  // ks_get_bulk (kseq.h:86-102) fails when the quality bytes reach the end of the file, also when they end exactly there: a last
  // record without a line end (or with a short quality string) ends the input and is not returned
  if ((size_t)(end - p) < seq.size() + 1) {
    if (!final) { *pp = rec; return 0; }
    r->dropped_name = name; *pp = end; return -2;
  }
  qual.assign((const char *)p, seq.size());
  p += seq.size();
  if (*p != '\n') { r->err = "Error:" + name + " this fastq file contains reads with different length"; return -1; }   // kseq.h:362-365
  *pp = p + 1;
  return 1;
}

// kseq_read4_fpc (libbwa/kseq.h:371-402) over memory: the record --frac_samp drops.  It counts every byte between the name line and
// the '+' (line ends included), skips the '+' line and then that count minus one bytes (ks_shift_bulk advances len - 1), and wants
// a line end next.  Returns as exact_next: 1 dropped, 0 end / more bytes needed, -1 error.
int exact_skip(fq_fastq *r, const uint8_t **pp, const uint8_t *end, bool final) {
  const uint8_t *p = *pp;
  while (p < end && *p != '@' && *p != '>') ++p;
  if (p == end) { *pp = end; return 0; }
  const uint8_t *rec = p;
  ++p;
  std::string name;
  while (p < end && !is_space(*p)) name.push_back((char)*p++);
  if (p == end && !final) { *pp = rec; return 0; }
  while (p < end && *p != '\n') ++p;
  if (p < end) ++p; else if (!final) { *pp = rec; return 0; }
  size_t count = 0;
  int c = -1;
  while (p < end && (c = *p) != '+' && c != '>' && c != '@') { ++count; ++p; }
  if (p == end) { if (!final) { *pp = rec; return 0; } c = -1; }
  if (c != '+') { *pp = p; return 1; }                     // (FASTA-shaped: returned as it is there; the next call starts at the '>' / '@')
  while (p < end && *p != '\n') ++p;
  if (p == end) { if (!final) { *pp = rec; return 0; } *pp = end; return 0; }
  ++p;
  const size_t skip = count ? count - 1 : 0;
  if ((size_t)(end - p) < skip + 1) {
    if (!final) { *pp = rec; return 0; }
    *pp = end; return 0;                                  // the bytes run out inside the quality string: the reader stops (-2 there)
  }
  p += skip;
  if (*p != '\n') { r->err = "Error:" + name + " this fastq file contains reads with different length"; return -1; }
  *pp = p + 1;
  return 1;
}

// next block from the producer (with the carry in front of it)
bool next_block(fq_fastq *r) {
  if (r->have_cur) {
    std::lock_guard<std::mutex> lk(r->mu);
    r->spare.push_back(std::move(r->cur));
    r->have_cur = false;
  }
  if (!r->started) { r->started = true; r->producer = std::thread(producer_main, r); }
  Block b;
  {
    std::unique_lock<std::mutex> lk(r->mu);
    r->cv_full.wait(lk, [&] { return !r->ready.empty(); });
    b = std::move(r->ready.front());
    r->ready.pop_front();
  }
  r->cv_free.notify_all();
  if (!b.err.empty()) { r->err = r->path + ": " + b.err; return false; }
  if (r->carry.size() > b.head) {   // (a carry larger than the room in front: a record of more than a megabyte)
    Block nb;
    nb.cap = r->carry.size() + b.n + 16; nb.data.reset(new uint8_t[nb.cap]); nb.head = r->carry.size(); nb.n = b.n; nb.last = b.last;
    memcpy(nb.data.get() + nb.head, b.data.get() + b.head, b.n);
    b = std::move(nb);
  }
  if (!r->carry.empty()) memcpy(b.data.get() + b.head - r->carry.size(), r->carry.data(), r->carry.size());
  r->cur_pos = b.head - r->carry.size();
  r->cur_end = b.head + b.n;
  r->carry.clear();
  r->cur = std::move(b);
  r->have_cur = true;
  return true;
}
}  // namespace

extern "C" int fq_fastq_open(const char *path, int threads, fq_fastq_t **out) {
  if (!path || !out) return FQ_EINVAL;
  *out = nullptr;
  std::unique_ptr<fq_fastq> r(new fq_fastq);
  r->path = path;
  r->threads = threads > 0 ? threads : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  uint8_t h[4096];
  struct stat sb;
  const bool regular = stat(path, &sb) == 0 && S_ISREG(sb.st_mode);
  size_t hn = 0;
  if (regular) {   // (a pipe cannot be looked at twice: it goes through gzread, which detects gzip by itself)
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return FQ_EIO;
    const ssize_t got = read(fd, h, sizeof h);
    hn = got > 0 ? (size_t)got : 0;
    if (looks_bgzf(h, hn)) { r->bgzf = true; r->fd = fd; lseek(fd, 0, SEEK_SET); }
    else if (hn >= 18 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8) {   // gzip, not BGZF: the stream decoder (FASTQUICK_ZLIB_INFLATE=1: gzread, to compare)
      r->gz_regular = true;
      const char *zenv = getenv("FASTQUICK_ZLIB_INFLATE");
      if (zenv && *zenv && *zenv != '0') close(fd);
      else {
        r->gs.on = true; r->fd = fd; lseek(fd, 0, SEEK_SET);
        r->gs.in.resize((size_t)8 << 20);
        r->gs.d.begin(r->gs.Z, r->gs.in.data(), 0, nullptr, 0);
      }
    }
    else close(fd);
  }
  if (r->bgzf) r->cbuf.resize((size_t)32 << 20);
  else if (r->gs.on) {}
  else {
    r->gz = gzopen(path, "rb");
    if (!r->gz) return FQ_EIO;
    gzbuffer(r->gz, r->gz_regular ? 1 << 16 : 1 << 20);      // (a gzip file's blocks are read by direct gzreads: see gz_fallback)
  }
  *out = r.release();
  return FQ_OK;
}
extern "C" void fq_fastq_close(fq_fastq_t *r) { delete r; }
extern "C" const char *fq_fastq_last_error(const fq_fastq_t *r) { return r ? r->err.c_str() : ""; }
extern "C" int fq_fastq_is_bgzf(const fq_fastq_t *r) { return r && r->bgzf ? 1 : 0; }
extern "C" int fq_fastq_configure(fq_fastq_t *r, int32_t batch_pairs, int32_t slot_mode, int64_t block_bytes) {
  if (!r || r->started || batch_pairs < 1 || slot_mode < 0 || slot_mode > 2) return FQ_EINVAL;
  r->batch_pairs = batch_pairs; r->slot_mode = slot_mode;
  if (block_bytes > 0) r->block_bytes = (size_t)std::max<int64_t>(block_bytes, 256);
  return FQ_OK;
}
int fq_fastq_resume(fq_fastq_t *r, int64_t file_offset, const uint8_t *text, size_t n_text, int64_t records_seen,
                    const uint8_t *slot_names, const uint8_t *slot_bases, const uint16_t *slot_lens, int shorter_after_longer) {
  if (!r || r->started || !r->bgzf || file_offset < 0 || (n_text && !text)) return FQ_EINVAL;
  if (lseek(r->fd, (off_t)file_offset, SEEK_SET) < 0) return FQ_EIO;
  r->cpos = r->cend = 0; r->file_eof = false;
  r->carry.assign(text, text + n_text);
  r->records_seen = records_seen;
  if (shorter_after_longer) r->shorter_after_longer.store(1, std::memory_order_relaxed);
  if (r->slot_mode != FQ_FASTQ_SLOTS_FRESH && (slot_names || slot_bases || slot_lens)) {
    const size_t B = (size_t)r->batch_pairs;
    for (int s = 0; s < 2; ++s) {
      r->slot_base[s].assign(B * 96, 0);
      r->slot_len[s].assign(B, 0);
      if (r->slot_mode == FQ_FASTQ_SLOTS_REUSED) r->slot_name[s].assign(B, std::string());
      for (size_t k = 0; k < B; ++k) {
        const size_t slot = (size_t)s * B + k;
        if (slot_bases) memcpy(&r->slot_base[s][k * 96], slot_bases + slot * 96, 96);
        if (slot_lens) r->slot_len[s][k] = slot_lens[slot];
        if (slot_names && r->slot_mode == FQ_FASTQ_SLOTS_REUSED) {
          // the slot's buffer as the reader keeps it: every byte up to the last one ever written (bytes behind a terminator included)
          const uint8_t *b = slot_names + slot * 304;
          size_t used = 304;
          while (used > 0 && b[used - 1] == 0) --used;
          r->slot_name[s][k].assign((const char *)b, used);
        }
      }
    }
  }
  return FQ_OK;
}
extern "C" int fq_fastq_set_sampling(fq_fastq_t *r, double frac) {
  if (!r || r->started || !(frac >= 0.0)) return FQ_EINVAL;
  r->frac = frac;
  return FQ_OK;
}
// the member decoder and the checksum on their own (tests compare them with zlib's)
extern "C" int fq_inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t out_len) {
  if ((!src && n) || (!dst && out_len)) return FQ_EINVAL;
  std::unique_ptr<fqz::Inflater> z(new fqz::Inflater);
  return fqz::inflate_raw(*z, src, n, dst, out_len) ? FQ_OK : FQ_EIO;
}
extern "C" uint32_t fq_crc32(const uint8_t *p, size_t n) { return p || !n ? fqz::crc32(p, n) : 0u; }
extern "C" int fq_fastq_unequal_lengths(const fq_fastq_t *r) { return r ? r->shorter_after_longer.load(std::memory_order_relaxed) : 0; }
extern "C" const char *fq_fastq_dropped_record(const fq_fastq_t *r) { return r && r->notice_dropped ? r->dropped_name.c_str() : nullptr; }

extern "C" int64_t fq_fastq_read(fq_fastq_t *r, int64_t max_reads, const fq_fastq_rows_t *o) {
  if (!r || !o || max_reads < 0 || !o->seq || !o->qual || !o->len || !o->names || o->stride < 1 || o->name_stride < 2) return FQ_EINVAL;
  if (!r->err.empty()) return FQ_EIO;
  int64_t produced = 0;
  if (r->slot_mode != FQ_FASTQ_SLOTS_FRESH) {
    for (int s = 0; s < 2; ++s) {
      if (r->slot_base[s].empty()) r->slot_base[s].assign((size_t)r->batch_pairs * 96, 0);
      if (r->slot_len[s].empty()) r->slot_len[s].assign((size_t)r->batch_pairs, 0);
      if (r->slot_mode == FQ_FASTQ_SLOTS_REUSED && r->slot_name[s].empty()) r->slot_name[s].resize((size_t)r->batch_pairs);
    }
  }
  std::string nm, sq, ql;
  while (produced < max_reads && !r->at_eof) {
    if (!r->have_cur || r->cur_pos == r->cur_end) {
      if (r->have_cur && r->cur.last) { r->at_eof = true; break; }
      if (!next_block(r)) return FQ_EIO;
      if (r->cur_pos == r->cur_end && r->cur.last) { r->at_eof = true; break; }
    }
    const uint8_t *base = r->cur.data.get();
    const uint8_t *p0 = base + r->cur_pos, *end = base + r->cur_end;
    const bool final = r->cur.last;
    // ---- lines of the block, by all threads ----
    std::vector<std::vector<uint32_t>> part;
    {
      const size_t n = (size_t)(end - p0);
      const int T = (int)std::min<size_t>((size_t)r->threads, std::max<size_t>(1, n >> 20));
      part.resize((size_t)T);
      par_for(r->pool_tok, T, (size_t)T, 1, [&](size_t lo, size_t hi, int) {
        for (size_t t = lo; t < hi; ++t) {
          const size_t per = (n + T - 1) / T, a = std::min(n, t * per), b = std::min(n, a + per);
          std::vector<uint32_t> &v = part[t];
          v.reserve((b - a) / 64 + 16);
          const uint8_t *q = p0 + a, *e = p0 + b;
          while (q < e && (q = (const uint8_t *)memchr(q, '\n', (size_t)(e - q))) != nullptr) { v.push_back((uint32_t)(q - p0)); ++q; }
        }
      });
    }
    std::vector<size_t> part_off(part.size() + 1, 0);
    for (size_t t = 0; t < part.size(); ++t) part_off[t + 1] = part_off[t] + part[t].size();
    const size_t n_lines = part_off.back();
    // one array of line ends (random access by record below)
    std::vector<uint32_t> flat;
    const uint32_t *nl = nullptr;
    if (part.size() == 1) nl = part[0].data();
    else { flat.resize(n_lines); for (size_t t = 0; t < part.size(); ++t) if (!part[t].empty()) memcpy(flat.data() + part_off[t], part[t].data(), part[t].size() * 4); nl = flat.data(); }
    // ---- --frac_samp: which of the block's records are kept is a serial walk of the batch's generator (a draw per record met)
    const bool sampling = r->frac < 1.0;
    std::vector<uint32_t> kept;                  // record index of the j-th kept record of this block (sampling only)
    size_t n_met = n_lines / 4;                   // records of the block this call looks at
    const fq_fastq::Sampler sampler0 = r->sampler;
    if (sampling) {
      const size_t want = (size_t)(max_reads - produced);
      size_t i = 0;
      for (; i < n_lines / 4 && kept.size() < want; ++i) if (r->sampler.keep(r->frac, r->batch_pairs)) kept.push_back((uint32_t)i);
      n_met = i;
    }
    const size_t n_fast = sampling ? n_met : (size_t)std::min<int64_t>((int64_t)(n_lines / 4), max_reads - produced);
    // ---- four-line records, checked and emitted by all threads ----
    std::vector<int64_t> row_of;                  // sampling: output row of record i, or -1 for a dropped one
    if (sampling) { row_of.assign(n_fast, -1); for (size_t j = 0; j < kept.size(); ++j) row_of[kept[j]] = (int64_t)j; }
    std::atomic<size_t> first_bad{n_fast};
    std::mutex err_mu;
    std::string emit_err;
    size_t emit_err_at = n_fast;
    par_for(r->pool_tok, r->threads, n_fast, 2048, [&](size_t lo, size_t hi, int) {
      std::string lerr;
      for (size_t i = lo; i < hi; ++i) {
        if (i >= first_bad.load(std::memory_order_relaxed)) break;
        const uint8_t *l0 = p0 + (i == 0 ? 0 : (size_t)nl[4 * i - 1] + 1);
        const uint8_t *e0 = p0 + nl[4 * i], *e1 = p0 + nl[4 * i + 1], *e2 = p0 + nl[4 * i + 2], *e3 = p0 + nl[4 * i + 3];
        const uint8_t *l1 = e0 + 1, *l2 = e1 + 1, *l3 = e2 + 1;
        bool ok = l0 < e0 && *l0 == '@' && l2 < e2 && *l2 == '+' && (e1 - l1) == (e3 - l3);
        if (ok) {   // a base line kseq reads as one token: printable, none of the record markers (no early exit: one table look-up per byte)
          static const struct Bad { uint8_t t[256]; Bad() { for (int c = 0; c < 256; ++c) t[c] = (uint8_t)(!is_graph(c) || c == '+' || c == '>' || c == '@'); } } bad;
          unsigned any = 0;
          for (const uint8_t *q = l1; q < e1; ++q) any |= bad.t[*q];
          ok = any == 0;
        }
        if (!ok) { size_t cur = first_bad.load(); while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {} break; }
        const uint8_t *nb = l0 + 1, *ne = nb;
        while (ne < e0 && !is_space(*ne)) ++ne;
        if (sampling && row_of[i] < 0) continue;            // dropped: a well-formed record is skipped whole by kseq_read4_fpc too
        if (emit(o, produced + (sampling ? row_of[i] : (int64_t)i), nb, (size_t)(ne - nb), l1, l3, (size_t)(e1 - l1), &lerr)) {
          std::lock_guard<std::mutex> lk(err_mu);
          if (i < emit_err_at) { emit_err_at = i; emit_err = lerr; }
          size_t cur = first_bad.load(); while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
          break;
        }
      }
    });
    const size_t n_ok = first_bad.load();
    if (emit_err_at == n_ok && !emit_err.empty()) r->err = emit_err;   // (an odd record in front of it goes the byte-wise way first)
    // ---- slot history for the records emitted so far: windows of 2 * batch_pairs records have distinct slots ----
    auto apply_slots = [&](int64_t row0, size_t count) {
      const size_t W = (size_t)2 * (size_t)r->batch_pairs;
      size_t done = 0;
      while (done < count) {
        const long long g0 = r->records_seen + (long long)done;
        const size_t w = std::min(count - done, W - (size_t)(g0 % (long long)W));
        par_for(r->pool_tok, r->threads, w, 4096, [&](size_t lo, size_t hi, int) { for (size_t k = lo; k < hi; ++k) slot_apply(r, o, row0 + (int64_t)(done + k), g0 + (long long)k); });
        done += w;
      }
      r->records_seen += (long long)count;
    };
    size_t n_emit = n_ok;
    if (sampling) {   // records 0 .. n_ok-1 were met; the generator goes back to where it stood in front of record n_ok
      n_emit = (size_t)(std::lower_bound(kept.begin(), kept.end(), (uint32_t)n_ok) - kept.begin());
      if (n_ok < n_met) { r->sampler = sampler0; for (size_t i = 0; i < n_ok; ++i) (void)r->sampler.keep(r->frac, r->batch_pairs); }
    }
    apply_slots(produced, n_emit);
    if (!r->err.empty()) return FQ_EIO;
    produced += (int64_t)n_emit;
    const uint8_t *p = p0 + (n_ok == 0 ? 0 : (size_t)nl[4 * n_ok - 1] + 1);
    // ---- what the four-line reading did not take: byte-wise, as kseq_read3_fpc reads it ----
    const bool fast_complete = n_ok == n_fast;
    if (!fast_complete || (n_fast == n_lines / 4 && produced < max_reads)) {
      fq_fastq::Sampler before_draw = r->sampler;
      // either an odd record (then the rest of the block goes this way), or fewer than four lines are left in the block
      while (produced < max_reads) {
        const uint8_t *q = p;
        before_draw = r->sampler;
        if (sampling && !r->sampler.keep(r->frac, r->batch_pairs)) {
          const int rs = exact_skip(r, &q, end, final);
          if (rs == 1) { p = q; continue; }
          if (rs == -1) return FQ_EIO;
          r->sampler = before_draw;            // the record is met again with the next block (or there was none)
          if (final) { p = end; r->at_eof = true; }
          else { r->carry.assign(q, end); p = end; }
          break;
        }
        const int rc = exact_next(r, &q, end, final, nm, sq, ql);
        if (rc == 1) {
          std::string lerr;
          if (emit(o, produced, (const uint8_t *)nm.data(), nm.size(), (const uint8_t *)sq.data(), (const uint8_t *)ql.data(), sq.size(), &lerr)) { r->err = lerr; return FQ_EIO; }
          apply_slots(produced, 1);
          ++produced;
          p = q;
          continue;
        }
        if (rc == -1) return FQ_EIO;
        if (rc == -2) { r->notice_dropped = true; p = end; r->at_eof = true; break; }
        // rc == 0: end of input, or the record continues in the next block
        r->sampler = before_draw;
        if (final) { p = end; r->at_eof = true; }
        else { r->carry.assign(q, end); p = end; }
        break;
      }
    }
    r->cur_pos = (size_t)(p - base);
  }
  return produced;
}
