// fq_cli.cpp -- `FASTQuick_amd align`: the reference's `FASTQuick align --sam_out` command line on top of the C ABI.
//
// Mirrors runAlign (src/FASTQuick.cpp:159-488): same flag names and meanings for the flags the hot path reads, the
// index prefix convention (<index_prefix>.FASTQuick.fa.*), SAM text on stdout in the --sam_out dialect, the summary
// notices on stderr.  The FASTQ tokenizer follows kseq_read3_fpc (libbwa/kseq.h:327-370): name up to the first white
// space, bases = printable characters up to the '+' line, quality = exactly as many characters as bases.
// Without --sam_out the records go to <out_prefix>.bam in genome coordinates (fq_bam_*: SetSamRecord / SetSamFileHeader); the QC
// files of StatCollector (<out_prefix>.InsertSizeTable .Pileup .DepthDist ... .Summary) are written in both modes (fq_qc_*) when the
// index carries its .SelectedSite.vcf / .dbSNP.subset.vcf / .gc.  Flank lengths and the original reference (for @SQ and the genome
// size) come from <index_prefix>.FASTQuick.fa.param as `FASTQuick index` wrote it (src/FASTQuick.cpp:376-465).
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <chrono>
#include <cstdio>
#include <map>
#include <functional>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"

namespace {
// (_exit after flushing: other threads may be inside the HIP runtime -- the index is staged beside the first read -- and must not meet the
//  process's static destructors half-way)
// part / worker files of a multi-device run: removed when the run dies (a finished run appends and removes them itself)
std::mutex g_tmp_mu;
std::vector<std::string> g_tmp_files;
void tmp_file(const std::string &p) { std::lock_guard<std::mutex> lk(g_tmp_mu); g_tmp_files.push_back(p); }
[[noreturn]] void die(const std::string &m) {
  fprintf(stderr, "FATAL ERROR - \n%s\n", m.c_str());
  { std::lock_guard<std::mutex> lk(g_tmp_mu); for (const auto &p : g_tmp_files) remove(p.c_str()); }
  fflush(nullptr);
  _exit(EXIT_FAILURE);
}
// FASTQUICK_TRACE=1: wall-clock marks of the run's phases on stderr (milliseconds since the process began)
const std::chrono::steady_clock::time_point g_t0 = std::chrono::steady_clock::now();
const bool g_trace = [] { const char *e = getenv("FASTQUICK_TRACE"); return e && *e && *e != '0'; }();
// device objects being destroyed beside the run's next steps: release_later() starts, release_join() waits (before the index they refer to goes, and
// at the end).  (The list is never destroyed itself: a die() while a release runs must not meet a joinable thread in a static destructor.)
std::mutex g_releases_mu;
std::vector<std::thread> &releases() { static std::vector<std::thread> *v = new std::vector<std::thread>; return *v; }
template <class F> void release_later(F f) { std::lock_guard<std::mutex> lk(g_releases_mu); releases().emplace_back(std::move(f)); }
void release_join() {
  std::vector<std::thread> mine;
  { std::lock_guard<std::mutex> lk(g_releases_mu); mine.swap(releases()); }
  for (auto &t : mine) t.join();
}
void mark(const char *what) { if (g_trace) fprintf(stderr, "TRACE - %9.1f ms  %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_t0).count(), what); }
void notice(const char *fmt, long long a) { fprintf(stderr, "NOTICE - "); fprintf(stderr, fmt, a); fputc('\n', stderr); }

// One FASTQ file through the library's front end (fq_fastq_*: parallel inflate + tokeniser with kseq_read3_fpc's tokens)
struct FastqReader {
  fq_fastq_t *h = nullptr;
  std::string path;
  bool told_dropped = false;
  FastqReader(const std::string &p, int threads, int batch_pairs, int slot_mode, double frac) : path(p) {
    if (fq_fastq_open(p.c_str(), threads, &h)) die("Open " + p + " failed!");
    fq_fastq_configure(h, batch_pairs, slot_mode, 0);
    if (fq_fastq_set_sampling(h, frac)) die("--frac_samp must not be negative");
  }
  FastqReader(const std::string &p, fq_fastq_t *adopt) : h(adopt), path(p) {}      // a reader the device front end handed over (fq_frontend_handover)
  ~FastqReader() { if (h) fq_fastq_close(h); }
  FastqReader(const FastqReader &) = delete;
};
// length of a file's first read (0: none), to size the rows
size_t first_read_len(const std::string &path) {
  fq_fastq_t *h = nullptr;
  if (fq_fastq_open(path.c_str(), 1, &h)) return 0;
  fq_fastq_configure(h, 1, FQ_FASTQ_SLOTS_FRESH, 1 << 16);
  std::vector<uint8_t> seq(65536), qual(65536);
  std::vector<char> nm(512);
  int32_t len = 0;
  fq_fastq_rows_t rows = {65536, 512, seq.data(), qual.data(), &len, nm.data()};
  const int64_t n = fq_fastq_read(h, 1, &rows);
  fq_fastq_close(h);
  return n == 1 ? (size_t)len : 0;
}

// One end of one chunk, in the layout fq_read_batch_t wants: fixed-stride rows, filled by that file's reader thread.
// grow-only byte storage that is not cleared on allocation: a chunk of the default size is 2.7 GB of rows, rows are cleared as
// records arrive, and value-initialising the rest cost seconds per run
template <class T> struct RawBuf {
  std::unique_ptr<T[]> p;
  size_t n = 0;
  void resize(size_t m) { if (m > n) { p.reset(new T[m]); n = m; } }     // (contents are not kept: callers fill what they use)
  size_t size() const { return n; }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
};
struct EndChunk {
  RawBuf<uint8_t> seq, qual;
  uint8_t *ext_seq = nullptr, *ext_qual = nullptr;   // rows elsewhere (the shared buffer of a pair chunk) instead of seq / qual
  std::vector<int32_t> len;
  RawBuf<char> names;
  int n = 0, stride = 0, name_stride = 0;
  bool eof = false;
  std::string error;
};
// Reads up to `cap` records of one FASTQ file into `c` (rows of `stride` bytes; a read longer than that is an error: the
// reference, too, wants reads of one length, kseq.h:362-365).  The read-slot history of the reference (names without terminator,
// bases behind a short read: SURVEY Q7 / Q8) is modelled by the reader (fq_fastq_configure).
void fill_chunk(FastqReader &r, EndChunk &c, long long cap, int stride, int name_stride) {
  c.n = 0; c.stride = stride; c.name_stride = name_stride; c.error.clear();
  // (rows are cleared as records arrive: a chunk of the default size is 2.7 GB of rows, a small input touches a few of them)
  if (!c.ext_seq && c.seq.size() < (size_t)cap * stride) { c.seq.resize((size_t)cap * stride); c.qual.resize((size_t)cap * stride); }
  if (c.len.size() < (size_t)cap) c.len.resize((size_t)cap);
  if (c.names.size() < (size_t)cap * name_stride) c.names.resize((size_t)cap * name_stride);
  fq_fastq_rows_t rows = {stride, name_stride, c.ext_seq ? c.ext_seq : c.seq.data(), c.ext_qual ? c.ext_qual : c.qual.data(), c.len.data(), c.names.data()};
  const int64_t n = fq_fastq_read(r.h, cap, &rows);
  if (n < 0) { c.error = fq_fastq_last_error(r.h); if (c.error.empty()) c.error = "reading " + r.path + " failed (" + std::to_string(n) + ")"; return; }
  c.n = (int)n;
  if (n < cap) c.eof = true;
  if (!r.told_dropped && fq_fastq_dropped_record(r.h)) {
    r.told_dropped = true;
    fprintf(stderr, "NOTICE - the last record (%s) has no line end after its quality string; like the reference's reader, it is not used\n", fq_fastq_dropped_record(r.h));
  }
}

struct Args {
  std::string fq1, fq2, out_prefix = "Empty", index_prefix = "Empty";
  bool sam_out = false;
  fq_opts_t o;
  int opte = -1;
  long long chunk_pairs = 16LL * 262144;
  int device = 0;
  std::string devices;   // --devices: several devices, the lines of --fq_list dealt over them
  int pack_threads = std::min(32, std::max(1, fq_host_cpus()));     // host threads of the FASTQ readers (half per file) and of the packer, from the CPUs the process may use; --t sets it
  bool clean_names = false;
  bool host_consumers = false;   // --host_consumers: SAM text and StatCollector's sums on the host's threads from the result arrays (round 5's way; the default runs them in kernels, fq_emit.h)
  bool host_reader = false;   // --host_reader: the FASTQ front end on the host's threads also for BGZF files (the default inflates and tokenises them on the device)
  bool strict = false;   // --strict_reference: stop where the output could differ from the reference's bytes (today: QUAL of reads of unequal lengths)
  std::string fq_list, rg = "@RG\\tID:foo\\tSM:bar";   // runAlign's default --RG (src/FASTQuick.cpp:170)
  bool cal_dup = true;
  double frac = 1.0;    // --frac_samp: gap_opt_t::frac (libbwa/bwtaln.c:47), the share of the records that is kept
  int read_len = 151;   // gap_opt_t::read_len (libbwa/bwtaln.c:48): the reference sizes its read buffers from it and has no flag for it
};

int usage() {
  fprintf(stderr, "Usage: FASTQuick_amd align --index_prefix P --fastq_1 R1.fq[.gz] [--fastq_2 R2.fq[.gz]] | --fq_list LIST  --out_prefix O [--sam_out] [--RG STR] [--cal_dup]\n"
                  "                       [--q INT] [--n FLOAT|INT] [--kmer_thresh INT] [--o INT] [--e INT] [--i INT] [--d INT] [--l INT] [--k INT]\n"
                  "                       [--m INT] [--R INT] [--N] [--L] [--I] [--max_isize INT] [--max_occ INT] [--is_sw] [--n_multi INT] [--N_multi INT]\n"
                  "                       [--ap_prior FLOAT] [--force_isize] [--frac_samp FLOAT] [--t INT] [--chunk_pairs INT] [--batch_pairs INT] [--device INT | --devices LIST] [--read_len INT] [--clean_names] [--strict_reference] [--host_reader] [--host_consumers]\n"
                  "       FASTQuick_amd index --ref REDUCED.FASTQuick.fa [--rollhash]\n");
  return 1;
}

// Where a worker's records go.  One device: straight to stdout / the BAM file, as they are produced.  Several devices: every FASTQ pair
// into part files of its own (SAM text; BAM records as bytes), which the main thread appends to the output in input order.
struct Sink {
  bool sam_out = false;
  FILE *sam_fp = nullptr;        // stdout, or the input's part file
  fq_bam_t *bam = nullptr;       // the file's writer (direct), or a formatter without a file
  FILE *bam_fp = nullptr;        // the input's part file of BAM records (null: direct)
  std::string what;
  void sam(const char *p, size_t n) { if (fwrite(p, 1, n, sam_fp) != n) die("writing " + what + " failed"); }
  void bam_add(fq_ctx_t *ctx) {
    if (!bam_fp) { if (fq_bam_add_last(bam, ctx)) die("writing " + what + " failed"); return; }
    const void *data = nullptr; int64_t len = 0;
    if (fq_bam_format_last(bam, ctx, &data, &len) || (len && fwrite(data, 1, (size_t)len, bam_fp) != (size_t)len)) die("writing " + what + " failed");
  }
  void flush() { if (sam_fp) fflush(sam_fp); if (bam_fp) fflush(bam_fp); }
};
// "0-3", "0,2,5", "0,0": the devices of --devices (a device named twice gets two workers)
std::vector<int> parse_devices(const std::string &v) {
  std::vector<int> d;
  auto ordinal = [&](const std::string &t) -> int {       // digits only: "a", "0-x", "1.5" are errors, not device 0
    char *e = nullptr;
    const long x = t.empty() ? -1 : strtol(t.c_str(), &e, 10);
    if (t.empty() || !e || *e || t.find_first_not_of("0123456789") != std::string::npos || x < 0 || x > 63) die("--devices: bad device ordinal '" + t + "' in " + v);
    return (int)x;
  };
  size_t at = 0;
  while (at < v.size()) {
    size_t end = v.find(',', at);
    if (end == std::string::npos) end = v.size();
    const std::string tok = v.substr(at, end - at);
    const size_t dash = tok.find('-');
    if (tok.empty()) die("--devices: empty entry in " + v);
    if (dash == std::string::npos) d.push_back(ordinal(tok));
    else { const int a = ordinal(tok.substr(0, dash)), b = ordinal(tok.substr(dash + 1)); if (b < a) die("--devices: bad range " + tok); for (int x = a; x <= b; ++x) d.push_back(x); }
    at = end + 1;
  }
  if (d.empty() || (!v.empty() && v.back() == ',')) die("--devices: empty entry in " + v);
  return d;
}
void append_file(const std::string &path, FILE *to, fq_bam_t *bam) {   // a part file onto the output (SAM text to `to`, BAM records to `bam`), then gone
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) die("cannot reopen " + path);
  std::vector<char> buf((size_t)8 << 20);
  size_t n;
  while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
    if (to ? fwrite(buf.data(), 1, n, to) != n : fq_bam_write_records(bam, buf.data(), (int64_t)n) != 0) die("appending " + path + " to the output failed");
  }
  fclose(f);
  remove(path.c_str());
}

// The one column of the reference's output that is not reproduced: said loudly, or refused.
void unequal_lengths_notice(const Args &A, const fq_fastq_t *a, const fq_fastq_t *b, bool seen_on_device = false) {
  if (!seen_on_device && !(a && fq_fastq_unequal_lengths(a)) && !(b && fq_fastq_unequal_lengths(b))) return;
  const char *msg = "reads of unequal lengths: the reference prints the QUAL column of a read that follows a longer read in its read slot with that read's "
                    "tail behind it (an unterminated buffer: src/BwtMapper.cpp:549-558, libbwa/bwase.c:401; longer than SEQ, not valid SAM); here QUAL has "
                    "the read's own length -- every other column, and every QC file, is the reference's";
  if (A.strict) die(std::string("--strict_reference: ") + msg);
  fprintf(stderr, "WARNING - %s\n", msg);
}
// One FASTQ pair (or one single-end file) through a device: an independent stream -- its own context (srand48, last_ii, position cache)
// and read slots, as PairEndMapper / SingleEndMapper set them up per call (src/BwtMapper.cpp:232-262).
// FASTQ front end: one reader thread per file tokenises the next chunk into flat buffers while the device aligns the current
// one (the reference, too, decodes the two files on two IO threads: BwtMapper.cpp:1873-1935).
// `ready`: called once the input's first chunk has been read and before anything needs the index, the QC consumer or the sink -- the
// one-device command line stages the index (about a second: the filter bitmaps are built on the device) on another thread meanwhile.
// The front end of an input: on the device for BGZF files (fq_frontend_*: the host reads compressed bytes, everything per byte of text
// happens in HBM), on the host's threads for anything else (fq_fastq_*) -- and from the record on at which the device hands over (a
// record that is not four plain lines, a file that ends inside a record: fq_frontend_handover gives host readers standing exactly there).
fq_frontend_t *open_device_front_end(const Args &A, const std::string &f1, const std::string &f2, int device, int slot_mode, int stride) {
  if (A.host_reader || A.frac < 1.0) return nullptr;       // (--frac_samp: the reference's generator is walked record by record, on the host)
  fq_frontend_t *fe = nullptr;
  const int rc = fq_frontend_open(device, f1.c_str(), f2.empty() ? nullptr : f2.c_str(), A.o.batch_pairs, A.chunk_pairs, slot_mode, stride, &fe);
  mark("front end open");
  if (rc == FQ_EIO) return nullptr;                          // not a regular BGZF file: the host reader's
  if (rc) die("cannot start the front end on device " + std::to_string(device) + " (" + std::to_string(rc) + ")");
  return fe;
}
void front_end_notice(fq_frontend_t *fe) {
  fq_frontend_stats_t st;
  fq_frontend_stats(fe, &st);
  fprintf(stderr, "NOTICE - front end on the device: %lld pairs, %lld BGZF members (%lld left to the host's decoder), %.1f MB -> %.1f MB of text; inflate %.1f ms ; lines, records, keys, slots %.1f ms\n",
          (long long)st.pairs, (long long)st.members, (long long)st.refused, 1e-6 * (double)st.comp_bytes, 1e-6 * (double)st.text_bytes, st.ms_inflate, st.ms_tokenise);
}

// One FASTQ pair (or one single-end file) through a device: an independent stream -- its own context (srand48, last_ii, position cache)
// and read slots, as PairEndMapper / SingleEndMapper set them up per call (src/BwtMapper.cpp:232-262).
// `ready`: called once the input's first chunk has been read and before anything needs the index, the QC consumer or the sink -- the
// one-device command line stages the index (about a second: the filter bitmaps are built on the device) on another thread meanwhile.
void align_input(Args A, const std::pair<std::string, std::string> &input, fq_index_t *const &ix_ref, fq_qc_t *const &qc_ref, Sink &out, const std::function<void()> &ready, int device) {
  int rc;
  A.fq1 = input.first; A.fq2 = input.second;
  const bool se = A.fq2.empty() || A.fq2 == "Empty";
  if (se) A.fq2.clear();
  if (se) fprintf(stderr, "NOTICE - Processing Single End mapping\t%s\n", A.fq1.c_str());
  else fprintf(stderr, "NOTICE - Processing Pair End mapping\t%s\t%s\n", A.fq1.c_str(), A.fq2.c_str());
  // BwtMapper::SingleEndMapper (src/BwtMapper.cpp:1266-1407): its reader hands out fresh zeroed buffers (bwa_read_seq_with_hash, :350-475),
  // so neither bases nor name tails of earlier reads linger
  const int slot_mode = se ? FQ_FASTQ_SLOTS_FRESH : A.clean_names ? FQ_FASTQ_SLOTS_CLEAN_NAMES : FQ_FASTQ_SLOTS_REUSED;
  int stride = 0;
  {   // rows hold read_len bases, or the first records' if those are longer -- probed only in regular files (a pipe cannot be read twice:
      // there a longer read is an error that asks for --read_len)
    size_t l = 0;
    struct stat s1, s2;
    if (stat(A.fq1.c_str(), &s1) == 0 && S_ISREG(s1.st_mode) && (se || (stat(A.fq2.c_str(), &s2) == 0 && S_ISREG(s2.st_mode))))
      l = std::max(first_read_len(A.fq1), se ? (size_t)0 : first_read_len(A.fq2));
    stride = (int)((std::max<size_t>(l, (size_t)std::max(A.read_len, 16)) + 15) & ~(size_t)15);
  }
  const int name_stride = 304;   // the reference's name buffers hold 302 bytes (bwaseqio.c:233)
  const int reader_threads = se ? A.pack_threads : std::max(1, A.pack_threads / 2);
  long long num_read = 0, filtered = 0, unmapped = 0, order_checked_reads = 0;
  double qc_ms = 0, out_ms = 0, read_all_ms = 0, read_wait_ms = 0, align_ms = 0, read_ms = 0, pack_ms = 0;
  std::vector<char> sam;
  fq_index_t *ix = nullptr;
  fq_ctx_t *ctx = nullptr, *ctx_a = nullptr;      // ctx: the context that holds the stream's state; ctx_a: the first one made (a second one may join it: ctx2)
  bool started = false;
  fq_qc_t *qc = nullptr;
  // the consumers of a context's records run in kernels inside its calls (fq_emit.h): the SAM text is formatted on the device, StatCollector's sums
  // stay there; the consumer threads below only move text and append
  auto device_consumers = [&](fq_ctx_t *cx) {
    if (A.host_consumers) return;
    // (nothing on the host reads the result arrays any more: they stay in HBM)
    if (fq_ctx_set_emit(cx, (A.sam_out ? FQ_EMIT_SAM : 0) | FQ_EMIT_DEVICE_ONLY)) die("fq_ctx_set_emit failed");
    if (!A.sam_out && out.bam && fq_ctx_attach_bam(cx, out.bam)) die("fq_ctx_attach_bam failed");
    if (qc && fq_ctx_attach_qc(cx, qc)) die("fq_ctx_attach_qc failed");
  };
  // A refusal ends the run behind the records of every call before it (the reference prints the batches before the one that aborts,
  // src/BwtMapper.cpp:2030-2092): consumers of the previous call that still run on their own threads are waited for, the sink is flushed, then die().
  std::function<void()> before_die;
  auto fail = [&](const std::string &m) { if (before_die) before_die(); out.flush(); die(m); };
  auto start = [&] {                    // from here on: the index, the QC consumer, the sink
    if (started) return;
    started = true;
    mark("first chunk there; waiting for the index");
    ready();
    mark("index ready");
    ix = ix_ref; qc = qc_ref;
    if (qc) fq_qc_begin_file(qc, A.fq1.c_str(), se ? A.fq1.c_str() : A.fq2.c_str());      // FileStatCollector(fq1[, fq2]): a single file is named twice
    fq_opts_t o = A.o;
    o.single_end = se ? 1 : 0;
    const int crc = fq_ctx_create(ix, &o, (int32_t)A.chunk_pairs, &ctx);
    if (crc) die("fq_ctx_create failed (" + std::to_string(crc) + "): option outside the supported range");
    ctx_a = ctx;
    device_consumers(ctx);
    mark("context created");
  };
  // the consumers of a call's records: StatCollector and the record writer (src/BwtMapper.cpp:2047-2050, 2075-2085).  The reference runs them
  // one after the other on its main thread; neither reads what the other writes here (each applies AddAlignment's contig-bridging mutation,
  // SURVEY Q10, to its own view of a record), so consume_on() may run them side by side
  std::mutex tm_mu;
  auto consume_qc = [&](fq_ctx_t *cx) {
    const auto tc0 = std::chrono::steady_clock::now();
    if (qc && fq_qc_add_last(qc, cx)) die(std::string("QC consumer failed: ") + fq_qc_last_error(qc));
    std::lock_guard<std::mutex> lk(tm_mu);
    qc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count();
  };
  auto consume_out = [&](fq_ctx_t *cx) {
    const auto tc1 = std::chrono::steady_clock::now();
    std::vector<char> text;
    if (A.sam_out && !A.host_consumers) {
      struct U { Sink *out; } u{&out};
      if (fq_sam_device_last(cx, [](void *user, const void *data, int64_t n) -> int { ((U *)user)->out->sam((const char *)data, (size_t)n); return 0; }, &u) < 0)
        die(std::string("fetching the SAM text failed: ") + fq_ctx_last_error(cx));
    } else if (A.sam_out) {
      const int64_t sz = fq_sam_format_last(cx, nullptr, 0);
      text.resize((size_t)sz + 1);
      fq_sam_format_last(cx, text.data(), sz + 1);
      out.sam(text.data(), (size_t)sz);
    } else out.bam_add(cx);
    std::lock_guard<std::mutex> lk(tm_mu);
    out_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count();
  };
  auto count = [&](const fq_result_batch_t &res, long long n_reads) {
    num_read += n_reads; filtered += res.n_both_filtered; unmapped += res.n_both_unmapped;
    fprintf(stderr, se ? "NOTICE - %lld sequences are loaded.\n" : "NOTICE - %lld sequences are processed.\n", num_read);
  };
  auto consume = [&](const fq_result_batch_t &res, long long n_reads) { consume_qc(ctx); consume_out(ctx); count(res, n_reads); };
  // src/BwtMapper.cpp:2087-2092: the first pair of a reference batch is compared when the running read count is a multiple of the batch
  // size (every full batch; a short last batch normally is not checked), over read_len name bytes.  Mates whose names differ elsewhere
  // pass, each printed under its own name (the reference's example input has such pairs).
  auto order_check = [&](int n, const std::function<const char *(int, int)> &first_name) {
    for (int i = 0; i < n; i += A.o.batch_pairs) {
      order_checked_reads += 2LL * std::min<long long>(A.o.batch_pairs, n - i);
      if (order_checked_reads % A.o.batch_pairs == 0 && strncmp(first_name(i / A.o.batch_pairs, 0), first_name(i / A.o.batch_pairs, 1), (size_t)A.read_len) != 0)
        fail("Abort, please make sure input pair of fastq files are in the same order!");
    }
  };

  // ---- the device's part of the stream ----
  std::unique_ptr<FastqReader> r1, r2;
  bool unequal_on_device = false;
  fq_frontend_t *fe = open_device_front_end(A, A.fq1, A.fq2, device, slot_mode, stride);
  fq_ctx_t *ctx2 = nullptr;             // the device's part runs on two contexts in turn: the consumers of one call's records work while the next call runs
  if (fe) {
    std::thread th_qc, th_out;
    fq_text_batch_t *tb_prev = nullptr;
    auto finish_prev = [&] {            // the previous call's consumers are done: its batch may be reused
      if (th_qc.joinable()) th_qc.join();
      if (th_out.joinable()) th_out.join();
      if (tb_prev) { fq_frontend_release(fe, tb_prev); tb_prev = nullptr; }
    };
    before_die = [&] { if (th_qc.joinable()) th_qc.join(); if (th_out.joinable()) th_out.join(); };
    fq_ctx_t *cur = nullptr, *other = nullptr;
    for (;;) {
      const auto tr0 = std::chrono::steady_clock::now();
      fq_text_batch_t *tb = nullptr;
      const int64_t n = fq_frontend_next(fe, &tb);
      const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count();
      if (!started) read_ms += waited; else read_wait_ms += waited;
      if (n == FQ_EFALLBACK) {
        finish_prev();
        fq_fastq_t *h[2] = {nullptr, nullptr};
        if ((rc = fq_frontend_handover(fe, reader_threads, h))) fail("the device front end could not hand " + A.fq1 + " over to the host reader (" + std::to_string(rc) + ")");
        r1.reset(new FastqReader(A.fq1, h[0]));
        if (!se) r2.reset(new FastqReader(A.fq2, h[1]));
        fprintf(stderr, "NOTICE - the FASTQ text from record %lld on is not four plain lines per record: read on by the host's reader\n", num_read / (se ? 1 : 2) + 1);
        break;
      }
      if (n < 0) fail(std::string(fq_frontend_last_error(fe)).empty() ? "the device front end failed (" + std::to_string(n) + ")" : fq_frontend_last_error(fe));
      if (n == 0) break;
      start();
      if (!cur) {
        cur = ctx;
        fq_opts_t o = A.o;
        o.single_end = se ? 1 : 0;
        if (fq_ctx_create(ix, &o, (int32_t)A.chunk_pairs, &ctx2)) fail("fq_ctx_create failed: option outside the supported range");
        device_consumers(ctx2);
        other = ctx2;
      } else {
        // the stream's order-dependent state -- drand48 stream, last_ii, (k,l) cache -- goes from the context of the last call to this one's
        if (fq_ctx_state_move(cur, other)) fail("handing the stream's state from one context to the other failed");
      }
      if (!se) order_check((int)n, [&](int sb, int e) { const char *nm = fq_text_batch_first_name(tb, sb, e); return nm ? nm : ""; });
      fq_result_batch_t res;
      const auto ta0 = std::chrono::steady_clock::now();
      if ((rc = fq_align_text(cur, tb, &res))) fail(std::string("fq_align_text failed: ") + fq_ctx_last_error(cur));
      align_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta0).count();
      mark("call done");
      finish_prev();
      mark("previous call's consumers done");
      count(res, se ? n : 2 * n);
      fq_ctx_t *cx = cur;
      th_qc = std::thread([&, cx] { consume_qc(cx); });
      th_out = std::thread([&, cx] { consume_out(cx); });
      tb_prev = tb;
      std::swap(cur, other);            // (`other` now names the context of the call just made: the one whose state goes on)
    }
    finish_prev();
    before_die = nullptr;
    if (other && cur) ctx = other;       // the context that holds the stream's state (the host readers' part, if any, goes on with it)
    unequal_on_device = fq_frontend_unequal_lengths(fe) != 0;
    front_end_notice(fe);
  } else {
    r1.reset(new FastqReader(A.fq1, reader_threads, A.o.batch_pairs, slot_mode, A.frac));
    if (!se) r2.reset(new FastqReader(A.fq2, reader_threads, A.o.batch_pairs, slot_mode, A.frac));
  }

  // ---- the host readers' part (all of it for files that are not BGZF): one reader thread per file tokenises the next chunk into flat
  //      buffers while the device aligns the current one (the reference, too, decodes the two files on two IO threads: BwtMapper.cpp:1873-1935)
  if (r1 && se) {
    EndChunk bufs1[2];
    fq_packed_batch_t *pk1 = nullptr;
    if (fq_packed_create((int32_t)((A.chunk_pairs + 1) / 2), stride, &pk1)) die("out of pinned host memory for the packed batch");
    fill_chunk(*r1, bufs1[0], A.chunk_pairs, stride, name_stride);
    for (int slot = 0;; slot ^= 1) {
      EndChunk &e0 = bufs1[slot];
      if (!e0.error.empty()) die(e0.error);
      const int n = e0.n;
      if (n == 0) break;
      start();
      const bool last = e0.eof;
      std::thread prefetch;
      if (!last) prefetch = std::thread(fill_chunk, std::ref(*r1), std::ref(bufs1[slot ^ 1]), A.chunk_pairs, stride, name_stride);
      fq_read_batch_t in = {n, stride, e0.seq.data(), e0.qual.data(), e0.len.data(), e0.names.data(), (int32_t)name_stride, nullptr};
      fq_result_batch_t res;
      // the packed boundary, as for pairs: filter keys of every read cross PCIe, full rows of the surviving reads only
      rc = fq_pack_single_reads_into(&in, A.pack_threads, pk1);
      if (rc) die("fq_pack_single_reads_into failed (" + std::to_string(rc) + ")");
      rc = fq_align_packed(ctx, pk1, &res);
      if (rc) die(std::string("fq_align_packed failed: ") + fq_ctx_last_error(ctx));
      consume(res, n);
      if (prefetch.joinable()) prefetch.join();
      if (last) break;
    }
    fq_packed_free(pk1);
  } else if (r1) {
    EndChunk bufs[2][2];   // [slot][end]
    // the two ends of a chunk back to back in one buffer, as fq_read_batch_t wants them ([end][pair][stride]): each file's reader
    // fills its half in place (end 1 behind the chunk_pairs rows of end 0; a short last chunk moves it down)
    RawBuf<uint8_t> pair_seq[2], pair_qual[2];
    for (int sl = 0; sl < 2; ++sl) {
      pair_seq[sl].resize((size_t)2 * A.chunk_pairs * stride); pair_qual[sl].resize((size_t)2 * A.chunk_pairs * stride);
      for (int e = 0; e < 2; ++e) { bufs[sl][e].ext_seq = pair_seq[sl].data() + (size_t)e * A.chunk_pairs * stride; bufs[sl][e].ext_qual = pair_qual[sl].data() + (size_t)e * A.chunk_pairs * stride; }
    }
    auto read_both = [&](int slot) {
      const auto tr0 = std::chrono::steady_clock::now();
      std::thread t0(fill_chunk, std::ref(*r1), std::ref(bufs[slot][0]), A.chunk_pairs, stride, name_stride);
      std::thread t1(fill_chunk, std::ref(*r2), std::ref(bufs[slot][1]), A.chunk_pairs, stride, name_stride);
      t0.join(); t1.join();
      read_all_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr0).count();
    };
    { const auto t0 = std::chrono::steady_clock::now(); read_both(0); read_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
    fq_packed_batch_t *pk = nullptr;   // packed-batch storage, reused from chunk to chunk (pinned once)
    for (int slot = 0;; slot ^= 1) {
      EndChunk &e0 = bufs[slot][0], &e1 = bufs[slot][1];
      if (!e0.error.empty()) die(e0.error);
      if (!e1.error.empty()) die(e1.error);
      const int n = std::min(e0.n, e1.n);
      if (n == 0) break;
      start();
      if (!pk && fq_packed_create((int32_t)A.chunk_pairs, stride, &pk)) die("out of pinned host memory for the packed batch");
      const bool last = e0.eof || e1.eof || e0.n != e1.n;
      std::thread prefetch;
      if (!last) prefetch = std::thread(read_both, slot ^ 1);          // next chunk while this one is on the device
      order_check(n, [&](int sb, int e) { return (const char *)&(e ? e1 : e0).names[(size_t)sb * A.o.batch_pairs * name_stride]; });
      if ((long long)n < A.chunk_pairs) {   // a short (last) chunk: end 1 moves down behind the n rows of end 0
        memmove(pair_seq[slot].data() + (size_t)n * stride, pair_seq[slot].data() + (size_t)A.chunk_pairs * stride, (size_t)n * stride);
        memmove(pair_qual[slot].data() + (size_t)n * stride, pair_qual[slot].data() + (size_t)A.chunk_pairs * stride, (size_t)n * stride);
      }
      std::vector<int32_t> len((size_t)2 * n);
      for (int e = 0; e < 2; ++e) memcpy(&len[(size_t)e * n], bufs[slot][e].len.data(), (size_t)n * 4);
      fq_read_batch_t in = {n, stride, pair_seq[slot].data(), pair_qual[slot].data(), len.data(), e0.names.data(), (int32_t)name_stride, e1.names.data()};
      fq_result_batch_t res;
      // the packed boundary: 24 bytes of filter keys per read cross PCIe, full rows only for the surviving pairs
      const auto tp0 = std::chrono::steady_clock::now();
      rc = fq_pack_reads_into(&in, A.pack_threads, pk);
      pack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
      if (rc) die("fq_pack_reads failed (" + std::to_string(rc) + ")");
      const auto ta0 = std::chrono::steady_clock::now();
      rc = fq_align_packed(ctx, pk, &res);
      if (rc) die(std::string("fq_align_packed failed: ") + fq_ctx_last_error(ctx));
      align_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ta0).count();
      consume(res, 2LL * n);
      const auto tw0 = std::chrono::steady_clock::now();
      if (prefetch.joinable()) prefetch.join();
      read_wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
      if (last) break;
    }
    if (pk) fq_packed_free(pk);
  }
  start();                              // (an input without a record: the consumers and the index are needed all the same)
  out.flush();
  unequal_lengths_notice(A, r1 ? r1->h : nullptr, r2 ? r2->h : nullptr, unequal_on_device);
  if (se) {
    notice("%lld sequences are filtered.", filtered);
    notice("%lld sequences are unmapped.", unmapped);
  } else {
    notice("%lld sequences are loaded.", num_read);
    notice("%lld sequences are filtered.", filtered * 2);
    notice("%lld sequences are unmapped.", unmapped * 2);
  }
  fq_stats_t st;
  fq_stats_get(ctx_a, &st);
  if (ctx2) {
    fq_stats_t s2;
    fq_stats_get(ctx2, &s2);
    for (int k = 0; k < 6; ++k) st.kernel_ms[k] += s2.kernel_ms[k];
    st.kernel_ms[FQ_K_EMIT] += s2.kernel_ms[FQ_K_EMIT]; st.kernel_ms[FQ_K_REC_KERNEL] += s2.kernel_ms[FQ_K_REC_KERNEL]; st.kernel_ms[FQ_K_MD_KERNEL] += s2.kernel_ms[FQ_K_MD_KERNEL];
    st.device_wait_ms += s2.device_wait_ms; st.host_cpu_ms += s2.host_cpu_ms;
    st.host_ms_total += s2.host_ms_total; st.wall_ms_total += s2.wall_ms_total;
  }
  fprintf(stderr, "NOTICE - device time (ms): prep %.1f width %.1f gap %.1f sa %.1f sw %.1f refine %.1f records %.1f md %.1f consumers' kernels %.1f ; host %.1f ; waited for the device %.1f ; calls' CPU %.1f ; wall %.1f\n", st.kernel_ms[0],
          st.kernel_ms[1], st.kernel_ms[2], st.kernel_ms[3], st.kernel_ms[4], st.kernel_ms[5], st.kernel_ms[FQ_K_REC_KERNEL], st.kernel_ms[FQ_K_MD_KERNEL], st.kernel_ms[FQ_K_EMIT], st.host_ms_total, st.device_wait_ms, st.host_cpu_ms, st.wall_ms_total);
  fprintf(stderr, "NOTICE - the calls of one context moved over PCIe (bytes per pair): to the device %.1f ; to the host %.1f (lists, counts, hit lists; with the consumers on the device their outputs leave it on streams of their own and are not in this figure)\n",
          st.pairs ? (double)st.h2d_bytes / (double)st.pairs : 0.0, st.pairs ? (double)st.d2h_bytes / (double)st.pairs : 0.0);
  fprintf(stderr, "NOTICE - consumers (ms): StatCollector %.1f ; %s writer %.1f ; first chunk read %.1f ; packing %.1f\n", qc_ms, A.sam_out ? "SAM" : "BAM", out_ms, read_ms, pack_ms);
  fprintf(stderr, "NOTICE - reading (ms): all chunks %.1f ; waited for %.1f ; alignment calls %.1f\n", read_all_ms, read_wait_ms, align_ms);
  if (qc) fq_qc_end_file(qc);
  mark("input done");
  // (the contexts' and the front end's gigabytes of device memory are given back beside whatever the caller does next -- the next input's first
  //  chunk, the QC files: 80 ms of a 67 M-pair run)
  release_later([ctx_a, ctx2, fe] {
    fq_ctx_destroy(ctx_a);
    mark("first context released");
    if (ctx2) fq_ctx_destroy(ctx2);
    if (fe) fq_frontend_close(fe);
    mark("contexts and front end released");
  });
}

// ---- ONE FASTQ pair over several devices: chunks of whole reference batches are dealt round-robin; every device runs filter,
//      search and SA walks of its chunk at once; the stream's order-dependent state -- the drand48 stream (srand48 once per FASTQ pair,
//      src/BwtMapper.cpp:1817), last_ii (:780-781), the (k,l) cache (:815-843) -- goes from the context that owns chunk b to the one that
//      owns b + 1 around the short serial part of each call (fq_ctx_set_serial_hooks / fq_ctx_state_export / _import).  The records come
//      out in chunk order through one writer; every chunk's StatCollector state is a segment, merged in order.
struct ShardRun {
  std::mutex mu;
  std::condition_variable cv;
  struct Chunk { RawBuf<uint8_t> seq, qual; EndChunk e[2]; std::vector<int32_t> len; int n = 0; long long index = -1; };   // index: -1 free, -2 being read
  std::vector<Chunk> slots;
  long long n_chunks = -1;                 // known once the reader has seen the end
  long long token_of = -1;                 // the stream's state after chunk token_of
  std::vector<char> token;
  struct Result { std::string sam; std::vector<char> bam, qc; long long pairs = 0, filtered = 0, unmapped = 0; };
  std::map<long long, Result> results;
  // a refusal (a reader's error, a call that fails on chunk b): the writer still emits every chunk before it, then the run dies with `error`
  long long fail_at = -1;
  std::string error;
  void refuse(long long b, const std::string &m) {
    { std::lock_guard<std::mutex> lk(mu); if (fail_at < 0 || b < fail_at) { fail_at = b; error = m; } }
    cv.notify_all();
  }
};
struct ShardHook { ShardRun *run; fq_ctx_t *ctx; long long b; };
void shard_before(void *u) {
  ShardHook *h = (ShardHook *)u;
  if (h->b == 0) return;
  std::unique_lock<std::mutex> lk(h->run->mu);
  h->run->cv.wait(lk, [&] { return h->run->token_of == h->b - 1; });
  if (fq_ctx_state_import(h->ctx, h->run->token.data(), (int64_t)h->run->token.size())) fq_ctx_mark_stream_broken(h->ctx);
}
void shard_after(void *u) {
  ShardHook *h = (ShardHook *)u;
  const int64_t need = fq_ctx_state_export(h->ctx, nullptr, 0);
  std::vector<char> t((size_t)std::max<int64_t>(need, 0));
  if (need > 0) fq_ctx_state_export(h->ctx, t.data(), need);
  { std::lock_guard<std::mutex> lk(h->run->mu); h->run->token.swap(t); h->run->token_of = h->b; }
  h->run->cv.notify_all();
}
template <class W>
void align_pair_sharded(const Args &A, const std::pair<std::string, std::string> &input, std::vector<W> &wk, FILE *sam_fp, fq_bam_t *bam, fq_qc_t *qc) {
  const size_t NW = wk.size();
  fprintf(stderr, "NOTICE - Processing Pair End mapping on %zu devices\t%s\t%s\n", NW, input.first.c_str(), input.second.c_str());
  const int slot_mode = A.clean_names ? FQ_FASTQ_SLOTS_CLEAN_NAMES : FQ_FASTQ_SLOTS_REUSED;
  FastqReader r1(input.first, std::max(1, A.pack_threads / 2), A.o.batch_pairs, slot_mode, A.frac), r2(input.second, std::max(1, A.pack_threads / 2), A.o.batch_pairs, slot_mode, A.frac);
  int stride = 0;
  {
    size_t l = 0;
    struct stat s1, s2;
    if (stat(input.first.c_str(), &s1) == 0 && S_ISREG(s1.st_mode) && stat(input.second.c_str(), &s2) == 0 && S_ISREG(s2.st_mode))
      l = std::max(first_read_len(input.first), first_read_len(input.second));
    stride = (int)((std::max<size_t>(l, (size_t)std::max(A.read_len, 16)) + 15) & ~(size_t)15);
  }
  const int name_stride = 304;
  ShardRun R;
  R.slots.resize(NW + 1);
  for (auto &c : R.slots) {
    c.seq.resize((size_t)2 * A.chunk_pairs * stride); c.qual.resize((size_t)2 * A.chunk_pairs * stride);
    for (int e = 0; e < 2; ++e) { c.e[e].ext_seq = c.seq.data() + (size_t)e * A.chunk_pairs * stride; c.e[e].ext_qual = c.qual.data() + (size_t)e * A.chunk_pairs * stride; }
  }
  if (qc) fq_qc_begin_file(qc, input.first.c_str(), input.second.c_str());
  // the reader: chunks in file order into free slots
  std::thread reader([&] {
    long long order_checked_reads = 0;
    for (long long b = 0;; ++b) {
      ShardRun::Chunk *c = nullptr;
      {
        std::unique_lock<std::mutex> lk(R.mu);
        R.cv.wait(lk, [&] { for (auto &s : R.slots) if (s.index == -1) { c = &s; return true; } return false; });
        c->index = -2;
      }
      std::thread t0(fill_chunk, std::ref(r1), std::ref(c->e[0]), A.chunk_pairs, stride, name_stride);
      fill_chunk(r2, c->e[1], A.chunk_pairs, stride, name_stride);
      t0.join();
      if (!c->e[0].error.empty()) { R.refuse(b, c->e[0].error); return; }
      if (!c->e[1].error.empty()) { R.refuse(b, c->e[1].error); return; }
      const int n = std::min(c->e[0].n, c->e[1].n);
      const bool last = n == 0 || c->e[0].eof || c->e[1].eof || c->e[0].n != c->e[1].n;
      if (n) {
        for (int i = 0; i < n; i += A.o.batch_pairs) {      // the name check of src/BwtMapper.cpp:2087-2092, at the reference's cadence
          order_checked_reads += 2LL * std::min<long long>(A.o.batch_pairs, n - i);
          if (order_checked_reads % A.o.batch_pairs == 0 &&
              strncmp(&c->e[0].names[(size_t)i * name_stride], &c->e[1].names[(size_t)i * name_stride], (size_t)A.read_len) != 0) {
            R.refuse(b, "Abort, please make sure input pair of fastq files are in the same order!");
            return;
          }
        }
        if ((long long)n < A.chunk_pairs) {
          memmove(c->seq.data() + (size_t)n * stride, c->seq.data() + (size_t)A.chunk_pairs * stride, (size_t)n * stride);
          memmove(c->qual.data() + (size_t)n * stride, c->qual.data() + (size_t)A.chunk_pairs * stride, (size_t)n * stride);
        }
        c->len.resize((size_t)2 * n);
        for (int e = 0; e < 2; ++e) memcpy(&c->len[(size_t)e * n], c->e[e].len.data(), (size_t)n * 4);
      }
      {
        std::lock_guard<std::mutex> lk(R.mu);
        c->n = n;
        c->index = n ? b : -1;
        if (last) R.n_chunks = n ? b + 1 : b;
      }
      R.cv.notify_all();
      if (last) break;
    }
  });
  // the workers: chunk b goes to device b mod NW
  auto work = [&](size_t w) {
    W &K = wk[w];
    fq_ctx_t *ctx = nullptr;
    if (fq_ctx_create(K.ix, &A.o, (int32_t)A.chunk_pairs, &ctx)) die("fq_ctx_create failed: option outside the supported range");
    fq_packed_batch_t *pk = nullptr;
    if (fq_packed_create((int32_t)A.chunk_pairs, stride, &pk)) die("out of pinned host memory for the packed batch");
    if (K.qc) { fq_qc_begin_file(K.qc, input.first.c_str(), input.second.c_str()); if (fq_qc_state_reset(K.qc)) die("QC consumer: cannot start a segment"); }
    if (!A.host_consumers) {
      if (fq_ctx_set_emit(ctx, (A.sam_out ? FQ_EMIT_SAM : 0) | FQ_EMIT_DEVICE_ONLY)) die("fq_ctx_set_emit failed");
      if (!A.sam_out && K.bam && fq_ctx_attach_bam(ctx, K.bam)) die("fq_ctx_attach_bam failed");
      if (K.qc && fq_ctx_attach_qc(ctx, K.qc)) die("fq_ctx_attach_qc failed");
    }
    std::vector<char> sam;
    for (long long b = (long long)w;; b += (long long)NW) {
      ShardRun::Chunk *c = nullptr;
      {
        std::unique_lock<std::mutex> lk(R.mu);
        R.cv.wait(lk, [&] { for (auto &s : R.slots) if (s.index == b) { c = &s; return true; } return (R.n_chunks >= 0 && b >= R.n_chunks) || R.fail_at >= 0; });
        if (!c) break;
      }
      const int n = c->n;
      fq_read_batch_t in = {n, stride, c->seq.data(), c->qual.data(), c->len.data(), c->e[0].names.data(), (int32_t)name_stride, c->e[1].names.data()};
      if (fq_pack_reads_into(&in, std::max(1, A.pack_threads / (int)NW), pk)) die("fq_pack_reads failed");
      ShardHook hook{&R, ctx, b};
      fq_ctx_set_serial_hooks(ctx, shard_before, shard_after, &hook);
      fq_result_batch_t res;
      if (fq_align_packed(ctx, pk, &res)) { R.refuse(b, std::string("fq_align_packed failed on device ") + std::to_string(K.device) + ": " + fq_ctx_last_error(ctx)); return; }
      ShardRun::Result out;
      out.pairs = n; out.filtered = res.n_both_filtered; out.unmapped = res.n_both_unmapped;
      if (K.qc) {
        if (fq_qc_add_last(K.qc, ctx)) die(std::string("QC consumer failed: ") + fq_qc_last_error(K.qc));
        const int64_t need = fq_qc_state_export(K.qc, nullptr, 0);
        out.qc.resize((size_t)std::max<int64_t>(need, 0));
        if (need < 0 || fq_qc_state_export(K.qc, out.qc.data(), need) != need || fq_qc_state_reset(K.qc)) die("QC consumer: export failed");
      }
      if (A.sam_out && !A.host_consumers) {
        if (fq_sam_device_last(ctx, [](void *user, const void *data, int64_t n) -> int { ((std::string *)user)->append((const char *)data, (size_t)n); return 0; }, &out.sam) < 0)
          die(std::string("fetching the SAM text failed: ") + fq_ctx_last_error(ctx));
      } else if (A.sam_out) {
        const int64_t sz = fq_sam_format_last(ctx, nullptr, 0);
        sam.resize((size_t)sz + 1);
        fq_sam_format_last(ctx, sam.data(), sz + 1);
        out.sam.assign(sam.data(), (size_t)sz);
      } else {
        const void *data = nullptr; int64_t len = 0;
        if (fq_bam_format_last(K.bam, ctx, &data, &len)) die("formatting BAM records failed");
        out.bam.assign((const char *)data, (const char *)data + len);
      }
      {
        std::lock_guard<std::mutex> lk(R.mu);
        c->index = -1;                       // the chunk's rows are free again (the consumers above were the last to read them)
        R.results.emplace(b, std::move(out));
      }
      R.cv.notify_all();
    }
    fq_ctx_destroy(ctx);
    fq_packed_free(pk);
  };
  std::vector<std::thread> th;
  for (size_t w = 0; w < NW; ++w) th.emplace_back(work, w);
  // the writer: results in chunk order
  long long num_read = 0, filtered = 0, unmapped = 0;
  for (long long b = 0;; ++b) {
    ShardRun::Result res;
    {
      std::unique_lock<std::mutex> lk(R.mu);
      R.cv.wait(lk, [&] { return R.results.count(b) || (R.n_chunks >= 0 && b >= R.n_chunks) || (R.fail_at >= 0 && b >= R.fail_at); });
      auto it = R.results.find(b);
      if (it == R.results.end() || (R.fail_at >= 0 && b >= R.fail_at)) break;
      res = std::move(it->second);
      R.results.erase(it);
    }
    if (A.sam_out) { if (fwrite(res.sam.data(), 1, res.sam.size(), sam_fp) != res.sam.size()) die("writing the SAM text failed"); }
    else if (fq_bam_write_records(bam, res.bam.data(), (int64_t)res.bam.size())) die("writing " + A.out_prefix + ".bam failed");
    if (qc && fq_qc_merge(qc, res.qc.data(), (int64_t)res.qc.size())) die(std::string("QC consumer: merge failed: ") + fq_qc_last_error(qc));
    num_read += 2 * res.pairs; filtered += res.filtered; unmapped += res.unmapped;
    fprintf(stderr, "NOTICE - %lld sequences are processed.\n", num_read);
  }
  {   // a refused run ends here, behind the records of the chunks before the refusal (threads that wait for a state that never comes end with the process)
    std::string err;
    { std::lock_guard<std::mutex> lk(R.mu); if (R.fail_at >= 0) err = R.error; }
    if (!err.empty()) { if (sam_fp) fflush(sam_fp); die(err); }
  }
  reader.join();
  for (auto &t : th) t.join();
  if (sam_fp) fflush(sam_fp);
  unequal_lengths_notice(A, r1.h, r2.h);
  notice("%lld sequences are loaded.", num_read);
  notice("%lld sequences are filtered.", filtered * 2);
  notice("%lld sequences are unmapped.", unmapped * 2);
  if (qc) fq_qc_end_file(qc);
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) return usage();
  const std::string cmd = argv[1];
  if (cmd == "index") {
    std::string ref;
    int rollhash = 0;
    for (int i = 2; i < argc; ++i) {
      if (!strcmp(argv[i], "--ref") && i + 1 < argc) ref = argv[++i];
      else if (!strcmp(argv[i], "--rollhash")) rollhash = 1;
      else return usage();
    }
    if (ref.empty()) return usage();
    const int rc = fq_index_build(ref.c_str(), rollhash);
    if (rc) die("fq_index_build failed (" + std::to_string(rc) + ")");
    return 0;
  }
  if (cmd != "align") return usage();
  Args A;
  fq_default_opts(&A.o);
  for (int i = 2; i < argc; ++i) {
    const std::string f = argv[i];
    auto need = [&](const char *) -> const char * { if (i + 1 >= argc) die("missing value for " + f); return argv[++i]; };
    if (f == "--fastq_1") A.fq1 = need("");
    else if (f == "--fastq_2") A.fq2 = need("");
    else if (f == "--out_prefix") A.out_prefix = need("");
    else if (f == "--index_prefix") A.index_prefix = need("");
    else if (f == "--sam_out") A.sam_out = true;
    else if (f == "--kmer_thresh") A.o.filter_thresh = atoi(need(""));
    else if (f == "--n") A.o.fnr = atof(need(""));
    else if (f == "--o") A.o.max_gapo = atoi(need(""));
    else if (f == "--e") A.opte = atoi(need(""));
    else if (f == "--i") A.o.indel_end_skip = atoi(need(""));
    else if (f == "--d") A.o.max_del_occ = atoi(need(""));
    else if (f == "--l") A.o.seed_len = atoi(need(""));
    else if (f == "--k") A.o.max_seed_diff = atoi(need(""));
    else if (f == "--m") A.o.max_entries = atoi(need(""));
    else if (f == "--t") { A.o.host_threads = atoi(need("")); if (A.o.host_threads > 0) A.pack_threads = A.o.host_threads; }   // accepted for command-line compatibility (see fastquick_amd.h)
    else if (f == "--R") A.o.max_top2 = atoi(need(""));
    else if (f == "--q") A.o.trim_qual = atoi(need(""));
    else if (f == "--N") { A.o.mode |= 0x10; A.o.max_top2 = 0x7fffffff; }
    else if (f == "--L") A.o.mode |= 4;
    else if (f == "--I") A.o.mode |= 0x200;   // BWA_MODE_IL13
    else if (f == "--max_isize") A.o.max_isize = atoi(need(""));
    else if (f == "--max_occ") A.o.max_occ = (uint32_t)atoi(need(""));
    else if (f == "--is_sw") A.o.is_sw = !A.o.is_sw;          // a bool flag on a default-1 int: it toggles (src/FASTQuick.cpp:278)
    else if (f == "--n_multi") A.o.n_multi = atoi(need(""));
    else if (f == "--N_multi") A.o.N_multi = atoi(need(""));
    else if (f == "--ap_prior") A.o.ap_prior = atof(need(""));
    else if (f == "--force_isize") A.o.force_isize = 1;
    else if (f == "--chunk_pairs") A.chunk_pairs = atoll(need(""));
    else if (f == "--read_len") A.read_len = atoi(need(""));
    else if (f == "--clean_names") A.clean_names = true;
    else if (f == "--strict_reference") A.strict = true;
    else if (f == "--host_reader") A.host_reader = true;
    else if (f == "--host_consumers") A.host_consumers = true;
    else if (f == "--batch_pairs") A.o.batch_pairs = atoi(need(""));   // READ_BUFFER_SIZE of the run to reproduce (default 262144)
    else if (f == "--device") A.device = atoi(need(""));
    else if (f == "--devices") A.devices = need("");
    else if (f == "--fq_list") A.fq_list = need("");
    else if (f == "--RG") A.rg = need("");
    else if (f == "--cal_dup") A.cal_dup = !A.cal_dup;        // (a bool flag on a default-1 field, like --is_sw)
    else if (f == "--frac_samp") A.frac = atof(need(""));
    else if (f == "--bam_in") die(f + " is not supported (the reference's BAM input is disabled, too)");
    else die("unknown option " + f);
  }
  if (A.o.fnr >= 1.0) { A.o.max_diff = (int)A.o.fnr; A.o.fnr = -1.0; }                  // src/FASTQuick.cpp:312-315
  if (A.opte > 0) { A.o.max_gape = A.opte; A.o.mode &= ~1; }                             // :316-319
  // The reference re-allocates a read slot when a read longer than read_len arrives (src/BwtMapper.cpp:536-546).  That read then fills the
  // whole slot, so with read_len >= 96 -- the reference's own value is 151 -- nothing an earlier read left can reach the 96-base window the
  // read filter looks at, and the re-allocation is invisible; a smaller --read_len would make it visible, and is refused.
  if (A.read_len < 96) die("--read_len must be at least 96 (the reference's is 151): below that its slot re-allocation (src/BwtMapper.cpp:536-546) would show in the read filter, and it is not modelled");
  if (A.out_prefix == "Empty") die("--out_prefix is required");
  if (A.index_prefix == "Empty") die("--index_prefix is required");
  // --fq_list: one FASTQ pair per line, '#' lines skipped (src/BwtMapper.cpp:232-262); every pair is an independent stream
  // (its own srand48, last_ii, position cache and read slots: PairEndMapper sets them up per call), all into one output
  std::vector<std::pair<std::string, std::string>> inputs;
  if (!A.fq_list.empty()) {
    FILE *fl = fopen(A.fq_list.c_str(), "r");
    if (!fl) die("Open file " + A.fq_list + " failed");
    char line[8192];
    while (fgets(line, sizeof line, fl)) {
      if (line[0] == '#') continue;
      char a[4096] = "", b[4096] = "";
      const int got = sscanf(line, "%4095s %4095s", a, b);
      if (got < 1) continue;
      inputs.emplace_back(a, got < 2 ? "" : b);      // one column: single-end (src/BwtMapper.cpp:255-262)
    }
    fclose(fl);
  } else {
    if (A.fq1.empty()) die("--fastq_1 (and --fastq_2 for paired-end reads), or --fq_list, is required");
    inputs.emplace_back(A.fq1, A.fq2);
  }
  if (A.o.batch_pairs < 1) die("--batch_pairs must be positive");
  A.chunk_pairs = std::max<long long>(A.o.batch_pairs, A.chunk_pairs / A.o.batch_pairs * A.o.batch_pairs);   // whole reference batches per chunk

  mark("options read");
  fq_runtime_configure(20, 1);   // hardware queues for the contexts' streams, sleeping waits: before the first HIP call (fastquick_amd.h)
  const std::string pre = A.index_prefix + ".FASTQuick.fa";
  // <index>.param: REFERENCE_PATH, TARGET_REGION_PATH, DBSNP_VCF_PATH, NUM_VAR_LONG, NUM_VAR_SHORT, SHORT_FLANK_LENGTH, LONG_FLANK_LENGTH
  fq_qc_opts_t qo;
  fq_qc_default_opts(&qo);
  qo.read_len = A.read_len; qo.cal_dup = A.cal_dup ? 1 : 0;
  std::string ref_path;
  {
    FILE *fp = fopen((pre + ".param").c_str(), "r");
    char key[256], val[4096];
    while (fp && fscanf(fp, "%255s %4095s", key, val) == 2) {
      if (!strcmp(key, "REFERENCE_PATH")) ref_path = val;
      else if (!strcmp(key, "SHORT_FLANK_LENGTH")) qo.flank_len = atoi(val);
      else if (!strcmp(key, "LONG_FLANK_LENGTH")) qo.flank_long_len = atoi(val);
    }
    if (fp) fclose(fp);
    if (!ref_path.empty()) {   // BwtIndexer::LoadContigSize (src/BwtIndexer.cpp:764-802): sums of column 2 of the .fai and of EVERY line of the .amb
      char line[8192], a[4096], b[4096];
      if (FILE *ff = fopen((ref_path + ".fai").c_str(), "r")) { while (fgets(line, sizeof line, ff)) if (sscanf(line, "%4095s %4095s", a, b) == 2) qo.genome_size += atoi(b); fclose(ff); }
      if (FILE *fa = fopen((ref_path + ".amb").c_str(), "r")) { while (fgets(line, sizeof line, fa)) if (sscanf(line, "%4095s %4095s", a, b) == 2) qo.genome_n_size += atoi(b); fclose(fa); }
    }
  }
  struct stat sb;
  const bool have_qc = stat((pre + ".SelectedSite.vcf").c_str(), &sb) == 0;
  if (!have_qc) fprintf(stderr, "NOTICE - %s.SelectedSite.vcf not found: the QC files are not written\n", pre.c_str());
  if (!A.sam_out && ref_path.empty()) die("BAM output needs the original reference's .fai: " + pre + ".param (REFERENCE_PATH) is missing; or pass --sam_out");
  const std::string fai = ref_path + ".fai";
  // one worker per entry of --devices: its own copy of the index on its device, its own consumers
  struct Worker { int device = 0; fq_index_t *ix = nullptr; fq_qc_t *qc = nullptr; fq_bam_t *bam = nullptr; std::thread th; };
  std::vector<int> devices = A.devices.empty() ? std::vector<int>{A.device} : parse_devices(A.devices);
  const bool shard_one_pair = devices.size() > 1 && inputs.size() == 1 && !inputs[0].second.empty() && inputs[0].second != "Empty";   // one pair, several devices
  if (!shard_one_pair && devices.size() > inputs.size()) devices.resize(std::max<size_t>(1, inputs.size()));   // (a worker per FASTQ pair at most)
  {   // every ordinal is checked before anything is started (a worker that fails to load its index would take the run down half-way)
    const int n_dev = fq_device_count();
    if (n_dev <= 0) die("no HIP device is visible; there is no CPU fallback");
    for (int d : devices) if (d >= n_dev) die("device " + std::to_string(d) + " does not exist (" + std::to_string(n_dev) + " visible)");
    mark("devices counted");
  }
  const size_t W = devices.size();
  std::vector<Worker> wk(W);
  auto open_worker = [&](size_t w, const std::string &qc_prefix, const char *bam_path) {
    Worker &K = wk[w];
    K.device = devices[w];
    const auto t_ix0 = std::chrono::steady_clock::now();
    int rc = fq_index_load(pre.c_str(), K.device, &K.ix);
    fprintf(stderr, "NOTICE - index staged on device %d in %.1f ms\n", K.device, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_ix0).count());
    if (rc) die("cannot load index " + pre + " onto HIP device " + std::to_string(K.device) + " (" + std::to_string(rc) + "); there is no CPU fallback");
    if (have_qc) {
      if (qc_prefix != A.out_prefix) tmp_file(qc_prefix + ".InsertSizeTable");
      rc = fq_qc_create(K.ix, pre.c_str(), qc_prefix.c_str(), &qo, &K.qc);
      if (rc) die("cannot set up the QC consumer from " + pre + ".SelectedSite.vcf / .dbSNP.subset.vcf / .gc (" + std::to_string(rc) + ")");
      mark("QC consumer set up");
    }
    if (!A.sam_out) {
      rc = fq_bam_create(K.ix, fai.c_str(), bam_path, A.rg.c_str(), &qo, &K.bam);
      if (rc) die("cannot open " + A.out_prefix + ".bam / " + fai + " (" + std::to_string(rc) + ")");
      mark("BAM writer set up");
    }
  };
  if (W == 1) {
    // ---- one device: the records go out as they are produced ----
    // The index is staged, and the consumers are set up, while the first input's first chunk is read and tokenised.
    Worker &K = wk[0];
    const std::string bam_path = A.out_prefix + ".bam";
    std::thread opener([&] { open_worker(0, A.out_prefix, A.sam_out ? nullptr : bam_path.c_str()); });
    Sink out;
    out.sam_out = A.sam_out; out.sam_fp = A.sam_out ? stdout : nullptr; out.what = A.sam_out ? "the SAM text" : bam_path;
    bool opened = false;
    const std::function<void()> ready = [&] {
      if (opened) return;
      opened = true;
      opener.join();
      out.bam = K.bam;
      if (A.sam_out) {
        const int64_t n = fq_sam_header(K.ix, nullptr, 0);
        std::vector<char> h((size_t)n + 1);
        fq_sam_header(K.ix, h.data(), n + 1);
        fwrite(h.data(), 1, (size_t)n, stdout);
      }
    };
    for (const auto &input : inputs) align_input(A, input, K.ix, K.qc, out, ready, devices[0]);   // (not K.device: the opener thread is still writing K)
    ready();
    // (the BAM file's last blocks and its close beside the QC files' writing: a third of a second of a deep run's tail)
    bool bam_bad = false;
    std::thread closer;
    if (K.bam) closer = std::thread([&] { bam_bad = fq_bam_close(K.bam) != 0; mark("BAM file closed"); });
    const bool qc_bad = K.qc && fq_qc_write(K.qc) != 0;
    if (closer.joinable()) closer.join();
    if (bam_bad) die("closing " + A.out_prefix + ".bam failed");
    if (qc_bad) die("writing the QC files failed");
    mark("QC files written");
    if (K.qc) { fq_qc_t *qc = K.qc; release_later([qc] { fq_qc_destroy(qc); mark("QC consumer released"); }); }      // (beside the contexts' release)
    release_join();
    fq_index_destroy(K.ix);
    mark("index released");
    return 0;
  }
  if (shard_one_pair) {
    // ---- one FASTQ pair over several devices: chunks dealt round-robin, the stream's state handed from context to context (align_pair_sharded)
    std::vector<std::thread> opn;
    for (size_t w = 0; w < W; ++w) opn.emplace_back([&, w] { open_worker(w, A.out_prefix + ".worker" + std::to_string(w), nullptr); });
    for (auto &t : opn) t.join();
    fq_bam_t *bam = nullptr;
    fq_qc_t *qc = nullptr;
    if (A.sam_out) {
      const int64_t n = fq_sam_header(wk[0].ix, nullptr, 0);
      std::vector<char> h((size_t)n + 1);
      fq_sam_header(wk[0].ix, h.data(), n + 1);
      fwrite(h.data(), 1, (size_t)n, stdout);
    } else if (fq_bam_create(wk[0].ix, fai.c_str(), (A.out_prefix + ".bam").c_str(), A.rg.c_str(), &qo, &bam)) die("cannot open " + A.out_prefix + ".bam / " + fai);
    if (have_qc && fq_qc_create(wk[0].ix, pre.c_str(), A.out_prefix.c_str(), &qo, &qc)) die("cannot set up the QC consumer");
    Args AW = A;
    if (AW.o.host_threads <= 0) AW.o.host_threads = std::max(2, std::min(16, 2 * fq_host_cpus() / (int)W));
    align_pair_sharded(AW, inputs[0], wk, A.sam_out ? stdout : nullptr, bam, qc);
    if (bam && fq_bam_close(bam)) die("closing " + A.out_prefix + ".bam failed");
    if (qc) {
      if (fq_qc_write(qc)) die("writing the QC files failed");
      fq_qc_destroy(qc);
    }
    release_join();
    for (size_t w = 0; w < W; ++w) {
      if (wk[w].bam) fq_bam_close(wk[w].bam);
      if (wk[w].qc) { fq_qc_destroy(wk[w].qc); remove((A.out_prefix + ".worker" + std::to_string(w) + ".InsertSizeTable").c_str()); }
      fq_index_destroy(wk[w].ix);
    }
    return 0;
  }
  // ---- several devices: the lines of --fq_list are dealt over them (the next free worker takes the next pair); every pair's records go
  //      to part files of its own and its StatCollector state to a segment (fq_qc_state_export); the main thread puts the parts behind
  //      each other and merges the segments (fq_qc_merge) in input order -- the files of the one-device run (src/BwtMapper.cpp:232-262,
  //      src/StatCollector.h:46-62: one StatCollector over all pairs of the list)
  {
    const size_t n_in = inputs.size();
    std::vector<std::vector<char>> segment(n_in);
    std::atomic<size_t> next{0};
    Args AW = A;
    AW.pack_threads = std::max(1, A.pack_threads / (int)W);                 // the host's CPUs are shared by the workers
    if (AW.o.host_threads <= 0) AW.o.host_threads = std::max(2, std::min(16, 2 * fq_host_cpus() / (int)W));
    auto part = [&](size_t i, const char *ext) { return A.out_prefix + ".part" + std::to_string(i) + ext; };
    auto work = [&](size_t w) {
      open_worker(w, A.out_prefix + ".worker" + std::to_string(w), nullptr);
      Worker &K = wk[w];
      if (K.qc && fq_qc_state_reset(K.qc)) die("QC consumer: cannot start a segment");
      for (size_t i; (i = next.fetch_add(1)) < n_in;) {
        Sink out;
        out.sam_out = A.sam_out; out.bam = K.bam;
        out.what = part(i, A.sam_out ? ".sam" : ".bamrec");
        tmp_file(out.what);
        FILE *f = fopen(out.what.c_str(), "wb");
        if (!f) die("cannot create " + out.what);
        if (A.sam_out) out.sam_fp = f; else out.bam_fp = f;
        fprintf(stderr, "NOTICE - device %d takes line %zu of the list\n", K.device, i + 1);
        align_input(AW, inputs[i], K.ix, K.qc, out, [] {}, K.device);
        if (fclose(f)) die("writing " + out.what + " failed");
        if (K.qc) {
          const int64_t need = fq_qc_state_export(K.qc, nullptr, 0);
          if (need < 0) die(std::string("QC consumer: export failed: ") + fq_qc_last_error(K.qc));
          segment[i].resize((size_t)need);
          if (fq_qc_state_export(K.qc, segment[i].data(), need) != need || fq_qc_state_reset(K.qc)) die("QC consumer: export failed");
        }
      }
    };
    for (size_t w = 0; w < W; ++w) wk[w].th = std::thread(work, w);
    for (auto &K : wk) K.th.join();
    // the output, in input order
    fq_bam_t *bam = nullptr;
    fq_qc_t *qc = nullptr;
    if (A.sam_out) {
      const int64_t n = fq_sam_header(wk[0].ix, nullptr, 0);
      std::vector<char> h((size_t)n + 1);
      fq_sam_header(wk[0].ix, h.data(), n + 1);
      fwrite(h.data(), 1, (size_t)n, stdout);
    } else if (fq_bam_create(wk[0].ix, fai.c_str(), (A.out_prefix + ".bam").c_str(), A.rg.c_str(), &qo, &bam)) die("cannot open " + A.out_prefix + ".bam / " + fai);
    if (have_qc && fq_qc_create(wk[0].ix, pre.c_str(), A.out_prefix.c_str(), &qo, &qc)) die("cannot set up the QC consumer");
    for (size_t i = 0; i < n_in; ++i) {
      append_file(part(i, A.sam_out ? ".sam" : ".bamrec"), A.sam_out ? stdout : nullptr, bam);
      if (qc && fq_qc_merge(qc, segment[i].data(), (int64_t)segment[i].size())) die(std::string("QC consumer: merge failed: ") + fq_qc_last_error(qc));
    }
    fflush(stdout);
    if (bam && fq_bam_close(bam)) die("closing " + A.out_prefix + ".bam failed");
    if (qc) {
      if (fq_qc_write(qc)) die("writing the QC files failed");
      fq_qc_destroy(qc);
    }
    release_join();
    for (size_t w = 0; w < W; ++w) {
      if (wk[w].bam) fq_bam_close(wk[w].bam);
      if (wk[w].qc) { fq_qc_destroy(wk[w].qc); remove((A.out_prefix + ".worker" + std::to_string(w) + ".InsertSizeTable").c_str()); }
      fq_index_destroy(wk[w].ix);
    }
  }
  return 0;
}
