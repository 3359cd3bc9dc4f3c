// fq_align.cpp -- host side of the per-batch hot path: stages the GPU kernels and carries the
// order-dependent state exactly as BwtMapper::PairEndMapper / PEworker do
// (src/BwtMapper.cpp:654-684, 1796-2104):
//   * the glibc drand48 stream consumed by bwa_aln2seq_core (libbwa/bwase.c:19-95), seeded
//     srand48(bns->seed) once per FASTQ pair (:1817)                                  [SURVEY Q2]
//   * per-batch insert-size inference with the last_ii fallback (bwape.c:49-117, :780) [Q3]
//   * the (k,l) -> positions cache for SA intervals >= 1000 wide (:815-843)            [Q6]
// Everything data-parallel (filter, widths, gap search, SA walks, SW, global DP, MD) runs on the
// GPU through fq_backend.h; libm-dependent scalar decisions stay on the host (Q4/Q5).
#include <algorithm>
#include <atomic>
#include <ctime>
#include <chrono>
#include <climits>
#include <cstdint>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <mutex>
#include <thread>
#include <vector>
#include <pthread.h>
#include <sched.h>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"
#include "fq_index.h"
#include "fq_pipeline.h"
#include "fq_pool.h"
#include "fq_text_batch.h"

using std::vector;

namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// bwa_cal_maxdiff, libbwa/bwtaln.c:58-70 (the int factorial wraps as in the reference)
int cal_maxdiff(int l, double err, double thres) {
  double elambda = exp(-l * err), sum = elambda, y = 1.0;
  int x = 1;
  for (int k = 1; k < 1000; ++k) {
    y *= l * err;
    x = (int)((unsigned)x * (unsigned)k);
    sum += elambda * y / x;
    if (1.0 - sum < thres) return k;
  }
  return 2;
}
// A stream's first calls are short ones (the front end hands out an eighth, a quarter, a half of a chunk first): a buffer that has to grow in such a call is
// sized for the calls that follow (tl_grow = pairs of a steady call / pairs of this one, set by the call; 1 outside), so that a context allocates its device and
// pinned buffers once instead of at every step of the ramp -- pinning a few hundred megabytes costs 0.4 ms per megabyte, and a hipFree waits for the device.
thread_local double tl_grow = 1.0;
struct GrowScope { double prev; explicit GrowScope(double g) : prev(tl_grow) { tl_grow = g; } ~GrowScope() { tl_grow = prev; } };
inline size_t grown(size_t n) { return tl_grow > 1.0 ? (size_t)((double)n * tl_grow * 1.02) : 0; }

// hipFree and hipHostFree wait for every stream of the device -- the other context's consumers streaming gigabytes of text to the host included: a buffer that
// had to grow by a hair in the middle of a call cost that call 100-250 ms now and then.  A buffer that grows during a call is therefore not freed there but retired
// to its context's list (tl_retired, set by the call), from which the growth of a LATER call may take it again (nothing of the call that retired it is in flight
// by then); the list is freed with the context, or at the start of a call once it holds more than kRetiredMax.
struct Retired {
  struct B { void *p; size_t bytes; uint64_t call; bool host; };
  std::vector<B> v;
  uint64_t call = 0;
  static constexpr size_t kRetiredMax = (size_t)6 << 30;
  size_t bytes(bool host) const { size_t t = 0; for (auto &b : v) if (b.host == host) t += b.bytes; return t; }
  void drain() { for (auto &b : v) { if (b.host) fqdev::hfree(b.p); else fqdev::dfree(b.p); } v.clear(); }
  void next_call() { ++call; if (bytes(false) > kRetiredMax || bytes(true) > kRetiredMax / 8) drain(); }
  ~Retired() { drain(); }
};
thread_local Retired *tl_retired = nullptr;
struct RetiredScope { Retired *prev; explicit RetiredScope(Retired *r) : prev(tl_retired) { tl_retired = r; } ~RetiredScope() { tl_retired = prev; } };
inline void retire(void *p, size_t bytes, bool host) {
  if (!p) return;
  if (tl_retired) tl_retired->v.push_back({p, bytes, tl_retired->call, host});
  else if (host) fqdev::hfree(p); else fqdev::dfree(p);
}
// a block of at least `bytes` (and not half as much again) that an earlier call retired, or a fresh one
inline void *obtain(size_t bytes, bool host, size_t *got) {
  if (tl_retired) {
    int best = -1;
    for (int i = 0; i < (int)tl_retired->v.size(); ++i) {
      const Retired::B &b = tl_retired->v[(size_t)i];
      if (b.host == host && b.call < tl_retired->call && b.bytes >= bytes && b.bytes <= bytes + bytes / 2 && (best < 0 || b.bytes < tl_retired->v[(size_t)best].bytes)) best = i;
    }
    if (best >= 0) { void *q = tl_retired->v[(size_t)best].p; *got = tl_retired->v[(size_t)best].bytes; tl_retired->v.erase(tl_retired->v.begin() + best); return q; }
  }
  void *q = host ? fqdev::hmalloc(bytes) : fqdev::dmalloc(bytes);
  if (!q && tl_retired && !tl_retired->v.empty()) { tl_retired->drain(); q = host ? fqdev::hmalloc(bytes) : fqdev::dmalloc(bytes); }      // (out of memory: what was kept goes first)
  *got = bytes;
  return q;
}

template <class T> struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  ~DevBuf() { fqdev::dfree(p); }
  bool take(size_t want) { size_t got = 0; p = (T *)obtain(want * sizeof(T), false, &got); cap = p ? got / sizeof(T) : 0; return p != nullptr; }
  // slack against regrowth: a quarter for small buffers, a sixteenth for the multi-GB ones (chunks of a stream differ by a per cent or two)
  static size_t slack(size_t n) { return std::max(std::min<size_t>(n / 4, (size_t)16 << 20), n / 16); }
  bool ensure(size_t n) {
    if (n <= cap) return true;
    retire(p, cap * sizeof(T), false);
    p = nullptr; cap = 0;
    if (const size_t g = grown(n)) { if (take(g + 64)) return true; }
    if (take(n + slack(n) + 64)) return true;
    return take(n);
  }
  // the consumers' multi-GB buffers (SAM text, BAM records and members): an eighth of slack
  bool ensure_roomy(size_t n) {
    if (n <= cap) return true;
    retire(p, cap * sizeof(T), false);
    p = nullptr; cap = 0;
    if (take(std::max(n + n / 8, grown(n)) + 64)) return true;
    if (take(n + n / 8 + 64)) return true;
    return take(n);
  }
  bool ensure_keep(size_t n, size_t keep) {   // as ensure, but the first `keep` elements survive a reallocation (copied on the compute stream)
    if (n <= cap) return true;
    DevBuf<T> q;
    if (!q.take(std::max(n + slack(n), grown(n)) + 64) && !q.take(n + 64)) return false;
    if (keep && (fqdev::d2d(q.p, p, keep * sizeof(T)) || fqdev::sync())) return false;      // (q's destructor frees the new block)
    retire(p, cap * sizeof(T), false);
    p = q.p; cap = q.cap;
    q.p = nullptr; q.cap = 0;
    return true;
  }
};
}  // namespace

template <class T> struct PinBuf {   // pinned host staging: device <-> host copies of per-call lists run at full PCIe rate and truly async
  T *p = nullptr;
  size_t cap = 0;
  ~PinBuf() { fqdev::hfree(p); }
  bool take(size_t want) { size_t got = 0; p = (T *)obtain(want * sizeof(T), true, &got); cap = p ? got / sizeof(T) : 0; return p != nullptr; }
  bool ensure(size_t n) {
    if (n <= cap) return true;
    retire(p, cap * sizeof(T), true);
    p = nullptr; cap = 0;
    if (take(std::max(n + n / 4, grown(n)) + 64)) return true;
    return take(n + 64);
  }
  bool ensure_keep(size_t n, size_t keep) {   // as ensure, but the first `keep` elements survive a reallocation
    if (n <= cap) return true;
    PinBuf<T> q;
    if (!q.take(std::max(n + n / 4, grown(n)) + 64) && !q.take(n + 64)) return false;
    if (keep) memcpy(q.p, p, keep * sizeof(T));
    retire(p, cap * sizeof(T), true);
    p = q.p; cap = q.cap;
    q.p = nullptr; q.cap = 0;
    return true;
  }
};

// Every host <-> device copy of a call goes through pinned memory: a copy from or to pageable memory takes the runtime's
// staged, blocking path, and mixing the two kinds on one stream made the pageable ones several times slower under load (the
// refine stage of a call went from 4 to 14 ms with sixteen contexts).  The arena is a bump allocator over pinned blocks,
// reset when a call starts; results of device -> host copies are handed to their (ordinary) destinations at the next sync.
struct PinArena {
  struct Block { uint8_t *p; size_t cap; };
  std::vector<Block> blocks;
  size_t used = 0, total = 0;
  double last_grow = 1.0;          // tl_grow of the call `total` was counted in
  struct Out { void *dst; const void *src; size_t bytes; };
  std::vector<Out> pending;
  ~PinArena() { for (auto &b : blocks) fqdev::hfree(b.p); }
  void reset() {
    pending.clear();
    if (blocks.size() > 1) {   // one block of the size the last call needed
      for (auto &b : blocks) retire(b.p, b.cap, true);
      blocks.clear();
      size_t want = std::max(total + total / 4, (size_t)((double)total * last_grow * 1.02));
      uint8_t *p = (uint8_t *)obtain(want, true, &want);
      if (p) blocks.push_back({p, want});
    }
    used = 0; total = 0; last_grow = tl_grow;
  }
  void *alloc(size_t bytes) {
    bytes = (bytes + 63) & ~(size_t)63;
    total += bytes;
    if (blocks.empty() || used + bytes > blocks.back().cap) {
      size_t cap = std::max<size_t>(std::max(bytes, grown(bytes)), (size_t)4 << 20);
      uint8_t *p = (uint8_t *)obtain(cap, true, &cap);
      if (!p) return nullptr;
      blocks.push_back({p, cap});
      used = 0;
    }
    void *r = blocks.back().p + used;
    used += bytes;
    return r;
  }
};

struct FqKnobs {   // experiment / test knobs (fq_ctx_set_tuning); defaults are what DESIGN.md measures
  uint32_t gap_long_pops = 256;    // lane kernel hands a search to the wavefront-per-read kernel after this many pops (queue dry); 1,024 -> 256: single-stream WGS call 19.4 -> 17.3 ms, search stage 8.0 -> 5.7 ms
  int gap_long_always = 0;         // ... whatever the state of the queue (tests)
  int64_t gap_skip_bound = -1;     // first round: reads whose lower bound on the differences (both strands) is at least this go straight to the full search; -1: the smallest bound that rules out a hit the round could settle, 0: none
  uint32_t gap_long_pops2 = 0;     // hand-over threshold of the second round of a device-filling call (0: no hand-over)
  int64_t gap_split_hard = 0;      // experiment: after the round without gap children, search reads whose lower bound is >= this in a launch of their own (0: off)
  int64_t gap_pipeline_min = -1;        // experiment (off): calls that search at least this many reads run the first round in segments, each segment's second round beside the next segment's first
  int gap_pipeline_segs = 8;
  int64_t gap_nogap_min = 131072;  // launches of at least this many reads begin with the round that searches without gap children (-1: never)
  uint32_t gap_pool = 2048;        // stack entries per lane of the lane kernel
  int gap_no_order = 0;
  int device_turns = 2;            // calls that search at least gap_nogap_min reads take turns on the device: 1 = the search kernels, 2 = the width kernel too, 0 = off.
                                   // Two streams of 4.2 M-pair on-target calls: search kernels 54.1 / 41.7 / 38.8 ms per launch with 0 / 1 / 2 (38.8 alone), 16.8 / 16.3 / 16.7 M pairs/s
  int device_turn_slots = 1;       // how many contexts may hold a turn at once (experiment: a few small search stages side by side instead of sixteen or one)
  int64_t device_turn_min = 1 << 20; // ... calls that search at least this many reads (launches that fill the device several times over; sixteen streams of a 100k-marker
                                   // WGS mix search 176 k reads per call: one residency of the first round, and taking turns for it serialised the streams)
  int gap_round2_refill = 16;      // refill group of the round after it (see the launch)
  int gap_round1_refill = 0;       // experiment: refill group of the round without gap children (0: whole wavefronts)
  int gap_round2_lane_major = 1;   // ... with every lane's stack pool contiguous (FqGapTier::lane_major)
  int gap_round2_waves = 2560;     // wavefronts of the round after the one without gap children (0: as many as fit): its reads are long searches, a trip of a
                                   // wavefront costs more the more wavefronts share its SIMD, and the launch lasts as long as its longest search
  int sw_wave_max = 4096;          // largest mate-SW window the wavefront kernel takes
  int host_threads = -1;           // -1: fq_opts_t::host_threads
  size_t host_par_min = 32768;     // below this many items a per-pair host phase stays on the calling thread
  int64_t md_mask_min = 4096;      // calls with at least this many records of compact rows compare them with the reference piece-wise before MD (fq_md_mask_piece)
  int64_t packed_bulk_min = -1;    // packed batches: upload the whole body instead of gathered survivor rows from this many survivor pairs (-1: n_pairs / 8)
  int trace = 0;
};

struct fq_ctx {
  FqWorkPool pool;                     // the workers of this context's host phases (parallel_chunks on the calling thread)
  const fq_index *ix = nullptr;
  fq_opts_t o{};
  FqKOpts ko{};
  FqKnobs kn;
  fqdev::State *dev = nullptr;
  int max_pairs = 0;
  int debug = 0;
  std::string err;
  // single-stream sharding: callbacks around the order-dependent part of a call (fq_ctx_set_serial_hooks)
  fq_serial_hook before_serial = nullptr, after_serial = nullptr;
  void *hook_user = nullptr;
  bool serial_open = false, serial_done = false;   // this call's `before` hook has run / its `after` hook has run
  bool stream_broken = false;   // a call of this stream failed (here or on the rank a state came from): exported states say so, and the ranks behind fail too instead of waiting or going on with a stale state
  // order-dependent state
  uint64_t rng = 0;
  fq_isize_t last_ii{};
  std::unordered_map<uint64_t, vector<uint32_t>> kl_cache;
  int g_log_n[256];
  uint8_t maxdiff_lut[FQ_LMAX + 2];
  // the batch of the current call: ASCII rows resident in HBM (fq_batch_upload) or a packed host batch (fq_align_packed)
  int n_pairs = 0, stride = 0;
  int in_kind = 0;                      // 0 none, 1 ASCII, 2 packed, 3 FASTQ text resident in HBM (fq_align_text)
  fq_read_batch_t hb{};
  fq_packed_batch_t pb{};
  const fq_text_batch *tb = nullptr;    // the text batch of the current call: valid until the next call (the caller releases it after the consumers have run)
  PinBuf<uint8_t> p_cseq, p_cqual;      // ... its surviving reads' rows on the host, compact (FqHostReads::compact)
  PinBuf<int32_t> p_clen;
  PinBuf<char> p_cnames;
  DevBuf<char> d_cnames;
  vector<uint16_t> h_len_all;           // every row's length (debug dumps of a text batch)
  int c_stride = 0, c_name_stride = 0;
  // packed input: two head buffers (the next batch's head uploads while this one is aligned)
  DevBuf<uint64_t> d_head[2];
  DevBuf<uint16_t> d_hlen[2];
  DevBuf<uint8_t> d_qlast[2];
  int head_slot = 0;                               // buffer of the current call
  const fq_packed_batch_t *pend[2] = {nullptr, nullptr};   // batch whose head was prefetched into buffer i and not yet aligned
  uint64_t pend_serial[2] = {0, 0};                        // ... and the serial of its content (a batch object that was packed again is another batch)
  bool is_pending(int i, const fq_packed_batch_t *b) const { return pend[i] == b && pend_serial[i] == b->serial; }
  DevBuf<uint8_t> d_body, d_pqual;
  DevBuf<uint64_t> d_exc;
  DevBuf<int32_t> d_row_map, d_crow, d_len_c, d_len_all;
  PinBuf<uint8_t> p_body, p_pqual;
  PinBuf<uint64_t> p_exc;
  PinBuf<uint16_t> p_hlen;
  DevBuf<uint16_t> d_blen;
  // device-resident reads (ASCII rows the kernels after the filter read): whole batch (ASCII input) or survivors only (packed)
  DevBuf<uint8_t> d_seq, d_qual, d_filtered, d_maxdiff;
  DevBuf<FqGapWork> d_winfo;
  DevBuf<int32_t> d_len, d_len_trim, d_read_list, d_sidx, d_pair_list, d_counts;
  DevBuf<uint64_t> d_counters;
  std::vector<uint64_t> h_counters;   // (its stripes, as read back)
  DevBuf<uint32_t> d_queue;
  // search workspaces
  DevBuf<int32_t> d_work;
  DevBuf<uint32_t> d_heads, d_naln, d_status;
  DevBuf<uint32_t> d_wfull;
  DevBuf<FqPos> d_prec;
  DevBuf<uint8_t> d_bid_end;
  DevBuf<int32_t> d_order;
  DevBuf<uint32_t> d_order_cnt;
  DevBuf<FqEntry> d_pool;
  // second set for the search rounds that run beside the first round of a large call (stageA_search)
  DevBuf<FqGapWork> d_winfo2; DevBuf<uint32_t> d_queue2, d_heads2, d_naln2, d_status2, d_wfull2, d_order_cnt2, d_cnt2; DevBuf<int32_t> d_work2, d_order2;
  DevBuf<FqPos> d_prec2; DevBuf<uint8_t> d_bid_end2; DevBuf<FqEntry> d_pool2; DevBuf<FqAln> d_aln2;
  DevBuf<FqAln> d_aln;
  DevBuf<FqAln> d_hits;                   // the hit lists of the call, every launch's behind the earlier ones'; by search index: d_aoff / d_an
  DevBuf<uint64_t> d_aoff; DevBuf<uint32_t> d_an;
  DevBuf<uint64_t> d_off;
  // the records of the call and their side arrays (fq_records.h)
  DevBuf<FqDRec> d_rec, d_grec;
  DevBuf<uint32_t> d_nocc, d_qfirst, d_cnt[3], d_isz;
  DevBuf<uint16_t> d_ntop;
  DevBuf<uint8_t> d_enum, d_cls;
  DevBuf<uint64_t> d_row0, d_scan[3], d_rng, d_grow0;
  DevBuf<int32_t> d_list, d_refmax;
  DevBuf<fq_multi_t> d_multi, d_omulti;
  DevBuf<FqSwTask> d_swslot; DevBuf<FqSwCand> d_swcand;
  DevBuf<FqPairRead> d_greads; DevBuf<FqPairOut> d_gout;
  DevBuf<FqRefTgt> d_reftgt;
  DevBuf<uint16_t> d_cigs, d_ocig;
  DevBuf<fq_isize_t> d_iis;
  DevBuf<fq_result_t> d_orec;
  DevBuf<char> d_omd;
  PinBuf<fq_result_t> p_orec; PinBuf<uint16_t> p_ocig; PinBuf<char> p_omd; PinBuf<fq_multi_t> p_omulti;   // the C-ABI arrays land here
  PinBuf<uint16_t> p_ntop; PinBuf<uint32_t> p_isz; PinBuf<int32_t> p_pairs;
  // SA
  DevBuf<FqAln> d_qaln;
  DevBuf<uint32_t> d_qlen, d_pos, d_qpos, d_qrow, d_qinfo;
  DevBuf<FqPairIsize> d_pisize; DevBuf<int32_t> d_plut, d_glogn; DevBuf<uint64_t> d_pscratch;
  DevBuf<uint64_t> d_qoff;
  // DP
  DevBuf<FqSwTask> d_swtask;
  DevBuf<FqSwOut> d_swout;
  DevBuf<FqRefTask> d_reftask;
  DevBuf<FqRefOut> d_refout;
  DevBuf<uint16_t> d_cig, d_cigarena;
  DevBuf<uint8_t> d_scratch;
  DevBuf<char> d_md;
  DevBuf<uint16_t> d_mdmask;
  // host staging
  vector<uint8_t> h_filtered;
  vector<int32_t> h_len_trim, h_sub_max;
  const int32_t *h_pair_list = nullptr;   // survivor pair -> pair of the batch (p_pairs, where the copy engine lands it)
  const FqSurvInfo *h_surv = nullptr;     // ... and the survivors' search indices (p_surv)
  Retired retired;                     // buffers that grew during a call: freed with the context (hipFree waits for the device)
  PinArena arena;
  PinBuf<int32_t> p_i32;
  PinBuf<FqSurvInfo> p_surv;
  PinBuf<uint32_t> p_u32a, p_u32b, p_pos;
  PinBuf<FqAln> p_aln;
  DevBuf<FqSurvInfo> d_surv;
  DevBuf<int32_t> d_sub_max;
  int64_t n_bases_in = 0;
  // the consumers on the device (fq_emit.h; fq_ctx_set_emit): compact qualities and names of the surviving reads where the input kind
  // leaves them on the host, the SAM text of the last call
  int emit_flags = 0;
  bool host_rows = true;               // the last text batch's surviving rows were copied to the host
  DevBuf<uint8_t> d_equal; PinBuf<uint8_t> p_equal;
  DevBuf<char> d_enames; PinBuf<char> p_enames;
  DevBuf<uint32_t> d_samlen, d_sammeta; DevBuf<uint64_t> d_samoff; DevBuf<char> d_samtext;
  uint64_t sam_bytes = 0;
  bool sam_ready = false;
  // the consumer side's own streams and pinned slices: what a call left in HBM leaves the device beside the next call.  Two lanes, because the
  // record writer (0: SAM text, BAM records) and StatCollector's host side (1) fetch at the same time on threads of their own
  fqdev::State *dev_emit[2] = {nullptr, nullptr};
  PinBuf<char> p_emit[2][2];
  // ... the BAM records of a call (fq_ctx_attach_bam), as bytes in HBM until the writer fetches them
  fq_bam *bam = nullptr;
  DevBuf<uint32_t> d_bamlen, d_bammeta, d_zsize; DevBuf<uint64_t> d_bamoff, d_zoff; DevBuf<uint8_t> d_bamrec, d_zstage, d_bamz;
  FqBamCallOut bam_out;
  PinBuf<uint64_t> p_ztotal;
  uint32_t emit_nb = 0;
  std::mutex emit_mu;                  // the fill kernels of the last call may still run (emit_pending): the first fetcher waits for them
  bool emit_pending = false;
  // ... StatCollector's part of a call (fq_ctx_attach_qc): per-call lists on the device, what comes back for the consumer's host side
  fq_qc *qc = nullptr;
  DevBuf<int32_t> d_qadded;
  DevBuf<uint32_t> d_istlen, d_ptcnt;
  DevBuf<uint64_t> d_istoff, d_ptoff, d_qcnt, d_dupkey;
  DevBuf<char> d_isttext; DevBuf<FqPileEntry> d_pile;
  PinBuf<uint64_t> p_qcnt, p_dupkey;
  FqQcCallOut qc_out;
  // results of the last batch
  FqBatchState st;
  struct CallVecs {            // per-call host arrays of one entry per searched read: kept here so that a call reuses the last call's pages
    std::vector<int32_t> work, next_work;
  } cv;
  fq_stats_t stats{};
  double wait_ms = 0;                  // time the calling thread has spent waiting for the device (sync_staged)
  ~fq_ctx();
};

extern "C" void fq_default_opts(fq_opts_t *o) {
  memset(o, 0, sizeof *o);
  o->s_mm = 3; o->s_gapo = 11; o->s_gape = 4;
  o->mode = FQ_MODE_GAPE | FQ_MODE_COMPREAD;
  o->indel_end_skip = 5; o->max_del_occ = 10; o->max_entries = 2000000;
  o->fnr = 0.02; o->max_diff = -1; o->max_gapo = 1; o->max_gape = 6;
  o->max_seed_diff = 2; o->seed_len = 32; o->max_top2 = 30; o->trim_qual = 0; o->filter_thresh = 3;
  o->max_isize = 500; o->force_isize = 0; o->max_occ = 100000; o->n_multi = 3; o->N_multi = 10; o->is_sw = 1;
  o->ap_prior = 1e-5; o->host_threads = 0; o->batch_pairs = 262144; o->single_end = 0; o->pad_opts = 0;
}

extern "C" const char *fq_version(void) { return "fastquick_amd 0.1 (gfx950)"; }
extern "C" const char *fq_ctx_last_error(const fq_ctx_t *c) { return c ? c->err.c_str() : "null context"; }

extern "C" int fq_ctx_create(const fq_index_t *ix, const fq_opts_t *opts, int32_t max_pairs, fq_ctx_t **out) {
  if (!ix || !opts || !out || max_pairs <= 0) return FQ_EINVAL;
  *out = nullptr;
  const fq_opts_t &o = *opts;
  // limits imposed by the 32-bit entry packing / 128 score buckets (fq_common.h)
  if (o.max_gapo < 0 || o.max_gapo > 3 || o.max_gape < 0 || o.max_gape > 15 || o.seed_len < 1 || o.seed_len > FQ_SEED_MAX) return FQ_EINVAL;
  if (o.s_mm <= 0 || o.s_gapo <= 0 || o.s_gape <= 0) return FQ_EINVAL;   // children must score strictly more than parents (Q1)
  if (o.fnr <= 0.0 && (o.max_diff < 0 || o.max_diff > 30)) return FQ_EINVAL;
  if (o.batch_pairs < 1) return FQ_EINVAL;
  if (o.max_seed_diff < 0 || o.max_seed_diff > 30) return FQ_EINVAL;       // 5-bit lower bounds in the packed position records
  if (o.max_entries < 1 || o.max_entries > (1 << 30)) return FQ_EINVAL;   // 32-bit live-entry counter in the search kernel
  if (o.max_occ > (1u << 30) || o.n_multi < 0 || o.N_multi < 0 || o.n_multi > 4096 || o.N_multi > 4096) return FQ_EINVAL;   // 32-bit row counts per pair (fq_enum_plan_thread)
  std::unique_ptr<fq_ctx> c(new fq_ctx);
  c->ix = ix; c->o = o; c->max_pairs = max_pairs;
  int md_max = 0;
  for (int l = 0; l <= FQ_LMAX; ++l) {
    int md = o.fnr > 0.0 ? cal_maxdiff(l, 0.02, o.fnr) : o.max_diff;
    if (md > 30) return FQ_EINVAL;
    c->maxdiff_lut[l] = (uint8_t)md;
    md_max = std::max(md_max, md);
  }
  if ((md_max + 1) * o.s_mm + o.max_gapo * o.s_gapo + o.max_gape * o.s_gape >= FQ_MAX_BUCKETS) return FQ_EINVAL;
  if (o.max_gapo > c->maxdiff_lut[FQ_LMIN]) return FQ_EINVAL;   // the per-slice max_gapo clamp (BwtMapper.cpp:79-80) must be a no-op whatever the reads of a slice
  c->g_log_n[0] = 0;
  for (int i = 1; i < 256; ++i) c->g_log_n[i] = (int)(4.343 * log(i) + 0.5);   // bwase_initialize, bwase.c:602
  c->rng = ((uint64_t)ix->seed << 16) | 0x330EULL;                             // srand48(bns->seed)
  c->last_ii.avg = -1.0;
  FqKOpts &k = c->ko;
  k.s_mm = o.s_mm; k.s_gapo = o.s_gapo; k.s_gape = o.s_gape; k.mode = o.mode;
  k.indel_end_skip = o.indel_end_skip; k.max_del_occ = o.max_del_occ; k.max_entries = o.max_entries;
  k.max_gapo = o.max_gapo; k.max_gape = o.max_gape; k.max_seed_diff = o.max_seed_diff; k.seed_len = o.seed_len;
  k.max_top2 = o.max_top2; k.trim_qual = o.trim_qual; k.filter_thresh = o.filter_thresh; k.n_buckets = FQ_MAX_BUCKETS;
  if (const char *e = getenv("FASTQUICK_CTX_TRACE")) c->kn.trace = atoi(e);     // per-stage wall times of every call on stderr (as the tuning key `trace`)
  c->dev = fqdev::state_create(ix->device);
  if (!c->dev || fqdev::bind(c->dev)) return FQ_ENODEV;
  if (!c->d_maxdiff.ensure(FQ_LMAX + 2) || !c->d_counters.ensure(FQ_C_STRIPES * FQ_C_STRIDE) || !c->d_counts.ensure(4) || !c->d_queue.ensure(4) || !c->d_glogn.ensure(256)) return FQ_ENOMEM;
  if (fqdev::h2d(c->d_maxdiff.p, c->maxdiff_lut, FQ_LMAX + 2) || fqdev::h2d(c->d_glogn.p, c->g_log_n, 256 * 4) || fqdev::dzero(c->d_counters.p, (size_t)FQ_C_STRIPES * FQ_C_STRIDE * 8) || fqdev::sync()) return FQ_ENODEV;
  *out = c.release();
  return FQ_OK;
}
fq_ctx::~fq_ctx() { for (auto *e : dev_emit) if (e) fqdev::state_destroy(e); fqdev::state_destroy(dev); }   // synchronises the context's streams before the buffers below are freed
extern "C" void fq_ctx_destroy(fq_ctx_t *c) { delete c; }

extern "C" int fq_ctx_set_tuning(fq_ctx_t *c, const char *key, int64_t v) {
  if (!c || !key) return FQ_EINVAL;
  const std::string k = key;
  fqdev::Tune *t = fqdev::tune(c->dev);
  if (k == "gap_long_pops") c->kn.gap_long_pops = (uint32_t)v;
  else if (k == "gap_long_always") c->kn.gap_long_always = (int)v;
  else if (k == "gap_pool") c->kn.gap_pool = (uint32_t)v;
  else if (k == "gap_nogap_min") c->kn.gap_nogap_min = v;
  else if (k == "gap_pipeline_min") c->kn.gap_pipeline_min = v;
  else if (k == "gap_pipeline_segs") c->kn.gap_pipeline_segs = (int)std::max<int64_t>(2, std::min<int64_t>(v, 64));
  else if (k == "gap_split_hard") c->kn.gap_split_hard = v;
  else if (k == "gap_long_pops2") c->kn.gap_long_pops2 = (uint32_t)v;
  else if (k == "gap_skip_bound") c->kn.gap_skip_bound = v;
  else if (k == "gap_no_order") c->kn.gap_no_order = (int)v;
  else if (k == "device_turns") c->kn.device_turns = (int)v;
  else if (k == "device_turn_min") c->kn.device_turn_min = v;
  else if (k == "device_turn_slots") c->kn.device_turn_slots = (int)std::max<int64_t>(1, v);
  else if (k == "gap_round2_waves") c->kn.gap_round2_waves = (int)v;
  else if (k == "gap_round2_lane_major") c->kn.gap_round2_lane_major = (int)v;
  else if (k == "gap_round1_refill") c->kn.gap_round1_refill = (int)v;
  else if (k == "gap_round2_refill") c->kn.gap_round2_refill = (int)v;
  else if (k == "sw_wave_max") c->kn.sw_wave_max = (int)v;
  else if (k == "host_threads") c->kn.host_threads = (int)v;
  else if (k == "host_par_min") c->kn.host_par_min = (size_t)v;
  else if (k == "packed_bulk_min") c->kn.packed_bulk_min = v;
  else if (k == "md_mask_min") c->kn.md_mask_min = v;
  else if (k == "trace") c->kn.trace = (int)v;
  else if (k == "gap_order_asc") t->gap_order_asc = (int)v;
  else if (k == "gap_waves_per_cu") t->gap_waves_per_cu = (int)v;
  else if (k == "gap_coop_waves") t->gap_coop_waves = (int)v;
  else if (k == "gap_refill_min") t->gap_refill_min = (int)v;
  else if (k == "filter_no_turns") t->filter_no_turns = (int)v;
  else if (k == "prep_priority") t->prep_priority = (int)v;
  else if (k == "refine_lanes") t->refine_lanes = (int)v;
  else if (k == "gap_generic_opts") t->gap_generic_opts = (int)v;
  else if (k == "sw_serial_reverse") t->sw_serial_reverse = (int)v;
  else if (k == "width_both_strands") t->width_both_strands = (int)v;
  else if (k == "spin_sync") t->spin_sync = (int)v;
  else return FQ_EINVAL;
  return FQ_OK;
}

extern "C" void fq_stats_get(const fq_ctx_t *c, fq_stats_t *out) { if (c && out) *out = c->stats; }
extern "C" void fq_stats_reset(fq_ctx_t *c) { if (c) memset(&c->stats, 0, sizeof c->stats); }

#define CK(expr)                                                  \
  do {                                                            \
    if ((expr)) { c->err = std::string(#expr) + ": " + fqdev::last_error(); return FQ_ENODEV; } \
  } while (0)
#define CKM(expr)                                                 \
  do {                                                            \
    if (!(expr)) { c->err = std::string("out of device memory: ") + #expr; return FQ_ENOMEM; } \
  } while (0)

// small per-call copies, staged through the context's pinned arena (see PinArena)
static int h2d_staged(fq_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  void *p = c->arena.alloc(bytes);
  if (!p) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  memcpy(p, src, bytes);
  if (fqdev::copy_pinned(dst, p, bytes, 1)) { c->err = std::string("h2d: ") + fqdev::last_error(); return FQ_ENODEV; }
  return 0;
}
static int d2h_staged(fq_ctx *c, void *dst, const void *src, size_t bytes) {   // dst is valid after the next sync_staged()
  if (!bytes) return 0;
  void *p = c->arena.alloc(bytes);
  if (!p) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  if (fqdev::copy_pinned(p, src, bytes, 0)) { c->err = std::string("d2h: ") + fqdev::last_error(); return FQ_ENODEV; }
  c->arena.pending.push_back({dst, p, bytes});
  return 0;
}
static int sync_staged(fq_ctx *c) {
  const double t_wait = now_ms();
  if (fqdev::sync()) { c->err = std::string("sync: ") + fqdev::last_error(); return FQ_ENODEV; }
  c->wait_ms += now_ms() - t_wait;       // (the thread sleeps here: fq_stats_t::device_wait_ms)
  for (auto &o : c->arena.pending) memcpy(o.dst, o.src, o.bytes);
  c->arena.pending.clear();
  return 0;
}
// The work counters as the kernels keep them (FQ_C_STRIPES copies, fq_common.h), folded: sums, maxima for the two that are maxima.
static void fold_counters(const uint64_t *striped, uint64_t *cnt) {
  for (int k = 0; k < FQ_C_COUNT; ++k) {
    const bool is_max = k == FQ_C_MAXPOPS || k == FQ_C_MAXTRIPS;
    uint64_t v = 0;
    for (int st = 0; st < FQ_C_STRIPES; ++st) { const uint64_t x = striped[(size_t)st * FQ_C_STRIDE + k]; v = is_max ? std::max(v, x) : v + x; }
    cnt[k] = v;
  }
}
#define CKS(expr) do { const int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" int fq_batch_upload(fq_ctx_t *c, const fq_read_batch_t *in) {
  if (!c || !in || in->n_pairs < 0 || !in->seq || !in->qual || !in->len) return FQ_EINVAL;
  if (in->n_pairs > c->max_pairs) { c->err = "batch larger than max_pairs_per_batch"; return FQ_ELIMIT; }
  if (fqdev::bind(c->dev)) return FQ_ENODEV;
  if (in->stride < 1 || in->stride > 4096) return FQ_EINVAL;
  const size_t n2 = (size_t)in->n_pairs * (c->o.single_end ? 1 : 2);   // rows the caller provides
  int64_t nb = 0;
  for (size_t i = 0; i < n2; ++i) nb += in->len[i];
  c->n_bases_in = nb;
  for (size_t i = 0; i < n2; ++i)
    if (in->len[i] < FQ_LMIN || in->len[i] > FQ_LMAX || in->len[i] > in->stride) { c->err = "read length outside [" + std::to_string(FQ_LMIN) + "," + std::to_string(FQ_LMAX) + "]"; return FQ_ELIMIT; }
  CKM(c->d_seq.ensure((size_t)in->n_pairs * 2 * in->stride + 64));
  CKM(c->d_qual.ensure((size_t)in->n_pairs * 2 * in->stride + 64));
  CKM(c->d_len.ensure((size_t)in->n_pairs * 2 + 1));
  CK(fqdev::h2d(c->d_seq.p, in->seq, n2 * in->stride));
  CK(fqdev::h2d(c->d_qual.p, in->qual, n2 * in->stride));
  CK(fqdev::h2d(c->d_len.p, in->len, n2 * 4));
  CK(fqdev::sync());
  c->n_pairs = in->n_pairs;
  c->stride = in->stride;
  c->hb = *in;
  c->in_kind = 1;
  c->stats.h2d_bytes += 2 * n2 * (size_t)in->stride + n2 * 4;
  return FQ_OK;
}

// A call of a sharded stream (fq_ctx_set_serial_hooks) that fails before its order-dependent part has begun -- here, in the checks of
// the batch -- still takes the stream's state from the rank before it and hands it on, marked: nobody waits for a state that
// never comes, nobody goes on with a stale one (run_call does the same for failures further on).
static int broken_stream_call(fq_ctx_t *c, int rc) {
  if (rc && c && (c->before_serial || c->after_serial)) {
    c->stream_broken = true;
    if (c->before_serial) c->before_serial(c->hook_user);
    if (c->after_serial) c->after_serial(c->hook_user);
  }
  return rc;
}
extern "C" int fq_align_batch(fq_ctx_t *c, const fq_read_batch_t *in, fq_result_batch_t *out) {
  int rc = fq_batch_upload(c, in);
  if (rc) return broken_stream_call(c, rc);
  return fq_align_resident(c, out);
}

// ---- host-side scalar stages ----------------------------------------------------------------------------
namespace {

// The per-pair host phases are independent across pairs (everything order-dependent -- the drand48 stream, the insert-size
// chain, the (k,l) position cache -- is handled serially before them), so large batches are split over a few threads.
// The worker threads of a call stay on the NUMA node of the thread that drives it: the per-read records are walked serially (main
// hit choice in read order, insert sizes) right after phases that touch them in parallel, and on a two-socket host every record a
// worker on the other socket touched comes back over the socket link (main-hit phase of a 1 M-pair on-target call: 26 vs 78 ms).
struct NodeCpus { std::vector<cpu_set_t> sets; std::vector<int> node_of_cpu; };
static const NodeCpus &node_cpus() {
  static const NodeCpus nc = [] {
    NodeCpus r;
    for (int node = 0; node < 64; ++node) {
      char path[96];
      snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
      FILE *f = fopen(path, "r");
      if (!f) break;
      char buf[4096];
      cpu_set_t set; CPU_ZERO(&set);
      if (fgets(buf, sizeof buf, f))
        for (char *tok = strtok(buf, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
          int a = 0, b = 0;
          const int k = sscanf(tok, "%d-%d", &a, &b);
          if (k == 1) b = a;
          if (k >= 1) for (int cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu) { CPU_SET(cpu, &set); if ((int)r.node_of_cpu.size() <= cpu) r.node_of_cpu.resize(cpu + 1, -1); r.node_of_cpu[cpu] = node; }
        }
      fclose(f);
      r.sets.push_back(set);
    }
    return r;
  }();
  return nc;
}
// For the duration of a call the driving thread narrows its affinity to the CPUs of the node it is on (within the mask it was
// given), and restores it on return; threads it creates inherit the mask.
struct NodePin {
  cpu_set_t old;
  bool changed = false;
  NodePin() {
    const NodeCpus &nc = node_cpus();
    const int cpu = sched_getcpu();
    if (nc.sets.size() < 2 || cpu < 0 || cpu >= (int)nc.node_of_cpu.size() || nc.node_of_cpu[cpu] < 0) return;
    if (pthread_getaffinity_np(pthread_self(), sizeof old, &old) != 0) return;
    cpu_set_t want;
    CPU_AND(&want, &old, &nc.sets[nc.node_of_cpu[cpu]]);
    if (CPU_COUNT(&want) == 0 || CPU_EQUAL(&want, &old)) return;
    changed = pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
  }
  ~NodePin() { if (changed) pthread_setaffinity_np(pthread_self(), sizeof old, &old); }
};
// Threads of a call's host phases when neither the options nor the tuning say.  The phases stream over the per-read records and
// stop scaling where the memory system does: at most 8 threads on a small host, 16 where there are 32 cores or more (one on-target
// call of 4.2 M pairs on the 32-core host of an MI355X: 419 / 365 / 354 ms with 8 / 16 / 24 threads).  Calls of other contexts
// in flight in this process share the cores: sixteen streams of 1 M-pair calls run 16.3 / 12.7 / 8.8 M pairs/s with 4 / 8 / 16
// threads each, two streams of 4.2 M-pair calls 15.2 / 16.6 / 13.3 M pairs/s with 8 / 16 / 24 -- so twice the cores (a call waits for the device about half
// of its time) are divided by the number of calls in flight (on the contexts of the same index) when the call starts.  (Results do not depend on the number of threads.)
// (counted per process: the calls of all contexts, whatever index and device they belong to, share the host's cores -- the command
// line's --devices drives one index per device from one process)
static std::atomic<int> g_calls_in_flight{0};
struct CallInFlight {
  explicit CallInFlight(const fq_index *) { g_calls_in_flight.fetch_add(1, std::memory_order_relaxed); }
  ~CallInFlight() { g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed); }
};
// CPUs this process may actually use: the hardware threads it sees, cut to the cgroup's CPU quota when there is one (a container
// that shows 256 hardware threads and is throttled to 16 CPUs runs 32 threads per call SLOWER than 16 and burns half as much again:
// 3.0 against 1.9 core-seconds per 4.2 M-pair on-target call, profiles/round3_host_quota_and_threads.txt)
inline unsigned effective_cpus() {
  static const unsigned n = [] {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota> <period>" or "max <period>"
      char q[64]; long long period = 0;
      if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const long long quota = atoll(q);
        if (quota > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
      }
      fclose(f);
    } else {
      long long quota = -1, period = 0;                                         // cgroup v1
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
      if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
      if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
    }
    // The ranks of one node share the host (one process per GPU: torchrun exports LOCAL_WORLD_SIZE): each sizes its threads by its share,
    // not by the whole allowance.  FASTQUICK_HOST_CPUS states the share outright.
    if (const char *e = getenv("FASTQUICK_HOST_CPUS")) { const int v = atoi(e); if (v > 0) return std::min(hw, (unsigned)v); }
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) { const int v = atoi(e); if (v > 1) hw = std::max(1u, hw / (unsigned)v); }
    return hw;
  }();
  return n;
}
inline int default_host_threads(const fq_index *ix) {
  const unsigned hw = effective_cpus();
  const unsigned cap = hw >= 128 ? 32u : hw >= 16 ? 16u : std::min(8u, hw);   // (one 4.2 M-pair call on a 2 x 64-core host under a quota of 16 CPUs: 430 / 331 / 339 / 614 ms with 24 / 32 / 48 / 64 threads, 301-354 with 16)
  (void)ix;
  const unsigned share = 2 * hw / (unsigned)std::max(1, g_calls_in_flight.load(std::memory_order_relaxed));   // (a call waits for the device about half of its time)
  return (int)std::max(std::min(2u, cap), std::min(cap, share));
}
// The calling thread of a call hands its passes to the context's worker pool (run_call sets tl_pool); the side threads a call
// starts (record set-up, the drand48 plan) have none and fork their own helpers, so that they never queue behind the main thread's.
static thread_local FqWorkPool *tl_pool = nullptr;
template <class F>
void parallel_chunks(size_t n, int threads, size_t par_min, F fn) {   // fn(lo, hi, thread index); below par_min items the phase stays on the calling thread
  if (threads <= 1 || n < par_min) { fn((size_t)0, n, 0); return; }
  const size_t per = (n + threads - 1) / threads;
  const int chunks = (int)((n + per - 1) / per);
  if (tl_pool) {
    tl_pool->run(chunks, [&](int t) { const size_t lo = (size_t)t * per, hi = std::min(n, lo + per); if (lo < hi) fn(lo, hi, t); });
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t) {
    const size_t lo = (size_t)t * per, hi = std::min(n, lo + per);
    if (lo >= hi) break;
    th.emplace_back([=]() { fn(lo, hi, t); });
  }
  for (auto &x : th) x.join();
}

// v = n copies of val, written by the call's host threads (a serial vector::assign of the per-read arrays of a 4 M-pair call is 20-30 ms
// of one core per call; the vectors live with the context, so resize() itself initialises nothing after the first call)
template <class T>
void par_assign(std::vector<T> &v, size_t n, T val, int threads, size_t par_min) {
  v.resize(n);
  T *p = v.data();
  parallel_chunks(n, threads, par_min, [=](size_t lo, size_t hi, int) { std::fill(p + lo, p + hi, val); });
}
// in place: v[i] = v[0] + ... + v[i] (v[0] is the caller's start value); chunk sums, their serial prefix, then the chunks again
template <class T>
void par_prefix(std::vector<T> &v, int threads, size_t par_min) {
  const size_t n = v.size();
  if (threads <= 1 || n < par_min) { for (size_t i = 1; i < n; ++i) v[i] += v[i - 1]; return; }
  const size_t per = (n + threads - 1) / threads;
  std::vector<T> tot((size_t)threads + 1, T(0));
  T *p = v.data();
  parallel_chunks(n, threads, par_min, [&](size_t lo, size_t hi, int t) { T a = T(0); for (size_t i = lo; i < hi; ++i) a += p[i]; tot[(size_t)t + 1] = a; });
  for (int t = 0; t < threads; ++t) tot[(size_t)t + 1] += tot[t];
  parallel_chunks(n, threads, par_min, [&](size_t lo, size_t hi, int t) { T a = tot[t]; for (size_t i = lo; i < hi; ++i) { a += p[i]; p[i] = a; } });
  (void)per;
}

// infer_isize, libbwa/bwape.c:49-117
// (isz: one sample per survivor pair, written by all of the call's threads -- the scan of the records is what this stage costs, and a
// reference batch per thread leaves most of them idle)
void infer_isize(const uint32_t *isz, int sp_lo, int sp_hi, int max_len_all, fq_isize_t *ii, double ap_prior, int64_t L) {
  ii->avg = ii->std = -1.0; ii->low = ii->high = ii->high_bayesian = 0; ii->ap_prior = 0;
  vector<uint64_t> is;
  for (int s = sp_lo; s < sp_hi; ++s) if (isz[s] != ~0u) is.push_back(isz[s]);
  const int tot = (int)is.size();
  int max_len = std::max(1, max_len_all);
  if (tot < 20) return;
  if (tot < 4096) std::sort(is.begin(), is.end());
  else {   // every value is below 100,000: a counting sort leaves the same sorted array
    vector<uint32_t> cnt(100000, 0);
    for (uint64_t v : is) ++cnt[v];
    size_t at = 0;
    for (uint32_t v = 0; v < 100000; ++v) for (uint32_t k = cnt[v]; k; --k) is[at++] = v;
  }
  const int p25 = (int)is[(int)(tot * 0.25 + 0.5)], p75 = (int)is[(int)(tot * 0.75 + 0.5)];
  const int tmp = (int)(p25 - 2.0 * (p75 - p25) + .499);
  ii->low = (uint32_t)(tmp > max_len ? tmp : max_len);
  ii->high = (uint32_t)(int)(p75 + 2.0 * (p75 - p25) + .499);
  uint64_t x = 0;
  int n = 0;
  for (int i = 0; i < tot; ++i) if (is[i] >= ii->low && is[i] <= ii->high) { ++n; x += is[i]; }
  ii->avg = (double)x / n;
  for (int i = 0; i < tot; ++i)
    if (is[i] >= ii->low && is[i] <= ii->high) { const double t = (is[i] - ii->avg) * (is[i] - ii->avg); ii->std += t; }   // sum starts at -1.0 as in the reference
  ii->std = sqrt(ii->std / n);
  double y;
  for (y = 1.0; y < 10.0; y += 0.01) if (.5 * erfc(y / M_SQRT2) < ap_prior / L * (y * ii->std + ii->avg)) break;
  ii->high_bayesian = (uint32_t)(y * ii->std + ii->avg + .499);
  uint64_t n_ap = 0;
  for (int i = 0; i < tot; ++i) if (is[i] > ii->high_bayesian) ++n_ap;
  ii->ap_prior = .01 * (n_ap + .01) / tot;
  if (ii->ap_prior < ap_prior) ii->ap_prior = ap_prior;
  if (std::isnan(ii->std) || p75 > 100000) { ii->low = ii->high = ii->high_bayesian = 0; ii->avg = ii->std = -1.0; }
}

// pairing + __pairing_aux/__pairing_aux2, libbwa/bwape.c:119-213, bwape.h:55-82: fq_pair_sweep (fq_kernels.h) is the one
// implementation -- fq_pair_rec_thread runs it per lane, and the host runs it for the pairs the kernel does not take.
// the insert-size term of bwape.h:62 for every insert size of one reference batch (libm, as the reference evaluates it per candidate)
void pair_penalty_lut(const fq_isize_t &ii, vector<int32_t> &lut) {
  if (!ii.high) return;
  for (uint32_t l = 0; l <= ii.high_bayesian; ++l) {
    const double v = -4.343 * log(0.5 * erfc(M_SQRT1_2 * fabs(l - ii.avg) / ii.std)) + 0.499;
    lut.push_back(v >= 2147483648.0 || v != v ? INT32_MIN : (int32_t)v);   // (what the x86 conversion leaves for an infinite value)
  }
}
}  // namespace

// ---- the batch ------------------------------------------------------------------------------------------
// One call = stage 0 (filter + ordered compaction; the only stage that sees every read), the search over the reads of surviving
// pairs, then the record stages (fq_records.h) over device-resident records.  The stage functions share the per-call state in `Call`.
namespace {
struct EmitPlan {                      // what emit_measure leaves for emit_fill
  FqSamArgs sam{}; FqBamArgs bam{}; FqQcArgs qc{};
  uint64_t sam_total = 0, bam_total = 0, ist_total = 0, pt_total = 0;
};
struct Call {
  fq_ctx *c;
  EmitPlan plan_emit;
  // the call's place among the calls that count for the same QC consumer (taken right behind its order-dependent part); a call that ends early
  // still takes its turn, so that the calls behind it are not left waiting
  fq_qc *gate = nullptr; uint64_t gate_ticket = 0; bool gate_entered = false, gate_left = false;
  void gate_take(fq_qc *q) { if (q) { gate = q; gate_ticket = fq_qc_gate_ticket(q); } }
  void gate_enter() { if (gate && !gate_entered) { fq_qc_gate_enter(gate, gate_ticket); gate_entered = true; } }
  void gate_leave() { if (gate && gate_entered && !gate_left) { fq_qc_gate_leave(gate); gate_left = true; } }
  void gate_release() { if (gate && !gate_left) { gate_enter(); gate_leave(); } }
  explicit Call(fq_ctx *ctx) : c(ctx), aln_off(ctx->st.aln_off), aln_n(ctx->st.aln_n) {}
  int n = 0, n2 = 0, B = 0, n_sub = 0, n_search = 0, n_surv = 0, max_len_all = 1, host_threads = 1;
  size_t par_min = 32768;
  // where the kernels after the filter find the reads: ASCII rows [row][dstride], trimmed lengths [row], search index -> row.
  // ASCII input: row = the read's row in the batch; packed input: row = 2 * survivor pair + end (only those were unpacked).
  const uint8_t *dseq = nullptr;
  int dstride = 0;
  const int32_t *dlen_trim = nullptr, *dread_list = nullptr;
  FqSamArgs emit{};                    // the consumers' view of the call (fq_emit.h), made once per call
  bool emit_ready = false;
  vector<int> sub_max_len, sub_lo;
  // (the arrays of one entry per searched read live in the batch state and keep their pages)
  vector<uint64_t> &aln_off;           // per search index s: its hit list is S.aln[aln_off[s] .. + aln_n[s])
  vector<uint32_t> &aln_n;
  FqRecArgs ra{};                      // the record stages' arguments, filled in as the stages go
  uint64_t n_q = 0, n_rows = 0;        // hits / SA rows enumerated
  const uint32_t *h_pos = nullptr;     // positions of the enumerated SA rows (pinned staging of the context; fetched when the host pairs)
  bool have_pos = false;
  vector<fq_isize_t> iis;
  vector<FqPairIsize> pis;             // insert-size penalty tables, one per reference batch
  vector<int32_t> lut;
  int64_t n_host_pairs = 0;            // pairs the host pairs (FqRecArgs::cls == 2), listed in c->d_list / host_list
  const int32_t *host_list = nullptr;
  const FqPairRead *host_reads = nullptr;
  const uint64_t *host_row0 = nullptr;
  uint64_t n_multi = 0, n_sw = 0, n_ref = 0, cig_used = 0;   // XA entries, mate-rescue tasks, refine tasks, CIGAR arena entries in use
  // the main-hit stage's plan (stageB1_plan): where every chunk of pairs enters the drand48 stream
  struct B1Plan { uint64_t *start = nullptr; size_t n_chunks = 0; uint64_t rng_end = 0; } plan;
  std::thread plan_thread;             // the plan is drawn up beside the SA stage when nothing else can move the stream's state
  ~Call() { if (plan_thread.joinable()) plan_thread.join(); gate_release(); }
  double t_trace = 0, t_wall0 = 0, t_host0 = 0, t_serial1 = 0, t_host1 = 0, cpu_trace = 0, tcpu_trace = 0, w_trace = 0;
  double w_call0 = 0, w_host0 = 0, w_serial1 = 0, w_host1 = 0, cpu_call0 = 0;   // the context's wait_ms at those marks; the calling thread's CPU time at the start
  int sidx(size_t idx) const { return c->h_surv[idx].sidx; }
  const FqAln *aln_of(size_t idx, int *n_out) const {
    const int s = sidx(idx);
    if (s < 0) { *n_out = 0; return nullptr; }
    *n_out = (int)aln_n[s];
    return c->st.aln.data() + aln_off[s];
  }
  void trace(const char *label) {   // wall time since the previous mark, and the CPU time the whole process used meanwhile (all threads: core-ms)
    if (!c->kn.trace) return;
    const double t = now_ms();
    timespec ts;
    clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts);
    const double cpu = 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
    const double tcpu = 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec;
    fprintf(stderr, "[fq] %-22s %8.3f ms   cpu %8.1f core-ms   this thread %6.3f ms   waited for the device %6.3f ms\n", label, t - t_trace, cpu - cpu_trace, tcpu - tcpu_trace, c->wait_ms - w_trace);
    t_trace = t; cpu_trace = cpu; tcpu_trace = tcpu; w_trace = c->wait_ms;
  }
};

// survivors of stage 0 -> host lists and the reference-batch boundaries among them
int stage0_lists(Call &K, bool have_len_trim) {
  fq_ctx *c = K.c;
  const int n_surv = K.n_surv;
  CKM(c->p_pairs.ensure((size_t)n_surv + 1) && c->p_surv.ensure((size_t)n_surv * 2 + 1));
  CK(fqdev::copy_pinned(c->p_pairs.p, c->d_pair_list.p, (size_t)n_surv * 4, 0));
  CK(fqdev::copy_pinned(c->p_surv.p, c->d_surv.p, (size_t)n_surv * 2 * sizeof(FqSurvInfo), 0));
  CKS(sync_staged(c));
  c->stats.d2h_bytes += (size_t)n_surv * (4 + 2 * sizeof(FqSurvInfo));
  c->h_pair_list = c->p_pairs.p; c->h_surv = c->p_surv.p;   // read where they land
  (void)have_len_trim;
  K.sub_lo.assign(K.n_sub + 1, 0);
  for (int sb = 0; sb <= K.n_sub; ++sb)   // (the list is ascending)
    K.sub_lo[sb] = (int)(std::lower_bound(c->h_pair_list, c->h_pair_list + n_surv, (int32_t)std::min<int64_t>((int64_t)sb * K.B, INT32_MAX)) - c->h_pair_list);
  K.sub_lo[K.n_sub] = n_surv;
  return FQ_OK;
}
void stage0_sub_max(Call &K) {
  K.max_len_all = 1;
  K.sub_max_len.assign(K.n_sub, 1);
  for (int sb = 0; sb < K.n_sub; ++sb) {
    K.sub_max_len[sb] = std::max(1, K.c->h_sub_max[sb]);
    K.max_len_all = std::max(K.max_len_all, K.sub_max_len[sb]);
  }
}

// ---- stage 0, ASCII rows resident in HBM: encode + trim + filter + ordered compaction (GPU) ---------------------------
// The call may carry several reference batches (READ_BUFFER_SIZE pairs each, src/BwtMapper.h:36): the GPU stages run
// over all of them at once, the order-dependent host stages walk them one reference batch at a time.
int stage0_ascii(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  const int n = K.n, n2 = K.n2, n_sub = K.n_sub, B = K.B;
  CKM(c->d_len_trim.ensure(n2) && c->d_filtered.ensure(n2 + 64) && c->d_read_list.ensure(n2) &&
      c->d_sidx.ensure(n2) && c->d_pair_list.ensure(n) && c->d_sub_max.ensure(n_sub));
  CK(fqdev::dzero(c->d_sub_max.p, (size_t)n_sub * 4));
  // (the filter kernels of all contexts of a device are chained on the device: fqdev::launch_prep)
  FqPrepArgs a{};
  a.ix = ix->dev; a.o = c->ko; a.seq = c->d_seq.p; a.qual = c->d_qual.p; a.len = c->d_len.p; a.stride = c->stride; a.n_reads = c->o.single_end ? n : n2;
  if (c->o.single_end) {   // the rows of the absent mates: filtered, so that the pair machinery carries each read alone
    CK(fqdev::dfill(c->d_filtered.p + n, 1, (size_t)n));
    CK(fqdev::dzero(c->d_len_trim.p + n, (size_t)n * 4));
  }
  a.len_trim = c->d_len_trim.p; a.filtered = c->d_filtered.p; a.sub_max = c->d_sub_max.p; a.n_pairs = n; a.batch_pairs = B;
  a.counters = c->d_counters.p;
  fqdev::time_begin(FQ_K_PREP);
  CK(fqdev::launch_prep(a));
  CK(fqdev::launch_compact(c->d_filtered.p, n, c->d_read_list.p, c->d_sidx.p, c->d_pair_list.p, c->d_counts.p));
  fqdev::time_end(FQ_K_PREP);
  // Only what concerns surviving pairs comes back to the host: in a WGS-like stream that is a fraction of a percent of the
  // batch.  (Debug mode also fetches the per-read arrays of the whole batch for the stage dump.)
  int32_t counts[2] = {0, 0};
  c->h_sub_max.resize(n_sub);
  CKS(d2h_staged(c, counts, c->d_counts.p, 8));
  CKS(d2h_staged(c, c->h_sub_max.data(), c->d_sub_max.p, (size_t)n_sub * 4));
  CKS(sync_staged(c));
  K.trace("  stage0: filter + compaction on the device");
  K.n_search = counts[0]; K.n_surv = counts[1];
  CKM(c->d_surv.ensure((size_t)K.n_surv * 2 + 1));
  CK(fqdev::launch_surv_gather(c->d_pair_list.p, K.n_surv, n, c->d_len_trim.p, c->d_filtered.p, c->d_sidx.p, c->d_surv.p));
  if (c->debug) {
    c->h_filtered.resize(n2);
    c->h_len_trim.resize(n2);
    CKS(d2h_staged(c, c->h_filtered.data(), c->d_filtered.p, n2));
    CKS(d2h_staged(c, c->h_len_trim.data(), c->d_len_trim.p, (size_t)n2 * 4));
  } else { c->h_filtered.clear(); c->h_len_trim.clear(); }
  int rc = stage0_lists(K, true);
  if (rc) return rc;
  stage0_sub_max(K);
  K.dseq = c->d_seq.p; K.dstride = c->stride; K.dlen_trim = c->d_len_trim.p; K.dread_list = c->d_read_list.p;
  return FQ_OK;
}

// ---- stage 0, packed host batch (SURVEY 8d boundary) ---------------------------------------------------------------------
// Only the 24-byte filter key of every read crosses PCIe (its upload may have been started by fq_packed_prefetch and then
// ran under the previous call's kernels); the full rows -- and, when --q trimming is on, the qualities -- follow for the
// reads of surviving pairs only, and are unpacked into the compact ASCII rows the later kernels read.
int head_upload(fq_ctx *c, const fq_packed_batch_t *b, int slot) {
  const size_t n2 = (size_t)b->n_pairs * (b->single_end ? 1 : 2);   // rows the batch holds
  CKM(c->d_head[slot].ensure(n2 * 3 + 8));
  CK(fqdev::h2d_copy(c->d_head[slot].p, b->head, n2 * 24));
  c->stats.h2d_bytes += n2 * 24;
  if (!b->uniform_len) {
    CKM(c->d_hlen[slot].ensure(n2 + 8));
    CK(fqdev::h2d_copy(c->d_hlen[slot].p, b->len, n2 * 2));
    c->stats.h2d_bytes += n2 * 2;
  }
  if (c->o.trim_qual >= 1 && b->qual_last) {
    CKM(c->d_qlast[slot].ensure(n2 + 8));
    CK(fqdev::h2d_copy(c->d_qlast[slot].p, b->qual_last, n2));
    c->stats.h2d_bytes += n2;
  }
  CK(fqdev::copy_record(slot));
  return FQ_OK;
}
int stage0_packed(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  const fq_packed_batch_t &pb = c->pb;
  const int n = K.n, n2 = K.n2, n_sub = K.n_sub, B = K.B;
  // A single-end batch (BwtMapper::SingleEndMapper) holds n reads in n rows; the pair machinery carries each read alone, the rows
  // of the absent mates (n .. 2n-1) are filtered, empty reads, as in the ASCII path.
  const bool se = c->o.single_end != 0;
  const int n_in = se ? n : n2;                  // rows of the batch's arrays
  const bool ragged = pb.uniform_len <= 0;
  const bool trim = c->o.trim_qual >= 1;
  const int slot = c->head_slot;   // chosen by fq_align_packed: the buffer the batch was prefetched into, else a free one (uploaded now)
  CK(fqdev::compute_wait_copy(slot));
  const bool have_qlast = trim && pb.qual_last != nullptr;
  CKM(c->d_filtered.ensure(n2 + 64) && c->d_read_list.ensure(n2) && c->d_sidx.ensure(n2) && c->d_pair_list.ensure(n) && c->d_sub_max.ensure((size_t)3 * n_sub));
  CK(fqdev::dzero(c->d_sub_max.p, (size_t)3 * n_sub * 4));
  if (se) CK(fqdev::dfill(c->d_filtered.p + n, 1, (size_t)n));
  FqPrepPackedArgs a{};
  a.qual_last = have_qlast ? c->d_qlast[slot].p : nullptr; a.sub_whole = c->d_sub_max.p + n_sub;
  a.ix = ix->dev; a.o = c->ko; a.head = c->d_head[slot].p; a.len = ragged ? c->d_hlen[slot].p : nullptr; a.uniform_len = pb.uniform_len;
  a.n_reads = n_in; a.filtered = c->d_filtered.p; a.sub_max = c->d_sub_max.p; a.n_pairs = n; a.batch_pairs = B; a.counters = c->d_counters.p;
  fqdev::time_begin(FQ_K_PREP);
  CK(fqdev::launch_prep_packed(a));
  CK(fqdev::launch_compact(c->d_filtered.p, n, c->d_read_list.p, c->d_sidx.p, c->d_pair_list.p, c->d_counts.p));
  fqdev::time_end(FQ_K_PREP);
  int32_t counts[2] = {0, 0};
  uint64_t lcnt[2] = {0, 0};   // FQ_C_BASES, FQ_C_BADLEN (ragged batches)
  c->h_sub_max.assign(n_sub, pb.uniform_len);   // longest untrimmed read per reference batch
  vector<int32_t> sub_whole(n_sub, 0);          // longest read trimming leaves whole (known without its quality row)
  CKS(d2h_staged(c, counts, c->d_counts.p, 8));
  if (have_qlast) CKS(d2h_staged(c, sub_whole.data(), c->d_sub_max.p + n_sub, (size_t)n_sub * 4));
  std::vector<uint64_t> striped(ragged ? (size_t)FQ_C_STRIPES * FQ_C_STRIDE : 0);
  if (ragged) {
    CKS(d2h_staged(c, c->h_sub_max.data(), c->d_sub_max.p, (size_t)n_sub * 4));
    CKS(d2h_staged(c, striped.data(), c->d_counters.p, striped.size() * 8));
  }
  CKS(sync_staged(c));
  if (ragged) {
    uint64_t folded[FQ_C_COUNT];
    fold_counters(striped.data(), folded);
    lcnt[0] = folded[FQ_C_BASES]; lcnt[1] = folded[FQ_C_BADLEN];
    if (lcnt[1]) { c->err = "read length outside [" + std::to_string(FQ_LMIN) + "," + std::to_string(FQ_LMAX) + "]"; return FQ_ELIMIT; }
    for (int sb = 0; sb < n_sub; ++sb)   // a ragged row longer than its packed row would be unpacked from its neighbour's bytes
      if (((int64_t)c->h_sub_max[sb] + 3) / 4 > (int64_t)pb.body_stride) { c->err = "a read is longer than body_stride holds"; return FQ_EINVAL; }
    c->n_bases_in = (int64_t)lcnt[0];
  } else c->n_bases_in = (int64_t)n_in * pb.uniform_len;
  const int n_search = counts[0], n_surv = counts[1], nrow = 2 * n_surv;
  K.n_search = n_search; K.n_surv = n_surv;
  const int64_t bulk_min = c->kn.packed_bulk_min >= 0 ? c->kn.packed_bulk_min : (int64_t)n / 8;
  const bool bulk = !se && n_surv > 0 && (int64_t)n_surv >= bulk_min;   // many survivors: upload the whole body, gather on the device (single-end batches: rows gathered on the host, the absent mates' among them as empty rows)
  CKM(c->d_surv.ensure((size_t)nrow + 1) && c->d_row_map.ensure((size_t)nrow + 1) && c->d_len_c.ensure((size_t)nrow + 1) && c->d_len_trim.ensure((size_t)nrow + 1));
  const bool need_crow = bulk && pb.n_exc > 0;   // batch row -> compact row (-1: not unpacked), for the exception list of the whole batch
  if (need_crow) { CKM(c->d_crow.ensure(n2)); CK(fqdev::dfill(c->d_crow.p, 0xff, (size_t)n2 * 4)); }
  // d_read_list is rewritten as search index -> compact row
  CK(fqdev::launch_surv_map(c->d_pair_list.p, n_surv, n, c->d_filtered.p, c->d_sidx.p, c->d_surv.p, c->d_row_map.p, c->d_read_list.p, need_crow ? c->d_crow.p : nullptr));
  int rc = stage0_lists(K, false);
  if (rc) return rc;
  // ---- the reads of surviving pairs: body rows (+ exceptions, + qualities when trimming) -> compact ASCII rows ----
  const int body_stride = pb.body_stride;
  int max_full = 1;
  for (int sb = 0; sb < n_sub; ++sb) max_full = std::max(max_full, c->h_sub_max[sb]);
  const int cstride = (max_full + 15) & ~15;
  CKM(c->d_seq.ensure((size_t)nrow * cstride + 64));
  FqUnpackArgs ua{};
  ua.body_stride = body_stride; ua.uniform_len = pb.uniform_len; ua.n_rows = nrow; ua.seq = c->d_seq.p; ua.stride = cstride;
  ua.len_out = c->d_len_c.p; ua.len_trim = c->d_len_trim.p;
  FqPatchArgs pa{};
  pa.seq = c->d_seq.p; pa.stride = cstride;
  FqTrimArgs ta{};
  ta.o = c->ko; ta.qual_stride = pb.qual_stride; ta.len = c->d_len_c.p; ta.n_rows = nrow; ta.len_trim = c->d_len_trim.p;
  ta.pair_list = c->d_pair_list.p; ta.batch_pairs = B; ta.sub_max = c->d_sub_max.p + 2 * n_sub;
  if (nrow && bulk) {
    CKM(c->d_body.ensure((size_t)n2 * body_stride + 64));
    CK(fqdev::h2d(c->d_body.p, pb.body, (size_t)n2 * body_stride));
    c->stats.h2d_bytes += (size_t)n2 * body_stride;
    ua.body = c->d_body.p; ua.row_map = c->d_row_map.p;
    if (ragged) ua.len = c->d_hlen[slot].p;
    if (pb.n_exc) {
      CKM(c->d_exc.ensure((size_t)pb.n_exc + 1));
      CK(fqdev::h2d(c->d_exc.p, pb.exc, (size_t)pb.n_exc * 8));
      c->stats.h2d_bytes += (size_t)pb.n_exc * 8;
      pa.exc = c->d_exc.p; pa.n_exc = pb.n_exc; pa.crow_of = c->d_crow.p;
    }
    if (trim) {
      CKM(c->d_pqual.ensure((size_t)n2 * pb.qual_stride + 64));
      CK(fqdev::h2d(c->d_pqual.p, pb.qual, (size_t)n2 * pb.qual_stride));
      c->stats.h2d_bytes += (size_t)n2 * pb.qual_stride;
      ta.qual = c->d_pqual.p; ta.row_map = c->d_row_map.p;
    }
  } else if (nrow) {
    // few survivors: gather their rows into pinned staging on the host (a few thousand rows per reference batch in a WGS stream)
    CKM(c->p_body.ensure((size_t)nrow * body_stride + 64) && c->d_body.ensure((size_t)nrow * body_stride + 64));
    const bool row_lens = ragged || se;   // per-row lengths travel with the rows (single-end: the absent mates' rows have none)
    if (row_lens) CKM(c->p_hlen.ensure((size_t)nrow + 8) && c->d_blen.ensure((size_t)nrow + 8));
    if (trim) CKM(c->p_pqual.ensure((size_t)nrow * pb.qual_stride + 64) && c->d_pqual.ensure((size_t)nrow * pb.qual_stride + 64));
    size_t ne = 0;
    vector<std::pair<size_t, size_t>> erange;   // exceptions of each compact row: [lo, hi) in pb.exc
    if (pb.n_exc) erange.resize(nrow);
    for (int t = 0; t < nrow; ++t) {
      const size_t r = (size_t)(t & 1) * (size_t)n + (size_t)c->h_pair_list[t >> 1];
      if (se && (t & 1)) {   // the absent mate of a single-end read: an empty row
        memset(c->p_body.p + (size_t)t * body_stride, 0, (size_t)body_stride);
        c->p_hlen.p[t] = 0;
        if (trim) memset(c->p_pqual.p + (size_t)t * pb.qual_stride, 0, (size_t)pb.qual_stride);
        if (pb.n_exc) erange[t] = {0, 0};
        continue;
      }
      memcpy(c->p_body.p + (size_t)t * body_stride, pb.body + r * (size_t)body_stride, (size_t)body_stride);
      if (row_lens) c->p_hlen.p[t] = ragged ? pb.len[r] : (uint16_t)pb.uniform_len;
      if (trim) memcpy(c->p_pqual.p + (size_t)t * pb.qual_stride, pb.qual + r * (size_t)pb.qual_stride, (size_t)pb.qual_stride);
      if (pb.n_exc) {
        const uint64_t *lo = std::lower_bound(pb.exc, pb.exc + pb.n_exc, (uint64_t)r << 32);
        const uint64_t *hi = std::lower_bound(lo, pb.exc + pb.n_exc, (uint64_t)(r + 1) << 32);
        erange[t] = {(size_t)(lo - pb.exc), (size_t)(hi - pb.exc)};
        ne += (size_t)(hi - lo);
      }
    }
    CK(fqdev::copy_pinned(c->d_body.p, c->p_body.p, (size_t)nrow * body_stride, 1));
    c->stats.h2d_bytes += (size_t)nrow * body_stride;
    ua.body = c->d_body.p; ua.row_map = nullptr;
    if (row_lens) { CK(fqdev::copy_pinned(c->d_blen.p, c->p_hlen.p, (size_t)nrow * 2, 1)); ua.len = c->d_blen.p; c->stats.h2d_bytes += (size_t)nrow * 2; }
    if (ne) {
      CKM(c->p_exc.ensure(ne + 1) && c->d_exc.ensure(ne + 1));
      size_t at = 0;
      for (int t = 0; t < nrow; ++t)
        for (size_t q = erange[t].first; q < erange[t].second; ++q) c->p_exc.p[at++] = ((uint64_t)t << 32) | (pb.exc[q] & 0xffffffffull);
      CK(fqdev::copy_pinned(c->d_exc.p, c->p_exc.p, ne * 8, 1));
      c->stats.h2d_bytes += ne * 8;
      pa.exc = c->d_exc.p; pa.n_exc = (int64_t)ne; pa.crow_of = nullptr;
    }
    if (trim) {
      CK(fqdev::copy_pinned(c->d_pqual.p, c->p_pqual.p, (size_t)nrow * pb.qual_stride, 1));
      c->stats.h2d_bytes += (size_t)nrow * pb.qual_stride;
      ta.qual = c->d_pqual.p; ta.row_map = nullptr;
    }
  }
  fqdev::time_begin(FQ_K_PREP);
  if (nrow) {
    CK(fqdev::launch_unpack(ua));
    if (pa.n_exc) CK(fqdev::launch_patch(pa));
    if (trim) CK(fqdev::launch_trim(ta));
  }
  fqdev::time_end(FQ_K_PREP);
  // the longest trimmed read among the survivors of every reference batch (the records themselves are set up on the device)
  vector<int32_t> surv_max(n_sub, 0);
  if (nrow && trim) { CKS(d2h_staged(c, surv_max.data(), c->d_sub_max.p + 2 * n_sub, (size_t)n_sub * 4)); CKS(sync_staged(c)); c->stats.d2h_bytes += (size_t)n_sub * 4; }
  // infer_isize's max_len is the longest trimmed read of the whole reference batch, filtered reads included (bwape.c:60-61).
  // Without trimming that is the longest read (known above).  With trimming, the trimmed lengths of the survivors are known, and
  // so is the length of every read that trimming leaves whole (from its last quality byte, fq_prep_packed_thread); every other
  // read is shorter than the batch's longest.  When those known lengths reach the batch's longest read they are the maximum;
  // otherwise -- and for the debug dump, which lists every read -- the qualities of all reads go to the device.
  for (int sb = 0; sb < n_sub; ++sb) surv_max[sb] = std::max(surv_max[sb], (int)sub_whole[sb]);
  bool need_all = trim && c->debug;
  if (trim) for (int sb = 0; sb < n_sub; ++sb) if (surv_max[sb] < c->h_sub_max[sb]) need_all = true;
  c->h_filtered.clear(); c->h_len_trim.clear();
  if (need_all) {
    CKM(c->d_pqual.ensure((size_t)n_in * pb.qual_stride + 64) && c->d_len_all.ensure(n2));
    CK(fqdev::h2d(c->d_pqual.p, pb.qual, (size_t)n_in * pb.qual_stride));
    c->stats.h2d_bytes += (size_t)n_in * pb.qual_stride;
    CK(fqdev::dzero(c->d_sub_max.p, (size_t)n_sub * 4));
    if (se) CK(fqdev::dzero(c->d_len_all.p + n, (size_t)n * 4));
    FqTrimAllArgs aa{};
    aa.o = c->ko; aa.qual = c->d_pqual.p; aa.qual_stride = pb.qual_stride; aa.len = ragged ? c->d_hlen[slot].p : nullptr; aa.uniform_len = pb.uniform_len;
    aa.n_reads = n_in; aa.n_pairs = n; aa.batch_pairs = B; aa.len_trim = c->d_len_all.p; aa.sub_max = c->d_sub_max.p;
    CK(fqdev::launch_trim_all(aa));
    CKS(d2h_staged(c, c->h_sub_max.data(), c->d_sub_max.p, (size_t)n_sub * 4));
    if (c->debug) { c->h_len_trim.resize(n2); CKS(d2h_staged(c, c->h_len_trim.data(), c->d_len_all.p, (size_t)n2 * 4)); }
    CKS(sync_staged(c));
  } else if (trim) {
    for (int sb = 0; sb < n_sub; ++sb) c->h_sub_max[sb] = surv_max[sb];
  }
  if (c->debug) {
    c->h_filtered.resize(n2);
    CKS(d2h_staged(c, c->h_filtered.data(), c->d_filtered.p, n2));
    CKS(sync_staged(c));
    if (c->h_len_trim.empty()) { c->h_len_trim.resize(n2); for (int r = 0; r < n2; ++r) c->h_len_trim[r] = r >= n_in ? 0 : ragged ? (int)pb.len[r] : pb.uniform_len; }
  }
  stage0_sub_max(K);
  K.dseq = c->d_seq.p; K.dstride = cstride; K.dlen_trim = c->d_len_trim.p; K.dread_list = c->d_read_list.p;
  return FQ_OK;
}

// ---- stage 0, FASTQ text resident in HBM (the device front end's batches, fq_frontend.cpp) --------------------------------------------
// The filter's keys are in place (fqt_piece_thread / fqt_slot_bases_thread wrote what fq_pack_reads_into writes on the host), so the filter
// and the compaction run as for a packed batch without an upload; the reads of surviving pairs are gathered from the text into the
// compact rows every later kernel reads, and come to the host for the consumers (bases, qualities, names of the survivors only).
int stage0_text(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  const fq_text_batch &tb = *c->tb;
  const int n = K.n, n2 = K.n2, n_sub = K.n_sub, B = K.B;
  const bool se = c->o.single_end != 0;
  const int n_in = se ? n : n2;
  const bool ragged = tb.uniform_len <= 0;
  const bool trim = c->o.trim_qual >= 1;
  CKM(c->d_filtered.ensure(n2 + 64) && c->d_read_list.ensure(n2) && c->d_sidx.ensure(n2) && c->d_pair_list.ensure(n) && c->d_sub_max.ensure((size_t)3 * n_sub));
  CK(fqdev::dzero(c->d_sub_max.p, (size_t)3 * n_sub * 4));
  if (se) CK(fqdev::dfill(c->d_filtered.p + n, 1, (size_t)n));
  FqPrepPackedArgs a{};
  a.qual_last = nullptr; a.sub_whole = nullptr;
  a.ix = ix->dev; a.o = c->ko; a.head = tb.d_head; a.len = ragged ? tb.d_hlen : nullptr; a.uniform_len = tb.uniform_len;
  a.n_reads = n_in; a.filtered = c->d_filtered.p; a.sub_max = c->d_sub_max.p; a.n_pairs = n; a.batch_pairs = B; a.counters = c->d_counters.p;
  fqdev::time_begin(FQ_K_PREP);
  CK(fqdev::launch_prep_packed(a));
  CK(fqdev::launch_compact(c->d_filtered.p, n, c->d_read_list.p, c->d_sidx.p, c->d_pair_list.p, c->d_counts.p));
  fqdev::time_end(FQ_K_PREP);
  int32_t counts[2] = {0, 0};
  uint64_t lcnt[2] = {0, 0};
  c->h_sub_max.assign(n_sub, tb.uniform_len);
  CKS(d2h_staged(c, counts, c->d_counts.p, 8));
  std::vector<uint64_t> striped(ragged ? (size_t)FQ_C_STRIPES * FQ_C_STRIDE : 0);
  if (ragged) {
    CKS(d2h_staged(c, c->h_sub_max.data(), c->d_sub_max.p, (size_t)n_sub * 4));
    CKS(d2h_staged(c, striped.data(), c->d_counters.p, striped.size() * 8));
  }
  CKS(sync_staged(c));
  K.trace("  stage0: filter + compaction of the text batch");
  if (ragged) {
    uint64_t folded[FQ_C_COUNT];
    fold_counters(striped.data(), folded);
    lcnt[0] = folded[FQ_C_BASES]; lcnt[1] = folded[FQ_C_BADLEN];
    if (lcnt[1]) { c->err = "read length outside [" + std::to_string(FQ_LMIN) + "," + std::to_string(FQ_LMAX) + "]"; return FQ_ELIMIT; }
    c->n_bases_in = (int64_t)lcnt[0];
  } else c->n_bases_in = (int64_t)n_in * tb.uniform_len;
  const int n_search = counts[0], n_surv = counts[1], nrow = 2 * n_surv;
  K.n_search = n_search; K.n_surv = n_surv;
  CKM(c->d_surv.ensure((size_t)nrow + 1) && c->d_row_map.ensure((size_t)nrow + 1) && c->d_len_c.ensure((size_t)nrow + 1) && c->d_len_trim.ensure((size_t)nrow + 1));
  CK(fqdev::launch_surv_map(c->d_pair_list.p, n_surv, n, c->d_filtered.p, c->d_sidx.p, c->d_surv.p, c->d_row_map.p, c->d_read_list.p, nullptr));
  int rc = stage0_lists(K, false);
  if (rc) return rc;
  K.trace("  stage0: survivors' lists D2H");
  // ---- the reads of surviving pairs: rows, qualities and names out of the text ----
  int max_full = 1;
  for (int sb = 0; sb < n_sub; ++sb) max_full = std::max(max_full, c->h_sub_max[sb]);
  const int cstride = (max_full + 15) & ~15;
  const int ns = tb.name_stride;
  c->c_stride = cstride; c->c_name_stride = ns;
  CKM(c->d_seq.ensure((size_t)nrow * cstride + 64) && c->d_pqual.ensure((size_t)nrow * cstride + 64) && c->d_cnames.ensure((size_t)nrow * ns + 64));
  // (the consumers on the device read the rows where they lie: FQ_EMIT_DEVICE_ONLY leaves them there)
  const bool host_rows = !(c->emit_flags & FQ_EMIT_DEVICE_ONLY) || c->debug;
  c->host_rows = host_rows;
  if (host_rows) CKM(c->p_cseq.ensure((size_t)nrow * cstride + 64) && c->p_cqual.ensure((size_t)nrow * cstride + 64) && c->p_clen.ensure((size_t)nrow + 8) && c->p_cnames.ensure((size_t)nrow * ns + 64));
  fqdev::time_begin(FQ_K_PREP);
  if (nrow) {
    FqTextGatherArgs g{};
    g.text[0] = tb.d_text[0]; g.text[1] = tb.d_text[1]; g.rec = tb.d_rec; g.names = tb.d_names; g.name_stride = ns; g.n_pairs = n; g.single_end = se ? 1 : 0;
    g.pair_list = c->d_pair_list.p; g.n_out = nrow; g.seq = c->d_seq.p; g.qual = c->d_pqual.p; g.stride = cstride;
    g.len_out = c->d_len_c.p; g.len_trim = c->d_len_trim.p; g.names_out = c->d_cnames.p;
    CK(fqdev::launch_text_gather(g));
    if (trim) {
      FqTrimArgs ta{};
      ta.o = c->ko; ta.qual = c->d_pqual.p; ta.qual_stride = cstride; ta.row_map = nullptr; ta.len = c->d_len_c.p; ta.n_rows = nrow; ta.len_trim = c->d_len_trim.p;
      ta.pair_list = c->d_pair_list.p; ta.batch_pairs = B; ta.sub_max = c->d_sub_max.p + 2 * n_sub;
      CK(fqdev::launch_trim(ta));
    }
    if (host_rows) {
      CK(fqdev::copy_pinned(c->p_cseq.p, c->d_seq.p, (size_t)nrow * cstride, 0));
      CK(fqdev::copy_pinned(c->p_cqual.p, c->d_pqual.p, (size_t)nrow * cstride, 0));
      CK(fqdev::copy_pinned(c->p_clen.p, c->d_len_c.p, (size_t)nrow * 4, 0));
      CK(fqdev::copy_pinned(c->p_cnames.p, c->d_cnames.p, (size_t)nrow * ns, 0));
      c->stats.d2h_bytes += (size_t)nrow * (2 * (size_t)cstride + 4 + (size_t)ns);
    }
  }
  fqdev::time_end(FQ_K_PREP);
  // infer_isize's max_len is the longest trimmed read of the whole reference batch, filtered reads included (libbwa/bwape.c:60-61): every
  // read's qualities are in HBM, so bwa_trim_read runs over all of them where they lie
  c->h_filtered.clear(); c->h_len_trim.clear(); c->h_len_all.clear();
  if (trim) {
    CKM(c->d_len_all.ensure(n2));
    CK(fqdev::dzero(c->d_sub_max.p, (size_t)n_sub * 4));
    if (se) CK(fqdev::dzero(c->d_len_all.p + n, (size_t)n * 4));
    FqTextTrimArgs aa{};
    aa.o = c->ko; aa.text[0] = tb.d_text[0]; aa.text[1] = tb.d_text[1]; aa.rec = tb.d_rec; aa.n_rows = n_in; aa.n_pairs = n; aa.batch_pairs = B;
    aa.len_trim = c->d_len_all.p; aa.sub_max = c->d_sub_max.p;
    CK(fqdev::launch_text_trim_all(aa));
    CKS(d2h_staged(c, c->h_sub_max.data(), c->d_sub_max.p, (size_t)n_sub * 4));
    if (c->debug) { c->h_len_trim.resize(n2); CKS(d2h_staged(c, c->h_len_trim.data(), c->d_len_all.p, (size_t)n2 * 4)); }
  }
  if (c->debug) {
    c->h_filtered.resize(n2);
    CKS(d2h_staged(c, c->h_filtered.data(), c->d_filtered.p, n2));
    c->h_len_all.assign(n2, 0);
    CKS(d2h_staged(c, c->h_len_all.data(), tb.d_hlen, (size_t)n_in * 2));
  }
  CKS(sync_staged(c));
  if (c->debug && c->h_len_trim.empty()) { c->h_len_trim.resize(n2); for (int r = 0; r < n2; ++r) c->h_len_trim[r] = r >= n_in ? 0 : (int)c->h_len_all[r]; }
  stage0_sub_max(K);
  K.trace("  stage0: rows, qualities, names gathered");
  K.dseq = c->d_seq.p; K.dstride = cstride; K.dlen_trim = c->d_len_trim.p; K.dread_list = c->d_read_list.p;
  return FQ_OK;
}

// ---- stage A: widths + gap search, tiered by stack-pool size (GPU) ----------------------------------
// S.aln: concatenated hit lists; per search index s: [aln_off[s], aln_off[s]+aln_n[s])
int stageA_search(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  const fq_opts_t &o = c->o;
  const int n_search = K.n_search, max_len_all = K.max_len_all;
  // the hit lists stay where the copy engine lands them (the context's pinned buffer): S.aln is a view of it
  c->st.aln.p = c->p_aln.p; c->st.aln.n = 0;
  par_assign(K.aln_off, (size_t)n_search + 1, (uint64_t)0, K.host_threads, K.par_min);
  par_assign(K.aln_n, (size_t)n_search, (uint32_t)0, K.host_threads, K.par_min);
  CKM(c->d_aoff.ensure((size_t)n_search + 1) && c->d_an.ensure((size_t)n_search + 1) && c->d_hits.ensure(1));
  CK(fqdev::dzero(c->d_an.p, ((size_t)n_search + 1) * 4));
  const int Lpad = (max_len_all + 1 + 7) & ~7;                  // exact widths per strand; rows are written 8 positions at a time
  const int Ppad = (max_len_all + 1 + FQ_POS_PAD + 7) & ~7;     // position records per strand (16-byte aligned rows)
  // tier 0: one read per lane, bounded stack and pop count; what it gives up on is searched again by one wavefront per read
  // (tier 1 with push-time pruning, tier 2 exactly as the reference: no pruning, n_entries exact).  The wavefront kernel never
  // reuses pool slots, so its pools hold every push of a search, not just the live entries.
  const uint32_t exact_pool = (uint32_t)std::min<uint64_t>(4ull * (uint64_t)o.max_entries + 4096ull, 0x7fffffffull);
  const FqGapTier lane_tier = {c->kn.gap_pool, 32u, 0, 0, c->kn.gap_long_pops, c->kn.gap_long_always, 0, 0};
  const FqGapTier wave_tier = {262144u, 512u, 0, 1, 0u, 0, 0, 0}, exact_tier = {exact_pool, 8192u, 1, 1, 0u, 0, 0, 0};
  // scores that can occur for the longest read of this call (children may exceed max_diff by one difference)
  const int nb_need = (c->maxdiff_lut[max_len_all] + 1) * o.s_mm + o.max_gapo * o.s_gapo + o.max_gape * o.s_gape + 1;
  // A launch that fills the device begins with the round that searches without gap children (FqGapLane, NOGAP): the reads it
  // cannot settle are searched in full by the next round.
  vector<FqGapTier> tiers;
  vector<size_t> chunk_reads;   // pools are per lane / per wavefront; only per-read outputs scale with the chunk
  if (c->kn.gap_nogap_min >= 0 && !c->kn.gap_long_always && !(o.mode & FQ_MODE_NONSTOP) && c->kn.gap_pool <= 65535u) {
    FqGapTier t = lane_tier; t.nogap = 1; t.long_pops = 0; tiers.push_back(t); chunk_reads.push_back((size_t)8 << 20);
  }
  const size_t n_first_tiers = tiers.size();
  tiers.push_back(lane_tier); chunk_reads.push_back((size_t)8 << 20);
  tiers.push_back(wave_tier); chunk_reads.push_back((size_t)1 << 20);
  tiers.push_back(exact_tier); chunk_reads.push_back(4096);
  vector<int32_t> &work = c->cv.work, &next_work = c->cv.next_work;
  work.resize((size_t)n_search);
  parallel_chunks((size_t)n_search, K.host_threads, K.par_min, [&](size_t lo, size_t hi, int) { for (size_t s = lo; s < hi; ++s) work[s] = (int32_t)s; });
  bool ran_nogap = false;
  size_t n_hard = 0;                      // leading items of `work` that the round without gap children left without any hit (fq_order_key)
  vector<uint8_t> bound_of;               // experiment: min over strands of k_width's lower bound, per read
  for (size_t tier = 0; tier < tiers.size() && !work.empty(); ++tier) {
    FqGapTier T = tiers[tier];
    if (T.nogap && (int64_t)work.size() < c->kn.gap_nogap_min) continue;   // a small launch is bound by its longest search: one round
    if (T.nogap) ran_nogap = true;
    // Handing long searches to the wavefront-per-read kernel pays when the launch is latency-bound (few reads: its duration is
    // its longest search); a launch that fills the device several times over hides its long searches behind the others.
    // The second round of such a call is a small launch, but of hard reads only: handing them over at 1,024 pops sends tens of
    // thousands of them to the wavefront kernel (12 + 30 ms instead of 19.5 ms for the 114 k reads an on-target call of 2.1 M leaves),
    // at 2,048 / 3,072 / 4,096 pops 15 + 14 / 17 + 12 / 20 + 9 ms: the rule stays with the size of the call.
    if (!T.coop && !T.long_always && n_search > 524288) T.long_pops = tier >= n_first_tiers && n_first_tiers > 0 && !work.empty() && work.size() <= 524288 ? c->kn.gap_long_pops2 : 0u;
    next_work.clear();
    vector<int32_t> next_easy;            // what a round without gap children leaves: reads with a hit already (short full searches) go behind the others
    size_t first_chunk = 0;   // experiment (gap_split_hard): the predicted-hard reads lead the work list and get a launch of their own
    if (!T.nogap && !T.coop && c->kn.gap_split_hard > 0 && !bound_of.empty()) {
      std::stable_partition(work.begin(), work.end(), [&](int32_t sidx) { return (int64_t)bound_of[sidx] >= c->kn.gap_split_hard; });
      while (first_chunk < work.size() && (int64_t)bound_of[work[first_chunk]] >= c->kn.gap_split_hard) ++first_chunk;
    }
    for (size_t c0 = 0, step_c = 0; c0 < work.size(); c0 += step_c) {
      step_c = c0 == 0 && first_chunk > 0 ? first_chunk : chunk_reads[tier];
      const int nw = (int)std::min(step_c, work.size() - c0);
      CKM(c->d_work.ensure(nw) && c->d_wfull.ensure((size_t)nw * 2 * Lpad) && c->d_prec.ensure((size_t)nw * 2 * Ppad) && c->d_winfo.ensure(nw) && c->d_bid_end.ensure((size_t)nw * 2) && c->d_order.ensure(nw) && c->d_order_cnt.ensure(2 * FQ_ORDER_KEYS + 2) &&
          c->d_aln.ensure((size_t)nw * T.aln_cap) && c->d_naln.ensure(nw) && c->d_status.ensure(nw) && c->d_off.ensure(nw + 1));
      CKM(c->p_i32.ensure(nw));
      memcpy(c->p_i32.p, work.data() + c0, (size_t)nw * 4);
      CK(fqdev::copy_pinned(c->d_work.p, c->p_i32.p, (size_t)nw * 4, 1));
      c->stats.h2d_bytes += (size_t)nw * 4;
      FqWidthArgs wa{};
      wa.ix = ix->dev; wa.o = c->ko; wa.seq = K.dseq; wa.stride = K.dstride; wa.len_trim = K.dlen_trim; wa.read_list = K.dread_list;
      wa.work = c->d_work.p; wa.n_work = nw; wa.wfull = c->d_wfull.p; wa.wstride = Lpad;
      wa.prec = c->d_prec.p; wa.pstride = Ppad; wa.winfo = c->d_winfo.p; wa.maxdiff_lut = c->d_maxdiff.p; wa.bid_end = c->d_bid_end.p; wa.counters = c->d_counters.p;
      // device-filling launches of several contexts take turns (fqdev::device_turn_begin)
      struct Turn { int slots = 1; bool held = false; void take() { if (!held) { fqdev::device_turn_begin(slots); held = true; } } void drop() { if (held) { fqdev::device_turn_end(); held = false; } } ~Turn() { drop(); } } turn;
      turn.slots = c->kn.device_turn_slots;
      const bool big_call = c->kn.device_turns > 0 && !T.coop && c->kn.gap_nogap_min >= 0 && (int64_t)n_search >= std::max(c->kn.gap_nogap_min, c->kn.device_turn_min) &&
                            (T.nogap || !(c->kn.device_turns == 3 || (c->kn.device_turns == 2 && g_calls_in_flight.load(std::memory_order_relaxed) >= 3)));
      // (The round after the one without gap children is latency-bound -- half of the vector ALUs idle -- and shares the device well: with three or
      //  more calls in flight there is always another context's stage to run beside it, and it leaves the turn to them: 3.2 -> 3.45 / 3.3 -> 3.6 x 10^7
      //  pairs/s with three / four on-target streams.  With two streams it keeps its turn: +3 % is not worth timing the kernel under a neighbour.
      //  device_turns = 3: never takes it.)
      K.trace("  A: work list, buffers");
      if (big_call && c->kn.device_turns >= 2) turn.take();
      fqdev::time_begin(FQ_K_WIDTH);
      CK(fqdev::launch_width(wa));
      CK(fqdev::launch_order(c->d_bid_end.p, nw, (int)(n_hard > c0 ? std::min<size_t>(n_hard - c0, (size_t)nw) : 0), c->d_order.p, c->d_order_cnt.p));   // long searches first
      fqdev::time_end(FQ_K_WIDTH);
      FqGapArgs ga{};
      ga.ix = ix->dev; ga.o = c->ko; ga.o.n_buckets = T.nogap ? std::min(nb_need, o.s_gapo + o.s_mm + 1) : nb_need; /* (no-gap round: parents below s_gapo only) */ ga.n_work = nw; ga.winfo = c->d_winfo.p; ga.order = c->kn.gap_no_order ? nullptr : c->d_order.p; ga.split = c->kn.gap_no_order ? nullptr : c->d_order_cnt.p + 2 * FQ_ORDER_KEYS;
      ga.wfull = c->d_wfull.p; ga.wstride = Lpad; ga.prec = c->d_prec.p; ga.pstride = Ppad;
      ga.pool = c->d_pool.p; ga.heads = c->d_heads.p; ga.tier = T; ga.aln = c->d_aln.p; ga.n_aln = c->d_naln.p; ga.status = c->d_status.p;
      ga.counters = c->d_counters.p; ga.queue = c->d_queue.p;
      if (T.nogap) {   // a hit of b mismatches settles a read only if (b + 1) * s_mm < s_gapo (FqGapLane::finish)
        ga.bid_end = c->d_bid_end.p;
        // (-1: calls of up to 4 M searched reads, where the next round is all tail and starts its long searches at once anyway; see FqGapArgs::skip_bound)
        ga.skip_bound = c->kn.gap_skip_bound >= 0 ? (int32_t)c->kn.gap_skip_bound : o.s_mm > 0 && n_search <= ((int64_t)4 << 20) ? std::max(1, (o.s_gapo + o.s_mm - 1) / o.s_mm - 1) : 0;
      }
      // The round after the one without gap children holds the hard reads only: their lengths differ by orders of magnitude, so a
      // wavefront that waits for all 64 lanes before it refills idles most of them (28.9 -> 24.2 ms for the 228 k reads a 4.2 M-read
      // call leaves); the first round keeps whole-wavefront refill, its reads finish together (refill by 16: 23.8 -> 25.8 ms).
      ga.refill_min = ran_nogap && !T.nogap && !T.coop ? c->kn.gap_round2_refill : T.nogap ? c->kn.gap_round1_refill : 0;
      if (ran_nogap && !T.nogap && !T.coop && c->kn.gap_round2_waves > 0) ga.max_waves = c->kn.gap_round2_waves;
      if (ran_nogap && !T.nogap && !T.coop) ga.tier.lane_major = c->kn.gap_round2_lane_major;
      // stack pools are the one large per-launch allocation (lanes x pool_cap x 16 B): when the device cannot hold them for
      // as many wavefronts as it could run, run fewer (the persistent lanes simply take more reads each)
      for (;;) {
        const size_t slots = (size_t)fqdev::gap_lane_slots(ga);
        GrowScope as_is(1.0);      // (per lane, not per read)
        if (c->d_heads.ensure(slots * FQ_MAX_BUCKETS) && c->d_pool.ensure(slots * T.pool_cap)) break;
        const int waves = (int)(T.coop ? slots : slots / 64);
        if (waves <= 1) { c->err = "out of device memory for the search pools"; return FQ_ENOMEM; }
        ga.max_waves = waves / 2;
      }
      ga.pool = c->d_pool.p; ga.heads = c->d_heads.p;
      // ---- experiment (gap_pipeline_min, off by default): the first round in segments of the (sorted: hard reads first) queue, and what
      //      each segment leaves unsettled searched in full on the context's second stream while the next segment's first round runs.
      //      Measured at 8.4 M reads per call: the search stage takes 93 / 93 / 114 ms with 2 / 4 / 8 segments against 82.6 ms for the two
      //      rounds one after the other -- at this size the second round is no longer a tail but 40 % of the stage's work, its
      //      persistent wavefronts take a quarter of the wave slots for as long as they live, and the first round loses more to that
      //      (it scales with occupancy: 16 -> 12 wavefronts per CU costs it 20 %) than the overlap saves.  Kept, bit-exact and tested.
      if (big_call) turn.take();
      uint32_t n2_total = 0;                 // reads whose second round ran in the pipeline
      vector<uint32_t> seg_cnt;
      bool piped = false;
      const FqGapTier T2 = [&] { FqGapTier t = lane_tier; t.long_pops = 0; return t; }();
      if (T.nogap && c0 == 0 && (size_t)nw == work.size() && c->kn.gap_pipeline_min >= 0 && (int64_t)nw >= c->kn.gap_pipeline_min && nw >= 64 * c->kn.gap_pipeline_segs) {
        const int S = c->kn.gap_pipeline_segs;
        const size_t cap2 = std::max<size_t>(65536, (size_t)nw / 4);
        FqGapArgs g2{};
        g2.ix = ix->dev; g2.o = c->ko; g2.o.n_buckets = nb_need; g2.tier = T2; g2.n_work = (int32_t)cap2;
        const size_t slots2 = (size_t)fqdev::gap_lane_slots(g2);
        piped = c->d_work2.ensure(cap2) && c->d_cnt2.ensure((size_t)S + 1) && c->d_wfull2.ensure(cap2 * 2 * Lpad) && c->d_prec2.ensure(cap2 * 2 * Ppad) && c->d_winfo2.ensure(cap2) &&
                c->d_bid_end2.ensure(cap2 * 2) && c->d_order2.ensure(cap2) && c->d_order_cnt2.ensure(2 * FQ_ORDER_KEYS + 2) && c->d_aln2.ensure(cap2 * T2.aln_cap) && c->d_naln2.ensure(cap2) &&
                c->d_status2.ensure(cap2) && c->d_queue2.ensure(4) && c->d_heads2.ensure(slots2 * FQ_MAX_BUCKETS) && c->d_pool2.ensure(slots2 * T2.pool_cap);
        if (piped) {
          CK(fqdev::dzero(c->d_cnt2.p, ((size_t)S + 1) * 4));
          fqdev::time_begin(FQ_K_GAP);
          bool room = true;
          for (int sg = 0; sg < S; ++sg) {
            ga.seg = sg; ga.n_seg = S;
            CK(fqdev::launch_gap(ga));
            if (!room) continue;
            CK(fqdev::launch_collect(ga.order, ga.split, nw, sg, S, c->d_status.p, c->d_work.p, c->d_work2.p + n2_total, c->d_cnt2.p + sg));
            uint32_t cnt = 0;
            CKS(d2h_staged(c, &cnt, c->d_cnt2.p + sg, 4));
            CKS(sync_staged(c));
            if ((size_t)n2_total + cnt > cap2) { room = false; continue; }      // (what does not fit stays for the ordinary next round)
            seg_cnt.push_back(cnt);
            if (cnt == 0) continue;
            CK(fqdev::stream_aux(1));
            CK(fqdev::stream_fork());
            FqWidthArgs w2{};
            w2.ix = ix->dev; w2.o = c->ko; w2.seq = K.dseq; w2.stride = K.dstride; w2.len_trim = K.dlen_trim; w2.read_list = K.dread_list;
            w2.work = c->d_work2.p + n2_total; w2.n_work = (int32_t)cnt; w2.wfull = c->d_wfull2.p; w2.wstride = Lpad; w2.prec = c->d_prec2.p; w2.pstride = Ppad;
            w2.winfo = c->d_winfo2.p; w2.maxdiff_lut = c->d_maxdiff.p; w2.bid_end = c->d_bid_end2.p; w2.counters = c->d_counters.p;
            CK(fqdev::launch_width(w2));
            CK(fqdev::launch_order(c->d_bid_end2.p, (int)cnt, 0, c->d_order2.p, c->d_order_cnt2.p));
            FqGapArgs gb{};
            gb.ix = ix->dev; gb.o = c->ko; gb.o.n_buckets = nb_need; gb.n_work = (int32_t)cnt; gb.winfo = c->d_winfo2.p;
            gb.order = c->kn.gap_no_order ? nullptr : c->d_order2.p; gb.split = c->kn.gap_no_order ? nullptr : c->d_order_cnt2.p + 2 * FQ_ORDER_KEYS;
            gb.wfull = c->d_wfull2.p; gb.wstride = Lpad; gb.prec = c->d_prec2.p; gb.pstride = Ppad; gb.pool = c->d_pool2.p; gb.heads = c->d_heads2.p; gb.tier = T2;
            gb.aln = c->d_aln2.p + (size_t)n2_total * T2.aln_cap; gb.n_aln = c->d_naln2.p + n2_total; gb.status = c->d_status2.p + n2_total;
            gb.counters = c->d_counters.p; gb.queue = c->d_queue2.p; gb.refill_min = 16;
            CK(fqdev::launch_gap(gb));
            CK(fqdev::stream_aux(0));
            n2_total += cnt;
          }
          CK(fqdev::stream_join());
          fqdev::time_end(FQ_K_GAP);
        }
      }
      if (!piped) {
        fqdev::time_begin(FQ_K_GAP);
        CK(fqdev::launch_gap(ga));
        fqdev::time_end(FQ_K_GAP);
      }
      CK(fqdev::launch_scan(c->d_naln.p, c->d_off.p, (uint32_t)nw));
      CKM(c->p_u32a.ensure((size_t)nw + 2) && c->p_u32b.ensure((size_t)nw + 2));
      uint32_t *h_status = c->p_u32a.p, *h_naln = c->p_u32b.p;
      uint64_t total = 0;
      CK(fqdev::copy_pinned(h_status, c->d_status.p, (size_t)nw * 4, 0));
      CK(fqdev::copy_pinned(h_naln, c->d_naln.p, (size_t)nw * 4, 0));
      uint64_t *h_off = (uint64_t *)c->arena.alloc(((size_t)nw + 1) * 8);      // exclusive prefix sums of the hit counts (launch_scan)
      if (!h_off) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
      CK(fqdev::copy_pinned(h_off, c->d_off.p, ((size_t)nw + 1) * 8, 0));
      vector<uint8_t> h_bid;
      if (T.nogap && c->kn.gap_split_hard > 0) { h_bid.resize((size_t)nw * 2); CKS(d2h_staged(c, h_bid.data(), c->d_bid_end.p, (size_t)nw * 2)); }
      CKS(sync_staged(c));
      turn.drop();
      K.trace("  A: width + search kernels, counts D2H");
      total = h_off[nw];
      if (!h_bid.empty()) {
        bound_of.assign(n_search, 0);
        for (int w = 0; w < nw; ++w) bound_of[work[c0 + w]] = std::min(h_bid[2 * w], h_bid[2 * w + 1]);
      }
      // the packed array holds exactly the lists of the reads that completed, in work order (a failed read reports no hits): it
      // lands behind the lists of the earlier launches, and every read finds its list at the offset the device's prefix sum gave it
      const uint64_t base = c->st.aln.n;
      // (a buffer that has to grow here is given room for the rounds behind this one as well: growing again means copying what it holds)
      const uint64_t hits_need = base + total + 1, hits_room = hits_need + hits_need / 2;
      CKM(c->d_hits.ensure_keep(c->d_hits.cap >= hits_need ? hits_need : hits_room, base) && c->p_aln.ensure_keep(c->p_aln.cap >= hits_need ? hits_need : hits_room, base));
      CK(fqdev::launch_pack_aln(c->d_aln.p, c->d_naln.p, c->d_off.p, T.aln_cap, (uint32_t)nw, c->d_hits.p + base));
      CK(fqdev::launch_aln_index(c->d_work.p, c->d_status.p, c->d_off.p, c->d_naln.p, base, c->d_aoff.p, c->d_an.p, nw));   // the device's own index of the lists (the records stay there)
      CK(fqdev::copy_pinned(c->p_aln.p + base, c->d_hits.p + base, total * sizeof(FqAln), 0));
      CKS(sync_staged(c));
      c->stats.d2h_bytes += (size_t)nw * 8 + total * sizeof(FqAln);
      c->st.aln.p = c->p_aln.p; c->st.aln.n = base + total;
      K.trace("  A: hit lists D2H");
      const int32_t *wk = work.data() + c0;
      parallel_chunks((size_t)nw, K.host_threads, K.par_min, [&](size_t lo, size_t hi, int) {
        for (size_t w = lo; w < hi; ++w) {
          if (h_status[w]) continue;
          const int s = wk[w];
          K.aln_off[s] = base + h_off[w];
          K.aln_n[s] = h_naln[w];
        }
      });
      vector<char> in_round2;
      vector<int32_t> list2;
      if (n2_total) {   // the reads whose second round already ran beside the first
        list2.resize(n2_total);
        CKS(d2h_staged(c, list2.data(), c->d_work2.p, (size_t)n2_total * 4));
        CKS(sync_staged(c));
        in_round2.assign((size_t)n_search, 0);
        for (int32_t sidx : list2) in_round2[sidx] = 1;
      }
      {   // the reads this launch left unsettled, in work order: listed per range of the launch, joined in range order
        const int TT = std::max(1, K.host_threads);
        vector<vector<int32_t>> left((size_t)TT), left_easy((size_t)TT);
        vector<uint64_t> n_left((size_t)TT * 8, 0);
        const bool by_class = T.nogap != 0;
        parallel_chunks((size_t)nw, TT, K.par_min, [&](size_t lo, size_t hi, int t) {
          for (size_t w = lo; w < hi; ++w)
            if (h_status[w]) {
              ++n_left[(size_t)t * 8];
              if (in_round2.empty() || !in_round2[wk[w]]) (by_class && !(h_status[w] & FQ_SF_NOHIT) ? left_easy[t] : left[t]).push_back(wk[w]);
            }
        });
        for (int t = 0; t < TT; ++t) {
          c->stats.tier_retries += n_left[(size_t)t * 8];
          next_work.insert(next_work.end(), left[t].begin(), left[t].end());
          next_easy.insert(next_easy.end(), left_easy[t].begin(), left_easy[t].end());
        }
      }
      if (n2_total) {
        CK(fqdev::launch_scan(c->d_naln2.p, c->d_off.p, n2_total));
        uint32_t *st2 = (uint32_t *)c->arena.alloc((size_t)n2_total * 4), *na2 = (uint32_t *)c->arena.alloc((size_t)n2_total * 4);
        uint64_t *off2 = (uint64_t *)c->arena.alloc(((size_t)n2_total + 1) * 8);
        if (!st2 || !na2 || !off2) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
        CK(fqdev::copy_pinned(st2, c->d_status2.p, (size_t)n2_total * 4, 0));
        CK(fqdev::copy_pinned(na2, c->d_naln2.p, (size_t)n2_total * 4, 0));
        CK(fqdev::copy_pinned(off2, c->d_off.p, ((size_t)n2_total + 1) * 8, 0));
        CKS(sync_staged(c));
        const uint64_t tot2 = off2[n2_total];
        const uint64_t base2 = c->st.aln.n;
        CKM(c->d_hits.ensure_keep(base2 + tot2 + 1, base2) && c->p_aln.ensure_keep(base2 + tot2 + 1, base2));
        CK(fqdev::launch_pack_aln(c->d_aln2.p, c->d_naln2.p, c->d_off.p, T2.aln_cap, n2_total, c->d_hits.p + base2));
        CK(fqdev::launch_aln_index(c->d_work2.p, c->d_status2.p, c->d_off.p, c->d_naln2.p, base2, c->d_aoff.p, c->d_an.p, (int)n2_total));
        CK(fqdev::copy_pinned(c->p_aln.p + base2, c->d_hits.p + base2, tot2 * sizeof(FqAln), 0));
        CKS(sync_staged(c));
        c->stats.d2h_bytes += (size_t)n2_total * 20 + tot2 * sizeof(FqAln);
        c->st.aln.p = c->p_aln.p; c->st.aln.n = base2 + tot2;
        for (uint32_t t = 0; t < n2_total; ++t) {
          const int sidx = list2[t];
          if (st2[t]) { next_work.push_back(sidx); ++c->stats.tier_retries; continue; }
          K.aln_off[sidx] = base2 + off2[t];
          K.aln_n[sidx] = na2[t];
        }
      }
    }
    n_hard = T.nogap ? next_work.size() : 0;
    next_work.insert(next_work.end(), next_easy.begin(), next_easy.end());
    work.swap(next_work);
    K.trace("  A: round bookkeeping");
  }
  if (!work.empty()) { c->err = "gap search: exact tier exhausted its pool (internal limit)"; return FQ_ELIMIT; }
  c->stats.reads_searched += n_search;
  return FQ_OK;
}

// ---- the record stages (fq_records.h): the reads' records live on the device from here to the result arrays ------------------------
#define REC(op, n) CK(fqdev::launch_rec((op), K.ra, (int64_t)(n)))
// one 64-bit total of a device-side prefix sum (launch_scan leaves it behind the sums), valid after the next sync_staged
static int fetch_u64(fq_ctx *c, uint64_t *dst, const uint64_t *src) { return d2h_staged(c, dst, src, 8); }

// fresh records, the rows of every read's hits, the plan of the SA enumeration; what the drand48 replay needs comes back
int stage_records(Call &K) {
  fq_ctx *c = K.c;
  const fq_opts_t &o = c->o;
  const int n_surv = K.n_surv;
  const size_t N = (size_t)n_surv * 2;
  FqRecArgs &A = K.ra;
  CKM(c->d_rec.ensure(N + 1) && c->d_nocc.ensure(N + 1) && c->d_ntop.ensure(N + 1) && c->d_enum.ensure(N + 1) && c->d_qfirst.ensure(N + 1) && c->d_row0.ensure(N + 1) &&
      c->d_cls.ensure((size_t)n_surv + 1) && c->d_isz.ensure((size_t)n_surv + 1) && c->d_cigs.ensure(64));
  for (int k = 0; k < 3; ++k) CKM(c->d_cnt[k].ensure(N + 1) && c->d_scan[k].ensure(N + 2));
  A.ix = c->ix->dev; A.n_surv = n_surv; A.n_pairs = K.n; A.batch_pairs = K.B; A.packed = c->in_kind >= 2 ? 1 : 0 /* compact rows: a packed batch, a text batch */; A.single_end = o.single_end ? 1 : 0;
  A.max_occ = o.max_occ; A.multi_cap = o.single_end ? 4u : (uint32_t)std::max(o.n_multi, o.N_multi) + 1;   // (single-end: N_OCC + 1, src/BwtMapper.cpp:33, 1344)
  A.n_multi = o.n_multi; A.N_multi = o.N_multi; A.max_isize = o.max_isize; A.s_mm = o.s_mm; A.is_sw = o.is_sw;
  A.pair_list = c->d_pair_list.p; A.surv = c->d_surv.p; A.len_trim = K.dlen_trim; A.full_len = A.packed ? c->d_len_c.p : c->d_len.p;
  A.hits = c->d_hits.p; A.aoff = c->d_aoff.p; A.an = c->d_an.p; A.maxdiff_lut = c->d_maxdiff.p; A.g_log_n = c->d_glogn.p;
  A.rec = c->d_rec.p; A.nocc = c->d_nocc.p; A.ntop = c->d_ntop.p; A.enumerated = c->d_enum.p; A.qfirst = c->d_qfirst.p; A.row0 = c->d_row0.p;
  A.pq_cnt = c->d_cnt[0].p; A.prow_cnt = c->d_cnt[1].p; A.pq = c->d_scan[0].p; A.prow = c->d_scan[1].p;
  A.counters = c->d_counters.p; A.cigs = c->d_cigs.p;
  REC(FQ_ROP_INIT, N);
  REC(FQ_ROP_NOCC, N);
  REC(FQ_ROP_ENUM_PLAN, n_surv);
  CK(fqdev::launch_scan(c->d_cnt[0].p, c->d_scan[0].p, (uint32_t)n_surv));
  CK(fqdev::launch_scan(c->d_cnt[1].p, c->d_scan[1].p, (uint32_t)n_surv));
  CKM(c->p_ntop.ensure(N + 1));
  CK(fqdev::copy_pinned(c->p_ntop.p, c->d_ntop.p, N * 2, 0));
  uint64_t tot[2] = {0, 0};
  CKS(fetch_u64(c, &tot[0], c->d_scan[0].p + n_surv));
  CKS(fetch_u64(c, &tot[1], c->d_scan[1].p + n_surv));
  CKS(sync_staged(c));
  c->stats.d2h_bytes += N * 2 + 16;
  K.n_q = tot[0]; K.n_rows = tot[1];
  if (K.n_q > 0xfffffff0ull) { c->err = "more than 2^32 hits to enumerate in one call"; return FQ_ELIMIT; }
  return FQ_OK;
}

// ---- SA rows -> positions for every enumerated row (GPU) -----------------------------------------------------------------
int stage_sa_rows(Call &K) {
  fq_ctx *c = K.c;
  FqRecArgs &A = K.ra;
  const uint64_t n_q = K.n_q, rows = K.n_rows;
  CKM(c->d_qaln.ensure(n_q + 1) && c->d_qlen.ensure(n_q + 1) && c->d_qoff.ensure(n_q + 2) && c->d_pos.ensure(rows + 1) && c->d_pscratch.ensure(rows + 1));
  A.qaln = c->d_qaln.p; A.qlen = c->d_qlen.p; A.qoff = c->d_qoff.p; A.n_q = n_q; A.n_rows = rows; A.pos = c->d_pos.p; A.pscratch = c->d_pscratch.p;
  REC(FQ_ROP_ENUM_FILL, K.n_surv);
  if (rows) {
    FqSaArgs sa{};
    sa.ix = c->ix->dev; sa.aln = c->d_qaln.p; sa.aln_len = c->d_qlen.p; sa.row_off = c->d_qoff.p; sa.n_aln = (uint32_t)n_q;
    sa.n_rows = rows; sa.pos = c->d_pos.p; sa.counters = c->d_counters.p;
    fqdev::time_begin(FQ_K_SA);
    CK(fqdev::launch_sa(sa));
    fqdev::time_end(FQ_K_SA);
    c->stats.sa_rows += rows;
  }
  return FQ_OK;
}

// ---- the plan of stage B1: the state of the drand48 stream at the first read of every chunk of pairs --------
// One read depends on the others only through HOW MANY numbers they drew: one per hit that shares the best score, plus one each
// time such a hit is taken (bwase.c:29-41) -- two for a read with a single best hit, unless the first of them is exactly 0.  Counting
// the best hits of every read is the device's work (fq_rec_nocc_thread); the replay of the draws is serial, one 16-bit count per
// read (and the hit lists of the reads with several best hits).  It can run beside the SA stage.
// x -> a x + c (mod 2^48): the generator's step, its powers, and how many steps a state is away from the state 0
struct Lcg48 { uint64_t a, c; };
static constexpr uint64_t kM48 = 0xFFFFFFFFFFFFULL;
static inline Lcg48 lcg_then(const Lcg48 &f, const Lcg48 &g) { return {(g.a * f.a) & kM48, (g.a * f.c + g.c) & kM48}; }   // f, then g
static Lcg48 lcg_pow(uint64_t n) {
  Lcg48 r{1, 0}, b{0x5DEECE66DULL, 0xBULL};
  for (; n; n >>= 1) { if (n & 1) r = lcg_then(r, b); b = lcg_then(b, b); }
  return r;
}
// The smallest n >= 1 with step^n(x) == 0.  The low k bits of the state sequence have period 2^k (a = 1 mod 4, c odd), so n is found bit
// by bit: of the two candidates that agree on the bits so far, one brings the low k bits to zero.
static uint64_t steps_to_zero(uint64_t x) {
  uint64_t n = 0;
  for (int k = 1; k <= 48; ++k) {
    const Lcg48 p = lcg_pow(n);
    const uint64_t xn = (p.a * x + p.c) & kM48;
    if (xn & ((1ULL << k) - 1)) n |= 1ULL << (k - 1);
  }
  const Lcg48 p = lcg_pow(n);
  if (((p.a * x + p.c) & kM48) != 0) return 0;          // (cannot happen; the caller then replays read by read)
  return n ? n : (1ULL << 48);
}
void stageB1_plan(Call &K, uint64_t rng0) {
  fq_ctx *c = K.c;
  const size_t N = (size_t)K.n_surv * 2;
  const uint16_t *ntop = c->p_ntop.p;
  uint64_t *start = K.plan.start;
  const size_t per = 2 * FQ_RNG_CHUNK_PAIRS;
  // two steps of the generator at once: X'' = A2 X + C2 (mod 2^48); the step between them matters only when it lands on 0
  constexpr uint64_t A1 = 0x5DEECE66DULL, C1 = 0xBULL, M48 = kM48, A2 = (A1 * A1) & M48, C2 = (A1 * C1 + C1) & M48;
  // A chunk whose reads all have exactly one best hit draws 2 * per numbers -- unless one of the first draws is exactly 0, i.e. unless
  // the stream passes through the state 0 inside the chunk.  How far the state 0 is away is known (steps_to_zero, kept up to date), so
  // such a chunk is one multiply-add instead of a chain of `per` dependent ones: almost every chunk of an on-target call.
  const Lcg48 jump = lcg_pow(2 * per);
  uint64_t to_zero = steps_to_zero(rng0);
  const bool can_jump = to_zero != 0 && per == 32;
  uint64_t x = rng0;
  auto advanced = [&](uint64_t steps) { if (to_zero) to_zero = to_zero > steps ? to_zero - steps : to_zero + (1ULL << 48) - steps; };
  for (size_t c0 = 0; c0 < N; c0 += per) {
    start[c0 / per] = x;
    const size_t c1 = std::min(N, c0 + per);
    if (can_jump && c1 - c0 == per && to_zero > 2 * per) {
      uint64_t w[8];
      memcpy(w, ntop + c0, 64);
      bool ones = true;
      for (int q = 0; q < 8; ++q) ones &= w[q] == 0x0001000100010001ULL;
      if (ones) { x = (jump.a * x + jump.c) & M48; advanced(2 * per); continue; }
    }
    for (size_t idx = c0; idx < c1; ++idx) {
      const int t = ntop[idx];
      if (t == 0) continue;
      if (t == 1) {                                                          // wdt >= 1: taken unless the draw is exactly 0
        const uint64_t x1 = (A1 * x + C1) & M48;
        x = x1 == 0 ? x1 : (A2 * x + C2) & M48;
        advanced(x1 == 0 ? 1 : 2);
        continue;
      }
      int na; const FqAln *a = K.aln_of(idx, &na);
      const int best = a[0].score;                                           // (fq_main_draws, counting the draws)
      uint32_t cnt = 0;
      uint64_t steps = 0;
      for (int i = 0; i < na; ++i) {
        if (a[i].score > best) break;
        const uint32_t wdt = a[i].l - a[i].k + 1;
        ++steps;
        if (fq_rng_step(x) * (double)(uint32_t)(wdt + cnt) > (double)(int)cnt) { (void)fq_rng_step(x); ++steps; }
        cnt += wdt;
      }
      advanced(steps);
    }
  }
  K.plan.rng_end = x;
}

// ---- stage B1: main hit choice consumes the drand48 stream in read order (Q2); the choice itself is the device's ------------
int stageB1_main_hit(Call &K) {
  fq_ctx *c = K.c;
  FqRecArgs &A = K.ra;
  const int n_surv = K.n_surv;
  if (K.plan_thread.joinable()) K.plan_thread.join();
  else stageB1_plan(K, c->rng);
  c->rng = K.plan.rng_end;
  CKM(c->d_rng.ensure(K.plan.n_chunks + 1));
  CK(fqdev::copy_pinned(c->d_rng.p, K.plan.start, K.plan.n_chunks * 8, 1));
  c->stats.h2d_bytes += K.plan.n_chunks * 8;
  A.rng_start = c->d_rng.p; A.isz = c->d_isz.p; A.cls = c->d_cls.p; A.flag32 = c->d_cnt[2].p; A.off64 = c->d_scan[2].p;
  fqdev::time_begin(FQ_K_SA);       // (counted with the stage that enumerates the hits' positions)
  REC(FQ_ROP_MAIN_HIT, n_surv);
  fqdev::time_end(FQ_K_SA);
  if (!c->o.single_end) {
    CK(fqdev::launch_scan(c->d_cnt[2].p, c->d_scan[2].p, (uint32_t)n_surv));
    CKM(c->p_isz.ensure((size_t)n_surv + 1));
    CK(fqdev::copy_pinned(c->p_isz.p, c->d_isz.p, (size_t)n_surv * 4, 0));
    uint64_t n2 = 0;
    CKS(fetch_u64(c, &n2, c->d_scan[2].p + n_surv));
    CKS(sync_staged(c));
    c->stats.d2h_bytes += (size_t)n_surv * 4 + 8;
    K.n_host_pairs = (int64_t)n2;
  }
  return FQ_OK;
}

// ---- stage B2: insert size per reference batch, with the last_ii fallback chain (Q3) -----------------------
void stageB2_isize(Call &K) {
  fq_ctx *c = K.c;
  const fq_opts_t &o = c->o;
  const int n_sub = K.n_sub;
  K.iis.assign(n_sub, fq_isize_t{});
  // the inference of a reference batch depends on nothing but its pairs' samples (one per survivor pair, written by fq_main_hit_thread);
  // only the fallback chain is sequential
  vector<fq_isize_t> raw(n_sub);
  {
    std::vector<std::thread> th;
    const int T = (size_t)K.n_surv >= K.par_min ? std::min(K.host_threads, n_sub) : 1;
    const uint32_t *isz = c->p_isz.p;
    auto work = [&](int t) { for (int sb = t; sb < n_sub; sb += T) infer_isize(isz, K.sub_lo[sb], K.sub_lo[sb + 1], K.sub_max_len[sb], &raw[sb], o.ap_prior, (int64_t)c->ix->dev.fm[0].seq_len); };
    if (tl_pool && T > 1) tl_pool->run(T, work);
    else {
      for (int t = 1; t < T; ++t) th.emplace_back(work, t);
      work(0);
      for (auto &x : th) x.join();
    }
  }
  fq_isize_t prev = c->last_ii;
  for (int sb = 0; sb < n_sub; ++sb) {
    fq_isize_t ii = raw[sb];
    if (ii.avg < 0.0 && prev.avg > 0.0) ii = prev;
    if (o.force_isize) { ii.low = ii.high = 0; ii.avg = ii.std = -1.0; }
    K.iis[sb] = ii;
    prev = ii;
  }
}

// ---- the pairs the host pairs, and the (k,l) position cache, filled in pair order (serial; part of the order-dependent state, Q6) -----
// fq_main_hit_thread marks the pairs a lane does not take (an interval of 1,000 rows or more, or more than FQ_PAIR_LANE_ROWS rows):
// their list, the fields of their records the pairing reads and their rows' positions come to the host.
int stage_kl_cache(Call &K) {
  fq_ctx *c = K.c;
  FqRecArgs &A = K.ra;
  const int64_t n2 = K.n_host_pairs;
  if (!n2) return FQ_OK;
  CKM(c->d_list.ensure((size_t)n2 + 1) && c->d_greads.ensure((size_t)2 * n2) && c->d_grow0.ensure((size_t)2 * n2) && c->p_pos.ensure(K.n_rows + 1));
  A.list = c->d_list.p; A.g_reads = c->d_greads.p; A.g_row0 = c->d_grow0.p; A.n_list = (int32_t)n2;
  REC(FQ_ROP_COMPACT, K.n_surv);
  REC(FQ_ROP_PAIR_GATHER, n2);
  int32_t *list = (int32_t *)c->arena.alloc((size_t)n2 * 4);
  FqPairRead *reads = (FqPairRead *)c->arena.alloc((size_t)2 * n2 * sizeof(FqPairRead));
  uint64_t *row0 = (uint64_t *)c->arena.alloc((size_t)2 * n2 * 8);
  if (!list || !reads || !row0) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  CK(fqdev::copy_pinned(list, c->d_list.p, (size_t)n2 * 4, 0));
  CK(fqdev::copy_pinned(reads, c->d_greads.p, (size_t)2 * n2 * sizeof(FqPairRead), 0));
  CK(fqdev::copy_pinned(row0, c->d_grow0.p, (size_t)2 * n2 * 8, 0));
  CK(fqdev::copy_pinned(c->p_pos.p, c->d_pos.p, K.n_rows * 4, 0));
  CKS(sync_staged(c));
  c->stats.d2h_bytes += (size_t)n2 * (4 + 2 * sizeof(FqPairRead) + 16) + K.n_rows * 4;
  K.host_list = list; K.host_reads = reads; K.host_row0 = row0; K.h_pos = c->p_pos.p; K.have_pos = true;
  // MIN_HASH_WIDTH: the positions of an interval >= 1000 wide are those of its first requester, in pair order (Q6)
  for (int64_t t = 0; t < n2; ++t)
    for (int j = 0; j < 2; ++j) {
      int na; const FqAln *a = K.aln_of((size_t)2 * list[t] + j, &na);
      uint64_t row = row0[2 * t + j];
      for (int k = 0; k < na; ++k) {
        const uint32_t wdt = a[k].l - a[k].k + 1;
        if (wdt >= 1000) {
          auto ins = c->kl_cache.emplace((uint64_t)a[k].k << 32 | a[k].l, vector<uint32_t>());
          if (ins.second) ins.first->second.assign(K.h_pos + row, K.h_pos + row + wdt);
        }
        row += wdt;
      }
    }
  return FQ_OK;
}

// ---- stage B3: pairing (a lane per pair; the host for the pairs listed above) + XA lists ------------------------------------
int stageB3_pairing(Call &K) {
  fq_ctx *c = K.c;
  const fq_opts_t &o = c->o;
  FqRecArgs &A = K.ra;
  const int n_surv = K.n_surv;
  const size_t N = (size_t)n_surv * 2;
  if (!o.single_end) {
    // insert-size penalty tables, one per reference batch
    K.pis.assign(K.n_sub, FqPairIsize{});
    K.lut.clear();
    for (int sb = 0; sb < K.n_sub; ++sb) {
      K.pis[sb].high = K.iis[sb].high; K.pis[sb].high_bayesian = K.iis[sb].high_bayesian; K.pis[sb].lut_off = (int32_t)K.lut.size(); K.pis[sb].pad = 0;
      pair_penalty_lut(K.iis[sb], K.lut);
    }
    CKM(c->d_pisize.ensure(K.pis.size()) && c->d_plut.ensure(K.lut.size() + 1));
    CKS(h2d_staged(c, c->d_pisize.p, K.pis.data(), K.pis.size() * sizeof(FqPairIsize)));
    if (!K.lut.empty()) CKS(h2d_staged(c, c->d_plut.p, K.lut.data(), K.lut.size() * 4));
    c->stats.h2d_bytes += K.lut.size() * 4;
    A.pisize = c->d_pisize.p; A.plut = c->d_plut.p;
    fqdev::time_begin(FQ_K_SA);
    REC(FQ_ROP_PAIR, n_surv);
    fqdev::time_end(FQ_K_SA);
    if (K.n_host_pairs) {   // the same routine on the host, over the rows the (k,l) cache stands for
      const int64_t n2 = K.n_host_pairs;
      FqPairOut *outs = (FqPairOut *)c->arena.alloc((size_t)2 * n2 * sizeof(FqPairOut));
      if (!outs) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
      parallel_chunks((size_t)n2, K.host_threads, 64, [&](size_t lo, size_t hi, int) {
        vector<uint64_t> arr;
        for (size_t t = lo; t < hi; ++t) {
          const int sp = K.host_list[t];
          const FqAln *aln[2]; int na[2];
          aln[0] = K.aln_of((size_t)2 * sp, &na[0]); aln[1] = K.aln_of((size_t)2 * sp + 1, &na[1]);
          arr.clear();
          for (int j = 0; j < 2; ++j) {
            uint64_t row = K.host_row0[2 * t + j];
            for (int k = 0; k < na[j]; ++k) {
              const FqAln &q = aln[j][k];
              const uint32_t wdt = q.l - q.k + 1;
              const uint32_t *ps = K.h_pos + row;
              uint32_t np = wdt;
              if (wdt >= 1000) { const vector<uint32_t> &v = c->kl_cache.find((uint64_t)q.k << 32 | q.l)->second; ps = v.data(); np = (uint32_t)v.size(); }
              for (uint32_t z = 0; z < np; ++z) arr.push_back((uint64_t)ps[z] << 32 | (uint64_t)(k << 1) | (uint64_t)j);
              row += wdt;
            }
          }
          std::sort(arr.begin(), arr.end());
          fq_pair_sweep(aln[0], aln[1], K.host_reads + 2 * t, arr.data(), (uint32_t)arr.size(), K.pis[c->h_pair_list[sp] / K.B], K.lut.data(), c->g_log_n, o.max_isize, o.s_mm, outs + 2 * t);
        }
      });
      CKM(c->d_gout.ensure((size_t)2 * n2));
      CK(fqdev::copy_pinned(c->d_gout.p, outs, (size_t)2 * n2 * sizeof(FqPairOut), 1));
      c->stats.h2d_bytes += (size_t)2 * n2 * sizeof(FqPairOut);
      A.g_out = c->d_gout.p;
      REC(FQ_ROP_PAIR_SCATTER, n2);
    }
  }
  // XA lists of every read (the single-end mapper's alternative hits)
  A.xcnt = c->d_cnt[0].p; A.xoff = c->d_scan[0].p;
  REC(FQ_ROP_XA_COUNT, N);
  CK(fqdev::launch_scan(c->d_cnt[0].p, c->d_scan[0].p, (uint32_t)N));
  uint64_t total = 0;
  CKS(fetch_u64(c, &total, c->d_scan[0].p + N));
  CKS(sync_staged(c));
  if (total > 0xfffffff0ull) { c->err = "XA entries exceed 32-bit offsets"; return FQ_ELIMIT; }
  CKM(c->d_multi.ensure(total + 1));
  A.multi = c->d_multi.p;
  REC(FQ_ROP_XA_FILL, N);
  K.n_multi = total;
  return FQ_OK;
}

// ---- stage C: mate rescue by Smith-Waterman (bwa_paired_sw, libbwa/bwape.c:463-625) ---------------------
// Windows and candidates on the device (fq_sw_plan_thread); the candidates' records come to the host for the accept / reject
// arithmetic (libm) and go back.
int stageC_mate_sw(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  const fq_opts_t &o = c->o;
  FqRecArgs &A = K.ra;
  if (!o.is_sw) return FQ_OK;
  const int n_surv = K.n_surv;
  CKM(c->d_iis.ensure(K.n_sub) && c->d_swslot.ensure((size_t)2 * n_surv + 1));
  CKS(h2d_staged(c, c->d_iis.p, K.iis.data(), (size_t)K.n_sub * sizeof(fq_isize_t)));
  A.iis = c->d_iis.p; A.swslot = c->d_swslot.p; A.swcnt = c->d_cnt[1].p; A.swoff = c->d_scan[1].p;
  REC(FQ_ROP_SW_PLAN, n_surv);
  CK(fqdev::launch_scan(c->d_cnt[1].p, c->d_scan[1].p, (uint32_t)n_surv));
  uint64_t total = 0;
  CKS(fetch_u64(c, &total, c->d_scan[1].p + n_surv));
  CKS(sync_staged(c));
  if (!total) return FQ_OK;
  if (total > 0x7ffffff0ull) { c->err = "mate-rescue tasks exceed 31-bit indices"; return FQ_ELIMIT; }
  const size_t nt_all = (size_t)total;
  K.n_sw = total;
  CKM(c->d_swtask.ensure(nt_all) && c->d_swcand.ensure(nt_all) && c->d_list.ensure(nt_all + 1) && c->d_grec.ensure(2 * nt_all));
  A.swtask = c->d_swtask.p; A.swcand = c->d_swcand.p; A.list = c->d_list.p; A.g_rec = c->d_grec.p; A.n_list = (int32_t)nt_all;
  REC(FQ_ROP_SW_FILL, n_surv);
  REC(FQ_ROP_REC_GATHER, nt_all);
  FqSwTask *tasks = (FqSwTask *)c->arena.alloc(nt_all * sizeof(FqSwTask));
  FqSwCand *cands = (FqSwCand *)c->arena.alloc(nt_all * sizeof(FqSwCand));
  FqDRec *grec = (FqDRec *)c->arena.alloc(2 * nt_all * sizeof(FqDRec));
  if (!tasks || !cands || !grec) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  CK(fqdev::copy_pinned(tasks, c->d_swtask.p, nt_all * sizeof(FqSwTask), 0));
  CK(fqdev::copy_pinned(cands, c->d_swcand.p, nt_all * sizeof(FqSwCand), 0));
  CK(fqdev::copy_pinned(grec, c->d_grec.p, 2 * nt_all * sizeof(FqDRec), 0));
  CKS(sync_staged(c));
  c->stats.d2h_bytes += nt_all * (sizeof(FqSwTask) + sizeof(FqSwCand) + 2 * sizeof(FqDRec));
  int max_q = 0;
  for (size_t t = 0; t < nt_all; ++t) max_q = std::max(max_q, (int)grec[2 * t + cands[t].k].len);
  vector<FqSwOut> souts(nt_all);
  vector<uint16_t> scig;
  const int cig_cap = FQ_CIG_CAP;
  {
    // Windows are a few hundred bases when the insert-size estimate is sane; a poor estimate (chimeric libraries) can ask for tens
    // of thousands.  Tasks whose window fits the wavefront kernel's LDS go there; the rest run one per lane out of global scratch.
    const int kWaveMax = c->kn.sw_wave_max;
    scig.resize(nt_all * cig_cap);
    for (int big = 0; big < 2; ++big) {
      vector<int> sel;
      int RL = 1;
      const int QL = std::max(max_q, 1);
      for (size_t t = 0; t < nt_all; ++t)
        if ((tasks[t].reglen > kWaveMax) == (big != 0)) { sel.push_back((int)t); RL = std::max(RL, tasks[t].reglen); }
      if (sel.empty()) continue;
      vector<FqSwTask> sub(sel.size());
      for (size_t q = 0; q < sel.size(); ++q) sub[q] = tasks[sel[q]];
      vector<FqSwOut> sub_out(sel.size());
      vector<uint16_t> sub_cig(sel.size() * cig_cap);
      const size_t sstride = fq_dp_scratch_bytes(RL, QL);
      const size_t chunk = std::max<size_t>(64, ((size_t)6 << 30) / sstride);
      for (size_t t0 = 0; t0 < sub.size(); t0 += chunk) {
        const int nt = (int)std::min(chunk, sub.size() - t0);
        CKM(c->d_swtask.ensure(nt) && c->d_swout.ensure(nt) && c->d_cig.ensure((size_t)nt * cig_cap) && c->d_scratch.ensure((size_t)nt * sstride));
        CKS(h2d_staged(c, c->d_swtask.p, sub.data() + t0, (size_t)nt * sizeof(FqSwTask)));
        FqSwArgs a{};
        a.ix = ix->dev; a.seq = K.dseq; a.stride = K.dstride; a.len_trim = K.dlen_trim; a.task = c->d_swtask.p; a.n_task = nt;
        a.out = c->d_swout.p; a.cigar = c->d_cig.p; a.cig_cap = cig_cap; a.scratch = c->d_scratch.p; a.scratch_stride = sstride; a.RL = RL; a.QL = QL;
        fqdev::time_begin(FQ_K_SW);
        CK(big ? fqdev::launch_sw_serial(a) : fqdev::launch_sw(a));
        fqdev::time_end(FQ_K_SW);
        CKS(d2h_staged(c, sub_out.data() + t0, c->d_swout.p, (size_t)nt * sizeof(FqSwOut)));
        CKS(d2h_staged(c, sub_cig.data() + t0 * cig_cap, c->d_cig.p, (size_t)nt * cig_cap * 2));
        CKS(sync_staged(c));
      }
      for (size_t q = 0; q < sel.size(); ++q) {
        souts[sel[q]] = sub_out[q];
        memcpy(&scig[(size_t)sel[q] * cig_cap], &sub_cig[q * cig_cap], (size_t)cig_cap * 2);
      }
    }
    c->stats.sw_tasks += nt_all;
  }
  // decisions (:556-617), per pair, using the kernel outputs; an accepted CIGAR goes into the call's device CIGAR arena
  vector<uint16_t> new_cigs;
  size_t ti = 0;
  while (ti < nt_all) {
    const int sp = cands[ti].sp;
    const size_t first = ti;
    const fq_isize_t &ii = K.iis[c->h_pair_list[sp] / K.B];
    FqDRec *p[2] = {&grec[2 * first], &grec[2 * first + 1]};
    const uint16_t *cigar[2] = {nullptr, nullptr};
    int n_cigar[2] = {0, 0}, mq_adjust[2] = {255, 255}, mapQ = 0;
    int64_t beg[2] = {0, 0};
    uint32_t cnt[2] = {0, 0};
    for (; ti < nt_all && cands[ti].sp == sp; ++ti) {
      const int k = cands[ti].k;
      const FqSwOut &O = souts[ti];
      beg[k] = O.beg; cnt[k] = O.cnt;
      if (O.n_cigar > 0) { cigar[k] = scig.data() + ti * cig_cap; n_cigar[k] = O.n_cigar; }
      else beg[k] = tasks[ti].beg;
      if (cigar[k] && p[k]->type != FQ_TYPE_NO_MATCH) {
        int clip = 0;
        if ((cigar[k][0] >> 14) == FQ_OP_S) clip += cigar[k][0] & 0x3fff;
        if ((cigar[k][n_cigar[k] - 1] >> 14) == FQ_OP_S) clip += cigar[k][n_cigar[k] - 1] & 0x3fff;
        int s_old = (int)((p[k]->n_mm * 9 + p[k]->n_gapo * 13 + p[k]->n_gape * 2) / 3. * 8. + .499);
        int s_new = (int)(((cnt[k] >> 16) * 9 + (cnt[k] >> 8 & 0xff) * 13 + (cnt[k] & 0xff) * 2 + (uint32_t)clip * 3) / 3. * 8. + .499);
        s_old = (int)(s_old + -4.343 * log(ii.ap_prior / ix->l_pac));
        s_new += (int)(-4.343 * log(.5 * erfc(M_SQRT1_2 * 1.5) + .499));
        if (s_old < s_new) { mq_adjust[k] = s_new - s_old; cigar[k] = nullptr; n_cigar[k] = 0; }
        else mq_adjust[k] = s_old - s_new;
      }
    }
    int k = -1;
    if (cigar[0] && cigar[1]) { k = p[0]->mapQ < p[1]->mapQ ? 0 : 1; mapQ = abs((int)p[1]->mapQ - (int)p[0]->mapQ); }
    else if (cigar[0]) { k = 0; mapQ = p[1]->mapQ; }
    else if (cigar[1]) { k = 1; mapQ = p[0]->mapQ; }
    if (k >= 0 && (int64_t)p[k]->pos != beg[k]) {
      int tmp = (int)p[1 - k]->mapQ - (int)p[k]->mapQ / 2 - 8;
      if (tmp <= 0) tmp = 1;
      if (mapQ > tmp) mapQ = tmp;
      p[k]->mapQ = p[1 - k]->mapQ = (uint8_t)mapQ;
      p[k]->seQ = p[1 - k]->seQ = (uint8_t)(p[1 - k]->seQ < mapQ ? p[1 - k]->seQ : mapQ);
      if (p[k]->mapQ > mq_adjust[k]) p[k]->mapQ = (uint8_t)mq_adjust[k];
      if (p[k]->seQ > mq_adjust[k]) p[k]->seQ = (uint8_t)mq_adjust[k];
      p[k]->cig_off = (uint32_t)(K.cig_used + new_cigs.size()); p[k]->n_cigar = (uint16_t)n_cigar[k];
      new_cigs.insert(new_cigs.end(), cigar[k], cigar[k] + n_cigar[k]);
      p[k]->type = FQ_TYPE_MATESW; p[k]->pos = (uint32_t)beg[k]; p[k]->seQ = p[1 - k]->seQ;   // __set_fixed (:525-533)
      p[k]->strand = (uint8_t)(1 - p[1 - k]->strand);
      p[k]->n_mm = (uint8_t)((cnt[k] >> 16) & 0xff); p[k]->n_gapo = (uint8_t)(cnt[k] >> 8 & 0xff); p[k]->n_gape = (uint8_t)(cnt[k] & 0xff);
      p[k]->extra_flag |= 2; p[1 - k]->extra_flag |= 2;
    }
    for (size_t q = first + 1; q < ti; ++q) { grec[2 * q] = grec[2 * first]; grec[2 * q + 1] = grec[2 * first + 1]; }   // (a pair with two tasks is listed twice)
  }
  if (!new_cigs.empty()) {
    CKM(c->d_cigs.ensure_keep(K.cig_used + new_cigs.size() + 1, K.cig_used));
    CKS(h2d_staged(c, c->d_cigs.p + K.cig_used, new_cigs.data(), new_cigs.size() * 2));
    K.cig_used += new_cigs.size();
    A.cigs = c->d_cigs.p;
  }
  CK(fqdev::copy_pinned(c->d_grec.p, grec, 2 * nt_all * sizeof(FqDRec), 1));
  c->stats.h2d_bytes += 2 * nt_all * sizeof(FqDRec) + new_cigs.size() * 2;
  REC(FQ_ROP_REC_SCATTER, nt_all);
  return FQ_OK;
}

// ---- stage D: gapped refinement (bwa_refine_gapped, libbwa/bwase.c:339-418), MD / NM ------------------------------
int stageD_refine(Call &K) {
  fq_ctx *c = K.c;
  const fq_index *ix = c->ix;
  FqRecArgs &A = K.ra;
  const size_t N = (size_t)K.n_surv * 2;
  A.rcnt = c->d_cnt[0].p; A.roff = c->d_scan[0].p;
  CKM(c->d_refmax.ensure(2));
  CK(fqdev::dzero(c->d_refmax.p, 8));
  REC(FQ_ROP_REF_COUNT, N);
  CK(fqdev::launch_scan(c->d_cnt[0].p, c->d_scan[0].p, (uint32_t)N));
  uint64_t total = 0;
  CKS(fetch_u64(c, &total, c->d_scan[0].p + N));
  CKS(sync_staged(c));
  K.trace("  D: refine task count");
  if (total) {
    if (K.cig_used + total * FQ_CIG_CAP > 0xfffffff0ull) { c->err = "refine tasks exceed 32-bit CIGAR offsets"; return FQ_ELIMIT; }
    const size_t nt_all = (size_t)total;
    CKM(c->d_reftask.ensure(nt_all) && c->d_reftgt.ensure(nt_all) && c->d_refout.ensure(nt_all) && c->d_cigs.ensure_keep(K.cig_used + nt_all * FQ_CIG_CAP + 1, K.cig_used));
    A.reftask = c->d_reftask.p; A.reftgt = c->d_reftgt.p; A.ref_max = c->d_refmax.p; A.refout = c->d_refout.p; A.ref_cig_base = (uint32_t)K.cig_used; A.n_ref = (int32_t)nt_all;
    A.cigs = c->d_cigs.p;
    REC(FQ_ROP_REF_FILL, N);
    int32_t mx[2] = {1, 1};
    CKS(d2h_staged(c, mx, c->d_refmax.p, 8));
    CKS(sync_staged(c));
    const int max_ref = std::max(1, mx[0]), max_q = std::max(1, mx[1]);
    const size_t sstride = fq_dp_scratch_bytes(max_ref, max_q);
    const size_t chunk = std::max<size_t>(64, ((size_t)6 << 30) / sstride);
    for (size_t t0 = 0; t0 < nt_all; t0 += chunk) {
      const int nt = (int)std::min(chunk, nt_all - t0);
      CKM(c->d_scratch.ensure((size_t)nt * sstride));
      FqRefineArgs a{};
      a.ix = ix->dev; a.seq = K.dseq; a.stride = K.dstride; a.len_trim = K.dlen_trim; a.task = c->d_reftask.p + t0; a.n_task = nt;
      a.out = c->d_refout.p + t0; a.cigar = c->d_cigs.p + K.cig_used + t0 * FQ_CIG_CAP; a.cig_cap = FQ_CIG_CAP; a.scratch = c->d_scratch.p; a.scratch_stride = sstride; a.RL = max_ref; a.QL = max_q;
      fqdev::time_begin(FQ_K_REFINE);
      CK(fqdev::launch_refine(a));
      fqdev::time_end(FQ_K_REFINE);
    }
    REC(FQ_ROP_REF_APPLY, nt_all);
    K.cig_used += nt_all * FQ_CIG_CAP;
    K.n_ref = total;
    c->stats.refine_tasks += nt_all;
  }
  // MD / NM for every mapped read (bwa_cal_md1), straight from the records into one slot per read
  const int md_cap = 3 * (K.max_len_all + 8) + 32;
  CKM(c->d_md.ensure(N * (size_t)md_cap + 1));
  A.seq = K.dseq; A.stride = K.dstride; A.md = c->d_md.p; A.md_cap = md_cap; A.cigs = c->d_cigs.p;
  fqdev::time_begin(FQ_K_REFINE);
  A.mdmask = nullptr;
  if (A.packed && (K.dstride & 15) == 0 && (int64_t)N >= c->kn.md_mask_min) {   // compact rows (device row = record index): the rows compared piece-wise first
    const size_t per_row = (size_t)K.dstride >> 4;
    if (N * per_row < 0x7fffffffull) {
      CKM(c->d_mdmask.ensure(N * per_row + 8));
      A.mdmask = c->d_mdmask.p;
      REC(FQ_ROP_MD_MASK, N * per_row);
    }
  }
  REC(FQ_ROP_MD, N);
  fqdev::time_end(FQ_K_REFINE);
  return FQ_OK;
}

// the records as they are now, in the host's vocabulary (debug: the stage dumps of the parity tests)
int snapshot_records(Call &K, vector<FqRead> &dst) {
  fq_ctx *c = K.c;
  const size_t N = (size_t)K.n_surv * 2;
  vector<FqDRec> rec(N);
  vector<fq_multi_t> multi((size_t)K.n_multi);
  vector<uint16_t> cigs((size_t)K.cig_used);
  CKS(d2h_staged(c, rec.data(), c->d_rec.p, N * sizeof(FqDRec)));
  CKS(d2h_staged(c, multi.data(), c->d_multi.p, multi.size() * sizeof(fq_multi_t)));
  CKS(d2h_staged(c, cigs.data(), c->d_cigs.p, cigs.size() * 2));
  CKS(sync_staged(c));
  dst.assign(N, FqRead());
  for (size_t i = 0; i < N; ++i) {
    const FqDRec &s = rec[i];
    FqRead &p = dst[i];
    p.r = (int)(i & 1) * K.n + c->h_pair_list[i >> 1];
    p.score = s.score; p.sa = s.sa; p.pos = s.pos; p.c1 = s.c1; p.c2 = s.c2; p.len = s.len; p.full_len = s.full_len; p.clip_len = s.clip_len;
    p.nm = (int16_t)s.nm; p.filtered = s.filtered; p.type = s.type; p.strand = s.strand; p.extra_flag = s.extra_flag;
    p.n_mm = s.n_mm; p.n_gapo = s.n_gapo; p.n_gape = s.n_gape; p.mapQ = s.mapQ; p.seQ = s.seQ; p.revived = s.revived != 0;
    p.cigar.assign(cigs.data() + (s.n_cigar ? s.cig_off : 0), cigs.data() + (s.n_cigar ? s.cig_off : 0) + s.n_cigar);
    for (uint32_t j = 0; j < s.n_multi; ++j) {
      const fq_multi_t &m = multi[s.multi_off + j];
      FqMulti q;
      q.pos = m.pos; q.gap = m.gap; q.mm = m.mm; q.strand = m.strand;
      if (m.n_cigar) q.cigar.assign(cigs.data() + m.cigar_off, cigs.data() + m.cigar_off + m.n_cigar);
      p.multi.push_back(q);
    }
  }
  return FQ_OK;
}

// ---- the consumers on the device (fq_emit.h): the SAM text of the call, formatted from the result arrays where they lie ----------------
// The reads' qualities and names are resident for a text batch (fq_align_text); for ASCII and packed batches the surviving reads' are gathered
// on the host and uploaded compact (row = 2 * survivor + end).
int emit_reads(Call &K, const uint8_t **qual, int *qual_stride, const char **names, int *name_stride) {
  fq_ctx *c = K.c;
  const size_t N = (size_t)K.n_surv * 2;
  if (c->in_kind == 3) { *qual = c->d_pqual.p; *qual_stride = c->c_stride; *names = c->d_cnames.p; *name_stride = c->c_name_stride; return FQ_OK; }
  const FqHostReads hb = fq_ctx_host_reads(c);
  if (N && !hb.has_qual()) { c->err = "SAM text on the device: the batch carries no qualities"; return FQ_EINVAL; }
  const int n = K.n;
  const bool se = c->o.single_end != 0;
  int max_len = 1, max_name = 1;
  std::vector<int> tl((size_t)std::max(1, K.host_threads), 1), tn((size_t)std::max(1, K.host_threads), 1);
  parallel_chunks(N, K.host_threads, K.par_min, [&](size_t lo, size_t hi, int t) {
    int ml = 1, mn = 1;
    for (size_t i = lo; i < hi; ++i) {
      const int e = (int)(i & 1), pair = c->h_pair_list[i >> 1];
      if (se && e) continue;
      ml = std::max(ml, hb.len((size_t)e * n + pair));
      if (hb.has_names()) mn = std::max(mn, (int)strnlen(hb.name_of(pair, e), (size_t)hb.name_stride));
    }
    tl[(size_t)t] = ml; tn[(size_t)t] = mn;
  });
  for (size_t t = 0; t < tl.size(); ++t) { max_len = std::max(max_len, tl[t]); max_name = std::max(max_name, tn[t]); }
  const int qs = (max_len + 15) & ~15, ns = (max_name + 1 + 7) & ~7;
  CKM(c->p_equal.ensure(N * qs + 64) && c->d_equal.ensure(N * qs + 64) && c->p_enames.ensure(N * ns + 64) && c->d_enames.ensure(N * ns + 64));
  parallel_chunks(N, K.host_threads, K.par_min, [&](size_t lo, size_t hi, int) {
    for (size_t i = lo; i < hi; ++i) {
      const int e = (int)(i & 1), pair = c->h_pair_list[i >> 1];
      uint8_t *q = c->p_equal.p + i * (size_t)qs;
      char *nm = c->p_enames.p + i * (size_t)ns;
      memset(nm, 0, (size_t)ns);
      if (se && e) { memset(q, 0, (size_t)qs); continue; }
      const size_t r = (size_t)e * n + pair;
      const int l = hb.len(r);
      memcpy(q, hb.qual(r), (size_t)l);
      memset(q + l, 0, (size_t)(qs - l));
      if (hb.has_names()) { const char *src = hb.name_of(pair, e); memcpy(nm, src, strnlen(src, (size_t)hb.name_stride)); }
      else nm[0] = '*';
    }
  });
  CK(fqdev::copy_pinned(c->d_equal.p, c->p_equal.p, N * qs, 1));
  CK(fqdev::copy_pinned(c->d_enames.p, c->p_enames.p, N * ns, 1));
  c->stats.h2d_bytes += N * ((size_t)qs + (size_t)ns);
  *qual = c->d_equal.p; *qual_stride = qs; *names = c->d_enames.p; *name_stride = ns;
  return FQ_OK;
}
// the call's records, reads and names as the consumers' kernels see them (qualities and names uploaded once per call)
int emit_args(Call &K, FqSamArgs &a) {
  fq_ctx *c = K.c;
  if (!K.emit_ready) {
    FqSamArgs &e = K.emit;
    e = FqSamArgs{};
    e.cg = c->ix->dev_contigs;
    e.n_surv = K.n_surv; e.n_pairs = K.n; e.packed = K.ra.packed; e.single_end = c->o.single_end; e.mode = c->o.mode; e.max_top2 = c->o.max_top2;
    e.pair_list = c->d_pair_list.p;
    e.rec = c->d_orec.p; e.cigar = c->d_ocig.p; e.md = c->d_omd.p; e.multi = c->d_omulti.p;
    e.seq = K.ra.seq; e.stride = K.ra.stride;
    CKS(emit_reads(K, &e.qual, &e.qual_stride, &e.names, &e.name_stride));
    K.emit_ready = true;
  }
  a = K.emit;
  return FQ_OK;
}
// The consumers' kernels run in two steps.  emit_measure, inside the call: every record's line / record length and every pair's decisions, the prefix
// sums that place them, ONE wait for the totals.  emit_fill, as the call's last act: the buffers, then the kernels that write -- SAM text, BAM records
// and their BGZF members, .InsertSizeTable lines, pileup entries, the per-base sums -- are only ENQUEUED: the call returns, and they run beside the next
// call's search kernels on another context.  Whoever fetches what they leave waits for them first (fq_ctx_emit_wait).
int emit_measure(Call &K) {
  fq_ctx *c = K.c;
  const size_t N = (size_t)K.n_surv * 2, P = (size_t)K.n_surv;
  EmitPlan &E = K.plan_emit;
  E = EmitPlan();
  c->sam_bytes = 0; c->sam_ready = false;
  c->bam_out = FqBamCallOut(); c->bam_out.owner = c->bam;
  c->qc_out = FqQcCallOut(); c->qc_out.owner = c->qc; c->qc_out.n_surv = K.n_surv;
  if (!N) return FQ_OK;
  if (c->emit_flags & FQ_EMIT_SAM) {
    CKS(emit_args(K, E.sam));
    CKM(c->d_samlen.ensure(N + 1) && c->d_samoff.ensure(N + 2) && c->d_sammeta.ensure(N + 1));
    E.sam.len = c->d_samlen.p; E.sam.off = c->d_samoff.p; E.sam.meta = c->d_sammeta.p;
    CK(fqdev::launch_sam(FQ_EOP_SAM_LEN, E.sam, (int64_t)N));
    CK(fqdev::launch_scan(c->d_samlen.p, c->d_samoff.p, (uint32_t)N));
    CKS(fetch_u64(c, &E.sam_total, c->d_samoff.p + N));
  }
  if (c->bam) {
    CKS(emit_args(K, E.bam.s));
    { const int rc = fq_bam_device_prepare(c->bam, &E.bam); if (rc) { c->err = "the BAM writer could not stage its tables on the device"; return rc; } }
    CKM(c->d_bamlen.ensure(N + 1) && c->d_bamoff.ensure(N + 2) && c->d_bammeta.ensure(N + 1));
    E.bam.len = c->d_bamlen.p; E.bam.off = c->d_bamoff.p; E.bam.meta = c->d_bammeta.p; E.bam.split = 1;
    CK(fqdev::launch_bam(FQ_EOP_BAM_LEN, E.bam, (int64_t)N));
    CK(fqdev::launch_scan(c->d_bamlen.p, c->d_bamoff.p, (uint32_t)N));
    CKS(fetch_u64(c, &E.bam_total, c->d_bamoff.p + N));
  }
  if (c->qc) {
    FqQcArgs &a = E.qc;
    CKS(emit_args(K, a.s));
    a.ix = c->ix->dev;
    K.gate_enter();
    { const int rc = fq_qc_device_prepare(c->qc, &a, K.n_surv); if (rc) { c->err = std::string("the QC consumer could not take the call: ") + fq_qc_last_error(c->qc); return rc; } }
    const size_t NC = (size_t)FQ_C_STRIPES * FQ_C_STRIDE;
    CKM(c->d_qadded.ensure(N + 1) && c->d_istlen.ensure(P + 1) && c->d_istoff.ensure(P + 2) && c->d_ptcnt.ensure(N + 1) && c->d_ptoff.ensure(N + 2) && c->d_qcnt.ensure(NC) && c->p_qcnt.ensure(NC));
    if (a.shard) { CKM(c->d_dupkey.ensure(P + 1) && c->p_dupkey.ensure(P + 1)); }
    CK(fqdev::dzero(c->d_qcnt.p, NC * 8));
    a.counters = c->d_qcnt.p; a.added = c->d_qadded.p; a.ist_len = c->d_istlen.p; a.ist_off = c->d_istoff.p; a.pt_cnt = c->d_ptcnt.p; a.pt_off = c->d_ptoff.p;
    a.dup_key = a.shard ? c->d_dupkey.p : nullptr;
    CK(fqdev::launch_qc(FQ_QOP_PAIR, a, (int64_t)P));
    CK(fqdev::launch_scan(c->d_istlen.p, c->d_istoff.p, (uint32_t)P));
    CK(fqdev::launch_scan(c->d_ptcnt.p, c->d_ptoff.p, (uint32_t)N));
    CKS(fetch_u64(c, &E.ist_total, c->d_istoff.p + P));
    CKS(fetch_u64(c, &E.pt_total, c->d_ptoff.p + N));
  }
  CKS(sync_staged(c));
  K.gate_leave();
  K.trace("consumers: lengths, decisions, prefix sums");
  return FQ_OK;
}
int emit_fill(Call &K) {
  fq_ctx *c = K.c;
  const size_t N = (size_t)K.n_surv * 2, P = (size_t)K.n_surv;
  EmitPlan &E = K.plan_emit;
  if (!N) {
    if (c->emit_flags & FQ_EMIT_SAM) c->sam_ready = true;
    c->bam_out.ready = c->bam != nullptr; c->qc_out.ready = c->qc != nullptr;
    return FQ_OK;
  }
  std::lock_guard<std::mutex> lk(c->emit_mu);
  if (c->emit_flags & FQ_EMIT_SAM) {
    CKM(c->d_samtext.ensure_roomy(E.sam_total + 64));
    E.sam.text = c->d_samtext.p;
    static const bool body_split = [] { const char *e = getenv("FASTQUICK_SAM_BODY"); return !(e && *e == '0'); }();
    E.sam.split = body_split ? 1 : 0;
    CK(fqdev::launch_sam(FQ_EOP_SAM_FILL, E.sam, (int64_t)N));      // heads and tags: a thread per record
    if (body_split) CK(fqdev::launch_sam(FQ_EOP_SAM_BODY, E.sam, (int64_t)N));      // SEQ / QUAL runs: a thread per sixteen bytes
    c->sam_bytes = E.sam_total;
    c->sam_ready = true;
  }
  if (c->bam) {
    const uint64_t total = E.bam_total;
    CKM(c->d_bamrec.ensure_roomy(total + 64));
    E.bam.out = c->d_bamrec.p;
    CK(fqdev::launch_bam(FQ_EOP_BAM_FILL, E.bam, (int64_t)N));      // fixed fields, name, CIGAR, tags: a thread per record
    CK(fqdev::launch_bam(FQ_EOP_BAM_BODY, E.bam, (int64_t)N));      // packed bases and qualities: a thread per sixteen bytes
    c->bam_out.bytes = total;
    c->emit_nb = 0;
    if (total && fq_bam_wants_members(c->bam)) {
      // the writer has a file: the records leave the device as finished BGZF members (fq_deflate.h: a wavefront per block of the record stream),
      // packed behind each other into a buffer sized for the worst case; their size comes back with the wait
      const uint32_t nb = (uint32_t)((total + FQD_BLOCK - 1) / FQD_BLOCK);
      CKM(c->d_zstage.ensure_roomy((size_t)nb * FQD_SLOT) && c->d_zsize.ensure((size_t)nb + 1) && c->d_zoff.ensure((size_t)nb + 2) && c->d_bamz.ensure_roomy((size_t)nb * FQD_SLOT) && c->p_ztotal.ensure(8));
      FqDeflateArgs z{c->d_bamrec.p, total, c->d_zstage.p, c->d_zsize.p, fqdev::crc_const(), nb};
      if (!z.crc) { c->err = std::string("BGZF on the device: ") + fqdev::last_error(); return FQ_ENODEV; }
      CK(fqdev::launch_deflate(z));
      CK(fqdev::launch_scan(c->d_zsize.p, c->d_zoff.p, nb));
      FqDeflatePackArgs pk{c->d_zstage.p, c->d_zsize.p, c->d_zoff.p, c->d_bamz.p, nb};
      CK(fqdev::launch_deflate_pack(pk));
      CK(fqdev::copy_pinned(c->p_ztotal.p, c->d_zoff.p + nb, 8, 0));
      c->emit_nb = nb;
    }
    c->bam_out.ready = true;
  }
  if (c->qc) {
    FqQcArgs &a = E.qc;
    const size_t NC = (size_t)FQ_C_STRIPES * FQ_C_STRIDE;
    CKM(c->d_isttext.ensure_roomy(E.ist_total + 64) && c->d_pile.ensure_roomy(E.pt_total + 1));
    a.ist_text = c->d_isttext.p; a.pt = c->d_pile.p;
    CK(fqdev::launch_qc(FQ_QOP_IST_FILL, a, (int64_t)P));
    CK(fqdev::launch_qc(FQ_QOP_PILE_FILL, a, (int64_t)N));
    CK(fqdev::launch_qc(FQ_QOP_BASE, a, (int64_t)N));
    // (the lines and the pileup entries stay in HBM: the consumer's host side fetches them in slices beside the next call -- fq_ctx_qc_stream)
    CK(fqdev::copy_pinned(c->p_qcnt.p, c->d_qcnt.p, NC * 8, 0));
    if (a.shard) CK(fqdev::copy_pinned(c->p_dupkey.p, c->d_dupkey.p, P * 8, 0));
    c->stats.d2h_bytes += NC * 8 + (a.shard ? P * 8 : 0);
    c->qc_out.ist_bytes = E.ist_total; c->qc_out.n_pile = E.pt_total; c->qc_out.dup_key = a.shard ? c->p_dupkey.p : nullptr;
    c->qc_out.ready = true;
  }
  CK(fqdev::copy_flush_now());
  c->emit_pending = true;
  static const bool emit_sync = [] { const char *e = getenv("FASTQUICK_EMIT_SYNC"); return e && *e && *e != '0'; }();   // A/B: the call waits for its consumers' kernels
  if (emit_sync) CKS(sync_staged(c));
  K.trace("consumers: fill kernels enqueued");
  return FQ_OK;
}

// ---- the C-ABI result arrays (written on the device, landed in pinned host memory), the work counters ---------------------------
int stage_finish(Call &K, fq_result_batch_t *out) {
  fq_ctx *c = K.c;
  FqBatchState &S = c->st;
  FqRecArgs &A = K.ra;
  const int n_sub = K.n_sub, n_surv = K.n_surv;
  const size_t N = (size_t)n_surv * 2;
  S.isize_sub = K.iis;
  S.isize = K.iis[n_sub - 1];
  uint64_t cc = 0, mm = 0, xx = 0;
  if (N) {
    A.fc_cnt = c->d_cnt[0].p; A.fm_cnt = c->d_cnt[1].p; A.fx_cnt = c->d_cnt[2].p;
    A.fc_off = c->d_scan[0].p; A.fm_off = c->d_scan[1].p; A.fx_off = c->d_scan[2].p;
    REC(FQ_ROP_FLAT_COUNT, N);
    for (int k = 0; k < 3; ++k) CK(fqdev::launch_scan(c->d_cnt[k].p, c->d_scan[k].p, (uint32_t)N));
    CKS(fetch_u64(c, &cc, c->d_scan[0].p + N));
    CKS(fetch_u64(c, &mm, c->d_scan[1].p + N));
    CKS(fetch_u64(c, &xx, c->d_scan[2].p + N));
    CKS(sync_staged(c));
    if (cc > 0xffffffffull || mm > 0xfffffffeull || xx > 0xffffffffull) { c->err = "result arenas exceed 32-bit offsets"; return FQ_ELIMIT; }
  }
  // FQ_EMIT_DEVICE_ONLY: the result arrays stay where the consumers' kernels read them (0.55 GB per 4.2 M-pair on-target call stays off PCIe)
  const bool host_arrays = !(c->emit_flags & FQ_EMIT_DEVICE_ONLY) || c->debug;
  CKM(c->d_orec.ensure(N + 1) && c->d_ocig.ensure(cc + 1) && c->d_omd.ensure(mm + 1) && c->d_omulti.ensure(xx + 1));
  if (host_arrays) CKM(c->p_orec.ensure(N + 1) && c->p_ocig.ensure(cc + 1) && c->p_omd.ensure(mm + 1) && c->p_omulti.ensure(xx + 1));
  if (N) {
    A.o_rec = c->d_orec.p; A.o_cigar = c->d_ocig.p; A.o_md = c->d_omd.p; A.o_multi = c->d_omulti.p;
    REC(FQ_ROP_FLAT_FILL, N);
    if ((c->emit_flags & FQ_EMIT_SAM) || c->qc || c->bam) CKS(emit_measure(K));
    if (host_arrays) {
      CK(fqdev::copy_pinned(c->p_orec.p, c->d_orec.p, N * sizeof(fq_result_t), 0));
      CK(fqdev::copy_pinned(c->p_ocig.p, c->d_ocig.p, cc * 2, 0));
      CK(fqdev::copy_pinned(c->p_omd.p, c->d_omd.p, mm, 0));
      CK(fqdev::copy_pinned(c->p_omulti.p, c->d_omulti.p, xx * sizeof(fq_multi_t), 0));
      c->stats.d2h_bytes += N * sizeof(fq_result_t) + cc * 2 + mm + xx * sizeof(fq_multi_t);
    }
  }
  uint64_t cnt[FQ_C_COUNT];
  c->h_counters.resize((size_t)FQ_C_STRIPES * FQ_C_STRIDE);
  CKS(d2h_staged(c, c->h_counters.data(), c->d_counters.p, c->h_counters.size() * 8));
  CKS(sync_staged(c));
  fold_counters(c->h_counters.data(), cnt);
  CK(fqdev::dzero(c->d_counters.p, c->h_counters.size() * 8));
  if (!N && ((c->emit_flags & FQ_EMIT_SAM) || c->qc || c->bam)) CKS(emit_measure(K));
  if (host_arrays) {
    if (!cc) c->p_ocig.p[0] = 0;
    if (!mm) c->p_omd.p[0] = 0;
    if (!xx) c->p_omulti.p[0] = fq_multi_t{};
  }
  K.trace("result arrays D2H");
  if (cnt[FQ_C_ERR_DRAW0]) {
    c->err = "the drand48 stream drew exactly 0 for the first best hit of a read (once in 2^48 draws): bwa_aln2seq_core then takes no hit and the reference's "
             "record keeps the SA row of the read slot's previous occupant (libbwa/bwase.c:29-41) -- its result is undefined, and so this call is refused";
    return FQ_ELIMIT;
  }
  if (cnt[FQ_C_ERR_CIGAR]) { c->err = "refine: CIGAR longer than the device slot"; return FQ_ELIMIT; }
  if (cnt[FQ_C_ERR_MD]) { c->err = "MD string longer than the device slot"; return FQ_ELIMIT; }
  if (host_arrays) { S.rec = c->p_orec.p; S.cigar = c->p_ocig.p; S.md = c->p_omd.p; S.multi = c->p_omulti.p; }
  else { S.rec = nullptr; S.cigar = nullptr; S.md = nullptr; S.multi = nullptr; }
  S.n_both_unmapped = (int)cnt[FQ_C_UNMAPPED];
  out->n_survivors = n_surv;
  out->n_both_filtered = K.n - n_surv;
  out->n_both_unmapped = S.n_both_unmapped;
  out->pair_idx = S.pair_idx;
  out->rec = S.rec;
  out->cigar = S.cigar;
  out->md = S.md;
  out->multi = S.multi;
  out->isize = K.iis[n_sub - 1];
  out->n_sub = n_sub;
  out->isize_sub = S.isize_sub.data();
  out->n_bases = c->n_bases_in;
  fqdev::time_collect(c->stats.kernel_ms, c->stats.kernel_launches, FQ_K_COUNT);
  c->stats.occ_block_touches += cnt[FQ_C_OCC_WIDTH] + cnt[FQ_C_OCC_GAP] + cnt[FQ_C_OCC_SA];
  c->stats.gap_occ_touches += cnt[FQ_C_OCC_GAP];
  c->stats.width_occ_touches += cnt[FQ_C_OCC_WIDTH];
  c->stats.md_reads += cnt[FQ_C_MD_READS];
  c->stats.host_pairs += (uint64_t)K.n_host_pairs;
  c->stats.gap_nogap_touches += cnt[FQ_C_OCC_NOGAP];
  c->stats.filter_probes += cnt[FQ_C_PROBES];
  c->stats.stack_pops += cnt[FQ_C_POPS];
  c->stats.stack_pushes += cnt[FQ_C_PUSHES];
  if (cnt[FQ_C_MAXPOPS] > c->stats.max_pops_per_read) c->stats.max_pops_per_read = cnt[FQ_C_MAXPOPS];
  c->stats.reads_over_4k_pops += cnt[FQ_C_POPS_GT4K];
  if (cnt[FQ_C_MAXTRIPS] > c->stats.max_wave_trips) c->stats.max_wave_trips = cnt[FQ_C_MAXTRIPS];
  c->stats.wave_trips += cnt[FQ_C_SUMTRIPS];
  c->stats.lane_trips += cnt[FQ_C_LANETRIPS];
  c->stats.sa_rows += cnt[FQ_C_SA_DIRECT];
  c->stats.pairs_on_device += cnt[FQ_C_PAIRS_DEV];
  for (int k = 0; k < 16; ++k) c->stats.dbg[k] += cnt[FQ_C_DBG0 + k];
  c->stats.pairs += K.n;
  // host_ms_*: the host's own time between the SA stage and the result arrays -- the order-dependent part (drand48 replay, insert sizes, (k,l)
  // cache) and what follows it -- WITHOUT the waits for the device inside those sections (round 3 and before: host work with a few waits
  // in it; since the records live on the device: waits with a little host work in them, and under sixteen streams the waits are queueing
  // behind the other streams' kernels).  device_wait_ms: all the time the call's thread slept in waits; host_cpu_ms: the CPU time it used.
  c->stats.host_ms_serial += (K.t_serial1 - K.t_host0) - (K.w_serial1 - K.w_host0);
  c->stats.host_ms_pair += (K.t_host1 - K.t_serial1) - (K.w_host1 - K.w_serial1);
  c->stats.host_ms_total += (K.t_host1 - K.t_host0) - (K.w_host1 - K.w_host0);
  c->stats.wall_ms_total += now_ms() - K.t_wall0;
  c->stats.device_wait_ms += c->wait_ms - K.w_call0;
  { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); c->stats.host_cpu_ms += 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec - K.cpu_call0; }
  K.trace("counters+timers");
  if ((c->emit_flags & FQ_EMIT_SAM) || c->qc || c->bam) CKS(emit_fill(K));
  return FQ_OK;
}
#undef REC

int run_call_stages(fq_ctx *c, fq_result_batch_t *out);
int run_call(fq_ctx *c, fq_result_batch_t *out) {
  c->serial_open = c->serial_done = false;
  const int rc = run_call_stages(c, out);
  if (rc && (c->before_serial || c->after_serial)) {
    // a sharded stream: the owner of the next batch waits for this call's state.  It gets one -- marked broken -- whether the call
    // failed before its order-dependent part, inside it or after it (then the next owner already has a good state, and the mark
    // reaches the ranks with the state after that)
    c->stream_broken = true;
    if (!c->serial_open && !c->serial_done && c->before_serial) c->before_serial(c->hook_user);
    if (!c->serial_done && c->after_serial) c->after_serial(c->hook_user);
  }
  if (rc) {   // stage_finish zeroes the device counters after it has read them: a call that ends early must not leave its counts (lengths
              // out of range, bases, work counters) to the next one
    fqdev::copy_discard();
    (void)fqdev::stream_aux(0);
    (void)fqdev::dzero(c->d_counters.p, (size_t)FQ_C_STRIPES * FQ_C_STRIDE * 8);
    (void)fqdev::sync();
    // uploads from the caller's pinned batch may still be in flight on the shared copy stream (a prefetched head):
    // the caller is free to repack or free that storage as soon as the failed call has returned
    (void)fqdev::copy_wait(0);
    (void)fqdev::copy_wait(1);
  }
  return rc;
}
int run_call_stages(fq_ctx *c, fq_result_batch_t *out) {
  CallInFlight in_flight(c->ix);
  Call K(c);
  K.t_trace = K.t_wall0 = now_ms();
  K.w_call0 = c->wait_ms;
  { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); K.cpu_call0 = 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; }
  NodePin pin;
  struct PoolScope { FqWorkPool *prev; explicit PoolScope(FqWorkPool *p) : prev(tl_pool) { tl_pool = p; } ~PoolScope() { tl_pool = prev; } } pool_scope(&c->pool);
  (void)fqdev::stream_aux(0);      // (an error return may have left the context on its second stream)
  // (a short first call of a long stream sizes what it allocates for the calls behind it: tl_grow)
  const double grow = c->in_kind == 3 && c->tb && c->n_pairs > 0 && std::min<double>((double)c->max_pairs, (double)c->tb->pairs_behind) > (double)c->n_pairs
                          ? std::min(16.0, std::min<double>((double)c->max_pairs, (double)c->tb->pairs_behind) / (double)c->n_pairs) : 1.0;
  GrowScope grow_scope(grow);
  c->retired.next_call();
  RetiredScope retired_scope(&c->retired);
  if (c->kn.trace) { fprintf(stderr, "[fq]   arena: %zu blocks, last call used %zu bytes:", c->arena.blocks.size(), c->arena.total); for (auto &b : c->arena.blocks) fprintf(stderr, " %zu", b.cap); fprintf(stderr, "\n"); }
  c->arena.reset();
  const fq_opts_t &o = c->o;
  FqBatchState &S = c->st;
  S.clear();
  S.n_pairs = c->n_pairs;
  memset(out, 0, sizeof *out);
  out->n_pairs = c->n_pairs;
  if (c->n_pairs == 0) {
    // an empty batch is a call like any other to the consumers: empty text, nothing counted, and not the previous call's outputs
    const int rc0 = emit_measure(K);
    return rc0 ? rc0 : emit_fill(K);
  }
  K.n = c->n_pairs; K.n2 = 2 * K.n; K.B = o.batch_pairs; K.n_sub = (K.n + K.B - 1) / K.B;
  K.par_min = c->kn.host_par_min;
  K.host_threads = c->kn.host_threads >= 0 ? c->kn.host_threads : o.host_threads > 0 ? o.host_threads : default_host_threads(c->ix);
  int rc = c->in_kind == 3 ? stage0_text(K) : c->in_kind == 2 ? stage0_packed(K) : stage0_ascii(K);
  if (rc) return rc;
  S.n_surv = K.n_surv;
  S.batch_pairs = K.B;
  S.sub_lo = K.sub_lo;
  S.pair_idx = c->h_pair_list;
  S.surv = c->h_surv;
  K.trace("stage0 prep+compact");
  if ((rc = stageA_search(K))) return rc;
  K.trace("stageA width+gap");
  if ((rc = stage_records(K))) return rc;
  K.trace("records, SA plan, best-hit counts D2H");
  // where every chunk of pairs enters the drand48 stream: drawn up beside the SA stage
  // (a sharded stream's hook may still hand this context the stream's state: then the plan waits for it)
  K.plan.n_chunks = ((size_t)K.n_surv + FQ_RNG_CHUNK_PAIRS - 1) / FQ_RNG_CHUNK_PAIRS;
  K.plan.start = (uint64_t *)c->arena.alloc((K.plan.n_chunks + 1) * 8);
  if (!K.plan.start) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  if (!c->before_serial && (size_t)K.n_surv * 2 >= K.par_min) K.plan_thread = std::thread([&K, c] { stageB1_plan(K, c->rng); });
  if ((rc = stage_sa_rows(K))) return rc;
  K.trace("SA enumerate+kernel");
  K.t_host0 = now_ms(); K.w_host0 = c->wait_ms;
  // ---- the order-dependent part of the call: drand48 stream, last_ii chain, (k,l) cache.  A stream sharded over ranks by
  //      reference batch hands this state from the owner of one batch to the owner of the next around it (fq_ctx_set_serial_hooks)
  if (c->before_serial) c->before_serial(c->hook_user);
  c->serial_open = true;
  if (c->stream_broken) { c->err = "the stream's state comes from a call that failed (on this rank or on the one before)"; return FQ_EIO; }
  if ((rc = stageB1_main_hit(K))) return rc;
  K.trace("B1 replay + main hit");
  if (o.single_end) K.iis.assign(K.n_sub, fq_isize_t{});     // no pairs: no insert sizes, no (k,l) cache, no pairing, no mate rescue
  else {
    stageB2_isize(K);
    c->last_ii = K.iis[K.n_sub - 1];
    if ((rc = stage_kl_cache(K))) return rc;
  }
  K.gate_take(c->qc);               // (inside the order-dependent part: the tickets are in the stream's order)
  if (c->after_serial) c->after_serial(c->hook_user);
  c->serial_done = true;
  if (c->stream_broken) { c->err = "the hook that hands the stream's state on failed (fq_ctx_mark_stream_broken)"; return FQ_EIO; }
  K.t_serial1 = now_ms(); K.w_serial1 = c->wait_ms;
  K.trace("B2 isize, (k,l) cache");
  if ((rc = stageB3_pairing(K))) return rc;
  K.trace("B3 pairing+XA");
  if (c->debug && (rc = snapshot_records(K, S.stage_P))) return rc;   // snapshot for the stage dump (tests)
  if (!o.single_end && (rc = stageC_mate_sw(K))) return rc;
  K.trace("C mate SW");
  if (c->debug && (rc = snapshot_records(K, S.stage_S))) return rc;
  if ((rc = stageD_refine(K))) return rc;
  K.trace("D refine+MD");
  K.t_host1 = now_ms(); K.w_host1 = c->wait_ms;
  return stage_finish(K, out);
}
}  // namespace

extern "C" int fq_host_cpus(void) { return (int)effective_cpus(); }
extern "C" int fq_runtime_configure(int hw_queues, int blocking_waits) { return fqdev::runtime_configure(hw_queues, blocking_waits); }
extern "C" int fq_device_count(void) { return fqdev::device_count(); }

extern "C" int fq_align_resident(fq_ctx_t *c, fq_result_batch_t *out) {
  if (!c || !out) return FQ_EINVAL;
  if (c->in_kind != 1) { c->err = "fq_align_resident: no batch uploaded (fq_batch_upload)"; return FQ_EINVAL; }
  if (fqdev::bind(c->dev)) return FQ_ENODEV;
  return run_call(c, out);
}

static int packed_check(fq_ctx_t *c, const fq_packed_batch_t *in) {
  if (!in || in->n_pairs < 0 || (in->n_pairs > 0 && (!in->head || !in->body || in->body_stride < 1 || (in->uniform_len <= 0 && !in->len)))) return FQ_EINVAL;
  if (in->n_exc < 0 || (in->n_exc > 0 && !in->exc)) return FQ_EINVAL;
  if (in->n_pairs > c->max_pairs) { c->err = "batch larger than max_pairs_per_batch"; return FQ_ELIMIT; }
  if (in->uniform_len > 0 && (in->uniform_len < FQ_LMIN || in->uniform_len > FQ_LMAX || (in->uniform_len + 3) / 4 > in->body_stride)) { c->err = "read length outside [" + std::to_string(FQ_LMIN) + "," + std::to_string(FQ_LMAX) + "]"; return FQ_ELIMIT; }
  if (c->o.trim_qual >= 1 && in->n_pairs > 0 && (!in->qual || in->qual_stride < 1)) { c->err = "quality trimming needs the batch's qualities"; return FQ_EINVAL; }
  if (in->n_pairs > 0 && in->qual_last && !in->qual) { c->err = "qual_last without the quality rows"; return FQ_EINVAL; }
  if ((in->single_end != 0) != (c->o.single_end != 0)) { c->err = c->o.single_end ? "a single-end context takes single-end batches (fq_pack_single_reads_into)" : "a paired-end context takes paired batches"; return FQ_EINVAL; }
  return FQ_OK;
}
extern "C" int fq_packed_prefetch(fq_ctx_t *c, const fq_packed_batch_t *next) {
  if (!c) return FQ_EINVAL;
  int rc = packed_check(c, next);
  if (rc) return rc;
  if (fqdev::bind(c->dev)) return FQ_ENODEV;
  if (next->n_pairs == 0 || c->is_pending(0, next) || c->is_pending(1, next)) return FQ_OK;
  for (int i = 0; i < 2; ++i) if (c->pend[i] == next) c->pend[i] = nullptr;   // the object was packed again since: what the buffer holds is stale
  // Calls are synchronous, so no kernel reads either buffer now: a buffer is free unless it holds a prefetched batch that has
  // not been aligned yet.  Usual order: prefetch(k+1), align(k) -- batch k waits in one buffer, k+1 goes into the other.
  const int slot = !c->pend[0] ? 0 : !c->pend[1] ? 1 : -1;
  if (slot < 0) return FQ_OK;   // both hold pending batches: this one is uploaded when its call comes
  rc = head_upload(c, next, slot);
  if (rc) return rc;
  c->pend[slot] = next; c->pend_serial[slot] = next->serial;
  return FQ_OK;
}
extern "C" int fq_packed_cancel(fq_ctx_t *c, const fq_packed_batch_t *b) {
  if (!c || !b) return FQ_EINVAL;
  for (int i = 0; i < 2; ++i)
    if (c->pend[i] == b) {
      if (fqdev::bind(c->dev)) return FQ_ENODEV;
      CK(fqdev::copy_wait(i));        // the copy engine reads the batch's arrays until its upload has finished
      c->pend[i] = nullptr;
    }
  return FQ_OK;
}
extern "C" int fq_align_packed(fq_ctx_t *c, const fq_packed_batch_t *in, fq_result_batch_t *out) {
  if (!c || !out) return FQ_EINVAL;
  int rc = packed_check(c, in);
  if (rc) return broken_stream_call(c, rc);
  if (fqdev::bind(c->dev)) return FQ_ENODEV;
  c->pb = *in;
  c->n_pairs = in->n_pairs;
  c->in_kind = 2;
  if (in->n_pairs > 0) {
    if (c->is_pending(0, in)) { c->head_slot = 0; c->pend[0] = nullptr; }
    else if (c->is_pending(1, in)) { c->head_slot = 1; c->pend[1] = nullptr; }
    else {   // not prefetched: into a buffer that holds no pending batch (a pending one is dropped if both do)
      for (int i = 0; i < 2; ++i) if (c->pend[i] == in) c->pend[i] = nullptr;   // (same object, older content)
      c->head_slot = !c->pend[0] ? 0 : 1;
      c->pend[c->head_slot] = nullptr;
      rc = head_upload(c, in, c->head_slot);
      if (rc) return broken_stream_call(c, rc);
    }
  }
  return run_call(c, out);
}

// Several streams of packed batches run to their ends inside the library: what every host program of more than one stream wrote for itself (a
// thread per stream: prefetch the next batch, align this one, hand the result on) -- PairEndMapper's producer / consumer pair per FASTQ pair
// (src/BwtMapper.cpp:232-262, 1840-1845: a thread pool over the lines of --fq_list).  One library thread per stream, asleep while the device
// works (the waits are blocking events); the caller's thread only waits for them.
extern "C" int fq_stream_run(fq_ctx_t *const *ctxs, int32_t n_streams, const fq_packed_batch_t *const *const *batches, const int32_t *n_batches, const int32_t *first,
                             int32_t n_calls, int32_t flags, fq_stream_call_fn on_call, void *user, int64_t *survivors_out) {
  if (!ctxs || n_streams < 1 || !batches || !n_batches || n_calls < 0) return FQ_EINVAL;
  for (int s = 0; s < n_streams; ++s) if (!ctxs[s] || !batches[s] || n_batches[s] < 1) return FQ_EINVAL;
  for (int s = 0; s < n_streams; ++s) for (int t = 0; t < s; ++t) if (ctxs[s] == ctxs[t]) return FQ_EINVAL;      // (a context is one stream's)
  std::vector<int> rcs((size_t)n_streams, FQ_OK);
  std::vector<int64_t> surv((size_t)n_streams, 0);
  static const bool trace_cpu = [] { const char *e = getenv("FASTQUICK_TRACE"); return e && *e && *e != '0'; }();     // the stream threads' CPU time, on stderr
  std::vector<double> cpu_pre((size_t)n_streams, 0.0), cpu_call((size_t)n_streams, 0.0);
  auto thread_cpu_ms = [] { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; };
  auto run = [&](int s) {
    fq_ctx_t *c = ctxs[s];
    const int nb = n_batches[s];
    int cur = (first ? first[s] : 0) % nb;
    if (cur < 0) cur += nb;
    fq_result_batch_t res;
    for (int k = 0; k < n_calls; ++k) {
      const int nxt = (cur + 1) % nb;
      int rc = FQ_OK;
      const bool pre = nxt != cur && (k + 1 < n_calls || (flags & FQ_STREAM_PREFETCH_BEYOND));
      const double c0 = trace_cpu ? thread_cpu_ms() : 0.0;
      if (pre) rc = fq_packed_prefetch(c, batches[s][nxt]);      // the next batch's upload runs under this batch's kernels
      const double c1 = trace_cpu ? thread_cpu_ms() : 0.0;
      if (!rc) rc = fq_align_packed(c, batches[s][cur], &res);
      if (trace_cpu) { cpu_pre[(size_t)s] += c1 - c0; cpu_call[(size_t)s] += thread_cpu_ms() - c1; }
      if (!rc) { surv[(size_t)s] += res.n_survivors; if (on_call) rc = on_call(user, s, k, &res); }
      if (rc) { rcs[(size_t)s] = rc; if (pre) (void)fq_packed_cancel(c, batches[s][nxt]); return; }
      cur = nxt;
    }
  };
  std::vector<std::thread> th;
  for (int s = 1; s < n_streams; ++s) th.emplace_back(run, s);
  run(0);                                                    // (the caller's thread is the first stream's)
  for (auto &t : th) t.join();
  if (survivors_out) for (int s = 0; s < n_streams; ++s) survivors_out[s] = surv[(size_t)s];
  if (trace_cpu && n_calls > 0) {
    double a = 0, b = 0;
    for (int s = 0; s < n_streams; ++s) { a += cpu_pre[(size_t)s]; b += cpu_call[(size_t)s]; }
    fprintf(stderr, "TRACE - fq_stream_run: %d streams x %d calls; CPU time of a stream's thread per call: prefetch %.3f ms, call %.3f ms\n", n_streams, n_calls,
            a / ((double)n_streams * n_calls), b / ((double)n_streams * n_calls));
  }
  for (int rc : rcs) if (rc) return rc;
  return FQ_OK;
}

// The whole hot path on a batch of the device front end (fq_frontend_next): nothing of the input crosses PCIe.  The batch must stay
// unreleased until the consumers of this call's records (fq_sam_format_last, fq_qc_add_last, fq_bam_*) have run.
extern "C" int fq_align_text(fq_ctx_t *c, const fq_text_batch *tb, fq_result_batch_t *out) {
  if (!c || !out || !tb) return FQ_EINVAL;
  int rc = FQ_OK;
  if (tb->n_pairs < 0 || (tb->n_pairs > 0 && (!tb->d_head || !tb->d_rec || !tb->d_text[0] || !tb->d_names))) rc = FQ_EINVAL;
  else if (tb->n_pairs > c->max_pairs) { c->err = "batch larger than max_pairs_per_batch"; rc = FQ_ELIMIT; }
  else if ((tb->single_end != 0) != (c->o.single_end != 0)) { c->err = "single-end / paired-end batch on a context of the other kind"; rc = FQ_EINVAL; }
  else if (tb->device != c->ix->device) { c->err = "the batch lives on another device than the context's index"; rc = FQ_EINVAL; }
  else if (tb->uniform_len > 0 && (tb->uniform_len < FQ_LMIN || tb->uniform_len > FQ_LMAX)) { c->err = "read length outside [" + std::to_string(FQ_LMIN) + "," + std::to_string(FQ_LMAX) + "]"; rc = FQ_ELIMIT; }
  if (rc) return broken_stream_call(c, rc);
  if (fqdev::bind(c->dev)) return FQ_ENODEV;
  c->tb = tb;
  c->n_pairs = tb->n_pairs;
  c->in_kind = 3;
  return run_call(c, out);
}

static const uint64_t kStateGood = 0x31545351465full, kStateBroken = 0x58545351465full;   // "_FQST1" / "_FQSTX"
// ---- order-dependent state of a stream, for sharding ONE FASTQ stream over ranks by reference batch -------------
// Layout: u64 mark (good / broken) | u64 rng | fq_isize_t last_ii | u64 n_entries | per entry: u64 key, u64 n, u32 pos[n]
extern "C" int64_t fq_ctx_state_export(const fq_ctx_t *c, void *buf, int64_t cap) {
  if (!c) return FQ_EINVAL;
  int64_t need = 8 + 8 + (int64_t)sizeof(fq_isize_t) + 8;
  for (const auto &kv : c->kl_cache) need += 16 + 4 * (int64_t)kv.second.size();
  if (!buf || cap < need) return need;
  uint8_t *p = (uint8_t *)buf;
  const uint64_t mark = c->stream_broken ? kStateBroken : kStateGood;
  memcpy(p, &mark, 8); p += 8;
  memcpy(p, &c->rng, 8); p += 8;
  memcpy(p, &c->last_ii, sizeof(fq_isize_t)); p += sizeof(fq_isize_t);
  const uint64_t n = c->kl_cache.size();
  memcpy(p, &n, 8); p += 8;
  for (const auto &kv : c->kl_cache) {
    const uint64_t m = kv.second.size();
    memcpy(p, &kv.first, 8); p += 8;
    memcpy(p, &m, 8); p += 8;
    memcpy(p, kv.second.data(), 4 * m); p += 4 * m;
  }
  return need;
}
// the same hand-over between two contexts of ONE process (the command line's two contexts in turn): the state moves, nothing is serialised -- the (k,l)
// cache of a repeat-rich stream is megabytes.  `from` is left with an empty cache; a broken stream stays broken on both.
extern "C" int fq_ctx_state_move(fq_ctx_t *to, fq_ctx_t *from) {
  if (!to || !from || to == from) return FQ_EINVAL;
  if (from->stream_broken) { to->stream_broken = true; to->err = "the stream's state comes from a call that failed"; return FQ_EIO; }
  to->rng = from->rng; to->last_ii = from->last_ii;
  to->kl_cache.swap(from->kl_cache);
  from->kl_cache.clear();
  return FQ_OK;
}
extern "C" int fq_ctx_state_import(fq_ctx_t *c, const void *buf, int64_t len) {
  if (!c || !buf || len < (int64_t)(24 + sizeof(fq_isize_t))) return FQ_EINVAL;
  const uint8_t *p = (const uint8_t *)buf, *end = p + len;
  uint64_t mark;
  memcpy(&mark, p, 8); p += 8;
  if (mark == kStateBroken) { c->stream_broken = true; c->err = "the stream's state comes from a call that failed on another rank"; return FQ_EIO; }
  if (mark != kStateGood) return FQ_EINVAL;
  // the whole token is parsed into temporaries and committed at the end: a malformed one (it comes from another rank) leaves the
  // context as it was -- and marks the stream broken, since the state it should have continued from never arrived
  uint64_t rng, n;
  fq_isize_t ii;
  memcpy(&rng, p, 8); p += 8;
  memcpy(&ii, p, sizeof(fq_isize_t)); p += sizeof(fq_isize_t);
  memcpy(&n, p, 8); p += 8;
  std::unordered_map<uint64_t, vector<uint32_t>> cache;
  bool ok = n <= (uint64_t)(end - p) / 16;
  for (uint64_t i = 0; ok && i < n; ++i) {
    if (end - p < 16) { ok = false; break; }
    uint64_t key, m;
    memcpy(&key, p, 8); p += 8;
    memcpy(&m, p, 8); p += 8;
    if (m > (uint64_t)(end - p) / 4) { ok = false; break; }      // (not 4 * m > ...: that wraps for m >= 2^62)
    vector<uint32_t> v((size_t)m);
    if (m) memcpy(v.data(), p, 4 * (size_t)m);
    p += 4 * (size_t)m;
    cache.emplace(key, std::move(v));
  }
  if (!ok) { c->stream_broken = true; c->err = "malformed stream state (fq_ctx_state_import)"; return FQ_EINVAL; }
  c->rng = rng; c->last_ii = ii; c->kl_cache.swap(cache);
  return FQ_OK;
}
// A hook that could not do its work (its transport failed, its language raised) says so here, from inside the hook: the call stops
// after `before`, and the state `after` exports carries the broken mark -- nobody goes on with a stale state.
extern "C" int fq_ctx_mark_stream_broken(fq_ctx_t *c) {
  if (!c) return FQ_EINVAL;
  c->stream_broken = true;
  return FQ_OK;
}
extern "C" int fq_ctx_set_serial_hooks(fq_ctx_t *c, fq_serial_hook before, fq_serial_hook after, void *user) {
  if (!c) return FQ_EINVAL;
  c->before_serial = before; c->after_serial = after; c->hook_user = user;
  return FQ_OK;
}

// ---- the consumers on the device ------------------------------------------------------------------------------------------------------
extern "C" int fq_ctx_set_emit(fq_ctx_t *c, int32_t flags) {
  if (!c || (flags & ~(FQ_EMIT_SAM | FQ_EMIT_DEVICE_ONLY))) return FQ_EINVAL;
  c->emit_flags = flags;
  return FQ_OK;
}
// The SAM text of the last call leaves the device in slices through two pinned buffers on streams of its own, so that it runs beside the
// next call on another context: sink(user, data, bytes) gets the slices in order.
// The consumers' fill kernels of the last call were only enqueued (emit_fill): whoever fetches what they leave waits here first -- on the context's
// own stream, which nothing else drives between two calls -- and takes the counts that came back with them.
int fq_ctx_emit_wait(fq_ctx_t *c) {
  std::lock_guard<std::mutex> lk(c->emit_mu);
  if (!c->emit_pending) return FQ_OK;
  if (fqdev::bind(c->dev) || fqdev::sync()) { c->err = std::string("waiting for the consumers' kernels: ") + fqdev::last_error(); return FQ_ENODEV; }
  if (c->qc && c->qc_out.ready)
    for (int k = 0; k < FQ_QC_C_COUNT; ++k) { uint64_t v = 0; for (int st = 0; st < FQ_C_STRIPES; ++st) v += c->p_qcnt.p[(size_t)st * FQ_C_STRIDE + k]; c->qc_out.cnt[k] = v; }
  if (c->bam && c->emit_nb) c->bam_out.z_bytes = c->p_ztotal.p[0];
  c->emit_pending = false;
  return FQ_OK;
}
static int64_t stream_device_bytes(fq_ctx_t *c, int lane, const char *src, uint64_t total, fq_sink_fn sink, void *user, const char *what, size_t granule = 1) {
  if (fq_ctx_emit_wait(c)) return FQ_ENODEV;
  if (!total) return 0;
  fqdev::State *&st = c->dev_emit[lane];
  PinBuf<char> *pin = c->p_emit[lane];
  if (!st) st = fqdev::state_create(c->ix->device);
  if (!st || fqdev::bind(st)) { c->err = std::string(what) + ": " + fqdev::last_error(); return FQ_ENODEV; }
  const size_t SL = std::min<uint64_t>(total, ((uint64_t)32 << 20) / granule * granule);     // (a slice holds whole items)
  if (!pin[0].ensure(SL) || !pin[1].ensure(SL)) { c->err = "out of pinned host memory"; return FQ_ENOMEM; }
  const uint64_t n_sl = (total + SL - 1) / SL;
  auto bytes_of = [&](uint64_t k) { return (size_t)std::min<uint64_t>(SL, total - k * SL); };
  if (fqdev::copy_pinned(pin[0].p, src, bytes_of(0), 0) || fqdev::sync()) { c->err = std::string(what) + ": " + fqdev::last_error(); return FQ_ENODEV; }
  for (uint64_t k = 0; k < n_sl; ++k) {
    if (k + 1 < n_sl && fqdev::copy_pinned(pin[(k + 1) & 1].p, src + (k + 1) * SL, bytes_of(k + 1), 0)) { c->err = std::string(what) + ": " + fqdev::last_error(); return FQ_ENODEV; }
    if (sink(user, pin[k & 1].p, (int64_t)bytes_of(k))) { (void)fqdev::sync(); c->err = std::string(what) + ": the sink failed"; return FQ_EIO; }
    if (fqdev::sync()) { c->err = std::string(what) + ": " + fqdev::last_error(); return FQ_ENODEV; }
  }
  return (int64_t)total;
}
extern "C" int64_t fq_sam_device_last(fq_ctx_t *c, fq_sink_fn sink, void *user) {
  if (!c || !sink) return FQ_EINVAL;
  if (!(c->emit_flags & FQ_EMIT_SAM) || !c->sam_ready) { c->err = "fq_sam_device_last: the last call did not format its SAM text on the device (fq_ctx_set_emit)"; return FQ_EINVAL; }
  return stream_device_bytes(c, 0, c->d_samtext.p, c->sam_bytes, sink, user, "fq_sam_device_last");
}
// the BAM records of the last call (fq_ctx_attach_bam), for the writer they were formatted for
int64_t fq_ctx_bam_stream(fq_ctx_t *c, fq_sink_fn sink, void *user, int members) {
  if (!c->bam || !c->bam_out.ready) { c->err = "the last call formatted no BAM records on the device"; return FQ_EINVAL; }
  if (fq_ctx_emit_wait(c)) return FQ_ENODEV;
  if (members) return stream_device_bytes(c, 0, (const char *)c->d_bamz.p, c->bam_out.z_bytes, sink, user, "BGZF members");
  return stream_device_bytes(c, 0, (const char *)c->d_bamrec.p, c->bam_out.bytes, sink, user, "BAM records");
}
extern "C" int fq_ctx_attach_bam(fq_ctx_t *c, fq_bam_t *b) {
  if (!c) return FQ_EINVAL;
  c->bam = b;
  c->bam_out = FqBamCallOut();
  return FQ_OK;
}
const FqBamCallOut *fq_ctx_bam_out(const fq_ctx_t *c) { return c->bam ? &c->bam_out : nullptr; }
extern "C" int fq_ctx_attach_qc(fq_ctx_t *c, fq_qc_t *q) {
  if (!c) return FQ_EINVAL;
  c->qc = q;
  c->qc_out = FqQcCallOut();
  return FQ_OK;
}
const FqQcCallOut *fq_ctx_qc_out(const fq_ctx_t *c) { return c->qc ? &c->qc_out : nullptr; }
// what the call left in HBM for the consumer's host side: which = 0 the .InsertSizeTable lines, 1 the pileup entries (whole entries per slice)
int64_t fq_ctx_qc_stream(fq_ctx_t *c, int which, fq_sink_fn sink, void *user) {
  if (!c->qc || !c->qc_out.ready) { c->err = "the last call counted nothing on the device"; return FQ_EINVAL; }
  if (which == 0) return stream_device_bytes(c, 1, c->d_isttext.p, c->qc_out.ist_bytes, sink, user, "InsertSizeTable lines");
  return stream_device_bytes(c, 1, (const char *)c->d_pile.p, c->qc_out.n_pile * sizeof(FqPileEntry), sink, user, "pileup entries", sizeof(FqPileEntry));
}
extern "C" int64_t fq_sam_device_bytes(const fq_ctx_t *c) { return c && (c->emit_flags & FQ_EMIT_SAM) && c->sam_ready ? (int64_t)c->sam_bytes : FQ_EINVAL; }

extern "C" int fq_ctx_set_debug(fq_ctx_t *c, int keep_stage_snapshots) {
  if (!c) return FQ_EINVAL;
  c->debug = keep_stage_snapshots;
  return FQ_OK;
}
void fq_ctx_all_reads(const fq_ctx_t *c, const uint8_t **filtered, const int32_t **len_trim) {
  *filtered = c->h_filtered.empty() ? nullptr : c->h_filtered.data();
  *len_trim = c->h_len_trim.empty() ? nullptr : c->h_len_trim.data();
}
// accessors used by fq_sam.cpp
const FqBatchState *fq_ctx_state(const fq_ctx_t *c) { return &c->st; }
const fq_index *fq_ctx_index(const fq_ctx_t *c) { return c->ix; }
FqHostReads fq_ctx_host_reads(const fq_ctx_t *c) {
  FqHostReads h;
  if (c->in_kind == 3) {
    h.compact = true; h.n_pairs = c->n_pairs; h.name_stride = c->c_name_stride; h.c_stride = c->c_stride; h.c_n_surv = c->st.n_surv;
    if (c->host_rows) { h.c_seq = c->p_cseq.p; h.c_qual = c->p_cqual.p; h.c_len = c->p_clen.p; h.c_names = c->p_cnames.p; }
    h.c_pair_idx = c->st.pair_idx;
    h.c_len_all = c->h_len_all.empty() ? nullptr : c->h_len_all.data(); h.c_uniform_len = c->tb ? c->tb->uniform_len : 0;
  } else if (c->in_kind == 2) { h.p = &c->pb; h.n_pairs = c->pb.n_pairs; h.names = c->pb.names; h.names_mate = c->pb.names_mate; h.name_stride = c->pb.name_stride; }
  else { h.a = &c->hb; h.n_pairs = c->hb.n_pairs; h.names = c->hb.names; h.names_mate = c->hb.names_mate; h.name_stride = c->hb.name_stride; }
  return h;
}
const fq_opts_t *fq_ctx_opts(const fq_ctx_t *c) { return &c->o; }
int64_t fq_ctx_last_bases(const fq_ctx_t *c) { return c->n_bases_in; }
