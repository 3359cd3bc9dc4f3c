// tests/emu/fq_emu_backend.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host-loop implementation of fastquick_amd/csrc/fq_backend.h so that the CPU-only test tier
// (`pytest -m "not gpu"`, no GPU in the build container) can exercise the host pipeline and the very same
// per-thread kernel bodies (fq_kernels.h).  It is built into tests/emu/libfq_emu.so by tests/emu/Makefile,
// is never built by __graft_entry__.build(), and is never loaded by the product (fastquick_amd/api.py only
// opens an explicitly passed path; its default is the HIP library, and loading fails loudly without it).
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fq_backend.h"

namespace fqdev {
static std::string g_err;
struct State { Tune tune; };
static thread_local State *g_bound = nullptr;   // (for the launcher's choice of search kernel: the tuning of the bound context)
int runtime_configure(int, int) { return 0; }
int device_count() { return 4; }            // four virtual devices (the CPU tier's multi-device tests use two)
State *state_create(int dev) { if (dev < 0 || dev >= device_count()) { g_err = "device ordinal out of range"; return nullptr; } return new State; }
void state_destroy(State *s) { if (g_bound == s) g_bound = nullptr; delete s; }
int bind(State *s) { g_bound = s; return 0; }
Tune *tune(State *s) { return &s->tune; }
void device_turn_begin(int) {}
void device_turn_end() {}
int h2d_copy(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int copy_record(int) { return 0; }
int copy_wait(int) { return 0; }
void copy_discard() {}
int compute_wait_copy(int) { return 0; }
const char *last_error() { return g_err.c_str(); }
bool is_real_gpu() { return false; }
void *dmalloc(size_t b) { return calloc(b ? b : 16, 1); }
void dfree(void *p) { free(p); }
void *hmalloc(size_t b) { return malloc(b ? b : 16); }
void hfree(void *p) { free(p); }
int h2d(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int d2h(void *d, const void *s, size_t n) { if (n) memcpy(d, s, n); return 0; }
int copy_pinned(void *d, const void *s, size_t n, int) { if (n) memcpy(d, s, n); return 0; }
int d2d(void *d, const void *s, size_t n) { if (n) memmove(d, s, n); return 0; }
int dzero(void *d, size_t n) { if (n) memset(d, 0, n); return 0; }
int dfill(void *d, int b, size_t n) { if (n) memset(d, b, n); return 0; }
int sync() { return 0; }
void time_begin(int) {}
void time_end(int) {}
void time_collect(double[], uint64_t[], int) {}

int launch_prep(const FqPrepArgs &a) { for (int r = 0; r < a.n_reads; ++r) fq_prep_thread(a, r); return 0; }
int launch_compact(const uint8_t *f, int n, int32_t *read_list, int32_t *sidx, int32_t *pair_list, int32_t *counts) {
  int ns = 0, np = 0;
  for (int p = 0; p < n; ++p) {
    const int f0 = f[p], f1 = f[n + p];
    if (!(f0 && f1)) pair_list[np++] = p;
    if (!f0) { read_list[ns] = p; sidx[p] = ns++; } else sidx[p] = -1;
    if (!f1) { read_list[ns] = n + p; sidx[n + p] = ns++; } else sidx[n + p] = -1;
  }
  counts[0] = ns; counts[1] = np;
  return 0;
}
int launch_surv_gather(const int32_t *pair_list, int n_surv, int n_pairs, const int32_t *len_trim, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out) {
  for (int t = 0; t < 2 * n_surv; ++t) fq_surv_gather_thread(pair_list, n_pairs, len_trim, filtered, sidx, out, t);
  return 0;
}
int launch_prep_packed(const FqPrepPackedArgs &a) { for (int r = 0; r < a.n_reads; ++r) fq_prep_packed_thread(a, r); return 0; }
int launch_surv_map(const int32_t *pair_list, int n_surv, int n_pairs, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out,
                    int32_t *row_map, int32_t *read_list_c, int32_t *crow_of) {
  for (int t = 0; t < 2 * n_surv; ++t) fq_surv_map_thread(pair_list, n_pairs, filtered, sidx, out, row_map, read_list_c, crow_of, t);
  return 0;
}
int launch_unpack(const FqUnpackArgs &a) { for (int t = 0; t < a.n_rows; ++t) fq_unpack_thread(a, t); return 0; }
int launch_patch(const FqPatchArgs &a) { for (int64_t q = 0; q < a.n_exc; ++q) fq_patch_thread(a, q); return 0; }
int launch_trim(const FqTrimArgs &a) { for (int t = 0; t < a.n_rows; ++t) fq_trim_thread(a, t); return 0; }
int launch_trim_all(const FqTrimAllArgs &a) { for (int r = 0; r < a.n_reads; ++r) fq_trim_all_thread(a, r); return 0; }
int launch_width(const FqWidthArgs &a) {
  uint8_t seed_bits[2 * FQ_SEED_MAX];
  for (int w = 0; w < a.n_work; ++w) {
    if (g_bound && g_bound->tune.width_both_strands) fq_width_read(a, w, seed_bits, 1);
    else { fq_width_strand(a, w, 1, seed_bits, 1); fq_width_strand(a, w, 0, seed_bits, 1); }      // (one strand at a time, as the device's kernel)
  }
  return 0;
}
int launch_order(const uint8_t *bid_end, int n, int n_hard, int32_t *order, uint32_t *cnt) {
  int at = 0;
  cnt[2 * FQ_ORDER_KEYS] = 0;
  for (int k = FQ_ORDER_KEYS - 1; k >= 0; --k) {
    for (int w = 0; w < n; ++w) if (fq_order_key(bid_end, w, n_hard) == k) order[at++] = w;
    if (k == FQ_ORDER_KEYS / 2) cnt[2 * FQ_ORDER_KEYS] = (uint32_t)at;
  }
  return 0;
}
struct SeqFetch { uint32_t *next; int n; uint32_t operator()(uint32_t k) const { const uint32_t at = *next; *next += k; return at; } };
// the lane kernels' two-block queue, walked front to back by the single host "wavefront"
struct SeqFetch2 { uint32_t *next; uint32_t lo, hi; uint64_t operator()(uint32_t k) const { const uint32_t at = lo + *next; *next += k; return at < hi ? (uint64_t)at | (uint64_t)hi << 32 : 0; } };   // one block here: the segment [lo, hi) of it
int gap_lane_slots(const FqGapArgs &a) { return a.n_work > 0 ? 1 : 0; }
#if defined(FQ_PROFILE)
}  // namespace fqdev
unsigned long long fq_prof[128];
int fq_prof_off = 0;
extern "C" unsigned long long *fq_emu_prof() { return fq_prof; }
namespace fqdev {
#endif
int launch_gap(const FqGapArgs &a_in) {
  FqGapArgs a = a_in;
#if defined(FQ_PROFILE)
  fq_prof_off = a.tier.coop ? 64 : a.tier.nogap ? 32 : 0;
#endif
  a.refill_min = 1;
  a.split = nullptr;   // one cursor over the whole order here: FqGapLane::queue_dry then looks at it alone
  uint32_t *next_p = a.queue; *next_p = 0;   // the same cursor the device kernels advance
  uint32_t seg_lo, seg_hi;
  fq_seg_range((uint32_t)a.n_work, a.seg, a.n_seg, &seg_lo, &seg_hi);
  if (a.tier.coop) {
    std::vector<uint32_t> heads(2 * FQ_MAX_BUCKETS);
    fq_gap_coop_wave(a, heads.data(), SeqFetch{next_p, a.n_work}, 0);
  } else if (a.tier.pool_cap <= 65535u) {   // same store policy the HIP launcher picks: 16-bit heads in (here: emulated) LDS
    std::vector<uint16_t> heads(a.o.n_buckets);
    std::vector<uint32_t> cold(FQ_COLD_N);
    FqGapStoreLds st = {heads.data(), 1, cold.data()};
    // the HIP launcher's rule: the kernels compiled for FASTQuick's own option block when the options match (FqOptsStock)
    const bool stock = FqOptsStock::matches(a.o) && !(g_bound && g_bound->tune.gap_generic_opts);
    if (a.tier.nogap) { if (stock) fq_gap_lanes<true, FqOptsStock>(a, st, SeqFetch2{next_p, seg_lo, seg_hi}, 0); else fq_gap_lanes<true>(a, st, SeqFetch2{next_p, seg_lo, seg_hi}, 0); }
    else { if (stock) fq_gap_lanes<false, FqOptsStock>(a, st, SeqFetch2{next_p, seg_lo, seg_hi}, 0); else fq_gap_lanes<false>(a, st, SeqFetch2{next_p, seg_lo, seg_hi}, 0); }
  } else {
    FqGapStoreGlobal st = {nullptr, {}};
    fq_gap_lanes<false>(a, st, SeqFetch2{next_p, seg_lo, seg_hi}, 0);
  }
  return 0;
}
int launch_scan(const uint32_t *in, uint64_t *out, uint32_t n) { uint64_t s = 0; for (uint32_t i = 0; i < n; ++i) { out[i] = s; s += in[i]; } out[n] = s; return 0; }
int launch_pack_aln(const FqAln *aln, const uint32_t *n_aln, const uint64_t *off, uint32_t cap, uint32_t n_work, FqAln *packed) {
  for (uint32_t w = 0; w < n_work; ++w) for (uint32_t j = 0; j < n_aln[w]; ++j) packed[off[w] + j] = aln[(size_t)w * cap + j];
  return 0;
}
int launch_sa(const FqSaArgs &a) { for (uint64_t q = 0; q < a.n_rows; ++q) fq_sa_thread(a, q); return 0; }
int stream_aux(int) { return 0; }
int stream_fork() { return 0; }
int stream_join() { return 0; }
int stream_mark(int) { return 0; }
int stream_wait_mark(int) { return 0; }
int launch_collect(const int32_t *order, const uint32_t *, int n_work, int seg, int n_seg, const uint32_t *status, const int32_t *work, int32_t *out, uint32_t *count) {
  uint32_t lo, hi;
  fq_seg_range((uint32_t)n_work, seg, n_seg, &lo, &hi);   // (one block here, as in launch_gap)
  for (uint32_t pos = lo; pos < hi; ++pos) { const int w = order ? order[pos] : (int)pos; if (status[w]) out[(*count)++] = work[w]; }
  return 0;
}
int launch_rec(int op, const FqRecArgs &a, int64_t n) {
  typedef void (*Body)(const FqRecArgs &, int);
  static const Body bodies[FQ_ROP_COUNT] = {fq_rec_init_thread, fq_rec_nocc_thread, fq_enum_plan_thread, fq_enum_fill_thread, fq_main_hit_thread, fq_compact_thread, fq_pair_rec_thread,
                                            fq_pair_gather_thread, fq_pair_scatter_thread, fq_xa_count_thread, fq_xa_fill_thread, fq_sw_plan_thread, fq_sw_fill_thread, fq_rec_gather_thread,
                                            fq_rec_scatter_thread, fq_ref_count_thread, fq_ref_fill_thread, fq_ref_apply_thread, fq_md_rec_thread, fq_md_mask_piece, fq_flat_count_thread, fq_flat_fill_thread};
  if (op < 0 || op >= FQ_ROP_COUNT) return -1;
  for (int64_t i = 0; i < n; ++i) bodies[op](a, (int)i);
  return 0;
}
int launch_sam(int op, const FqSamArgs &a, int64_t n) {
  const int pieces = (2 * a.stride + 1 + FQ_SAM_PIECE - 1) / FQ_SAM_PIECE;
  for (int64_t i = 0; i < n; ++i) {
    if (op == FQ_EOP_SAM_LEN) fq_sam_len_thread(a, (int)i);
    else if (op == FQ_EOP_SAM_FILL) fq_sam_fill_thread(a, (int)i);
    else if (op == FQ_EOP_SAM_BODY) { for (int c = 0; c < pieces; ++c) fq_sam_body_piece(a, (int)i, c); }
    else return -1;
  }
  return 0;
}
int copy_flush_now() { return 0; }
int launch_deflate(const FqDeflateArgs &a) {
  static thread_local FqdLds lds;
  for (uint32_t b = 0; b < a.n_blocks; ++b) a.bsize[b] = fqd_member(a, b, lds);
  return 0;
}
int launch_deflate_pack(const FqDeflatePackArgs &a) {
  for (uint64_t t = 0; t < (uint64_t)a.n_blocks * (FQD_SLOT / 16); ++t) fqd_pack_piece(a, t);
  return 0;
}
int launch_bam(int op, const FqBamArgs &a, int64_t n) {
  const int pieces = ((a.s.stride + 1) / 2 + a.s.stride + FQ_SAM_PIECE - 1) / FQ_SAM_PIECE;
  for (int64_t i = 0; i < n; ++i) {
    if (op == FQ_EOP_BAM_LEN) fq_bam_len_thread(a, (int)i);
    else if (op == FQ_EOP_BAM_FILL) fq_bam_fill_thread(a, (int)i);
    else if (op == FQ_EOP_BAM_BODY) { for (int c = 0; c < pieces; ++c) fq_bam_body_piece(a, (int)i, c); }
    else return -1;
  }
  return 0;
}
int launch_qc(int op, const FqQcArgs &a, int64_t n) {
  if (op == FQ_QOP_BASE) {
    std::vector<uint32_t> hist(4 * 256, 0);
    for (int64_t i = 0; i < n; ++i) fq_qc_base_record(a, (int)i, 0, 1, hist.data());
    for (int b = 0; b < 4 * 256; ++b) a.hist[b] += hist[b];
    return 0;
  }
  for (int64_t i = 0; i < n; ++i) {
    if (op == FQ_QOP_PAIR) fq_qc_pair_thread(a, (int)i);
    else if (op == FQ_QOP_IST_FILL) fq_qc_ist_fill_thread(a, (int)i);
    else if (op == FQ_QOP_PILE_FILL) fq_qc_pile_fill_thread(a, (int)i);
    else return -1;
  }
  return 0;
}
int launch_dup_rehash(const uint64_t *old, uint64_t old_cap, uint64_t *tab, uint64_t mask) {
  for (uint64_t i = 0; i < old_cap; ++i) fq_dupset_rehash_thread(old, tab, mask, (int64_t)i);
  return 0;
}
int launch_aln_index(const int32_t *work, const uint32_t *status, const uint64_t *off, const uint32_t *naln, uint64_t base, uint64_t *aoff, uint32_t *an, int n) {
  for (int w = 0; w < n; ++w) fq_aln_index_thread(work, status, off, naln, base, aoff, an, w);
  return 0;
}
int launch_sw(const FqSwArgs &a) { for (int t = 0; t < a.n_task; ++t) fq_sw_thread(a, t); return 0; }
int launch_sw_serial(const FqSwArgs &a) { return launch_sw(a); }
int launch_refine(const FqRefineArgs &a) { for (int t = 0; t < a.n_task; ++t) fq_refine_thread(a, t); return 0; }
const FqzCrcConst *crc_const() { static const FqzCrcConst *c = [] { FqzCrcConst *p = new FqzCrcConst; fqz_crc_const_make(p); return p; }(); return c; }
int launch_inflate(const FqInflateArgs &a) {
  FqzLds *lds = new FqzLds;
  for (int m = 0; m < a.n_mem; ++m) a.status[m] = fqz_inflate_member(a, m, *lds);
  delete lds;
  return 0;
}
int launch_inflate2(const FqInflateArgs &a, const FqInflateArgs &b) { if (a.n_mem > 0 && launch_inflate(a)) return -3; return b.n_mem > 0 ? launch_inflate(b) : 0; }
int launch_nl_index(const uint8_t *text, uint32_t n, uint32_t lo, uint32_t *nl, uint32_t cap, uint32_t *count) {
  uint32_t c = 0;
  for (uint32_t i = lo; i < n; ++i) if (text[i] == '\n') { if (c < cap) nl[c] = i; ++c; }
  *count = c;
  return 0;
}
int launch_tok_rec(const FqTokArgs &a) { for (int i = 0; i < a.n_rec; ++i) fqt_stat_commit(a, fqt_rec_thread(a, i)); return 0; }
int launch_tok_pieces(const FqTokArgs &a) { const int64_t n = (int64_t)a.n_rec * ((a.max_len + 31) >> 5); for (int64_t g = 0; g < n; ++g) fqt_piece_thread(a, g); return 0; }
int launch_slot_bases(const FqSlotArgs &a) { if (a.n_rec > 0) for (int s = 0; s < a.n_slots; ++s) fqt_slot_bases_thread(a, s); return 0; }
int launch_slot_names(const FqSlotArgs &a) {
  if (a.n_rec <= 0) return 0;
  if (a.plain_names) { for (int i = 0; i < a.n_rec; ++i) fqt_names_plain_thread(a, i); if (a.mode != 0) return 0; }
  for (int s = 0; s < a.n_slots; ++s) fqt_slot_names_thread(a, s);
  return 0;
}
int launch_text_gather(const FqTextGatherArgs &a) { const int64_t n = (int64_t)a.n_out * (a.stride >> 4); for (int64_t g = 0; g < n; ++g) fqt_gather_piece(a, g); return 0; }
int launch_text_trim_all(const FqTextTrimArgs &a) { for (int r = 0; r < a.n_rows; ++r) fqt_trim_all_thread(a, r); return 0; }
int dfill32(void *dst, uint32_t v, size_t n_words) { for (size_t i = 0; i < n_words; ++i) ((uint32_t *)dst)[i] = v; return 0; }
int launch_bitmap_kmers(const FqBitmapArgs &a) {
  for (int64_t idx = 0; idx < 2 * a.l_pac; ++idx) fq_bitmap_kmer_thread(a, idx, [&](int t, uint32_t x) { a.bitmap[t][x >> 5] |= 1u << (x & 31); });
  return 0;
}
int launch_bitmap_scatter(uint8_t *bitmap, const uint32_t *bits, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) bitmap[bits[i] >> 3] |= (uint8_t)(1u << (bits[i] & 7));
  return 0;
}
}  // namespace fqdev
