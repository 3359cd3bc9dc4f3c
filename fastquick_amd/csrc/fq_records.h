// fq_records.h -- the alignment records of a call, resident on the device from the end of the gap search to the result arrays.
//
// Round 3 kept one host object per read (std::vector members and all) and walked the 8.4 M of an on-target call a dozen times: an
// on-target call cost 2 core-seconds of host work and the device idled 70 % of the time.  Here the records are 64-byte device structs;
// main-hit choice, SA enumeration, insert-size samples, pairing, XA selection, mate-rescue windows, refinement and MD task lists and the
// flattening into the C-ABI arrays are kernels of one thread per read or per pair over them.  The host keeps what is serial or
// libm-dependent by definition: the replay of the drand48 stream (one 16-bit count per read), infer_isize per reference batch, the
// (k,l) cache, the accept / reject arithmetic of the mate rescue, and pairing for the pairs a lane does not take -- all of them on
// sparse lists the device compacts.
//
// Every body is a FQ_HD function of (args, index); fq_device.hip wraps each in a __global__ kernel, tests/emu loops over it.
// Reference citations are paths under the Griffan/FASTQuick tree.
#pragma once
#include "fq_kernels.h"
#include "../../include/fastquick_amd.h"

// bwa_seq_t fields the hot path writes (libbwa/bwtaln.h:57-86), one cache line per read.  Index of a record: 2 * survivor pair + end.
struct alignas(64) FqDRec {
  uint32_t pos, sa, c1, c2;
  int32_t score;
  int32_t sidx;                 // search index or -1 (filtered: never searched)
  int16_t len, full_len, clip_len, main_aln;
  uint8_t type, strand, filtered, extra_flag;
  uint8_t n_mm, n_gapo, n_gape, mapQ;
  uint8_t seQ, revived, has_md, pad0;
  uint16_t nm, n_cigar;
  uint32_t cig_off;             // the read's CIGAR in the call's device CIGAR arena (n_cigar entries; none: implicit <len>M)
  uint32_t multi_off, n_multi;  // XA hits in the call's device multi arena
  int32_t md_len;               // MD string in slot [index][md_cap]
};
static_assert(sizeof(FqDRec) == 64, "one record per cache line");

struct FqSwCand { int32_t sp, k; };          // mate-rescue task: survivor pair, which mate is being placed
struct FqRefTgt { int32_t idx, multi; };      // refine task: record, XA entry (-1: the main hit)
#define FQ_RNG_CHUNK_PAIRS 16                 // the host replays the drand48 stream and hands the device its state at every 16th pair
#define FQ_PAIR_LANE_ROWS 64u                 // a lane pairs at most this many rows (both reads together); the rest goes to the host
#define FQ_CIG_CAP 64                         // CIGAR entries per refine / mate-rescue slot

struct FqRecArgs {
  FqDevIndex ix;
  int32_t n_surv, n_pairs, batch_pairs, packed, single_end;
  uint32_t max_occ, multi_cap;
  int32_t n_multi, N_multi, max_isize, s_mm, is_sw;
  // stage 0 / A products
  const int32_t *pair_list;      // [n_surv] survivor pair -> pair of the batch
  const FqSurvInfo *surv;        // [2 n_surv]
  const int32_t *len_trim;       // by device row
  const int32_t *full_len;       // by device row
  int32_t *sub_surv_max;         // optional: longest trimmed survivor per reference batch (atomic max)
  const FqAln *hits;             // all hit lists of the call
  const uint64_t *aoff;          // by search index
  const uint32_t *an;
  const uint8_t *maxdiff_lut;
  const int32_t *g_log_n;
  // records and their side arrays
  FqDRec *rec;
  uint32_t *nocc;                // [2 n_surv] rows of all hits (saturating)
  uint16_t *ntop;                // hits that share the best score (saturating): what the drand48 replay needs
  uint8_t *enumerated;
  uint32_t *qfirst;              // first hit of the read in the enumerated list
  uint64_t *row0;                // first row of the read in pos
  uint32_t *pq_cnt, *prow_cnt;   // per pair: hits / rows to enumerate
  const uint64_t *pq, *prow;     // their exclusive prefix sums
  FqAln *qaln; uint32_t *qlen; uint64_t *qoff;
  uint64_t n_q, n_rows;
  const uint32_t *pos;           // positions of the enumerated rows (k_sa)
  uint64_t *counters;
  // main hit
  const uint64_t *rng_start;     // state of the drand48 stream at pair 16 c
  uint32_t *isz;                 // [n_surv] insert-size sample or ~0
  uint8_t *cls;                  // [n_surv] 0: no pairing, 1: paired by a lane, 2: by the host (Q6 intervals, many rows)
  uint32_t *flag32;              // [n_surv] scratch for compactions (count per item)
  const uint64_t *off64;         // ... its exclusive prefix sums
  int32_t *list;                 // compacted indices
  // pairing
  uint64_t *pscratch;            // as long as pos
  const FqPairIsize *pisize;     // per reference batch
  const int32_t *plut;
  FqPairRead *g_reads;           // gather / apply of the host's pairs: [2 n_list]
  uint64_t *g_row0;
  const FqPairOut *g_out;
  int32_t n_list;
  // XA
  uint32_t *xcnt;                // [2 n_surv]
  const uint64_t *xoff;
  fq_multi_t *multi;             // device multi arena (cigar_off: into the device CIGAR arena)
  // mate rescue
  const fq_isize_t *iis;         // per reference batch
  FqSwTask *swslot;              // [2 n_surv]: candidate k of pair sp at 2 sp + k (pad = 1) or none (pad = 0)
  uint32_t *swcnt;               // [n_surv]
  const uint64_t *swoff;
  FqSwTask *swtask; FqSwCand *swcand;
  FqDRec *g_rec;                 // gather / scatter of whole records: [2 n_list], pairs named by list
  // refinement
  uint32_t *rcnt;                // [2 n_surv] refine tasks of a record
  const uint64_t *roff;
  FqRefTask *reftask; FqRefTgt *reftgt;
  int32_t *ref_max;              // [2]: longest reference window, longest query (atomic max)
  const FqRefOut *refout;
  uint32_t ref_cig_base;         // first CIGAR slot of the refine tasks in the arena
  int32_t n_ref;
  uint16_t *cigs;                // the call's device CIGAR arena
  // MD
  const uint8_t *seq; int32_t stride;
  char *md; int32_t md_cap;
  uint16_t *mdmask;              // [2 n_surv][stride / 16] mismatch bits of ungapped reads (NULL: every read walks its row)
  // flatten
  uint32_t *fc_cnt, *fm_cnt, *fx_cnt;          // [2 n_surv] CIGAR entries, MD bytes, XA entries of a record
  const uint64_t *fc_off, *fm_off, *fx_off;
  fq_result_t *o_rec; uint16_t *o_cigar; char *o_md; fq_multi_t *o_multi;
};

FQ_HD int fq_rec_row(const FqRecArgs &A, int idx) {   // row of the read's bases on the device
  return A.packed ? idx : (idx & 1) * A.n_pairs + A.pair_list[idx >> 1];
}
FQ_HD bool fq_rec_mapped(const FqDRec &p) { return p.type == FQ_TYPE_UNIQUE || p.type == FQ_TYPE_REPEAT; }

// ---- fresh records for the reads of surviving pairs (bwa_read_seq_with_hash_dev's output state, flags of src/BwtMapper.cpp:749) ----
FQ_HD void fq_rec_init_thread(const FqRecArgs &A, int idx) {
  const int sp = idx >> 1, e = idx & 1;
  const int dr = fq_rec_row(A, idx);
  const FqSurvInfo si = A.surv[idx];
  FqDRec p;
  p.pos = p.sa = p.c1 = p.c2 = 0; p.score = 0; p.sidx = si.sidx; p.main_aln = 0;
  const int lt = A.len_trim[dr];
  p.len = p.clip_len = (int16_t)lt;
  p.full_len = (int16_t)((A.single_end && e) ? 0 : A.full_len[dr]);
  p.type = 0; p.strand = 0; p.filtered = (uint8_t)si.filtered;
  p.extra_flag = (uint8_t)(A.single_end ? 0 : (1 | (e == 0 ? 64 : 128)));   // SAM_FPD | SAM_FR1 / FR2; the single-end mapper sets none
  p.n_mm = p.n_gapo = p.n_gape = p.mapQ = p.seQ = p.revived = p.has_md = p.pad0 = 0;
  p.nm = p.n_cigar = 0; p.cig_off = 0; p.multi_off = p.n_multi = 0; p.md_len = 0;
  A.rec[idx] = p;
  if (A.sub_surv_max) FQ_ATOMIC_MAX32(&A.sub_surv_max[A.pair_list[sp] / A.batch_pairs], lt);
}

// ---- per read: rows of all its hits, and the hits that share the best score --------------------------------------------
FQ_HD void fq_rec_nocc_thread(const FqRecArgs &A, int idx) {
  const int s = A.surv[idx].sidx;
  uint32_t nocc = 0, ntop = 0;
  if (s >= 0) {
    const FqAln *a = A.hits + A.aoff[s];
    const uint32_t na = A.an[s];
    uint64_t t = 0;
    for (uint32_t k = 0; k < na; ++k) t += (uint64_t)(a[k].l - a[k].k) + 1;
    nocc = t > 0xffffffffull ? 0xffffffffu : (uint32_t)t;
    while (ntop < na && a[ntop].score <= a[0].score) ++ntop;     // (lists are in discovery order: best score first)
  }
  A.nocc[idx] = nocc;
  A.ntop[idx] = (uint16_t)(ntop > 65535u ? 65535u : ntop);
}

// ---- SA rows to resolve: every row of every hit of reads that can need them --------------------------------------------
// eligible(read) = n_occ <= max(n_multi, N_multi) + 1  (XA listing, libbwa/bwase.c:47-55)  or
//                  both mates have hits and both n_occ <= max_occ (pair enumeration, src/BwtMapper.cpp:797-811)
FQ_HD void fq_enum_plan_thread(const FqRecArgs &A, int sp) {
  const int s0 = A.surv[2 * sp].sidx, s1 = A.surv[2 * sp + 1].sidx;
  const uint32_t na0 = s0 >= 0 ? A.an[s0] : 0u, na1 = s1 >= 0 ? A.an[s1] : 0u;
  const uint32_t o0 = A.nocc[2 * sp], o1 = A.nocc[2 * sp + 1];
  const bool pair_ok = na0 > 0 && na1 > 0 && o0 <= A.max_occ && o1 <= A.max_occ;
  uint32_t nq = 0, nr = 0;
  const bool e0 = na0 > 0 && (pair_ok || o0 <= A.multi_cap), e1 = na1 > 0 && (pair_ok || o1 <= A.multi_cap);
  if (e0) { nq += na0; nr += o0; }
  if (e1) { nq += na1; nr += o1; }
  A.enumerated[2 * sp] = e0 ? 1 : 0; A.enumerated[2 * sp + 1] = e1 ? 1 : 0;
  A.pq_cnt[sp] = nq; A.prow_cnt[sp] = nr;
}
FQ_HD void fq_enum_fill_thread(const FqRecArgs &A, int sp) {
  uint64_t q = A.pq[sp], r0 = A.prow[sp];
  for (int e = 0; e < 2; ++e) {
    const int idx = 2 * sp + e;
    A.qfirst[idx] = (uint32_t)q; A.row0[idx] = r0;
    if (!A.enumerated[idx]) continue;
    const int s = A.surv[idx].sidx;
    const FqAln *a = A.hits + A.aoff[s];
    const uint32_t na = A.an[s], len = (uint32_t)A.rec[idx].len;
    for (uint32_t k = 0; k < na; ++k) {
      A.qaln[q] = a[k]; A.qlen[q] = len; A.qoff[q] = r0;
      ++q; r0 += (uint64_t)(a[k].l - a[k].k) + 1;
    }
  }
  if (sp == A.n_surv - 1) A.qoff[A.n_q] = A.n_rows;
}

// ---- main hit: bwa_aln2seq_core with set_main (libbwa/bwase.c:19-46), bwa_approx_mapQ (:102-111) ------------------------------
// glibc drand48: X' = (0x5DEECE66D X + 0xB) mod 2^48, value X' / 2^48
FQ_HD double fq_rng_step(uint64_t &x) {
  x = (0x5DEECE66DULL * x + 0xBULL) & 0xFFFFFFFFFFFFULL;
  return (double)x * (1.0 / 281474976710656.0);
}
// what the choice of one read's main hit draws from the stream (nothing else of a read reaches the reads behind it)
FQ_HD void fq_main_draws(uint64_t &x, const FqAln *a, uint32_t na) {
  if (na == 0) return;
  const int best = a[0].score;
  uint32_t cnt = 0;
  for (uint32_t i = 0; i < na; ++i) {
    if (a[i].score > best) break;
    const uint32_t wdt = a[i].l - a[i].k + 1;
    if (fq_rng_step(x) * (double)(uint32_t)(wdt + cnt) > (double)(int)cnt) (void)fq_rng_step(x);
    cnt += wdt;
  }
}
// Returns false when NO hit was taken: the first best hit is taken "unless the draw is exactly 0" -- once in 2^48 draws -- and then the
// reference's record keeps the SA row its read slot held before (unmodellable), which it resolves and reads the reference at, wherever
// that is.  The caller un-maps the record (no kernel may follow a wild position) and the call fails, loudly (FQ_C_ERR_DRAW0).
FQ_HD bool fq_main_choose(uint64_t &x, const FqAln *a, uint32_t na, FqDRec &s) {
  if (na == 0) { s.type = FQ_TYPE_NO_MATCH; s.c1 = s.c2 = 0; return true; }
  const int best = a[0].score;
  uint32_t i, cnt = 0;
  bool taken = false;
  for (i = 0; i < na; ++i) {
    const FqAln p = a[i];
    if (p.score > best) break;
    const uint32_t wdt = p.l - p.k + 1;
    if (fq_rng_step(x) * (double)(uint32_t)(wdt + cnt) > (double)(int)cnt) {
      s.n_mm = (uint8_t)(p.info & 0xff); s.n_gapo = (uint8_t)((p.info >> 8) & 0xff); s.n_gape = (uint8_t)((p.info >> 16) & 0xff); s.strand = (uint8_t)((p.info >> 24) & 1);
      s.score = p.score;
      s.sa = p.k + (uint32_t)((double)wdt * fq_rng_step(x));
      s.main_aln = (int16_t)i;
      taken = true;
    }
    cnt += wdt;
  }
  s.c1 = cnt & 0xfffffff;
  for (; i < na; ++i) cnt += a[i].l - a[i].k + 1;
  s.c2 = (cnt - s.c1) & 0xfffffff;
  s.type = s.c1 > 1 ? FQ_TYPE_REPEAT : FQ_TYPE_UNIQUE;
  return taken;
}
FQ_HD int fq_approx_mapq(const FqRecArgs &A, const FqDRec &p) {
  const int mm = A.maxdiff_lut[p.len];
  if (p.c1 == 0) return 23;
  if (p.c1 > 1) return 0;
  if (p.n_mm == mm) return 25;
  if (p.c2 == 0) return 37;
  const int n = p.c2 >= 255 ? 255 : (int)p.c2;
  return 23 < A.g_log_n[n] ? 0 : 23 - A.g_log_n[n];
}
// the insert size infer_isize looks at for one pair, or ~0 (libbwa/bwape.c:62-71: both ends mapQ >= 20, below 100,000)
FQ_HD uint32_t fq_isize_sample(const FqDRec &a, const FqDRec &b) {
  if (a.mapQ >= 20 && b.mapQ >= 20) {
    const uint64_t x = a.pos < b.pos ? (uint64_t)(uint32_t)(b.pos + (uint32_t)b.len - a.pos) : (uint64_t)(uint32_t)(a.pos + (uint32_t)a.len - b.pos);
    if (x < 100000) return (uint32_t)x;
  }
  return ~0u;
}
// One pair per thread.  The stream is consumed in read order; a read depends on the reads before it only through the state they
// leave, so a thread starts from the state the host's replay left at its chunk of 16 pairs and draws (without choosing) for the
// pairs before its own.
FQ_HD void fq_main_hit_thread(const FqRecArgs &A, int sp) {
  const int c0 = sp - sp % FQ_RNG_CHUNK_PAIRS;
  uint64_t x = A.rng_start[sp / FQ_RNG_CHUNK_PAIRS];
  for (int idx = 2 * c0; idx < 2 * sp; ++idx) {
    const int t = A.ntop[idx];
    if (t == 0) continue;
    if (t == 1) {                                   // wdt >= 1: the hit is taken (a second draw) unless the first draw is exactly 0
      const uint64_t x1 = (0x5DEECE66DULL * x + 0xBULL) & 0xFFFFFFFFFFFFULL;
      x = x1 == 0 ? x1 : (0x5DEECE66DULL * x1 + 0xBULL) & 0xFFFFFFFFFFFFULL;
      continue;
    }
    const int s = A.surv[idx].sidx;
    fq_main_draws(x, A.hits + A.aoff[s], A.an[s]);
  }
  FqDRec p[2];
  for (int e = 0; e < 2; ++e) {
    const int idx = 2 * sp + e;
    p[e] = A.rec[idx];
    if (p[e].filtered) continue;
    const int s = p[e].sidx;
    const FqAln *a = s >= 0 ? A.hits + A.aoff[s] : nullptr;
    const uint32_t na = s >= 0 ? A.an[s] : 0u;
    if (!fq_main_choose(x, a, na, p[e])) {
      p[e].type = FQ_TYPE_NO_MATCH; p[e].c1 = p[e].c2 = 0;
      FQ_ATOMIC_ADD64(&A.counters[FQ_C_ERR_DRAW0], 1);
    }
    if (fq_rec_mapped(p[e])) {
      if (A.enumerated[idx]) {
        uint64_t row = A.row0[idx];
        for (int k = 0; k < p[e].main_aln; ++k) row += (uint64_t)(a[k].l - a[k].k) + 1;
        p[e].pos = A.pos[row + (p[e].sa - a[p[e].main_aln].k)];
      } else {                                       // main hit of a very repetitive read whose rows were not enumerated (bwa_cal_pac_pos_core)
        uint32_t steps = 0;
        p[e].pos = p[e].strand ? fq_sa_lookup(A.ix.fm[0], p[e].sa, &steps) : A.ix.fm[1].seq_len - (fq_sa_lookup(A.ix.fm[1], p[e].sa, &steps) + (uint32_t)p[e].len);
        FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_SA], steps);
        FQ_ATOMIC_ADD64(&A.counters[FQ_C_SA_DIRECT], 1);
      }
      p[e].seQ = p[e].mapQ = (uint8_t)fq_approx_mapq(A, p[e]);
    }
    A.rec[idx] = p[e];
  }
  if (A.single_end) return;
  A.isz[sp] = fq_isize_sample(p[0], p[1]);
  // who pairs this pair (libbwa/bwape.c:119-213 over the rows enumerated above): a lane when both reads are mapped, enumerated and hold
  // at most FQ_PAIR_LANE_ROWS rows together (no interval of 1,000 rows or more among them: those take their positions from the
  // (k,l) cache); the host otherwise
  const uint32_t o0 = A.nocc[2 * sp], o1 = A.nocc[2 * sp + 1];
  uint8_t cls = 0;
  if (fq_rec_mapped(p[0]) && fq_rec_mapped(p[1]) && o0 <= A.max_occ && o1 <= A.max_occ)
    cls = (A.enumerated[2 * sp] && A.enumerated[2 * sp + 1] && (uint64_t)o0 + o1 <= FQ_PAIR_LANE_ROWS) ? 1 : 2;
  A.cls[sp] = cls;
  A.flag32[sp] = cls == 2 ? 1u : 0u;
  FQ_WAVE_COUNT(&A.counters[FQ_C_PAIRS_DEV], cls == 1);
}

// out[off[i] ..] = i for every i with cnt[i] != 0 (ordered compaction; off = exclusive prefix sums of cnt != 0 ? 1 : 0 -- callers
// pass 0/1 counts)
FQ_HD void fq_compact_thread(const FqRecArgs &A, int i) {
  if (A.flag32[i]) A.list[A.off64[i]] = i;
}

// ---- pairing on the device --------------------------------------------------------------------------------------------------
FQ_HD FqPairRead fq_pair_read_of(const FqDRec &p) {
  FqPairRead r;
  r.pos = p.pos; r.len = p.len; r.full_len = p.full_len; r.bits = (uint32_t)(p.strand & 1) | (uint32_t)p.mapQ << 8 | (uint32_t)p.seQ << 16;
  return r;
}
FQ_HD void fq_pair_apply(FqDRec &q, const FqPairOut &o) {
  if (!((o.bits >> 24) & 1u)) return;            // no proper pair: the record stays as it is
  q.mapQ = (uint8_t)(o.bits & 0xffu); q.seQ = (uint8_t)((o.bits >> 8) & 0xffu);
  q.extra_flag |= 2;
  if ((o.bits >> 25) & 1u) {                     // the pair's hit is not the read's main hit: the record moves (bwape.c:196-211)
    q.n_mm = (uint8_t)(o.info & 0xff); q.n_gapo = (uint8_t)((o.info >> 8) & 0xff); q.n_gape = (uint8_t)((o.info >> 16) & 0xff);
    q.strand = (uint8_t)((o.bits >> 16) & 1u); q.score = o.score;
    q.pos = o.pos;
  }
}
// one pair: list its rows, sort (a handful of entries for all but repeats: insertion sort where the rows lie), sweep
FQ_HD void fq_pair_rec_thread(const FqRecArgs &A, int sp) {
  if (A.cls[sp] != 1) return;
  FqDRec p0 = A.rec[2 * sp], p1 = A.rec[2 * sp + 1];
  const uint32_t q0 = A.qfirst[2 * sp], q1 = A.qfirst[2 * sp + 1];
  const uint32_t na0 = A.an[p0.sidx], na1 = A.an[p1.sidx];
  const FqAln *aln0 = A.qaln + q0, *aln1 = A.qaln + q1;
  uint64_t *arr = A.pscratch + A.row0[2 * sp];
  uint32_t n = 0;
  for (int j = 0; j < 2; ++j) {
    const uint32_t q = j ? q1 : q0, na = j ? na1 : na0;
    for (uint32_t k = 0; k < na; ++k) {
      const uint32_t wdt = A.qaln[q + k].l - A.qaln[q + k].k + 1;
      const uint32_t *ps = A.pos + A.qoff[q + k];
      for (uint32_t z = 0; z < wdt; ++z) {
        const uint64_t x = (uint64_t)ps[z] << 32 | (uint64_t)(k << 1) | (uint64_t)j;
        uint32_t at = n++;
        while (at > 0 && arr[at - 1] > x) { arr[at] = arr[at - 1]; --at; }
        arr[at] = x;
      }
    }
  }
  FqPairRead r[2] = {fq_pair_read_of(p0), fq_pair_read_of(p1)};
  FqPairOut out[2];
  fq_pair_sweep(aln0, aln1, r, arr, n, A.pisize[A.pair_list[sp] / A.batch_pairs], A.plut, A.g_log_n, A.max_isize, A.s_mm, out);
  fq_pair_apply(p0, out[0]); fq_pair_apply(p1, out[1]);
  A.rec[2 * sp] = p0; A.rec[2 * sp + 1] = p1;
}
// the host's pairs (A.list): what its routine reads of the records, and its outcome back into them
FQ_HD void fq_pair_gather_thread(const FqRecArgs &A, int t) {
  const int sp = A.list[t];
  for (int e = 0; e < 2; ++e) { A.g_reads[2 * t + e] = fq_pair_read_of(A.rec[2 * sp + e]); A.g_row0[2 * t + e] = A.row0[2 * sp + e]; }
}
FQ_HD void fq_pair_scatter_thread(const FqRecArgs &A, int t) {
  const int sp = A.list[t];
  for (int e = 0; e < 2; ++e) { FqDRec p = A.rec[2 * sp + e]; fq_pair_apply(p, A.g_out[2 * t + e]); A.rec[2 * sp + e] = p; }
}

// ---- XA lists: bwa_aln2seq_core without set_main (libbwa/bwase.c:47-95), called per read by bwa_cal_pac_pos_pe
//      (src/BwtMapper.cpp:857-871); the single-end mapper selects its alternative hits with the main one (N_OCC, :1344) ------------
FQ_HD int fq_xa_limit(const FqRecArgs &A, int idx, const FqDRec &p, const FqDRec &mate) {   // n_multi of the call, or -1: no call
  if (A.single_end) return (idx & 1) || p.type == FQ_TYPE_NO_MATCH || p.filtered ? -1 : 3;
  if (!(A.N_multi || A.n_multi) || p.type == FQ_TYPE_NO_MATCH) return -1;
  // a pair whose reads are both mapped and one of them has more than max_occ rows is skipped before its lists (BwtMapper.cpp:797-811)
  const int sp = idx >> 1;
  if (fq_rec_mapped(p) && fq_rec_mapped(mate) && (A.nocc[2 * sp] > A.max_occ || A.nocc[2 * sp + 1] > A.max_occ)) return -1;
  if (!(p.extra_flag & 2) && mate.type != FQ_TYPE_NO_MATCH) return (int)(p.c1 + p.c2) - 1 > A.N_multi ? A.n_multi : A.N_multi;
  return A.n_multi;
}
// the rows of the read's hits that are not its main hit's row, in hit order, at most nm of them; fill != 0 writes them
FQ_HD uint32_t fq_xa_select(const FqRecArgs &A, int idx, const FqDRec &p, int nm, fq_multi_t *out) {
  if (nm <= 0 || p.sidx < 0) return 0;
  const uint32_t n_occ = A.nocc[idx];
  if (n_occ > (uint32_t)nm + 1u) return 0;
  if (!A.enumerated[idx]) return 0;                // (n_occ <= multi_cap: always enumerated)
  const FqAln *a = A.hits + A.aoff[p.sidx];
  const uint32_t na = A.an[p.sidx];
  uint32_t n = 0;
  uint64_t row = A.row0[idx];
  for (uint32_t k = 0; k < na; ++k) {
    const FqAln q = a[k];
    const uint32_t wdt = q.l - q.k + 1;
    for (uint32_t t = 0; t < wdt; ++t, ++row) {
      if (q.k + t == p.sa || n >= (uint32_t)nm) continue;
      if (out) {
        fq_multi_t m;
        m.pos = A.pos[row]; m.cigar_off = 0; m.n_cigar = 0;
        m.gap = (uint8_t)(((q.info >> 8) & 0xff) + ((q.info >> 16) & 0xff)); m.mm = (uint8_t)(q.info & 0xff);
        m.strand = (uint8_t)((q.info >> 24) & 1); m.pad[0] = m.pad[1] = m.pad[2] = 0;
        out[n] = m;
      }
      ++n;
    }
  }
  return n;
}
FQ_HD void fq_xa_count_thread(const FqRecArgs &A, int idx) {
  const FqDRec p = A.rec[idx], mate = A.rec[idx ^ 1];
  A.xcnt[idx] = fq_xa_select(A, idx, p, fq_xa_limit(A, idx, p, mate), nullptr);
}
FQ_HD void fq_xa_fill_thread(const FqRecArgs &A, int idx) {
  const uint32_t n = A.xcnt[idx];
  FqDRec p = A.rec[idx];
  p.multi_off = (uint32_t)A.xoff[idx]; p.n_multi = n;
  if (n) { const FqDRec mate = A.rec[idx ^ 1]; (void)fq_xa_select(A, idx, p, fq_xa_limit(A, idx, p, mate), A.multi + A.xoff[idx]); }
  A.rec[idx] = p;
}

// ---- mate rescue: the windows of bwa_paired_sw (libbwa/bwape.c:477-533) -------------------------------------------------------------
FQ_HD void fq_sw_plan_thread(const FqRecArgs &A, int sp) {
  FqSwTask none; none.read = 0; none.use_rc = 0; none.beg = 0; none.reglen = 0; none.pad = 0;   // pad: 1 = a candidate
  A.swslot[2 * sp] = none; A.swslot[2 * sp + 1] = none;
  A.swcnt[sp] = 0;
  const fq_isize_t ii = A.iis[A.pair_list[sp] / A.batch_pairs];
  if (ii.avg < 0.0) return;                      // bwa_paired_sw returns before touching anything (bwape.c:477)
  FqDRec p[2] = {A.rec[2 * sp], A.rec[2 * sp + 1]};
  bool touched = false;
  for (int j = 0; j < 2; ++j) if (p[j].filtered) { p[j].filtered = 0; p[j].revived = 1; touched = true; }   // expand_seq: revived because its mate passed (:485-499)
  if (touched) { A.rec[2 * sp] = p[0]; A.rec[2 * sp + 1] = p[1]; }
  if (!((p[0].mapQ >= 17 || p[1].mapQ >= 17) && (p[0].extra_flag & 2) == 0)) return;
  uint32_t n = 0;
  for (int k = 0; k < 2; ++k) {
    const FqDRec &pref = p[1 - k], &pm = p[k];
    if (pref.type == FQ_TYPE_NO_MATCH) continue;
    int64_t beg, end;
    FqSwTask T;
    T.pad = 1;
    if (pref.strand == 0) {   // __set_rght_coor (:511-516)
      beg = (int64_t)((int64_t)pref.pos + ii.avg - 3 * ii.std - pm.len * 1.5);
      end = (int64_t)(beg + 6 * ii.std + 2 * pm.len);
      // the macro assigns `_pref->pos + _pref->len` in 32-bit unsigned arithmetic (it wraps for a hit hanging over the start of the
      // reference, pos = 2^32-1) after comparing in 64 bits
      if (beg < (int64_t)pref.pos + pref.len) beg = (int64_t)(uint32_t)((uint32_t)pref.pos + (uint32_t)pref.len);
      if (end > A.ix.l_pac) end = A.ix.l_pac;
      T.use_rc = 1;
    } else {                  // __set_left_coor (:518-523)
      beg = (int64_t)((int64_t)pref.pos + pref.len - ii.avg - 3 * ii.std - pm.len * 0.5);
      end = (int64_t)(beg + 6 * ii.std + 2 * pm.len);
      if (beg < 0) beg = 0;
      if (end > (int64_t)pref.pos) end = pref.pos;
      T.use_rc = 0;
    }
    T.read = fq_rec_row(A, 2 * sp + k); T.beg = beg; T.reglen = (int)(end - beg);
    A.swslot[2 * sp + k] = T;
    ++n;
  }
  A.swcnt[sp] = n;
}
FQ_HD void fq_sw_fill_thread(const FqRecArgs &A, int sp) {
  uint64_t at = A.swoff[sp];
  for (int k = 0; k < 2; ++k) {
    FqSwTask T = A.swslot[2 * sp + k];
    if (!T.pad) continue;
    T.pad = 0;
    A.swtask[at] = T;
    FqSwCand c; c.sp = sp; c.k = k;
    A.swcand[at] = c;
    A.list[at] = sp;                       // (fq_rec_gather_thread / fq_rec_scatter_thread: the candidates' records go to the host and back)
    ++at;
  }
}
// whole records of the pairs named by A.list, to the host and back (mate-rescue decisions)
FQ_HD void fq_rec_gather_thread(const FqRecArgs &A, int t) {
  const int sp = A.list[t];
  A.g_rec[2 * t] = A.rec[2 * sp]; A.g_rec[2 * t + 1] = A.rec[2 * sp + 1];
}
FQ_HD void fq_rec_scatter_thread(const FqRecArgs &A, int t) {
  const int sp = A.list[t];
  A.rec[2 * sp] = A.g_rec[2 * t]; A.rec[2 * sp + 1] = A.g_rec[2 * t + 1];
}

// ---- refinement tasks: bwa_refine_gapped (libbwa/bwase.c:339-418) -- XA hits with gaps, then the main hit -----------------------------
FQ_HD bool fq_ref_main(const FqDRec &s) { return !(s.type == FQ_TYPE_NO_MATCH || s.type == FQ_TYPE_MATESW || s.n_gapo == 0); }
FQ_HD void fq_ref_count_thread(const FqRecArgs &A, int idx) {
  const FqDRec s = A.rec[idx];
  uint32_t n = 0;
  if (!s.filtered) {
    for (uint32_t j = 0; j < s.n_multi; ++j) if (A.multi[s.multi_off + j].gap) ++n;
    if (fq_ref_main(s)) ++n;
  }
  A.rcnt[idx] = n;
}
FQ_HD void fq_ref_fill_thread(const FqRecArgs &A, int idx) {
  if (!A.rcnt[idx]) return;
  const FqDRec s = A.rec[idx];
  uint64_t t = A.roff[idx];
  const int dr = fq_rec_row(A, idx);
  int max_ref = 1;
  for (uint32_t j = 0; j < s.n_multi; ++j) {
    const fq_multi_t q = A.multi[s.multi_off + j];
    if (!q.gap) continue;
    FqRefTask T; T.read = dr; T.strand = q.strand; T.pos = q.pos; T.ext = (q.strand ? 1 : -1) * (int)q.gap;
    FqRefTgt g; g.idx = idx; g.multi = (int32_t)j;
    A.reftask[t] = T; A.reftgt[t] = g; ++t;
    if (s.len + q.gap > max_ref) max_ref = s.len + q.gap;
  }
  if (fq_ref_main(s)) {
    FqRefTask T; T.read = dr; T.strand = s.strand; T.pos = s.pos; T.ext = (s.strand ? 1 : -1) * (s.n_gapo + s.n_gape);
    FqRefTgt g; g.idx = idx; g.multi = -1;
    A.reftask[t] = T; A.reftgt[t] = g;
    if (s.len + s.n_gapo + s.n_gape > max_ref) max_ref = s.len + s.n_gapo + s.n_gape;
  }
  if (max_ref > FQ_LOAD_RELAXED(&A.ref_max[0])) FQ_ATOMIC_MAX32(&A.ref_max[0], max_ref);     // (almost every lane finds the maximum in place already)
  if ((int)s.len > FQ_LOAD_RELAXED(&A.ref_max[1])) FQ_ATOMIC_MAX32(&A.ref_max[1], (int)s.len);
}
FQ_HD void fq_ref_apply_thread(const FqRecArgs &A, int t) {   // a task owns its record's field
  const FqRefOut O = A.refout[t];
  const FqRefTgt g = A.reftgt[t];
  if (O.n_cigar <= 0) { FQ_ATOMIC_ADD64(&A.counters[FQ_C_ERR_CIGAR], 1); return; }
  const uint32_t off = A.ref_cig_base + (uint32_t)t * FQ_CIG_CAP;
  if (g.multi >= 0) {
    fq_multi_t *q = A.multi + A.rec[g.idx].multi_off + g.multi;
    q->pos = O.pos; q->cigar_off = off; q->n_cigar = (uint16_t)O.n_cigar;
  } else {
    FqDRec *s = A.rec + g.idx;
    s->pos = O.pos; s->cig_off = off; s->n_cigar = (uint16_t)O.n_cigar;
  }
}

// ---- MD / NM of every mapped read (bwa_cal_md1), straight from the records ------------------------------------------------------
// Almost every read is ungapped, and a thread that walks its own 150-byte row touches a cache line per lane and load.  So the rows
// are compared with the reference by one thread per 16-base piece (consecutive threads, consecutive 16-byte loads: whole lines), which
// leaves one mismatch bit per base; the read's thread then only visits the set bits.  (Compact rows only: device row = record index.)
FQ_HD void fq_md_mask_piece(const FqRecArgs &A, int g) {
  const int per_row = A.stride >> 4;
  const int idx = g / per_row, cidx = g - idx * per_row;
  const FqDRec s = A.rec[idx];
  uint32_t m = 0;
  const int slen = s.len, j0 = 16 * cidx;
  if (s.type != FQ_TYPE_NO_MATCH && s.n_cigar == 0 && j0 < slen) {
    const FqU4 v = *(const FqU4 *)(A.seq + (size_t)idx * (size_t)A.stride + (size_t)j0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int jj = j0 + j;
      if (jj < slen) {
        const int code = (int)fq_nt4_fast((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
        const int sc = s.strand ? fq_comp(code) : code;
        const int c = fq_pac_base(A.ix.pac, (int64_t)(uint32_t)(s.pos + (uint32_t)(s.strand ? slen - 1 - jj : jj)));
        if (sc > 3 || c != sc) m |= 1u << j;
      }
    }
  }
  A.mdmask[g] = (uint16_t)m;
}
FQ_HD void fq_md_rec_thread(const FqRecArgs &A, int idx) {
  FqDRec s = A.rec[idx];
  if (s.type == FQ_TYPE_NO_MATCH) return;
  if (A.mdmask && s.n_cigar == 0) {             // ungapped, mismatch bits ready: <matches><ref base><matches>... in alignment order
    const int per_row = A.stride >> 4, slen = s.len, cap = A.md_cap - 1;
    const uint16_t *mk = A.mdmask + (size_t)idx * (size_t)per_row;
    char *dst = A.md + (size_t)idx * (size_t)A.md_cap;
    int at = 0, nm = 0, prev = -1;
    for (int t = 0; t < per_row; ++t) {
      const int cidx = s.strand ? per_row - 1 - t : t;
      uint32_t m = mk[cidx];
      while (m) {
        const int b = s.strand ? 31 - __builtin_clz(m) : FQ_CTZ32(m);
        m &= ~(1u << b);
        const int jj = 16 * cidx + b, yy = s.strand ? slen - 1 - jj : jj;
        at = fq_put_int(dst, at, cap, yy - prev - 1);
        if (at < cap) dst[at] = "ACGTN"[fq_pac_base(A.ix.pac, (int64_t)(uint32_t)(s.pos + (uint32_t)yy))];
        ++at; ++nm; prev = yy;
      }
    }
    at = fq_put_int(dst, at, cap, slen - 1 - prev);
    if (at > cap) { FQ_ATOMIC_ADD64(&A.counters[FQ_C_ERR_MD], 1); at = 0; }
    s.has_md = 1; s.md_len = at; s.nm = (uint16_t)(nm & 0xfff);
    A.rec[idx] = s;
    FQ_WAVE_COUNT(&A.counters[FQ_C_MD_READS], true);
    return;
  }
  FqMdTask T;
  T.read = fq_rec_row(A, idx); T.strand = s.strand; T.pos = s.pos; T.n_cigar = s.n_cigar; T.cigar_off = s.cig_off; T.len = s.len;
  int at = 0, nm = 0;
  fq_md_core(A.ix, A.seq + (size_t)T.read * (size_t)A.stride, T, A.cigs, A.md + (size_t)idx * (size_t)A.md_cap, A.md_cap - 1, &at, &nm);
  if (at > A.md_cap - 1) { FQ_ATOMIC_ADD64(&A.counters[FQ_C_ERR_MD], 1); at = 0; }
  s.has_md = 1; s.md_len = at; s.nm = (uint16_t)(nm & 0xfff);
  A.rec[idx] = s;
  FQ_WAVE_COUNT(&A.counters[FQ_C_MD_READS], true);
}

// ---- the C-ABI arrays, with bwa_correct_trimmed (libbwa/bwase.c:298-337) applied to every record as they are written ----------------
// entries of the record's final CIGAR: the soft clip of a trimmed read joins an S at that end, or is appended (to <len>M when
// the record has no CIGAR)
FQ_HD uint32_t fq_flat_ncigar(const FqRecArgs &A, const FqDRec &s) {
  uint32_t n = s.n_cigar;
  if (s.len == s.full_len) return n;
  if (n) {
    const uint16_t edge = s.strand == 0 ? A.cigs[s.cig_off + n - 1] : A.cigs[s.cig_off];
    return (edge >> 14) == FQ_OP_S ? n : n + 1;
  }
  return 2;
}
FQ_HD void fq_flat_count_thread(const FqRecArgs &A, int idx) {
  const FqDRec s = A.rec[idx];
  uint32_t cc = fq_flat_ncigar(A, s);
  for (uint32_t j = 0; j < s.n_multi; ++j) cc += A.multi[s.multi_off + j].n_cigar;
  A.fc_cnt[idx] = cc;
  A.fm_cnt[idx] = s.has_md ? (uint32_t)s.md_len + 1u : 0u;
  A.fx_cnt[idx] = s.n_multi;
}
FQ_HD void fq_flat_fill_thread(const FqRecArgs &A, int idx) {
  const FqDRec s = A.rec[idx];
  fq_result_t o;
  o.pos = s.pos; o.sa = s.sa; o.c1 = s.c1; o.c2 = s.c2; o.score = s.score;
  o.len = s.len != s.full_len ? s.full_len : s.len; o.full_len = s.full_len; o.clip_len = s.clip_len;
  o.type = s.type; o.strand = s.strand; o.filtered = s.filtered; o.extra_flag = s.extra_flag;
  o.n_mm = s.n_mm; o.n_gapo = s.n_gapo; o.n_gape = s.n_gape; o.mapQ = s.mapQ; o.seQ = s.seQ; o.revived = s.revived; o.nm = s.nm;
  uint32_t ca = (uint32_t)A.fc_off[idx];
  o.cigar_off = ca;
  const uint32_t nc = fq_flat_ncigar(A, s);
  o.n_cigar = (uint16_t)nc; o.n_multi = (uint16_t)s.n_multi;
  {
    uint16_t *dst = A.o_cigar + ca;
    const uint16_t *src = A.cigs + s.cig_off;
    const uint32_t n = s.n_cigar;
    if (s.len == s.full_len) { for (uint32_t k = 0; k < n; ++k) dst[k] = src[k]; }
    else {
      const int clip = s.full_len - s.len;
      if (s.strand == 0) {
        if (n && (src[n - 1] >> 14) == FQ_OP_S) { for (uint32_t k = 0; k < n; ++k) dst[k] = src[k]; dst[n - 1] = (uint16_t)(src[n - 1] + clip); }
        else {
          uint32_t k = 0;
          if (!n) dst[k++] = (uint16_t)(FQ_OP_M << 14 | s.len); else for (; k < n; ++k) dst[k] = src[k];
          dst[k] = (uint16_t)(FQ_OP_S << 14 | clip);
        }
      } else {
        if (n && (src[0] >> 14) == FQ_OP_S) { for (uint32_t k = 0; k < n; ++k) dst[k] = src[k]; dst[0] = (uint16_t)(src[0] + clip); }
        else {
          dst[0] = (uint16_t)(FQ_OP_S << 14 | clip);
          if (!n) dst[1] = (uint16_t)(FQ_OP_M << 14 | s.len); else for (uint32_t k = 0; k < n; ++k) dst[k + 1] = src[k];
        }
      }
    }
    ca += nc;
  }
  if (s.has_md) {
    const uint32_t ma = (uint32_t)A.fm_off[idx];
    const char *src = A.md + (size_t)idx * (size_t)A.md_cap;
    char *dst = A.o_md + ma;
    for (int k = 0; k < s.md_len; ++k) dst[k] = src[k];
    dst[s.md_len] = 0;
    o.md_off = ma;
  } else o.md_off = 0xffffffffu;
  uint32_t xa = (uint32_t)A.fx_off[idx];
  o.multi_off = xa;
  for (uint32_t j = 0; j < s.n_multi; ++j) {
    fq_multi_t m = A.multi[s.multi_off + j];
    const uint16_t *src = A.cigs + m.cigar_off;
    for (uint32_t k = 0; k < m.n_cigar; ++k) A.o_cigar[ca + k] = src[k];
    m.cigar_off = ca;
    ca += m.n_cigar;
    A.o_multi[xa++] = m;
  }
  A.o_rec[idx] = o;
  FQ_WAVE_COUNT(&A.counters[FQ_C_UNMAPPED], (idx & 1) && s.type == FQ_TYPE_NO_MATCH && A.rec[idx - 1].type == FQ_TYPE_NO_MATCH);
}

// by search index: where a read's hit list lies in the call's list arena (work item w of a launch completed with status 0)
FQ_HD void fq_aln_index_thread(const int32_t *work, const uint32_t *status, const uint64_t *off, const uint32_t *naln, uint64_t base, uint64_t *aoff, uint32_t *an, int w) {
  if (status[w]) return;
  const int s = work[w];
  aoff[s] = base + off[w];
  an[s] = naln[w];
}

enum {
  FQ_ROP_INIT = 0, FQ_ROP_NOCC, FQ_ROP_ENUM_PLAN, FQ_ROP_ENUM_FILL, FQ_ROP_MAIN_HIT, FQ_ROP_COMPACT, FQ_ROP_PAIR, FQ_ROP_PAIR_GATHER, FQ_ROP_PAIR_SCATTER,
  FQ_ROP_XA_COUNT, FQ_ROP_XA_FILL, FQ_ROP_SW_PLAN, FQ_ROP_SW_FILL, FQ_ROP_REC_GATHER, FQ_ROP_REC_SCATTER, FQ_ROP_REF_COUNT, FQ_ROP_REF_FILL, FQ_ROP_REF_APPLY,
  FQ_ROP_MD, FQ_ROP_MD_MASK, FQ_ROP_FLAT_COUNT, FQ_ROP_FLAT_FILL, FQ_ROP_COUNT
};
