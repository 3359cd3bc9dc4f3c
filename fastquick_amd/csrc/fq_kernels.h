// fq_kernels.h -- per-thread bodies of the HIP kernels of the FASTQuick-align hot path.
//
// Every body is a FQ_HD function of (args, thread index); fq_device.hip wraps each in a
// __global__ kernel for gfx950.  (tests/emu builds the same bodies into a host loop so that the
// host-side pipeline logic can be exercised by the CPU-only test tier; that library is test
// infrastructure and is never loaded by the product.)
//
// Reference citations are paths under the Griffan/FASTQuick tree.
#pragma once
#include "fq_common.h"

#define FQ_LAMBDA_INLINE __attribute__((always_inline))
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_POPC64(x) __popcll(x)
#define FQ_POPC32(x) __popc(x)
#define FQ_CTZ32(x) (__ffs((int)(x)) - 1)
// (the work counters: into the workgroup's stripe of the array, fq_common.h)
#define FQ_C_STRIPE_OFF ((size_t)(blockIdx.x & (FQ_C_STRIPES - 1)) * FQ_C_STRIDE)
#define FQ_ATOMIC_ADD64(p, v) atomicAdd((unsigned long long *)(p) + FQ_C_STRIPE_OFF, (unsigned long long)(v))
#define FQ_ATOMIC_MAX64(p, v) atomicMax((unsigned long long *)(p) + FQ_C_STRIPE_OFF, (unsigned long long)(v))
#define FQ_LOAD_RELAXED(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// (a maximum that almost every caller leaves as it is: a look first, past the CU's L1, and the atomic only when it would raise it)
#define FQ_ATOMIC_MAX32(p, v) do { const int fq_v_ = (int)(v); if (fq_v_ > FQ_LOAD_RELAXED((const int *)(p))) atomicMax((int *)(p), fq_v_); } while (0)
// counter += number of active lanes for which pred holds: one atomic per wavefront instead of one per lane
#define FQ_WAVE_COUNT(p, pred)                                                                                      \
  do {                                                                                                              \
    const unsigned long long m_ = __ballot((pred) ? 1 : 0);                                                         \
    if ((pred) && __lane_id() == (unsigned)(__ffsll((long long)m_) - 1)) atomicAdd((unsigned long long *)(p) + FQ_C_STRIPE_OFF, (unsigned long long)__popcll(m_)); \
  } while (0)
#else
#define FQ_WAVE_COUNT(p, pred) do { if (pred) *(p) += 1; } while (0)
#define FQ_LOAD_RELAXED(p) (*(p))
#define FQ_ATOMIC_MAX32(p, v) (*(p) = *(p) > (int32_t)(v) ? *(p) : (int32_t)(v))
#define FQ_ATOMIC_MAX64(p, v) (*(p) = *(p) > (uint64_t)(v) ? *(p) : (uint64_t)(v))
#define FQ_POPC64(x) __builtin_popcountll(x)
#define FQ_POPC32(x) __builtin_popcount(x)
#define FQ_CTZ32(x) __builtin_ctz(x)
#define FQ_ATOMIC_ADD64(p, v) (*(p) += (v))
#endif

struct FqU4 { uint32_t x, y, z, w; };   // one 16-byte load / store

// nst_nt4_table, libbwa/bntseq.c:38-55 (A0 C1 G2 T3, '-' 5, anything else 4)
FQ_HD int fq_nt4(uint8_t ch) {
  switch (ch) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
  }
}
FQ_HD int fq_comp(int c) { return c < 4 ? 3 - c : c; }
FQ_HD int fq_pac_base(const uint8_t *pac, int64_t k) { return pac[k >> 2] >> ((~k & 3) << 1) & 3; }

// ---- Occ / rank: bwt_occ, bwt_occ4, bwt_2occ, bwt_2occ4 (libbwa/bwt.h:98-222) ---------------
FQ_HD uint32_t fq_adj(const FqFM &f, uint32_t k) { return k >= f.primary ? k - 1 : k; }

FQ_HD uint32_t fq_occ1(const FqFM &f, uint32_t k, int c) {
  if (k == 0xffffffffu) return 0;
  k = fq_adj(f, k);
  const FqOccBlk *b = f.blk + (k >> 6);
  const uint64_t hi = b->hi, lo = b->lo;
  const uint64_t m = ~0ull << (63 - (k & 63));
  const uint64_t x = ((c & 2) ? hi : ~hi) & ((c & 1) ? lo : ~lo) & m;
  return b->cnt[c] + (uint32_t)FQ_POPC64(x);
}
FQ_HD void fq_occ4(const FqFM &f, uint32_t k, uint32_t o[4]) {
  if (k == 0xffffffffu) { o[0] = o[1] = o[2] = o[3] = 0; return; }
  k = fq_adj(f, k);
  const FqOccBlk *b = f.blk + (k >> 6);
  const uint64_t hi = b->hi, lo = b->lo;
  const uint64_t m = ~0ull << (63 - (k & 63));
  o[0] = b->cnt[0] + (uint32_t)FQ_POPC64(~hi & ~lo & m);
  o[1] = b->cnt[1] + (uint32_t)FQ_POPC64(~hi & lo & m);
  o[2] = b->cnt[2] + (uint32_t)FQ_POPC64(hi & ~lo & m);
  o[3] = b->cnt[3] + (uint32_t)FQ_POPC64(hi & lo & m);
}
// reference-defined touch count of a paired lookup: distinct 128-row blocks read
FQ_HD uint32_t fq_touch2(const FqFM &f, uint32_t k, uint32_t l, bool single_base) {
  bool vk = k != 0xffffffffu && !(single_base && k == f.seq_len);
  bool vl = l != 0xffffffffu && !(single_base && l == f.seq_len);
  if (k == l) return vk ? 1u : 0u;
  if (vk && vl) return (fq_adj(f, k) >> 7) == (fq_adj(f, l) >> 7) ? 1u : 2u;
  return (vk ? 1u : 0u) + (vl ? 1u : 0u);
}

FQ_HD uint32_t fq_touch2p(uint32_t primary, uint32_t seq_len, uint32_t k, uint32_t l, bool single_base) {
  const bool vk = k != 0xffffffffu && !(single_base && k == seq_len);
  const bool vl = l != 0xffffffffu && !(single_base && l == seq_len);
  const uint32_t ak = k >= primary ? k - 1 : k, al = l >= primary ? l - 1 : l;
  if (k == l) return vk ? 1u : 0u;
  if (vk && vl) return (ak >> 7) == (al >> 7) ? 1u : 2u;
  return (vk ? 1u : 0u) + (vl ? 1u : 0u);
}

// bwt_invPsi (bwt.h:66-70) + bwt_sa (bwt.c:69-79): one block fetch yields both the symbol and its rank
FQ_HD uint32_t fq_sa_lookup(const FqFM &f, uint32_t k, uint32_t *steps) {
  uint32_t sa = 0;
  while (k % f.sa_intv != 0) {
    ++sa;
    if (k == f.primary) { k = 0; continue; }
    const uint32_t ka = fq_adj(f, k);
    const FqOccBlk *b = f.blk + (ka >> 6);
    const uint64_t hi = b->hi, lo = b->lo;
    const int sh = 63 - (int)(ka & 63);
    const int c = (int)((hi >> sh) & 1) << 1 | (int)((lo >> sh) & 1);
    const uint64_t m = ~0ull << sh;
    const uint64_t x = ((c & 2) ? hi : ~hi) & ((c & 1) ? lo : ~lo) & m;
    k = f.L2[c] + b->cnt[c] + (uint32_t)FQ_POPC64(x);
  }
  *steps += sa;
  return sa + f.sa[k / f.sa_intv];
}

// register-friendly 4-way select (a runtime-indexed local array would be placed in scratch memory)
// Written with masks, not ?: -- hipcc turns "cond ? mem_a : mem_b" into a load from a *selected address*, and a variable
// address into a local object pins that object (and everything around it) in scratch memory.
FQ_HD uint32_t fq_pick2(uint32_t x1, uint32_t x0, int a) { const uint32_t m = 0u - (uint32_t)(a & 1); return (x1 & m) | (x0 & ~m); }   // a ? x1 : x0
FQ_HD uint64_t fq_pick2p(uint64_t x1, uint64_t x0, int a) { const uint64_t m = 0ull - (uint64_t)(a & 1); return (x1 & m) | (x0 & ~m); }
FQ_HD uint32_t fq_sel4v(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, int c) {   // c in 0..3; arguments are values (registers)
  const uint32_t lo = (c & 1) ? v1 : v0, hi = (c & 1) ? v3 : v2;
  return (c & 2) ? hi : lo;
}
FQ_HD uint32_t fq_sel4(const uint32_t *v, int c) { return fq_sel4v(v[0], v[1], v[2], v[3], c & 3); }

// raw 32-byte Occ block fetch shared by the single-base and the four-base rank.  The two bit planes are kept as 32-bit
// halves and the prefix mask is built from 32-bit shifts: 64-bit shifts are quarter-rate on CDNA4 and this code sits on the
// dependent chain of every search step.
struct FqBlkRaw { uint32_t c0, c1, c2, c3; uint32_t lo0, lo1, hi0, hi1; uint32_t mk0, mk1; bool valid; };   // *0 = bases 0..31 (bits 63..32 of the plane)
FQ_HD FqBlkRaw fq_blk_none() {
  FqBlkRaw r;
  r.valid = false;
  r.c0 = r.c1 = r.c2 = r.c3 = 0; r.lo0 = r.lo1 = r.hi0 = r.hi1 = 0; r.mk0 = r.mk1 = 0;
  return r;
}
FQ_HD FqBlkRaw fq_blk_load(const FqOccBlk *blk, uint32_t primary, uint32_t k) {
  FqBlkRaw r = fq_blk_none();
  r.valid = k != 0xffffffffu;
  if (r.valid) {
    k = k >= primary ? k - 1 : k;
    const FqOccBlk *b = blk + (k >> 6);
    r.c0 = b->cnt[0]; r.c1 = b->cnt[1]; r.c2 = b->cnt[2]; r.c3 = b->cnt[3];
    const uint64_t lo = b->lo, hi = b->hi;
    r.lo0 = (uint32_t)(lo >> 32); r.lo1 = (uint32_t)lo; r.hi0 = (uint32_t)(hi >> 32); r.hi1 = (uint32_t)hi;
    const uint32_t t = k & 63;                     // bases 0..t of the block are counted
    const uint32_t t0 = t < 32 ? t : 31, t1 = t < 32 ? 0 : t - 32;
    r.mk0 = 0xffffffffu << (31 - t0);
    r.mk1 = t < 32 ? 0u : (0xffffffffu << (31 - t1));
  }
  return r;
}
FQ_HD FqBlkRaw fq_blk_load(const FqFM &f, uint32_t k) { return fq_blk_load(f.blk, f.primary, k); }
FQ_HD uint32_t fq_popc2(uint32_t a, uint32_t b) { return (uint32_t)(FQ_POPC32(a) + FQ_POPC32(b)); }
FQ_HD uint32_t fq_blk_occ1(const FqBlkRaw &r, int c) {
  if (!r.valid) return 0;
  const uint32_t fh = 0u - (uint32_t)(((c >> 1) & 1) ^ 1), fl = 0u - (uint32_t)((c & 1) ^ 1);   // flip masks: select base c
  const uint32_t x0 = (r.hi0 ^ fh) & (r.lo0 ^ fl) & r.mk0, x1 = (r.hi1 ^ fh) & (r.lo1 ^ fl) & r.mk1;
  return fq_sel4v(r.c0, r.c1, r.c2, r.c3, c) + fq_popc2(x0, x1);
}
FQ_HD void fq_blk_occ4(const FqBlkRaw &r, uint32_t o[4]) {
  if (!r.valid) { o[0] = o[1] = o[2] = o[3] = 0; return; }
  const uint32_t nh0 = ~r.hi0 & r.mk0, nh1 = ~r.hi1 & r.mk1, h0 = r.hi0 & r.mk0, h1 = r.hi1 & r.mk1;
  o[0] = r.c0 + fq_popc2(nh0 & ~r.lo0, nh1 & ~r.lo1);
  o[1] = r.c1 + fq_popc2(nh0 & r.lo0, nh1 & r.lo1);
  o[2] = r.c2 + fq_popc2(h0 & ~r.lo0, h1 & ~r.lo1);
  o[3] = r.c3 + fq_popc2(h0 & r.lo0, h1 & r.lo1);
}
// Occ(c, k-1) and Occ(c, l) of one backward-extension step.  Once the interval is narrow both rows sit in the same 64-base block
// most of the time: the block is fetched once and only the prefix mask differs.
FQ_HD void fq_occ1_pair(const FqFM &f, uint32_t km1, uint32_t l, int c, uint32_t *ok, uint32_t *ol) {
  const FqBlkRaw bk = fq_blk_load(f, km1);
  FqBlkRaw bl;
  const uint32_t ak = km1 >= f.primary ? km1 - 1 : km1, al = l >= f.primary ? l - 1 : l;
  if (bk.valid && l != 0xffffffffu && (ak >> 6) == (al >> 6)) {
    bl = bk;
    const uint32_t t = al & 63, t0 = t < 32 ? t : 31, t1 = t < 32 ? 0 : t - 32;
    bl.mk0 = 0xffffffffu << (31 - t0);
    bl.mk1 = t < 32 ? 0u : (0xffffffffu << (31 - t1));
  } else bl = fq_blk_load(f, l);
  *ok = fq_blk_occ1(bk, c);
  *ol = fq_blk_occ1(bl, c);
}

// The two blocks of one search step (rows k-1 and l).  A request for a block is what the search kernels are bound by (their time
// follows the number of busy lanes x requests, not the number of loop iterations): once the interval is narrow -- after the first
// dozen bases of a read -- both rows sit in the same 64-base block, which is then fetched once and only re-masked.
FQ_HD void fq_blk_load_pair(const FqOccBlk *blk, uint32_t primary, uint32_t km1, uint32_t l, FqBlkRaw &bk, FqBlkRaw &bl) {
  bk = fq_blk_load(blk, primary, km1);
  const uint32_t ak = km1 >= primary ? km1 - 1 : km1, al = l >= primary ? l - 1 : l;
  if (bk.valid && l != 0xffffffffu && (ak >> 6) == (al >> 6)) {
    bl = bk;
    const uint32_t t = al & 63, t0 = t < 32 ? t : 31, t1 = t < 32 ? 0 : t - 32;
    bl.mk0 = 0xffffffffu << (31 - t0);
    bl.mk1 = t < 32 ? 0u : (0xffffffffu << (31 - t1));
  } else bl = fq_blk_load(blk, primary, l);
}

// ---- K_prep: encode + quality trim + k-mer filter ----------------------------------------------
// bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:526-588), bwa_trim_read (libbwa/bwaseqio.c:75-88),
// IsReadInHashByCountMoreChunck + CountKmerHitInHash (src/BwtIndexer.cpp:441-456, 261-313).
struct FqPrepArgs {
  FqDevIndex ix;
  FqKOpts o;
  const uint8_t *seq, *qual;
  const int32_t *len;
  int32_t stride, n_reads;
  int32_t *len_trim;     // out: p->len (== clip_len)
  uint8_t *filtered;     // out: 1 = filtered
  int32_t *sub_max;      // out: [pair / batch_pairs] max trimmed length over both ends (zeroed by the caller)
  int32_t n_pairs, batch_pairs;
  uint64_t *counters;
};
FQ_HD uint32_t fq_kmer_project(uint64_t kmer, int t) {
  switch (t) {
    case 0: return (uint32_t)(kmer >> 32);
    case 1: return (uint32_t)kmer;
    case 2: return (uint32_t)(((kmer >> 48) << 16) | (kmer & 0xffff));
    case 3: return (uint32_t)(kmer >> 16);
    case 4: return (uint32_t)(((kmer >> 48) << 16) | ((kmer >> 16) & 0xffff));
    default: return (uint32_t)((((kmer >> 32) & 0xffff) << 16) | (kmer & 0xffff));
  }
}
// branch-free nst_nt4_table for one ASCII byte: A/a 0, C/c 1, G/g 2, T/t 3, '-' 5, anything else 4
FQ_HD uint32_t fq_nt4_fast(uint32_t ch) {
  const uint32_t u = ch & 0xDFu;                                  // fold case
  const uint32_t code = ((ch >> 1) ^ (ch >> 2)) & 3u;              // A0 C1 G2 T3 for the four letters
  const bool acgt = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
  return acgt ? code : (ch == 0x2Du ? 5u : 4u);
}

FQ_HD void fq_prep_thread(const FqPrepArgs &A, int r) {
  const uint8_t *row = A.seq + (size_t)r * (size_t)A.stride;
  const int full = A.len[r];
  int len = full;
  if (A.o.trim_qual >= 1) {
    const uint8_t *q = A.qual + (size_t)r * (size_t)A.stride;
    int s = 0, mx = 0, max_l = full - 1;
    const int qsub = (A.o.mode & FQ_MODE_IL13) ? 31 : 0;
    for (int l = full - 1; l >= 34; --l) {
      s += A.o.trim_qual - ((int)(uint8_t)(q[l] - qsub) - 33);
      if (s < 0) break;
      if (s > mx) { mx = s; max_l = l; }
    }
    len = max_l + 1;
  }
  A.len_trim[r] = len;
  {   // per-reference-batch maximum: one atomic per wavefront (its reads almost always belong to one batch)
    const int slot = (r >= A.n_pairs ? r - A.n_pairs : r) / A.batch_pairs;
#if defined(__HIP_DEVICE_COMPILE__)
    const int slot0 = __builtin_amdgcn_readfirstlane(slot);
    if (__ballot(1) == ~0ull && __ballot(slot != slot0) == 0) {   // full wavefront, one slot
      int mx = len;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
      if (__builtin_amdgcn_readfirstlane(r) == r) FQ_ATOMIC_MAX32(&A.sub_max[slot0], mx);
    } else FQ_ATOMIC_MAX32(&A.sub_max[slot], len);
#else
    if (A.sub_max[slot] < len) A.sub_max[slot] = len;
#endif
  }
  uint8_t filt = 0;
  if (A.o.filter_thresh != 0) {
    // the first 96 bases as 24 little-endian words: six 16-byte loads when rows are 16-byte aligned
    uint32_t w[24];
    if ((((uintptr_t)row) & 15) == 0 && full >= 96) {
#if defined(__HIP_DEVICE_COMPILE__)
      const uint4 *v = (const uint4 *)row;
#pragma unroll
      for (int j = 0; j < 6; ++j) { const uint4 t = v[j]; w[4 * j] = t.x; w[4 * j + 1] = t.y; w[4 * j + 2] = t.z; w[4 * j + 3] = t.w; }
#else
      for (int j = 0; j < 24; ++j) w[j] = (uint32_t)row[4 * j] | (uint32_t)row[4 * j + 1] << 8 | (uint32_t)row[4 * j + 2] << 16 | (uint32_t)row[4 * j + 3] << 24;
#endif
    } else {
      for (int j = 0; j < 24; ++j) {
        uint32_t x = 0;
        for (int b = 0; b < 4; ++b) { const int p = 4 * j + b; x |= (uint32_t)(p < full || (p < A.stride && row[p]) ? row[p] : (uint8_t)'A') << (8 * b); }   // beyond the read: what the row holds (the slot's earlier bases), 0 = never written = code 0
        w[j] = x;
      }
    }
    uint64_t kmer[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      uint64_t k = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t x = w[8 * ch + j];
#pragma unroll
        for (int b = 0; b < 4; ++b) k = (k << 2) | (uint64_t)fq_nt4_fast((x >> (8 * b)) & 0xffu);   // N (4) is OR-ed unmasked, as the reference does
      }
      kmer[ch] = k;
    }
    // The probes are issued before any is consumed (memory-level parallelism).  An off-target read needs the verdict "fewer than
    // thresh hits among 18": with thresh >= 3 that is certain once the first 16 all miss, so the last two are only fetched for the
    // few reads that have a hit by then.  (The probe count reported below stays the reference's.)
    const int th = A.o.filter_thresh;
    uint8_t byte[18];
    uint32_t bit[18], off[18];          // byte offsets into the six tables (their bases are kernel arguments: scalar registers)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const uint32_t x = fq_kmer_project(kmer[ch], t);
        off[6 * ch + t] = x >> 3;
        bit[6 * ch + t] = x & 7u;
      }
#pragma unroll
    for (int q = 0; q < 16; ++q) byte[q] = A.ix.bitmap[q % 6][off[q]];
    byte[16] = byte[17] = 0;
    if (th < 3) { byte[16] = A.ix.bitmap[4][off[16]]; byte[17] = A.ix.bitmap[5][off[17]]; }
    int c16 = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) c16 += (byte[q] >> bit[q]) & 1;
    if (th >= 3 && c16 + 2 >= th) { byte[16] = A.ix.bitmap[4][off[16]]; byte[17] = A.ix.bitmap[5][off[17]]; }
    int cnt[3] = {0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int t = 0; t < 6; ++t) cnt[ch] += (byte[6 * ch + t] >> bit[6 * ch + t]) & 1;
    // early-exit semantics of IsReadInHashByCountMoreChunck: pass as soon as the running count reaches thresh
    const bool p0 = cnt[0] >= th, p1 = cnt[0] + cnt[1] >= th, p2 = cnt[0] + cnt[1] + cnt[2] >= th;
    filt = p2 ? 0 : 1;
    const uint32_t probes = p0 ? 6u : p1 ? 12u : 18u;   // what the reference would have issued (algorithmic count)
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_PROBES], probes);
  }
  A.filtered[r] = filt;
}

// ---- the filter's bitmaps from the reduced reference in HBM (index load) ----------------------------------------------------------------
// BwtIndexer::AddSeq2HashCore (src/BwtIndexer.cpp:96-160) over a flank record and over its reverse complement: every 32-mer of the string goes
// into the six tables; the 32-mers that cover the string's middle position L / 2 are entered twice, with the marker's two alleles there (the
// record's name carries them: CHR:POS@REF/ALT).  One thread per (strand, position of the reference): the 32-mer ENDING at its position.  The
// host lists the bits the same way where a base or an allele is not one of ACGT (the reference draws those from rand(): fq_index.cpp).
struct FqBitmapArgs {
  const uint8_t *pac;        // 2-bit reference, 4 bases per byte, first base in the top bits (bntseq)
  int64_t l_pac;
  const int64_t *rec_off;    // [n_rec + 1] first base of every record; rec_off[n_rec] = l_pac
  const uint8_t *alleles;    // [n_rec] allele codes: a0 | a1 << 2
  int32_t n_rec;
  uint32_t *bitmap[6];       // 2^32 bits each, as 32-bit words (bit x of a table = bit x & 31 of word x >> 5: bit x & 7 of byte x >> 3)
};
template <class SetBit>
FQ_HD void fq_bitmap_kmer_thread(const FqBitmapArgs &A, int64_t idx, SetBit set_bit) {
  const int strand = (int)(idx >= A.l_pac);
  const int64_t q = strand ? idx - A.l_pac : idx;
  int lo = 0, hi = A.n_rec;                                   // the record that holds base q
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (A.rec_off[mid] <= q) lo = mid; else hi = mid; }
  const int64_t r0 = A.rec_off[lo];
  const int64_t L = A.rec_off[lo + 1] - r0, i = q - r0;       // position i of the record's string (of its reverse complement for strand 1)
  if (L < 64 || i < 31) return;
  const int64_t half = L / 2;
  uint64_t d = 0;
  for (int j = 31; j >= 0; --j) {                              // bases i - 31 .. i of the string, the oldest in the top bits
    const int64_t k = i - j;
    const uint32_t c = strand ? 3u - (uint32_t)fq_pac_base(A.pac, r0 + (L - 1 - k)) : (uint32_t)fq_pac_base(A.pac, r0 + k);
    d = (d << 2) | c;
  }
  const bool var = i >= half && i < half + 32;                // the window covers the middle position: once per allele
  const int n_ver = var ? 2 : 1;
  for (int v = 0; v < n_ver; ++v) {
    uint64_t km = d;
    if (var) {
      const int sh = 2 * (int)(i - half);                      // the middle position's base sits (i - half) bases above the newest
      const uint64_t a = (A.alleles[lo] >> (2 * v)) & 3u;
      km = (km & ~((uint64_t)3 << sh)) | (a << sh);
    }
    for (int t = 0; t < 6; ++t) set_bit(t, fq_kmer_project(km, t));
  }
}

// ---- K_prep, packed input: the k-mer filter over the three 32-mers of a read's first 96 bases --------------------------------
// A packed batch (fq_packed_batch_t) carries, for every read, the three 64-bit k-mers exactly as IsReadInHashByCountMoreChunck
// forms them (kmer = kmer << 2 | code over S[32i .. 32i+31], a non-ACGT code OR-ed in unmasked: src/BwtIndexer.cpp:441-456): that
// is the 2-bit packing of the first 96 bases in the filter's own bit order, 24 bytes per read, and all the filter needs.  The
// full reads of the few surviving pairs follow later (fq_unpack_thread).  Layout: head[ch][r], ch = 0..2 (three arrays of n_reads
// words, so that a wavefront's loads are contiguous).
struct FqPrepPackedArgs {
  FqDevIndex ix;
  FqKOpts o;
  const uint64_t *head;    // [3][n_reads]
  const uint16_t *len;     // [n_reads] or NULL: every read has uniform_len bases
  int32_t uniform_len;
  int32_t n_reads;
  uint8_t *filtered;       // out: 1 = filtered
  int32_t *sub_max;        // out: [pair / batch_pairs] longest read (untrimmed) over both ends; only written when len != NULL
  const uint8_t *qual_last;// [n_reads] quality of each read's last base, or NULL
  int32_t *sub_whole;      // out: [pair / batch_pairs] longest read that bwa_trim_read leaves whole (written when qual_last != NULL)
  int32_t n_pairs, batch_pairs;
  uint64_t *counters;      // FQ_C_PROBES, FQ_C_BASES (ragged), FQ_C_BADLEN (ragged)
};
FQ_HD void fq_prep_packed_thread(const FqPrepPackedArgs &A, int r) {
  if (A.qual_last) {
    // bwa_trim_read (libbwa/bwaseqio.c:75-88) scans from the last base while the running sum of (trim_qual - q) stays >= 0: a read
    // whose last base is above the threshold (or that is shorter than 35 bases) keeps its length
    const int len = A.len ? (int)A.len[r] : A.uniform_len;
    const int qsub = (A.o.mode & FQ_MODE_IL13) ? 31 : 0;
    const bool whole = len - 1 < 34 || A.o.trim_qual - ((int)(uint8_t)(A.qual_last[r] - qsub) - 33) < 0;
    const int slot = (r >= A.n_pairs ? r - A.n_pairs : r) / A.batch_pairs;
#if defined(__HIP_DEVICE_COMPILE__)
    int mx = whole ? len : 0;   // one atomic per wavefront and reference batch in the common case
    const int slot0 = __builtin_amdgcn_readfirstlane(slot);
    if (__ballot(slot != slot0) == 0) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
      if (mx > 0 && __lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) FQ_ATOMIC_MAX32(&A.sub_whole[slot0], mx);
    } else if (mx > 0) FQ_ATOMIC_MAX32(&A.sub_whole[slot], mx);
#else
    if (whole && A.sub_whole[slot] < len) A.sub_whole[slot] = len;
#endif
  }
  if (A.len) {
    const int len = A.len[r];
    const int slot = (r >= A.n_pairs ? r - A.n_pairs : r) / A.batch_pairs;
    FQ_ATOMIC_MAX32(&A.sub_max[slot], len);
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_BASES], (uint64_t)len);
    if (len < FQ_LMIN || len > FQ_LMAX) FQ_ATOMIC_ADD64(&A.counters[FQ_C_BADLEN], 1);
  }
  uint8_t filt = 0;
  if (A.o.filter_thresh != 0) {
    uint64_t kmer[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) kmer[ch] = A.head[(size_t)ch * (size_t)A.n_reads + (size_t)r];
    const int th = A.o.filter_thresh;
    uint8_t byte[18];
    uint32_t bit[18], off[18];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const uint32_t x = fq_kmer_project(kmer[ch], t);
        off[6 * ch + t] = x >> 3;
        bit[6 * ch + t] = x & 7u;
      }
    // as in fq_prep_thread: 16 probes in flight, the last two only when they can change the verdict
#pragma unroll
    for (int q = 0; q < 16; ++q) byte[q] = A.ix.bitmap[q % 6][off[q]];
    byte[16] = byte[17] = 0;
    if (th < 3) { byte[16] = A.ix.bitmap[4][off[16]]; byte[17] = A.ix.bitmap[5][off[17]]; }
    int c16 = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) c16 += (byte[q] >> bit[q]) & 1;
    if (th >= 3 && c16 + 2 >= th) { byte[16] = A.ix.bitmap[4][off[16]]; byte[17] = A.ix.bitmap[5][off[17]]; }
    int cnt[3] = {0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int t = 0; t < 6; ++t) cnt[ch] += (byte[6 * ch + t] >> bit[6 * ch + t]) & 1;
    const bool p0 = cnt[0] >= th, p1 = cnt[0] + cnt[1] >= th, p2 = cnt[0] + cnt[1] + cnt[2] >= th;
    filt = p2 ? 0 : 1;
    const uint32_t probes = p0 ? 6u : p1 ? 12u : 18u;
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t ps = probes;   // one atomic per wavefront
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) ps += (uint32_t)__shfl_xor((int)ps, d);
    if (__lane_id() == (unsigned)(__ffsll((long long)__ballot(1)) - 1)) FQ_ATOMIC_ADD64(&A.counters[FQ_C_PROBES], ps);
#else
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_PROBES], probes);
#endif
  }
  A.filtered[r] = filt;
}

// ---- survivors of a packed batch: 2-bit rows -> the ASCII rows every later kernel reads --------------------------------------------
// Compact row t = 2*sp + e (survivor pair sp, end e).  Its packed row is body[src(t)], src(t) = row_map ? row_map[t] : t:
// the host either gathered the survivors' rows into a staging buffer (few survivors: identity map) or uploaded the whole batch
// (row_map = the read's row in the batch).  Base i of a row sits in byte i>>2 at bit 2*(i&3); a non-ACGT base holds 0 there and is
// listed in the exception array, applied afterwards by fq_patch_thread.
struct FqUnpackArgs {
  const uint8_t *body;
  int32_t body_stride;
  const int32_t *row_map;   // NULL: identity
  const uint16_t *len;      // indexed like body rows; NULL: uniform_len
  int32_t uniform_len;
  int32_t n_rows;           // compact rows
  uint8_t *seq;             // out: [n_rows][stride] ASCII
  int32_t stride;
  int32_t *len_out;         // out: [n_rows] read length (also the untrimmed len_trim)
  int32_t *len_trim;        // out: [n_rows]
};
// One 16-byte piece of the output: bases [16 c, 16 c + 16) of compact row t = four bytes of the packed row.  The output rows are
// dense (stride a multiple of 16, fq_align.cpp), so consecutive pieces are consecutive 16-byte stores: the kernel writes whole lines
// (a thread per row wrote its row byte by byte, 64 rows per store instruction: 14.8 ms for 8.4 M rows against 1.6 ms).
FQ_HD void fq_unpack_piece(const FqUnpackArgs &A, int64_t g) {
  const int per_row = A.stride >> 4;
  const int t = (int)(g / per_row), cidx = (int)(g - (int64_t)t * per_row);
  const size_t src = A.row_map ? (size_t)A.row_map[t] : (size_t)t;
  const int len = A.len ? (int)A.len[src] : A.uniform_len;
  const uint8_t *b = A.body + src * (size_t)A.body_stride + 4 * (size_t)cidx;
  const int p0 = 16 * cidx;
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) if (p0 + 4 * j < len) v |= (uint32_t)b[j] << (8 * j);     // (bytes past the row's last base are not read)
  FqU4 o;
  uint32_t w[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int p = p0 + 4 * q + j;
      const uint32_t code = (v >> (8 * q + 2 * j)) & 3u;
      const uint32_t ch = p < len ? (0x54474341u >> (8 * code)) & 0xffu : 0u;            // "ACGT"[code]
      x |= ch << (8 * j);
    }
    w[q] = x;
  }
  o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
  *(FqU4 *)(A.seq + (size_t)t * (size_t)A.stride + (size_t)p0) = o;
  if (cidx == 0) { A.len_out[t] = len; A.len_trim[t] = len; }
}
FQ_HD void fq_unpack_thread(const FqUnpackArgs &A, int t) {      // a whole row (host-loop build)
  const int per_row = A.stride >> 4;
  for (int cidx = 0; cidx < per_row; ++cidx) fq_unpack_piece(A, (int64_t)t * per_row + cidx);
}
// exception e: row << 32 | pos << 8 | code (4: 'N', any other letter; 5: '-'); row is a batch row, mapped through crow_of (compact
// row or -1) when given, else already a compact row
struct FqPatchArgs {
  const uint64_t *exc;
  int64_t n_exc;
  const int32_t *crow_of;
  uint8_t *seq;
  int32_t stride;
};
FQ_HD void fq_patch_thread(const FqPatchArgs &A, int64_t q) {
  const uint64_t e = A.exc[q];
  int64_t row = (int64_t)(e >> 32);
  if (A.crow_of) row = A.crow_of[row];
  if (row < 0) return;
  const int pos = (int)((e >> 8) & 0xffffu), code = (int)(e & 0xffu);
  if (pos < A.stride) A.seq[(size_t)row * (size_t)A.stride + (size_t)pos] = (uint8_t)(code == 5 ? '-' : 'N');
}
// bwa_trim_read (libbwa/bwaseqio.c:75-88) for the rows that were unpacked; qual rows are indexed like body rows
struct FqTrimArgs {
  FqKOpts o;
  const uint8_t *qual;
  int32_t qual_stride;
  const int32_t *row_map;   // compact row -> qual row (NULL: identity)
  const int32_t *len;       // [n_rows] compact
  int32_t n_rows;
  int32_t *len_trim;        // out [n_rows]
  const int32_t *pair_list; // compact row t belongs to survivor pair t >> 1, pair pair_list[t >> 1] of the batch
  int32_t batch_pairs;
  int32_t *sub_max;         // out (zeroed by the caller): longest trimmed survivor per reference batch
};
FQ_HD int fq_trim_len(const FqKOpts &o, const uint8_t *q, int full) {
  int s = 0, mx = 0, max_l = full - 1;
  const int qsub = (o.mode & FQ_MODE_IL13) ? 31 : 0;
  for (int l = full - 1; l >= 34; --l) {
    s += o.trim_qual - ((int)(uint8_t)(q[l] - qsub) - 33);
    if (s < 0) break;
    if (s > mx) { mx = s; max_l = l; }
  }
  return max_l + 1;
}
FQ_HD void fq_trim_thread(const FqTrimArgs &A, int t) {
  const size_t src = A.row_map ? (size_t)A.row_map[t] : (size_t)t;
  const int lt = fq_trim_len(A.o, A.qual + src * (size_t)A.qual_stride, A.len[t]);
  A.len_trim[t] = lt;
  if (A.sub_max) FQ_ATOMIC_MAX32(&A.sub_max[A.pair_list[t >> 1] / A.batch_pairs], lt);
}
// the same over every read of a batch (debug dumps, and the rare call in which no surviving read of some reference batch kept its
// full length, so that the batch's longest trimmed read may be a filtered one): len_trim[r] for all rows, sub_max per reference batch
struct FqTrimAllArgs {
  FqKOpts o;
  const uint8_t *qual;
  int32_t qual_stride;
  const uint16_t *len;      // NULL: uniform_len
  int32_t uniform_len;
  int32_t n_reads, n_pairs, batch_pairs;
  int32_t *len_trim;        // out [n_reads]
  int32_t *sub_max;         // out (zeroed by the caller)
};
FQ_HD void fq_trim_all_thread(const FqTrimAllArgs &A, int r) {
  const int full = A.len ? (int)A.len[r] : A.uniform_len;
  const int lt = fq_trim_len(A.o, A.qual + (size_t)r * (size_t)A.qual_stride, full);
  A.len_trim[r] = lt;
  FQ_ATOMIC_MAX32(&A.sub_max[(r >= A.n_pairs ? r - A.n_pairs : r) / A.batch_pairs], lt);
}

// ---- what the host needs to know about the reads of surviving pairs (everything else stays on the device) ----
struct FqSurvInfo { int32_t len_trim, filtered, sidx; };
FQ_HD void fq_surv_gather_thread(const int32_t *pair_list, int n_pairs, const int32_t *len_trim, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out, int t) {
  const int sp = t >> 1, e = t & 1;
  const int r = e * n_pairs + pair_list[sp];
  FqSurvInfo v;
  v.len_trim = len_trim[r]; v.filtered = filtered[r]; v.sidx = sidx[r];
  out[t] = v;
}

// packed batches: the same for compact rows -- out[t].len_trim is filled in later; row_map[t] = the read's row in the batch,
// read_list_c[s] = t for searched reads (search index -> compact row), crow_of[r] = t when the whole batch was uploaded
FQ_HD void fq_surv_map_thread(const int32_t *pair_list, int n_pairs, const uint8_t *filtered, const int32_t *sidx, FqSurvInfo *out,
                              int32_t *row_map, int32_t *read_list_c, int32_t *crow_of, int t) {
  const int sp = t >> 1, e = t & 1;
  const int r = e * n_pairs + pair_list[sp];
  FqSurvInfo v;
  v.len_trim = 0; v.filtered = filtered[r]; v.sidx = sidx[r];
  out[t] = v;
  row_map[t] = r;
  if (v.sidx >= 0) read_list_c[v.sidx] = t;
  if (crow_of) crow_of[r] = t;
}

// ---- read access helpers: seq[0] is the reversed read, seq[1] its complement (reverse complement
// of the read), exactly the two arrays bwa_read_seq_with_hash_dev leaves in p->seq / p->rseq ------
struct FqReadView {
  const uint8_t *row;
  int len;
};
FQ_HD int fq_base(const FqReadView &v, int a, int i) {
  const int c = (int)fq_nt4_fast(v.row[v.len - 1 - i]);
  return (a && c < 4) ? 3 - c : c;
}

// ---- K_width: bwt_cal_width (libbwa/bwtaln.c:73-97), one thread per read: the seed chains, then the full chains, of both strands ----
// Outputs are written eight positions at a time (one 16-byte store of position records, two of widths): a thread's rows are
// private, so narrower stores would reach HBM as partial lines.
struct FqWidthArgs {
  FqDevIndex ix;
  FqKOpts o;
  const uint8_t *seq;
  int32_t stride;
  const int32_t *len_trim;
  const int32_t *read_list;   // s -> r
  const int32_t *work;        // w -> s (NULL: identity)
  int32_t n_work;
  uint32_t *wfull;            // [w][2][wstride]   width[a][p].w, exact (gap_shadow needs it); wstride % 8 == 0
  int32_t wstride;
  FqPos *prec;                // [w][2][pstride]   packed per-position records the search loop reads; pstride % 8 == 0
  int32_t pstride;
  FqGapWork *winfo;           // [w] what the search kernel needs to start read w (one 8-byte load)
  const uint8_t *maxdiff_lut; // [len] -> max_diff (bwa_cal_maxdiff, libbwa/bwtaln.c:43-58)
  uint8_t *bid_end;           // [w][2] lower bound on the differences of the whole read, per strand (width[len-1].bid): scheduling hint
  uint64_t *counters;
};
// One thread walks both strands of its read: the two chains (and the two seed chains before them) are independent, so every step
// has two Occ requests in flight instead of one -- the kernel is bound by the latency of a dependent step, not by requests.
// seed_bits[(strand * seed_len + ii) * seed_bits_stride]: 2 * seed_len bytes of thread-private scratch (LDS on the device)
FQ_HD void fq_width_read(const FqWidthArgs &A, int w, uint8_t *seed_bits, int seed_bits_stride) {
  const int s = A.work ? A.work[w] : w;
  const int r = A.read_list[s];
  FqReadView v = {A.seq + (size_t)r * (size_t)A.stride, A.len_trim[r]};
  uint32_t *ow = A.wfull + (size_t)w * 2 * (size_t)A.wstride;
  FqPos *prec = A.prec + (size_t)w * 2 * (size_t)A.pstride;
  uint32_t touches = 0;
  const bool use_seed = v.len > A.o.seed_len;
  const int seed_off = v.len - A.o.seed_len;
  uint32_t k[2], l[2], wprev[2];
  int bid[2];
  if (use_seed) {   // bwt_cal_width over the last seed_len bases (src/BwtMapper.cpp:131-137)
#pragma unroll
    for (int a = 0; a < 2; ++a) { k[a] = 0; l[a] = A.ix.fm[a].seq_len; wprev[a] = 0; bid[a] = 0; }
    uint64_t seed8 = 0;
    for (int i = 0; i < A.o.seed_len; ++i) {
      // seed position i is byte seed_len-1-i of the row: eight positions per 8-byte load, as below
      if ((i & 7) == 0 && i + 8 <= A.o.seed_len) memcpy(&seed8, v.row + (A.o.seed_len - 8 - i), 8);
      const bool fast = (i | 7) < A.o.seed_len;
      const int c0 = fast ? (int)fq_nt4_fast((uint32_t)(seed8 >> (8 * (7 - (i & 7)))) & 0xffu) : 0;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const FqFM &f = A.ix.fm[a];
        int c;
        if (fast) c = (a && c0 < 4) ? 3 - c0 : c0;
        else c = fq_base(v, a, seed_off + i);
        if (c < 4) {
          touches += fq_touch2(f, k[a] - 1, l[a], true);
          uint32_t ok, ol;
          fq_occ1_pair(f, k[a] - 1, l[a], c, &ok, &ol);
          k[a] = f.L2[c] + ok + 1;
          l[a] = f.L2[c] + ol;
        }
        if (k[a] > l[a] || c > 3) { k[a] = 0; l[a] = f.seq_len; ++bid[a]; }
        const uint32_t wcur = l[a] - k[a] + 1;
        seed_bits[(a * A.o.seed_len + i) * seed_bits_stride] = (uint8_t)((uint32_t)(bid[a] < 31 ? bid[a] : 31) | (i >= 1 && wcur == wprev[a] ? 1u << 5 : 0u));
        wprev[a] = wcur;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) { k[a] = 0; l[a] = A.ix.fm[a].seq_len; wprev[a] = 0; bid[a] = 0; }
  int namb = 0;
  for (int i0 = 0; i0 < v.len; i0 += 8) {
    uint32_t wv[2][8], pv[2][8];
    // the eight bases of this group sit in eight consecutive bytes of the row (the search walks the read backwards): one load
    uint64_t bases8 = 0;
    const bool whole = i0 + 8 <= v.len;
    if (whole) memcpy(&bases8, v.row + (v.len - 8 - i0), 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j;
      const int c0 = whole ? (int)fq_nt4_fast((uint32_t)(bases8 >> (8 * (7 - j))) & 0xffu) : 0;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        wv[a][j] = 0; pv[a][j] = 0;
        if (i < v.len) {
          const FqFM &f = A.ix.fm[a];
          int c;
          if (whole) c = (a && c0 < 4) ? 3 - c0 : c0;
          else c = fq_base(v, a, i);
          if (a == 0) namb += c > 3;
          if (c < 4) {
            touches += fq_touch2(f, k[a] - 1, l[a], true);
            uint32_t ok, ol;
            fq_occ1_pair(f, k[a] - 1, l[a], c, &ok, &ol);
            k[a] = f.L2[c] + ok + 1;
            l[a] = f.L2[c] + ol;
          }
          if (k[a] > l[a] || c > 3) { k[a] = 0; l[a] = f.seq_len; ++bid[a]; }
          const uint32_t wcur = l[a] - k[a] + 1;
          const uint32_t seedbits = (use_seed && i >= seed_off) ? (uint32_t)seed_bits[(a * A.o.seed_len + (i - seed_off)) * seed_bits_stride] << 6 : 0u;
          wv[a][j] = wcur;
          pv[a][j] = seedbits | (uint32_t)(bid[a] < 31 ? bid[a] : 31) | (i >= 1 && wcur == wprev[a] ? 1u << 5 : 0u) | (uint32_t)c << 12;
          wprev[a] = wcur;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uint32_t *owa = ow + (size_t)a * (size_t)A.wstride;
      FqPos *pra = prec + (size_t)a * (size_t)A.pstride;
      FqU4 q;
      q.x = wv[a][0]; q.y = wv[a][1]; q.z = wv[a][2]; q.w = wv[a][3];
      *(FqU4 *)(owa + i0) = q;
      q.x = wv[a][4]; q.y = wv[a][5]; q.z = wv[a][6]; q.w = wv[a][7];
      *(FqU4 *)(owa + i0 + 4) = q;
      q.x = pv[a][0] | pv[a][1] << 16; q.y = pv[a][2] | pv[a][3] << 16; q.z = pv[a][4] | pv[a][5] << 16; q.w = pv[a][6] | pv[a][7] << 16;
      *(FqU4 *)(pra + i0) = q;
    }
  }
  A.bid_end[2 * w] = (uint8_t)(bid[0] < 255 ? bid[0] : 255);
  A.bid_end[2 * w + 1] = (uint8_t)(bid[1] < 255 ? bid[1] : 255);
  const uint32_t md = A.maxdiff_lut[v.len];
  FqGapWork gw;
  gw.r = r;
  gw.meta = (uint32_t)v.len | md << 16 | (namb > (int)md ? 1u << 24 : 0u);
  A.winfo[w] = gw;
  FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_WIDTH], touches);
}

// The same for ONE strand of the read (the device's default since round 4): the two strands walk two different Occ tables of 3.3 MB each
// (10k markers) and an XCD's L2 holds 4 MB -- a thread that interleaves both chains makes every XCD cache both tables and miss in half of
// its lookups.  One thread per (read, strand), the threads of strand a in workgroups that share XCDs (fq_device.hip, k_width_strand):
// each XCD's L2 then serves one table.  Same arithmetic, same outputs; strand 0's thread also writes the read's start record.
FQ_HD void fq_width_strand(const FqWidthArgs &A, int w, int a, uint8_t *seed_bits, int seed_bits_stride) {
  const int s = A.work ? A.work[w] : w;
  const int r = A.read_list[s];
  FqReadView v = {A.seq + (size_t)r * (size_t)A.stride, A.len_trim[r]};
  uint32_t *owa = A.wfull + (size_t)w * 2 * (size_t)A.wstride + (size_t)a * (size_t)A.wstride;
  FqPos *pra = A.prec + (size_t)w * 2 * (size_t)A.pstride + (size_t)a * (size_t)A.pstride;
  const FqFM &f = A.ix.fm[a];
  uint32_t touches = 0;
  const bool use_seed = v.len > A.o.seed_len;
  const int seed_off = v.len - A.o.seed_len;
  uint32_t k = 0, l = f.seq_len, wprev = 0;
  int bid = 0;
  if (use_seed) {   // bwt_cal_width over the last seed_len bases (src/BwtMapper.cpp:131-137)
    uint64_t seed8 = 0;
    for (int i = 0; i < A.o.seed_len; ++i) {
      if ((i & 7) == 0 && i + 8 <= A.o.seed_len) memcpy(&seed8, v.row + (A.o.seed_len - 8 - i), 8);
      const bool fast = (i | 7) < A.o.seed_len;
      int c;
      if (fast) { const int c0 = (int)fq_nt4_fast((uint32_t)(seed8 >> (8 * (7 - (i & 7)))) & 0xffu); c = (a && c0 < 4) ? 3 - c0 : c0; }
      else c = fq_base(v, a, seed_off + i);
      if (c < 4) {
        touches += fq_touch2(f, k - 1, l, true);
        uint32_t ok, ol;
        fq_occ1_pair(f, k - 1, l, c, &ok, &ol);
        k = f.L2[c] + ok + 1;
        l = f.L2[c] + ol;
      }
      if (k > l || c > 3) { k = 0; l = f.seq_len; ++bid; }
      const uint32_t wcur = l - k + 1;
      seed_bits[i * seed_bits_stride] = (uint8_t)((uint32_t)(bid < 31 ? bid : 31) | (i >= 1 && wcur == wprev ? 1u << 5 : 0u));
      wprev = wcur;
    }
  }
  k = 0; l = f.seq_len; wprev = 0; bid = 0;
  int namb = 0;
  for (int i0 = 0; i0 < v.len; i0 += 8) {
    uint32_t wv[8], pv[8];
    uint64_t bases8 = 0;
    const bool whole = i0 + 8 <= v.len;
    if (whole) memcpy(&bases8, v.row + (v.len - 8 - i0), 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j;
      wv[j] = 0; pv[j] = 0;
      if (i < v.len) {
        int c;
        if (whole) { const int c0 = (int)fq_nt4_fast((uint32_t)(bases8 >> (8 * (7 - j))) & 0xffu); c = (a && c0 < 4) ? 3 - c0 : c0; }
        else c = fq_base(v, a, i);
        namb += c > 3;
        if (c < 4) {
          touches += fq_touch2(f, k - 1, l, true);
          uint32_t ok, ol;
          fq_occ1_pair(f, k - 1, l, c, &ok, &ol);
          k = f.L2[c] + ok + 1;
          l = f.L2[c] + ol;
        }
        if (k > l || c > 3) { k = 0; l = f.seq_len; ++bid; }
        const uint32_t wcur = l - k + 1;
        const uint32_t seedbits = (use_seed && i >= seed_off) ? (uint32_t)seed_bits[(i - seed_off) * seed_bits_stride] << 6 : 0u;
        wv[j] = wcur;
        pv[j] = seedbits | (uint32_t)(bid < 31 ? bid : 31) | (i >= 1 && wcur == wprev ? 1u << 5 : 0u) | (uint32_t)c << 12;
        wprev = wcur;
      }
    }
    FqU4 q;
    q.x = wv[0]; q.y = wv[1]; q.z = wv[2]; q.w = wv[3];
    *(FqU4 *)(owa + i0) = q;
    q.x = wv[4]; q.y = wv[5]; q.z = wv[6]; q.w = wv[7];
    *(FqU4 *)(owa + i0 + 4) = q;
    q.x = pv[0] | pv[1] << 16; q.y = pv[2] | pv[3] << 16; q.z = pv[4] | pv[5] << 16; q.w = pv[6] | pv[7] << 16;
    *(FqU4 *)(pra + i0) = q;
  }
  A.bid_end[2 * w + a] = (uint8_t)(bid < 255 ? bid : 255);
  if (a == 0) {
    const uint32_t md = A.maxdiff_lut[v.len];
    FqGapWork gw;
    gw.r = r;
    gw.meta = (uint32_t)v.len | md << 16 | (namb > (int)md ? 1u << 24 : 0u);
    A.winfo[w] = gw;
  }
  FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_WIDTH], touches);
}

// Scheduling key for the search kernel: the smaller of the two strands' lower bounds on the number of differences, and which
// strand it belongs to.  A read whose bound is 0 has an exact match and a search of nearly constant shape (~180 pops at 150 bp);
// mean and tail of the search length grow with the bound.  The queue hands out reads sorted by key and a wavefront refills all
// its lanes at once (FQ_REFILL_MIN 64), so the 64 searches of a wavefront start together and -- being alike -- stay in step:
// every divergent path of the loop is then executed for many lanes or not at all.  (The kernel is instruction-issue bound; on
// the on-target workload this ordering took it from 94 to 62 ms.)  Descending order: the long, irregular searches run while
// the device is full, the launch ends on the uniform ones.
// The key's top bit is the better strand: the queue is two blocks, reads that will walk mostly the reverse BWT and reads that will
// walk mostly the forward one (strand a searches bwt[1 - a]), each sorted by bound.  The wavefronts of half of the XCDs draw from
// one block first, the others from the other (fq_gap_lanes): an XCD's 4 MB L2 then serves mostly one 3.3 MB Occ table instead of
// being shared by two.
// Second round of a device-filling call (the reads the round without gap children could not settle): the first n_hard work items
// are the reads that round left WITHOUT a hit below s_gapo (FQ_SF_NOHIT).  Their full searches are the long ones -- three quarters
// of the round's pops, up to 5.5 k pops each, against at most 1.2 k for a read that already has a three-mismatch hit -- and nothing
// the width kernel knows tells them apart, so the class leads the key inside each strand block: every long search starts when the
// launch does and the launch ends on the short ones, instead of lasting as long as a late long search.
#define FQ_ORDER_KEYS 32
FQ_HD int fq_order_key(const uint8_t *bid_end, int w, int n_hard) {
  const int a = bid_end[2 * w], b = bid_end[2 * w + 1];
  int k = a < b ? a : b;
  if (k > FQ_ORDER_KEYS / 4 - 1) k = FQ_ORDER_KEYS / 4 - 1;
  return (b <= a ? 0 : FQ_ORDER_KEYS / 2) + (w < n_hard ? FQ_ORDER_KEYS / 4 : 0) + k;
}

// ---- K_gap: bwt_match_gap (libbwa/bwtgap.c:104-264) --------------------------------------------
// One lane runs the whole bounded best-first search of one read.  The score-bucketed LIFO stack
// (gap_stack_t, bwtgap.c:13-79) is a per-read pool of 16-byte entries in HBM with one linked
// chain per score bucket; a 128-bit occupancy mask in registers replaces the "scan for the next
// non-empty bucket" loop of gap_pop.  Pop order is identical to the reference's.
//
// Non-exact tiers drop, at push time, children that the reference would pop only to discard
// (score > best_score + s_mm once a hit exists, or more differences than max_diff can ever allow);
// both conditions are monotone, so the sequence of processed entries is unchanged.  The dropped
// children still count in n_live, which therefore over-estimates stack->n_entries: if it ever
// crosses max_entries the read is flagged and re-run in the exact tier, which keeps every entry.
struct FqGapArgs {
  FqDevIndex ix;
  FqKOpts o;
  int32_t n_work;
  const FqGapWork *winfo; // [w] read index, length, max_diff, too-many-N flag: written by k_width
  uint32_t *wfull;       // exact widths written by k_width; read and updated only by gap_shadow
  int32_t wstride;
  FqPos *prec;           // packed per-position records (fq_common.h): all the search loop reads about the read
  int32_t pstride;
  FqEntry *pool;         // [n_lane_slots][pool_cap]: one stack pool per persistent lane
  uint32_t *heads;       // [n_lane_slots][FQ_MAX_BUCKETS] (HBM-heads variant)
  FqGapTier tier;
  FqAln *aln;
  uint32_t *n_aln;
  uint32_t *status;
  uint64_t *counters;
  const int32_t *order;  // queue position -> work item (NULL: identity): two blocks by better strand, long searches first (fq_order_key)
  const uint32_t *split; // *split = length of the first block of `order` (NULL: one block)
  uint32_t *queue;       // work-queue cursors, one per block (zeroed before each launch)
  int32_t seg, n_seg;    // n_seg > 0: this launch takes only segment `seg` of `n_seg` of each queue block (fq_seg_range); the blocks are sorted,
                         // long searches first, so the early segments hold the hard reads
  int32_t refill_min;    // idle lanes of a wavefront wait until this many can be (re)initialised together
  int32_t max_waves;     // 0: as many wavefronts as the device holds; else a cap (the host lowers it when pool memory is short)
  const uint8_t *bid_end; // [w][2] k_width's lower bound on the differences of each strand (the sort key's source); used with skip_bound
  int32_t skip_bound;    // round without gap children: a read whose bound on BOTH strands is at least this cannot be settled there (its best
                         // possible hit scores skip_bound * s_mm, and (skip_bound + 1) * s_mm >= s_gapo): it is flagged for the full search
                         // at once instead of being searched twice.  0: every read is searched.  (Any choice is exact: the full search decides.)
                         // Measured (profiles/round4_search_loop_experiments.txt): first round 39.9 -> 37.6 ms at 8.4 M reads, but the next round
                         // loses the first round's knowledge of which of these reads have a hit at all (its long searches no longer all start
                         // first): 21.5 -> 24.0 ms, stage 61.4 -> 61.9; at 2.1 M reads per call the stage goes 26.4 -> 24.6 ms.
};

// positions [lo, hi) of a queue block of `len` entries that belong to segment seg of n_seg
FQ_HD void fq_seg_range(uint32_t len, int seg, int n_seg, uint32_t *lo, uint32_t *hi) {
  if (n_seg <= 0) { *lo = 0; *hi = len; return; }
  *lo = (uint32_t)((uint64_t)len * (uint64_t)seg / (uint64_t)n_seg);
  *hi = (uint32_t)((uint64_t)len * (uint64_t)(seg + 1) / (uint64_t)n_seg);
}
// Where a read's bucket heads live during the search: HBM (any pool size) or lane-interleaved LDS with 16-bit slots
// (pool <= 65535 entries), which takes the head read-modify-write of every push off the global-memory latency chain.
// The lane's seldom-touched state lives beside the heads (FqGapLane: store.cold(i)): a register each in the HBM-heads variant,
// lane-interleaved LDS words in the other.
enum { FQ_COLD_TPOPS, FQ_COLD_TPUSHES, FQ_COLD_TTOUCH, FQ_COLD_TMAXPOPS, FQ_COLD_TGT4K, FQ_COLD_W, FQ_COLD_MAXDIFF_OPT, FQ_COLD_BEST_CNT, FQ_COLD_N };
struct FqGapStoreGlobal {
  uint32_t *head;
  uint32_t cold_[FQ_COLD_N];
  FQ_HD uint32_t &cold(int i) { return cold_[i]; }
  FQ_HD void begin_lane(const FqGapArgs &A, int lane_slot) { head = A.heads + (size_t)lane_slot * FQ_MAX_BUCKETS; }
  FQ_HD uint32_t head_get(int b) const { return head[b]; }
  FQ_HD void head_set(int b, uint32_t slot) const { head[b] = slot; }
};
struct FqGapStoreLds {
  uint16_t *head;       // element b at head[b*stride]
  int stride;
  uint32_t *coldp;      // word i at coldp[i*stride]
  FQ_HD uint32_t &cold(int i) const { return coldp[i * stride]; }
  FQ_HD void begin_lane(const FqGapArgs &, int) {}
  FQ_HD uint32_t head_get(int b) const { return head[b * stride]; }
  FQ_HD void head_set(int b, uint32_t slot) const { head[b * stride] = (uint16_t)slot; }
};

// wavefront-level helpers.  The host-loop build (tests/emu) runs one lane at a time: a "wavefront" of one.
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_WAVE_SIZE 64
#define FQ_LANE_ID() ((int)__lane_id())
#define FQ_BALLOT(pred) ((uint64_t)__ballot(pred))
#define FQ_READLANE32(x, l) ((uint32_t)__builtin_amdgcn_readlane((int)(x), (l)))
#define FQ_CTZ64(x) (__ffsll((long long)(x)) - 1)
#define FQ_SHFL_UP1(x) ((uint32_t)__shfl_up((int)(x), 1))
#else
#define FQ_WAVE_SIZE 1
#define FQ_LANE_ID() 0
#define FQ_BALLOT(pred) ((uint64_t)((pred) ? 1 : 0))
#define FQ_READLANE32(x, l) ((uint32_t)(x))
#define FQ_CTZ64(x) __builtin_ctzll(x)
#define FQ_SHFL_UP1(x) ((uint32_t)(x))
#endif
#define FQ_REFILL_MIN 64  // idle lanes of a wavefront wait until this many can be (re)initialised together (64: whole wavefront)
// test-only instrumentation hooks (tests/emu builds may define FQ_PROFILE; empty in the product)
#if defined(FQ_PROFILE) && !defined(__HIP_DEVICE_COMPILE__)
extern unsigned long long fq_prof[128];
extern int fq_prof_off;   // 0: full search, 32: the round without gap children, 64: wavefront-per-read kernel
#define FQ_PROF(i) (++fq_prof[fq_prof_off + (i)])
#else
#define FQ_PROF(i) ((void)0)
#endif


// One lane = one search at a time; a lane that finishes pulls the next read from a queue, so a wavefront stays busy
// whatever the spread of search lengths inside its packet of 64 reads.
//
// A loop iteration ("trip") costs every lane exactly one global-memory round trip, and all loads of a trip are issued
// before any is consumed:
//   * a lane that holds a current entry in registers (the exact-match child of the entry it expanded in the previous trip
//     -- the reference pushes that child last, into the bucket its parent just left, so it is always the next one popped --
//     or the running state of an exact-match tail, bwt_match_exact_alt as a state machine) fetches the two 32-byte Occ
//     blocks of rows k-1 and l together with its width records, ranks all four bases once, and from the result takes the
//     match child (next current entry) and, unless it is in a tail, the gap / mismatch children;
//   * a lane without a current entry pops one: bucket head from LDS, 16-byte entry from its pool in HBM.
// Children are written per group (gap children, mismatch children): the members of a group share their score bucket, the
// push-time prune decision and all of the packed state word except position and state, so a child costs one slot
// allocation and one 16-byte store; the bucket head and occupancy mask are updated once per group.
// Completed alignments are collected in a wavefront-uniform section after the step: gap_shadow's sweep over the width
// array (up to read-length dependent loads and stores in the reference's loop) is done by all 64 lanes together.
//
// The per-read state is a plain struct with inlined member functions; per-strand index fields are chosen with masks:
// closures capturing by reference, "cond ? mem_a : mem_b", or a runtime index into the kernel-argument struct make hipcc
// keep the whole state in scratch memory (680 B/lane, every access a memory round trip).
// NOGAP: the first round of a device-filling launch searches without the gap children.  Children of a gap cost at least s_gapo
// more than their parent, so every entry the search pops from a bucket below s_gapo is an M-state entry reached by matches and
// mismatches only, and those buckets hold the same entries in the same order whether gap children are pushed or not.  If the
// first alignment is found there and best_score + s_mm < s_gapo, the reference stops (bwtgap.c:147) before it could pop a gap
// child: the hit list is complete -- for a 150 bp read with up to two mismatches, nine reads in ten of an on-target set.  The
// gap children it would have pushed in the meantime (five per step at most) only count towards max_entries, so they are added to
// n_live; any other read (no hit below s_gapo, a pop from a bucket >= s_gapo, non-stop mode) is flagged FQ_SF_NEEDGAP and
// searched again, in full, by the next round.  The step of this round has no gap group: three quarters of the pushes and a
// third of the instructions of the first walk down a read are gone.
// Opt: where the lane reads the search options from.  FqKOpts: the launch's option block (any options).  FqOptsStock: FASTQuick's own
// option block (fq_default_opts; BwtMapper's defaults) as compile-time constants -- the command line has no switch for most of
// them, so this is the kernel a FASTQuick run executes.  The options then cost no scalar registers (the generic kernel keeps ~100
// uniform values alive across its loop and spills 70-90 of them to vector-register lanes) and their tests fold away.
struct FqOptsStock {
  static constexpr int32_t s_mm = 3, s_gapo = 11, s_gape = 4, mode = FQ_MODE_GAPE | FQ_MODE_COMPREAD;
  static constexpr int32_t indel_end_skip = 5, max_del_occ = 10, max_entries = 2000000;
  static constexpr int32_t max_gapo = 1, max_gape = 6, max_seed_diff = 2, seed_len = 32, max_top2 = 30;
  FQ_HD FqOptsStock() {}
  FQ_HD FqOptsStock(const FqKOpts &) {}
  static bool matches(const FqKOpts &k) {   // host side: may this launch use the stock kernel?
    return k.s_mm == s_mm && k.s_gapo == s_gapo && k.s_gape == s_gape && (k.mode & (FQ_MODE_GAPE | FQ_MODE_LOGGAP | FQ_MODE_NONSTOP)) == (mode & (FQ_MODE_GAPE | FQ_MODE_LOGGAP | FQ_MODE_NONSTOP)) &&
           k.indel_end_skip == indel_end_skip && k.max_del_occ == max_del_occ && k.max_entries == max_entries && k.max_gapo == max_gapo && k.max_gape == max_gape &&
           k.max_seed_diff == max_seed_diff && k.seed_len == seed_len && k.max_top2 == max_top2;
  }
};
// In the round without gap children only buckets below s_gapo + s_mm are ever occupied: with the stock options one mask word.
template <class O> struct FqNogapOneWord { static constexpr bool value = false; };
template <> struct FqNogapOneWord<FqOptsStock> { static constexpr bool value = FqOptsStock::s_gapo + FqOptsStock::s_mm <= 32; };
template <class St, bool NOGAP = false, class Opt = FqKOpts>
struct FqGapLane {
  static constexpr bool kOneWord = NOGAP && FqNogapOneWord<Opt>::value;
  const FqGapArgs &A;
  St store;
  Opt o;
  bool gape_mode, nonstop;
  static constexpr bool exact = false;   // the exact tier (no push-time pruning) is the wavefront kernel's: launch_gap refuses it here
  uint32_t seq_len, L2_0, L2_1, L2_2, L2_3;          // identical for both strands (checked at index load)
  const FqOccBlk *blk_s0;                             // the table strand 0 walks (the reversed text's: bwtgap.c:148)
  ptrdiff_t blk_delta;                                // strand 1's table, in bytes from blk_s0
  uint32_t primary0, primary1;
  // per-read constants
  bool active, done;
  int len;                                            // (position ii = i - (len - seed_len) inside the seed)
  // (the work item, max_diff as the options give it, best_cnt and the lane's totals are touched once or twice per read: they live with
  //  the bucket heads -- store.cold(FQ_COLD_*) -- and leave their registers to the loop)
  FqEntry *pool;
  uint32_t pstep = 1;                                 // pool slot s at pool[s * pstep]
  FqPos *prec;                                        // this read's position records, strand 0 (strand 1 at + pstride)
  uint32_t pw0, pw1, pw2, pw3;                        // window of eight position records, positions [wbase, wbase + 8)
  int wbase;
  // search state
  uint32_t m0, m1, m2, m3, bump, spare, status, n_aln;
  int32_t n_live;
  int best_score, max_diff;
  uint32_t c_pops, c_pushes, c_touch;
  // current entry (valid when has_cur or hit_pending)
  bool has_cur, tail, hit_pending, too_many_n;
  uint32_t ck_, cl_, cpk;
  int cscore;
  // The entry below the one popped last (its `next`), fetched ahead: consecutive pops of a bucket find their entry in registers,
  // cost no trip of their own and are expanded at once.  A pool slot keeps its content for as long as it is on the stack, so the
  // copy is good whenever the bucket's head still names that slot.
  // (The round without gap children does it too since its kernel fits 128 VGPRs with them: 33 of a read's 36 pops there are consecutive
  // pops of one bucket -- the mismatch children of the walk down the read -- and each was a trip of its own: 42.0 -> 39.9 ms.)
  // (Fetching the entry's window of position records ahead as well was tried: loads return in order, so the extra request only
  // moves the wait into the next trip -- 19.1 -> 21.7 ms for the second round of an on-target call.)
  uint32_t pf_slot = FQ_NIL, pfk = 0, pfl = 0, pfpk = 0, pfnext = 0;
#if defined(FQ_GAP_INSTR)
  uint32_t ran = 0;   // instrumented builds: which paths the lane's last trip took (1 entry load, 2 tail, 4 general step, 8 fetched-ahead pop, 16 ended at a check)
#define FQ_RAN(x) (ran |= (x))
#else
#define FQ_RAN(x) ((void)0)
#endif
  FQ_HD void flush_counters() {   // at the end of the wavefront's life
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t_pops = store.cold(FQ_COLD_TPOPS), t_pushes = store.cold(FQ_COLD_TPUSHES), t_touch = store.cold(FQ_COLD_TTOUCH);
    const uint32_t t_maxpops = store.cold(FQ_COLD_TMAXPOPS), t_gt4k = store.cold(FQ_COLD_TGT4K);
    uint64_t v[3] = {t_pops, t_pushes, t_touch};
    uint32_t mx = t_maxpops, g4 = t_gt4k;
    for (int d = 32; d >= 1; d >>= 1) {
      for (int q = 0; q < 3; ++q) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v[q], d), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v[q] >> 32), d);
        v[q] += (uint64_t)lo | (uint64_t)hi << 32;
      }
      const uint32_t om = (uint32_t)__shfl_xor((int)mx, d);
      mx = om > mx ? om : mx;
      g4 += (uint32_t)__shfl_xor((int)g4, d);
    }
    if (FQ_LANE_ID() == 0) {
      FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS], v[0]);
      FQ_ATOMIC_ADD64(&A.counters[FQ_C_PUSHES], v[1]);
      FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_GAP], v[2]);
      if (NOGAP) FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_NOGAP], v[2]);
      FQ_ATOMIC_MAX64(&A.counters[FQ_C_MAXPOPS], mx);
      if (g4) FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS_GT4K], g4);
    }
#else
    const uint32_t t_pops = store.cold(FQ_COLD_TPOPS), t_pushes = store.cold(FQ_COLD_TPUSHES), t_touch = store.cold(FQ_COLD_TTOUCH);
    const uint32_t t_maxpops = store.cold(FQ_COLD_TMAXPOPS), t_gt4k = store.cold(FQ_COLD_TGT4K);
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS], t_pops);
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_PUSHES], t_pushes);
    FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_GAP], t_touch);
    if (NOGAP) FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_NOGAP], t_touch);
    FQ_ATOMIC_MAX64(&A.counters[FQ_C_MAXPOPS], t_maxpops);
    if (t_gt4k) FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS_GT4K], t_gt4k);
#endif
  }

  FQ_HD FqGapLane(const FqGapArgs &A_, const St &st, int lane_slot) : A(A_), store(st), o(A_.o) {
    gape_mode = (o.mode & FQ_MODE_GAPE) != 0; nonstop = (o.mode & FQ_MODE_NONSTOP) != 0;
    const FqFM &f0 = A_.ix.fm[0], &f1 = A_.ix.fm[1];
    seq_len = f0.seq_len; L2_0 = f0.L2[0]; L2_1 = f0.L2[1]; L2_2 = f0.L2[2]; L2_3 = f0.L2[3];
    blk_s0 = f1.blk; blk_delta = (const char *)f0.blk - (const char *)f1.blk; primary0 = f0.primary; primary1 = f1.primary;
    active = done = false; len = 0;
    for (int q = 0; q < FQ_COLD_N; ++q) store.cold(q) = 0;
    prec = A_.prec; pw0 = pw1 = pw2 = pw3 = 0; wbase = 0x7fff;
    // Stack storage belongs to the lane, not to the read.  The pools of a wavefront's lanes are interleaved entry by entry (slot s of
    // lane t at [s * 64 + t]): lanes that started together push and pop the same slots at about the same time, so a 128-byte line
    // holds one slot of eight neighbouring lanes instead of eight slots of one lane -- the pushes of a step reach memory as whole
    // lines, and the line a lane's pop brings in serves its neighbours' pops from the cache.
    // (The round after the one without gap children holds unlike reads: their lanes do not push and pop in step, a slot's line is
    // one lane's alone, and every 16-byte push and pop moved a line to or from HBM -- at the rate of random HBM accesses the round
    // ran at.  There a lane's pool is contiguous (tier.lane_major): the children of a step share a line, and so do the entries
    // popped after each other.)
    pstep = A_.tier.lane_major ? 1u : (uint32_t)FQ_WAVE_SIZE;
    pool = A_.tier.lane_major ? A_.pool + (size_t)lane_slot * (size_t)A_.tier.pool_cap
                              : A_.pool + (size_t)(lane_slot / FQ_WAVE_SIZE) * (size_t)A_.tier.pool_cap * FQ_WAVE_SIZE + (size_t)(lane_slot % FQ_WAVE_SIZE);
    store.begin_lane(A_, lane_slot);
    m0 = m1 = m2 = m3 = bump = status = n_aln = 0; spare = FQ_NIL; n_live = 0;
    best_score = max_diff = 0; c_pops = c_pushes = c_touch = 0;
    has_cur = tail = hit_pending = too_many_n = false; ck_ = cl_ = cpk = 0; cscore = 0;
  }
  // both blocks of the work queue handed out (the cursors overshoot their block's length once it is exhausted)
  FQ_HD bool queue_dry() const {
    const uint32_t sp = A.split ? *A.split : (uint32_t)A.n_work;
    return FQ_LOAD_RELAXED(A.queue) >= sp && FQ_LOAD_RELAXED(A.queue + 1) >= (uint32_t)A.n_work - sp;
  }
  FQ_HD bool bucket_test(int b) const { if constexpr (kOneWord) return ((m0 >> b) & 1u) != 0; else return ((fq_sel4v(m0, m1, m2, m3, b >> 5) >> (b & 31)) & 1u) != 0; }
  FQ_HD void bucket_set(int b) {
    if constexpr (kOneWord) m0 |= 1u << b;
    else { const uint32_t bit = 1u << (b & 31); const int q = b >> 5; m0 |= bit & (0u - (uint32_t)(q == 0)); m1 |= bit & (0u - (uint32_t)(q == 1)); m2 |= bit & (0u - (uint32_t)(q == 2)); m3 |= bit & (0u - (uint32_t)(q == 3)); }
  }
  FQ_HD void bucket_clr(int b) {
    if constexpr (kOneWord) m0 &= ~(1u << b);
    else { const uint32_t bit = 1u << (b & 31); const int q = b >> 5; m0 &= ~(bit & (0u - (uint32_t)(q == 0))); m1 &= ~(bit & (0u - (uint32_t)(q == 1))); m2 &= ~(bit & (0u - (uint32_t)(q == 2))); m3 &= ~(bit & (0u - (uint32_t)(q == 3))); }
  }
  FQ_HD bool stack_empty() const { if constexpr (kOneWord) return m0 == 0; else return (m0 | m1 | m2 | m3) == 0; }
  FQ_HD int lowest_bucket() const { if constexpr (kOneWord) return FQ_CTZ32(m0); else return m0 ? FQ_CTZ32(m0) : m1 ? 32 + FQ_CTZ32(m1) : m2 ? 64 + FQ_CTZ32(m2) : 96 + FQ_CTZ32(m3); }

  // ---- grouped push: children that share a score bucket -----------------------------------------------------------------
  // Non-exact tiers drop, at push time, children the reference would pop only to discard (see FqGapArgs).
  FQ_HD bool group_open(int score, int diffs, uint32_t &prev) {
    if (!exact) {
      if (n_aln > 0 && !nonstop && score > best_score + o.s_mm) { FQ_PROF(10); return false; }
      if (max_diff - diffs < 0) { FQ_PROF(11); return false; }
    }
    prev = FQ_NIL;
    if (bucket_test(score)) prev = store.head_get(score);
    return true;
  }
  FQ_HD void group_put(uint32_t k, uint32_t l, uint32_t pk, uint32_t &prev) {
    const bool us = spare != FQ_NIL;
    const uint32_t slot = us ? spare : bump;
    bump += us ? 0u : 1u;
    spare = FQ_NIL;
    FqEntry e;
    e.k = k; e.l = l; e.pk = pk; e.next = prev;
    // In the round without gap children the stack entries stream out (most are never read back): a non-temporal store keeps them
    // from displacing the Occ blocks from the 4 MB L2 of the XCD.  The full search pops what it pushes, soon: plain stores leave
    // the entries where the pops find them (second round of an 8.4 M-read call 35.6 -> 34.2 ms).
#if defined(__HIP_DEVICE_COMPILE__)
    if (NOGAP) {
      typedef uint32_t fq_v4u __attribute__((ext_vector_type(4)));
      fq_v4u v;
      v.x = e.k; v.y = e.l; v.z = e.pk; v.w = e.next;
      __builtin_nontemporal_store(v, (fq_v4u *)(pool + (size_t)slot * pstep));
    } else pool[(size_t)slot * pstep] = e;
#else
    pool[(size_t)slot * pstep] = e;
#endif
    prev = slot;
    ++c_pushes; FQ_PROF(12);
  }
  FQ_HD void group_close(int score, uint32_t prev) { store.head_set(score, prev); bucket_set(score); }

  FQ_HD void finish() {
    // failed reads are re-run in a larger tier; expose no partial list
    if (NOGAP && status == 0 && !(n_aln > 0 && !nonstop && best_score + o.s_mm < o.s_gapo) && !too_many_n) status |= FQ_SF_NEEDGAP;
    if (NOGAP && (status & FQ_SF_NEEDGAP) && n_aln == 0) status |= FQ_SF_NOHIT;   // scheduling hint for the next round (fq_order_key)
    const int w = (int)store.cold(FQ_COLD_W);
    A.n_aln[w] = status ? 0u : n_aln;
    A.status[w] = status;
    if (status == 0) {   // work counters describe completed searches only: a read that is searched again (larger tier, next round) counts once.
      // They are summed per lane here and reach the global counters once per wavefront (flush_counters): five atomics per read on
      // five shared addresses cost an on-target launch 12 % of its time.
      store.cold(FQ_COLD_TPOPS) += c_pops; store.cold(FQ_COLD_TPUSHES) += c_pushes; store.cold(FQ_COLD_TTOUCH) += c_touch;
      if (c_pops > store.cold(FQ_COLD_TMAXPOPS)) store.cold(FQ_COLD_TMAXPOPS) = c_pops;
      if (c_pops > 4096) store.cold(FQ_COLD_TGT4K) += 1;
    }
    active = false; has_cur = false; tail = false; hit_pending = false; pf_slot = FQ_NIL;
  }

  FQ_HD void begin(int w) {
    store.cold(FQ_COLD_W) = (uint32_t)w;
    const FqGapWork gw = A.winfo[w];                 // written by k_width: read index, length, max_diff, "too many N"
    len = (int)(gw.meta & 0xffffu);
    const int max_diff_opt = (int)((gw.meta >> 16) & 0xffu);
    store.cold(FQ_COLD_MAXDIFF_OPT) = (uint32_t)max_diff_opt;
    prec = A.prec + (size_t)w * 2 * (size_t)A.pstride;
    wbase = 0x7fff;
    m0 = m1 = m2 = m3 = 0; bump = 0; spare = FQ_NIL; status = 0; n_aln = 0; n_live = 0;
    best_score = (max_diff_opt + 1) * o.s_mm + (o.max_gapo + 1) * o.s_gapo + (o.max_gape + 1) * o.s_gape;
    max_diff = max_diff_opt; store.cold(FQ_COLD_BEST_CNT) = 0;
    c_pops = c_pushes = c_touch = 0;
    has_cur = tail = hit_pending = false; pf_slot = FQ_NIL;
    active = true;
    too_many_n = ((gw.meta >> 24) & 1u) != 0;
    if (too_many_n) { finish(); return; }   // "too many N" early-out of bwt_match_gap (bwtgap.c:118-124)
    if (NOGAP && A.skip_bound > 0) {
      const int ba = A.bid_end[2 * (size_t)w], bb = A.bid_end[2 * (size_t)w + 1];
      if ((ba < bb ? ba : bb) >= A.skip_bound) { status |= FQ_SF_NEEDGAP; finish(); return; }   // (finish() adds FQ_SF_NOHIT: a long search of the next round)
    }
    // the two roots (bwtgap.c:139-140): strand 0 first, so strand 1 is popped first
    // (Starting on strand 1's root without its trip through the pool -- push and pop done here, strand 0's root left as fetched-ahead
    // content -- saves two of a read's ~350 trips and was 1.3 ms SLOWER per 8.4 M reads, A/B on one device: 41.2 against 39.9 ms.)
    uint32_t prev = FQ_NIL;
    group_put(0, seq_len, fq_pack(len, 0, FQ_ST_M, 0, 0, 0, 0), prev);
    group_put(0, seq_len, fq_pack(len, 1, FQ_ST_M, 0, 0, 0, 0), prev);
    group_close(0, prev);
    n_live = 2;
  }

  // gap_pop (bwtgap.c:66-79) of entry (ek, el, epk) from slot `slot` of bucket b, and the checks between a pop and its expansion
  // (bwtgap.c:147-165).  True: the entry is the lane's current entry; false: the search ended, the entry was discarded, or it is
  // a completed alignment (hit_pending).
  FQ_HD bool take_entry(int b, uint32_t slot, uint32_t ek, uint32_t el, uint32_t epk, uint32_t enext) {
    if (enext == FQ_NIL) bucket_clr(b); else store.head_set(b, enext);
    spare = slot;
    pf_slot = FQ_NIL;
    --n_live; ++c_pops;                                                    // the window stays: siblings popped back to back share it
    if (!nonstop && b > best_score + o.s_mm) { finish(); return false; }   // bwtgap.c:147
    // A long search is handed over to the wavefront-per-read kernel, but only once the work queue has run dry: before that a
    // busy lane costs nothing, afterwards the whole launch waits for it.
    if (A.tier.long_pops && c_pops > A.tier.long_pops && (A.tier.long_always || ((c_pops & 63u) == 0 && queue_dry()))) {
      status |= FQ_SF_LONG; finish(); return false;
    }
    const int n_mm = (int)(epk >> 12) & 31, n_gapo = (int)(epk >> 17) & 3, n_gape = (int)(epk >> 19) & 15;
    const int m = max_diff - (n_mm + n_gapo + (gape_mode ? n_gape : 0));
    if (m < 0) { FQ_PROF(4); return false; }
    ck_ = ek; cl_ = el; cpk = epk; cscore = b;
    if ((epk & 511u) == 0) { FQ_PROF(5); hit_pending = true; return false; }
    has_cur = true;
    return true;
  }
  FQ_HD void fetch_ahead(uint32_t nslot) {
    pf_slot = nslot;
    if (nslot != FQ_NIL) {
      const FqU4 v = *(const FqU4 *)(pool + (size_t)nslot * pstep);
      pfk = v.x; pfl = v.y; pfpk = v.z; pfnext = v.w;
    }
  }

  // one trip of an active lane
  FQ_HD void step() {
    FQ_PROF(0);
#if defined(FQ_GAP_INSTR)
    ran = 0;
#endif
    bool popping = !has_cur;
    int b = 0;
    uint32_t slot = 0;
    if (popping) {
      if (stack_empty()) { finish(); return; }
      if (n_live > o.max_entries) { if (!exact) status |= FQ_SF_ENTRY_LIMIT; finish(); return; }   // bwtgap.c:144
      b = lowest_bucket();
      if (NOGAP && b >= o.s_gapo) { status |= FQ_SF_NEEDGAP; finish(); return; }   // the full search may hold a gap child at or below this bucket
      slot = store.head_get(b);
      if (slot == pf_slot) {   // the entry was fetched ahead: pop it now and expand it in this same trip
        const uint32_t ek = pfk, el = pfl, epk = pfpk, enext = pfnext;
        const bool go = take_entry(b, slot, ek, el, epk, enext);
        FQ_RAN(8);
        if (active) fetch_ahead(enext);
        if (!go) return;
        popping = false;
      }
    }
    // ---- loads: a popping lane fetches its 16-byte stack entry; a lane with a current entry fetches the two Occ blocks of
    //      rows k-1 and l and, when positions i0-1 / i0-2 have left its window, the next eight position records
    const int a = (int)(cpk >> 9) & 1, i0 = (int)(cpk & 511u);            // i0 >= 1 whenever has_cur
    const int need_lo = i0 >= 2 ? i0 - 2 : 0;
    const int sa = a << 10;                                                  // wbase carries the window's strand in bit 10
    const uint32_t primary = fq_pick2(primary0, primary1, a);
    const bool reload = !popping && (need_lo + sa < wbase || i0 - 1 + sa > wbase + 7);
    const int nb = i0 >= 8 ? ((i0 - 7) & ~1) : 0;                            // 4-byte aligned window holding i0-2 and i0-1
    const FqPos *pp = prec + (size_t)a * (size_t)A.pstride + nb;
    const uintptr_t pa = (uintptr_t)fq_pick2p((uint64_t)(uintptr_t)(pool + (size_t)slot * pstep), (uint64_t)(uintptr_t)pp, popping ? 1 : 0);
    FqU4 vA;
    vA.x = vA.y = vA.z = vA.w = 0;
    if (popping || reload) vA = *(const FqU4 *)pa;
    FqBlkRaw bk, bl;
    if (!popping) {
      // strand a searches the other strand's BWT (bwtgap.c:148); an offset from one table rather than a choice of two pointers: the
      // address stays a global-memory address (a pointer rebuilt from an integer is loaded with flat instructions)
      const FqOccBlk *blk = (const FqOccBlk *)((const char *)blk_s0 + (a ? blk_delta : (ptrdiff_t)0));
      fq_blk_load_pair(blk, primary, ck_ - 1, cl_, bk, bl);
    } else {
      bk = fq_blk_none(); bl = fq_blk_none();
    }
    if (popping) {   // ---- the entry arrives from the pool; it is expanded in the next trip -------------------------------------
      FQ_PROF(3); FQ_RAN(1);
      take_entry(b, slot, vA.x, vA.y, vA.z, vA.w);
      if (active) fetch_ahead(vA.w);
      return;
    }
    // ---- a lane with a current entry ------------------------------------------------------------------------------------------
    if (reload) { pw0 = vA.x; pw1 = vA.y; pw2 = vA.z; pw3 = vA.w; wbase = nb + sa; }
    const int o1 = (i0 - 1 + sa) - wbase, o2 = need_lo + sa - wbase;
    const uint32_t rec1 = (fq_sel4v(pw0, pw1, pw2, pw3, o1 >> 1) >> ((o1 & 1) << 4)) & 0xffffu;   // position i0-1
    const uint32_t rec2 = (fq_sel4v(pw0, pw1, pw2, pw3, o2 >> 1) >> ((o2 & 1) << 4)) & 0xffffu;   // position i0-2 (if any)
    const int cbase = (int)(rec1 >> 12) & 7;                                 // seq[a][i0-1]
    const int b0 = (int)(rec1 & 31u), b1 = (int)(rec2 & 31u);                // width[i0-1].bid, width[i0-2].bid
    const bool seeded = (i0 - 1) - (len - o.seed_len) > 0 && len > o.seed_len;   // (use_seed and the seed's offset, from len: a register each otherwise)
    const int st = (int)(cpk >> 10) & 3, n_mm = (int)(cpk >> 12) & 31, n_gapo = (int)(cpk >> 17) & 3, n_gape = (int)(cpk >> 19) & 15;
    const int diffs = n_mm + n_gapo + (gape_mode ? n_gape : 0);
    const int m = max_diff - diffs;
    if (!tail) {
      if (m < b0) { FQ_PROF(6); FQ_RAN(16); has_cur = false; return; }        // bwtgap.c:155
      if (m == 0 && (st == FQ_ST_M || gape_mode || n_gape == o.max_gape)) { tail = true; FQ_PROF(7); }   // no difference left: exact tail
    }
    uint32_t ok4[4], ol4[4];
    fq_blk_occ4(bk, ok4);
    fq_blk_occ4(bl, ol4);
    const uint32_t kk0 = L2_0 + ok4[0] + 1, kk1 = L2_1 + ok4[1] + 1, kk2 = L2_2 + ok4[2] + 1, kk3 = L2_3 + ok4[3] + 1;
    const uint32_t ll0 = L2_0 + ol4[0], ll1 = L2_1 + ol4[1], ll2 = L2_2 + ol4[2], ll3 = L2_3 + ol4[3];
    const uint32_t vmask = (kk0 <= ll0 ? 1u : 0u) | (kk1 <= ll1 ? 2u : 0u) | (kk2 <= ll2 ? 4u : 0u) | (kk3 <= ll3 ? 8u : 0u);
    const uint32_t touch = fq_touch2p(primary, seq_len, ck_ - 1, cl_, tail);
    const int i = i0 - 1;
    const bool mvalid = cbase < 4 && ((vmask >> (cbase & 3)) & 1u) != 0;
    const uint32_t mk = fq_sel4v(kk0, kk1, kk2, kk3, cbase & 3), ml = fq_sel4v(ll0, ll1, ll2, ll3, cbase & 3);
    const uint32_t fpk = (cpk & ~0xC00u) - 1u;                                // the match child: same counts, position i, state M
    if (tail) {   // one base of bwt_match_exact_alt (libbwa/bwt.c:102-117)
      FQ_PROF(1); FQ_RAN(2);
      if (cbase < 4) c_touch += touch;
      if (!mvalid) { has_cur = false; tail = false; return; }
      ck_ = mk; cl_ = ml; cpk = fpk;
      if (i == 0) { has_cur = false; tail = false; hit_pending = true; }
      return;
    }
    FQ_PROF(8); FQ_RAN(4);
    c_touch += touch;
    const int last_diff = (int)(cpk >> 23);
    bool allow_diff = true, allow_M = true;
    if (i > 0) {
      if (b1 > m - 1) allow_diff = false;             // width[i-1].bid, width[i].bid of the child position i (bwtgap.c:200-210)
      else if (b1 == m - 1 && b0 == m - 1 && ((rec1 >> 5) & 1u)) allow_M = false;
      if (seeded) {                                   // ii = i - seed_off > 0
        const int m_seed = o.max_seed_diff - diffs;
        const int s1 = (int)(rec2 >> 6) & 31, s0 = (int)(rec1 >> 6) & 31;   // seed_width[ii-1].bid, seed_width[ii].bid
        if (s1 > m_seed - 1) allow_diff = false;
        else if (s1 == m_seed - 1 && s0 == m_seed - 1 && ((rec1 >> 11) & 1u)) allow_M = false;
      }
    }
    if (bump + 10u > A.tier.pool_cap) { status |= FQ_SF_POOL_OVERFLOW; finish(); return; }   // room for every child of this entry
    const uint32_t k = ck_, l = cl_;
    if (allow_diff) {
      int tmp;
      if (o.mode & FQ_MODE_LOGGAP) { uint32_t vv = (uint32_t)(n_gape + n_gapo); int lg = 0; while (vv >>= 1) ++lg; tmp = lg / 2 + 1; }
      else tmp = n_gapo + n_gape;
      if (NOGAP) n_live += 5;   // what the gap group below would have pushed at most (they only count towards max_entries)
      if (!NOGAP && i >= o.indel_end_skip + tmp && len - i >= o.indel_end_skip + tmp) {   // ---- gap children (bwtgap.c:212-243)
        const bool is_open = st == FQ_ST_M;
        const bool can = is_open ? n_gapo < o.max_gapo : n_gape < o.max_gape;
        const bool has_I = can && st != FQ_ST_D;
        const bool has_D = can && st != FQ_ST_I && (is_open || n_gape + n_gapo < max_diff || (l - k + 1) < (uint32_t)o.max_del_occ);
        const uint32_t cnt = (has_I ? 1u : 0u) + (has_D ? (uint32_t)FQ_POPC32(vmask) : 0u);
        if (cnt) {
          FQ_PROF(14);
          n_live += (int32_t)cnt;
          const int go2 = n_gapo + (is_open ? 1 : 0), ge2 = n_gape + (is_open ? 0 : 1);
          const int score = cscore + (is_open ? o.s_gapo : o.s_gape);
          uint32_t prev;
          if (group_open(score, n_mm + go2 + (gape_mode ? ge2 : 0), prev)) {
            FQ_PROF(15);
            const uint32_t common = (cpk & ((1u << 9) | (31u << 12))) | (uint32_t)go2 << 17 | (uint32_t)ge2 << 19;
            const uint32_t pkI = common | (uint32_t)i | (uint32_t)FQ_ST_I << 10 | (uint32_t)i << 23;
            const uint32_t pkD = common | (uint32_t)(i + 1) | (uint32_t)FQ_ST_D << 10 | (uint32_t)(i + 1) << 23;
            if (has_I) group_put(k, l, pkI, prev);
            if (has_D) {
              if (vmask & 1u) group_put(kk0, ll0, pkD, prev);
              if (vmask & 2u) group_put(kk1, ll1, pkD, prev);
              if (vmask & 4u) group_put(kk2, ll2, pkD, prev);
              if (vmask & 8u) group_put(kk3, ll3, pkD, prev);
            }
            group_close(score, prev);
          }
        }
      }
      if (allow_M) {   // ---- mismatch children, bases (c+1)&3, (c+2)&3, (c+3)&3 and, for an ambiguous read base, (c+4)&3 (bwtgap.c:245-252)
        const uint32_t mmask = cbase < 4 ? (vmask & ~(1u << cbase)) : vmask;
        if (mmask) {
          FQ_PROF(16);
          n_live += (int32_t)FQ_POPC32(mmask);
          const int score = cscore + o.s_mm;
          uint32_t prev;
          if (group_open(score, diffs + 1, prev)) {
            FQ_PROF(17);
            const uint32_t pkM = ((cpk & ((1u << 9) | (31u << 12) | (3u << 17) | (15u << 19))) + (1u << 12)) | (uint32_t)i | (uint32_t)i << 23;
#pragma unroll
            for (int j = 1; j <= 4; ++j) {
              const int cc = (cbase + j) & 3;
              if ((mmask >> cc) & 1u) {
                group_put(fq_sel4v(kk0, kk1, kk2, kk3, cc), fq_sel4v(ll0, ll1, ll2, ll3, cc), pkM, prev);
              }
            }
            group_close(score, prev);
          }
        }
      }
    }
    // ---- the match child stays in registers (pushed last by the reference, hence popped next: bwtgap.c:246-258) ----
    if (!mvalid) { has_cur = false; return; }
    FQ_PROF(2);
    ++c_pushes; ++c_pops;
    if (n_live + 1 > o.max_entries) { if (!exact) status |= FQ_SF_ENTRY_LIMIT; finish(); return; }   // the loop-top check of its pop
    ck_ = mk; cl_ = ml; cpk = fpk;
    if (i == 0) { has_cur = false; hit_pending = true; }
  }

  // Completed alignments of this trip (bwtgap.c:166-198); called by every lane of the wavefront when any lane has one.
  FQ_HD void collect_hits() {
    bool keep_going = true, add = false;
    uint32_t x = 0;
    int ld = 0, a = 0;
    const int w = (int)store.cold(FQ_COLD_W);
    if (hit_pending) {
      a = (int)(cpk >> 9) & 1;
      const int n_mm = (int)(cpk >> 12) & 31, n_gapo = (int)(cpk >> 17) & 3, n_gape = (int)(cpk >> 19) & 15;
      if (n_aln == 0) {
        best_score = cscore;
        const int best_diff = n_mm + n_gapo + (gape_mode ? n_gape : 0);
        const int max_diff_opt = (int)store.cold(FQ_COLD_MAXDIFF_OPT);
        if (!nonstop) max_diff = best_diff + 1 > max_diff_opt ? max_diff_opt : best_diff + 1;
      }
      if (cscore == best_score) store.cold(FQ_COLD_BEST_CNT) += cl_ - ck_ + 1;
      else if ((int)store.cold(FQ_COLD_BEST_CNT) > o.max_top2) keep_going = false;
      if (keep_going) {
        add = true;
        if (n_gapo) {
          const FqAln *al = A.aln + (size_t)w * (size_t)A.tier.aln_cap;
          for (uint32_t j = 0; j < n_aln; ++j)
            if (al[j].k == ck_ && al[j].l == cl_) { add = false; break; }
        }
        if (add) { x = cl_ - ck_ + 1; ld = (int)(cpk >> 23); }
      }
    }
    // gap_shadow (bwtgap.c:81-91) over width[0..last_diff) of the hit's strand, one hit at a time, all lanes sweeping; the
    // packed position records (bid, and the "same width as the previous position" bit up to position last_diff) follow
    uint64_t sm = FQ_BALLOT(add && ld > 0);
    if (add && ld > 0) wbase = 0x7fff;                                        // the records change under the window
    while (sm) {
      const int L = FQ_CTZ64(sm);
      sm &= sm - 1;
      const uint64_t p64 = (uint64_t)(uintptr_t)(A.wfull + ((size_t)w * 2 + (size_t)a) * (size_t)A.wstride);
      const uint64_t q64 = (uint64_t)(uintptr_t)(prec + (size_t)a * (size_t)A.pstride);
      uint32_t *const ww = (uint32_t *)(uintptr_t)((uint64_t)FQ_READLANE32((uint32_t)p64, L) | (uint64_t)FQ_READLANE32((uint32_t)(p64 >> 32), L) << 32);
      FqPos *const pr = (FqPos *)(uintptr_t)((uint64_t)FQ_READLANE32((uint32_t)q64, L) | (uint64_t)FQ_READLANE32((uint32_t)(q64 >> 32), L) << 32);
      const int n = (int)FQ_READLANE32((uint32_t)ld, L);
      const uint32_t xx = FQ_READLANE32(x, L);
      uint32_t jj = 0, carry_w = 0;
      for (int base = 0; base <= n; base += FQ_WAVE_SIZE) {
        const int t = base + FQ_LANE_ID();
        const bool in = t < n, in2 = t <= n;                                   // position n itself keeps its width, but not its neighbour's
        uint32_t nw = in2 ? ww[t] : 0u;
        const bool gt = in && nw > xx, eq = in && nw == xx;
        const uint64_t em = FQ_BALLOT(eq);
        if (gt) nw -= xx;
        else if (eq) nw = seq_len - (jj + (uint32_t)FQ_POPC64(em & (((uint64_t)1 << FQ_LANE_ID()) - 1)) + 1);
        if (gt || eq) ww[t] = nw;
        jj += (uint32_t)FQ_POPC64(em);
        uint32_t wleft = FQ_SHFL_UP1(nw);                                      // new width of position t-1
        if (FQ_LANE_ID() == 0) wleft = carry_w;
        carry_w = FQ_READLANE32(nw, FQ_WAVE_SIZE - 1);
        if (in2) {
          uint32_t rec = (uint32_t)pr[t] & ~0x20u;
          if (eq) rec = (rec & ~0x1fu) | 1u;                                   // bid = 1 (bwtgap.c:87)
          pr[t] = (FqPos)(rec | ((t >= 1 && wleft == nw) ? 1u << 5 : 0u));
        }
      }
    }
    if (hit_pending) {
      hit_pending = false;
      if (!keep_going) finish();
      else if (add) {
        if (n_aln >= A.tier.aln_cap) { status |= FQ_SF_ALN_OVERFLOW; finish(); }
        else {
          const int n_mm = (int)(cpk >> 12) & 31, n_gapo = (int)(cpk >> 17) & 3, n_gape = (int)(cpk >> 19) & 15;
          FqAln h;
          h.info = (uint32_t)n_mm | (uint32_t)n_gapo << 8 | (uint32_t)n_gape << 16 | (uint32_t)a << 24;
          h.k = ck_; h.l = cl_; h.score = cscore;
          A.aln[(size_t)w * (size_t)A.tier.aln_cap + n_aln++] = h;
        }
      }
    }
  }
};

// fetch(n): reserves n consecutive queue positions of one block of the queue; returns first | limit << 32 -- positions below
// `limit` are valid, limit == 0: the queue is exhausted
template <bool NOGAP, class Opt = FqKOpts, class St, class Fetch>
FQ_HD void fq_gap_lanes(const FqGapArgs &A, const St &store0, Fetch fetch, int lane_slot) {
  FqGapLane<St, NOGAP, Opt> L(A, store0, lane_slot);
  uint32_t trips = 0, lane_trips = 0;
#if defined(FQ_GAP_INSTR) && defined(__HIP_DEVICE_COMPILE__)
  uint32_t ib[16];
  for (int q = 0; q < 16; ++q) ib[q] = 0;
  const uint64_t t_life0 = clock64();
#endif
  for (;; ++trips) {
    // (re)fill idle lanes, in groups: one queue reservation per group
    const bool want = !L.active && !L.done;
    const uint64_t wm = FQ_BALLOT(want), am = FQ_BALLOT(L.active);
    if (wm == 0 && am == 0) {
#if defined(FQ_GAP_INSTR) && defined(__HIP_DEVICE_COMPILE__)
#if FQ_GAP_INSTR == 2
      ib[14] = (uint32_t)((clock64() - t_life0) >> 4);   // lifetime of this wavefront
#endif
#if defined(FQ_GAP_INSTR_ROUND2)
      if (!NOGAP)      // (statistics of the full search only: what the first round left)
#elif defined(FQ_GAP_INSTR_ROUND1)
      if (NOGAP)
#endif
      if (FQ_LANE_ID() == 0) for (int q = 0; q < 16; ++q) FQ_ATOMIC_ADD64(&A.counters[FQ_C_DBG0 + q], ib[q]);
#endif
      L.flush_counters();
      if (FQ_LANE_ID() == 0) { FQ_ATOMIC_MAX64(&A.counters[FQ_C_MAXTRIPS], trips); FQ_ATOMIC_ADD64(&A.counters[FQ_C_SUMTRIPS], trips); }
      FQ_ATOMIC_ADD64(&A.counters[FQ_C_LANETRIPS], lane_trips);
      break;
    }
    if (wm != 0 && (FQ_POPC64(wm) >= A.refill_min || am == 0)) {
      const int leader = FQ_CTZ64(wm);
      uint64_t got_l = 0;
      if (FQ_LANE_ID() == leader) got_l = fetch((uint32_t)FQ_POPC64(wm));
      const uint32_t base = FQ_READLANE32((uint32_t)got_l, leader), limit = FQ_READLANE32((uint32_t)(got_l >> 32), leader);
      if (want) {
        const uint32_t wq = base + (uint32_t)FQ_POPC64(wm & (((uint64_t)1 << FQ_LANE_ID()) - 1));
        if (limit == 0) L.done = true;
        else if (wq < limit) L.begin(A.order ? A.order[wq] : (int)wq);   // (a lane past the end of the block asks again with the next group)
      }
    }
#if defined(FQ_GAP_INSTR) && defined(__HIP_DEVICE_COMPILE__)
    {   // which paths this trip will execute, and for how many lanes
      const uint64_t a2 = FQ_BALLOT(L.active);
      const int na = (int)FQ_POPC64(a2);
      ib[0] += 1; ib[1] += (uint32_t)na;
#if FQ_GAP_INSTR == 1
      ib[2 + (na == 0 ? 0 : na <= 8 ? 1 : na <= 16 ? 2 : na <= 32 ? 3 : na <= 48 ? 4 : 5)] += 1;     // 2..7: trips by active lanes (0, 1-8, 9-16, 17-32, 33-48, 49-64)
#endif
#if FQ_GAP_INSTR == 1
      const uint64_t pop = FQ_BALLOT(L.active && !L.has_cur), tl = FQ_BALLOT(L.active && L.has_cur && L.tail), ex = FQ_BALLOT(L.active && L.has_cur && !L.tail);
      ib[8] += pop != 0; ib[9] += tl != 0; ib[10] += ex != 0;
      ib[11] += (uint32_t)FQ_POPC64(pop); ib[12] += (uint32_t)FQ_POPC64(tl); ib[13] += (uint32_t)FQ_POPC64(ex);
      ib[14] += (pop != 0) + (tl != 0) + (ex != 0) == 3;
#endif
    }
#endif
#if defined(FQ_GAP_INSTR) && FQ_GAP_INSTR == 2 && defined(__HIP_DEVICE_COMPILE__)
    // timing mode: shader-clock cycles a wavefront spends in the step and in hit collection (dbg[2..7] instead of the histogram)
    const bool pop_any = FQ_BALLOT(L.active && !L.has_cur) != 0;
    bool rl_any;
    {   // will a lane fetch a new window of position records in this trip (the condition of FqGapLane::step)
      const int i0_ = (int)(L.cpk & 511u), sa_ = (int)((L.cpk >> 9) & 1u) << 10, lo_ = (i0_ >= 2 ? i0_ - 2 : 0) + sa_;
      rl_any = FQ_BALLOT(L.active && L.has_cur && (lo_ < L.wbase || i0_ - 1 + sa_ > L.wbase + 7)) != 0;
    }
    const uint64_t tc0 = clock64();
    if (L.active) { ++lane_trips; L.step(); }
    const bool any_hit = FQ_BALLOT(L.hit_pending) != 0;
    const uint64_t tc1 = clock64();
    const uint64_t shm = FQ_BALLOT(L.hit_pending && (L.cpk >> 23) > 0);
    if (any_hit) L.collect_hits();
    const uint64_t tc2 = clock64();
    ib[15] += any_hit;
    ib[2] += (uint32_t)((tc1 - tc0) >> 4); ib[3] += (uint32_t)((tc2 - tc1) >> 4);
    if (pop_any) { ib[6] += (uint32_t)((tc1 - tc0) >> 4); ib[7] += 1; }     // trips in which at least one lane pops a stack entry
    else if (rl_any) { ib[10] += (uint32_t)((tc1 - tc0) >> 4); ib[11] += 1; }   // no pop, but a window of position records is fetched
    else { ib[8] += (uint32_t)((tc1 - tc0) >> 4); ib[9] += 1; }                 // only Occ blocks
    if (FQ_POPC64(FQ_BALLOT(L.active || L.hit_pending)) <= 2 && !any_hit) { ib[12] += (uint32_t)((tc1 - tc0) >> 4); ib[13] += 1; }   // sparse wavefronts (the tail of a launch): at most two lanes at work
    ib[4] += (uint32_t)FQ_POPC64(shm); ib[5] += shm != 0;
#else
#if defined(FQ_GAP_INSTR) && FQ_GAP_INSTR == 3 && defined(__HIP_DEVICE_COMPILE__)
    const bool was_active = L.active;
#endif
    if (L.active) { ++lane_trips; L.step(); }
    const bool any_hit = FQ_BALLOT(L.hit_pending) != 0;
#if defined(FQ_GAP_INSTR) && defined(__HIP_DEVICE_COMPILE__)
    ib[15] += any_hit;
#if FQ_GAP_INSTR == 3
    {   // the set of paths this trip ran (entry load / tail / general step) and the lanes on each
      const uint32_t rn = was_active ? L.ran : 0u;
      const uint64_t mp = FQ_BALLOT((rn & 1u) != 0), mt = FQ_BALLOT((rn & 2u) != 0), mg = FQ_BALLOT((rn & 4u) != 0), mf = FQ_BALLOT((rn & 8u) != 0), md = FQ_BALLOT((rn & 16u) != 0);
      ib[2 + ((mp | mf) != 0) + 2 * (mt != 0) + 4 * (mg != 0)] += 1;
      ib[10] += (uint32_t)FQ_POPC64(mp); ib[11] += (uint32_t)FQ_POPC64(mt); ib[12] += (uint32_t)FQ_POPC64(mg); ib[13] += (uint32_t)FQ_POPC64(mf); ib[14] += (uint32_t)FQ_POPC64(md);
    }
#endif
#endif
    if (any_hit) L.collect_hits();
#endif
  }
}


// ---- the same search, one read per wavefront ------------------------------------------------------------------------------
// A search that pops tens of thousands of entries keeps one lane of the kernel above busy for that many dependent round
// trips, long after the rest of its launch has drained.  Such reads (FQ_SF_LONG from the lane kernel) are searched again
// here with the lanes of a wavefront working on ONE read.
//
// What makes that exact: all entries popped while bucket b is the lowest non-empty one were pushed before b was first
// popped (children score strictly more than their parent, except the exact-match child, which is popped immediately after
// its parent).  So the reference's pop sequence inside bucket b is: top entry, then its whole exact-match chain, then the
// next entry and its chain, ...  The chains of different entries do not see each other -- except through a completed
// alignment, which moves best_score / max_diff and lets gap_shadow edit the width array.  A round therefore gives the top
// T <= 64 entries of bucket b to T lanes, runs every chain to its end twice (pass 1 counts the children each chain sends to
// each higher bucket and notices alignments, pass 2 writes them), and commits only lanes 0..j*, j* being the first lane
// whose chain completed an alignment; the entries of the lanes behind it stay on the stack and are redone after the
// alignment has been recorded.  Children are appended per bucket in lane order, each lane's in step order: exactly the
// order in which the reference pushes them.  A bucket is a stack of runs (contiguous arrays in the read's pool, the header
// slot in front of each run links to the run below); it only grows while lower buckets are being worked on and only
// shrinks afterwards, so runs are never appended to once popping has begun.
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_SHFL_UPN(x, d) ((uint32_t)__shfl_up((int)(x), (d)))
#define FQ_SHFL_XORN(x, d) ((uint32_t)__shfl_xor((int)(x), (d)))
#define FQ_WAVE_FENCE() __threadfence_block()
#else
#define FQ_SHFL_UPN(x, d) ((uint32_t)(x))
#define FQ_SHFL_XORN(x, d) ((uint32_t)(x))
#define FQ_WAVE_FENCE() ((void)0)
#endif
FQ_HD uint32_t fq_wave_sum(uint32_t x) {
  for (int d = 1; d < FQ_WAVE_SIZE; d <<= 1) x += FQ_SHFL_XORN(x, d);
  return x;
}
// inclusive prefix sum over the lanes of the wavefront
FQ_HD uint32_t fq_wave_incl_scan(uint32_t x) {
  for (int d = 1; d < FQ_WAVE_SIZE; d <<= 1) { const uint32_t y = FQ_SHFL_UPN(x, d); if (FQ_LANE_ID() >= d) x += y; }
  return x;
}

struct FqGapCoop {
  const FqGapArgs &A;
  FqKOpts o;
  bool gape_mode, nonstop, exact;
  uint32_t seq_len, L2_0, L2_1, L2_2, L2_3;
  const FqOccBlk *blk0, *blk1;
  uint32_t primary0, primary1;
  uint32_t *heads;       // LDS: [bucket] first entry of the top run (0 = empty), [FQ_MAX_BUCKETS + bucket] entries left in it
  FqEntry *pool;
  // per-read state, identical in every lane
  int w, len, max_diff_opt, seed_off;
  bool use_seed;
  FqPos *prec;
  uint32_t m0, m1, m2, m3, bump, status, n_aln;
  int32_t n_live;
  int best_score, max_diff, best_cnt;
  uint32_t c_pops, c_pushes, c_touch;
  // per-lane chain state
  bool on, hit, tail, limit_hit;
  uint32_t ck_, cl_, cpk;
  uint32_t pw0, pw1, pw2, pw3;
  int wbase;
  uint32_t cnt0, cnt1, cnt2;          // children per slot: counted in pass 1, write cursor in pass 2
  uint32_t live_add, ch_pops, ch_pushes, ch_touch;
  int cur_b;                          // bucket of this round
  int slot_open, slot_ext, slot_mm;   // which of the (at most three) target runs a class of children goes to
  int32_t live_run;                   // sequential rounds only: stack->n_entries as the reference sees it

  FQ_HD FqGapCoop(const FqGapArgs &A_, uint32_t *heads_, int wave_slot) : A(A_), o(A_.o), heads(heads_) {
    gape_mode = (o.mode & FQ_MODE_GAPE) != 0; nonstop = (o.mode & FQ_MODE_NONSTOP) != 0; exact = A_.tier.exact != 0;
    const FqFM &f0 = A_.ix.fm[0], &f1 = A_.ix.fm[1];
    seq_len = f0.seq_len; L2_0 = f0.L2[0]; L2_1 = f0.L2[1]; L2_2 = f0.L2[2]; L2_3 = f0.L2[3];
    blk0 = f0.blk; blk1 = f1.blk; primary0 = f0.primary; primary1 = f1.primary;
    pool = A_.pool + (size_t)wave_slot * (size_t)A_.tier.pool_cap;
    w = len = max_diff_opt = seed_off = 0; use_seed = false; prec = A_.prec;
    m0 = m1 = m2 = m3 = bump = status = n_aln = 0; n_live = 0; best_score = max_diff = best_cnt = 0;
    c_pops = c_pushes = c_touch = 0;
    on = hit = tail = limit_hit = false; ck_ = cl_ = cpk = 0; pw0 = pw1 = pw2 = pw3 = 0; wbase = 0x7fff;
    cnt0 = cnt1 = cnt2 = 0; live_add = ch_pops = ch_pushes = ch_touch = 0; cur_b = 0;
    slot_open = 0; slot_ext = 1; slot_mm = 2; live_run = 0;
    // classes of children whose scores coincide share a run, so that their relative order is kept
    if (o.s_gape == o.s_gapo) slot_ext = slot_open;
    if (o.s_mm == o.s_gapo) slot_mm = slot_open; else if (o.s_mm == o.s_gape) slot_mm = slot_ext;
  }
  FQ_HD void bucket_set(int b) { const uint32_t bit = 1u << (b & 31); const int q = b >> 5; if (q == 0) m0 |= bit; else if (q == 1) m1 |= bit; else if (q == 2) m2 |= bit; else m3 |= bit; }
  FQ_HD void bucket_clr(int b) { const uint32_t bit = ~(1u << (b & 31)); const int q = b >> 5; if (q == 0) m0 &= bit; else if (q == 1) m1 &= bit; else if (q == 2) m2 &= bit; else m3 &= bit; }
  FQ_HD void slot_add(int s, uint32_t n) { cnt0 += s == 0 ? n : 0u; cnt1 += s == 1 ? n : 0u; cnt2 += s == 2 ? n : 0u; }
  FQ_HD uint32_t slot_take(int s) {   // returns the cursor of slot s and advances it
    const uint32_t at = fq_sel4v(cnt0, cnt1, cnt2, 0u, s);   // (a ?: chain over members would pin the whole state in scratch memory)
    slot_add(s, 1u);
    return at;
  }
  // children of one group: counted, and in pass 2 written behind the lane's cursor of the group's run
  FQ_HD bool group_kept(int score, int diffs) const {
    if (exact) return true;
    if (n_aln > 0 && !nonstop && score > best_score + o.s_mm) return false;
    return max_diff - diffs >= 0;
  }
  FQ_HD void child(bool write, int slot, uint32_t k, uint32_t l, uint32_t pk) {
    if (write) { FqEntry e; e.k = k; e.l = l; e.pk = pk; e.next = 0; pool[slot_take(slot)] = e; }
    else slot_add(slot, 1u);
    ++ch_pushes;
  }

  // one step of this lane's chain (bwtgap.c:149-258 for the entry in ck_/cl_/cpk)
  FQ_HD void chain_step(bool write, bool sequential) {
    const int a = (int)(cpk >> 9) & 1, i0 = (int)(cpk & 511u);
    const int need_lo = i0 >= 2 ? i0 - 2 : 0;
    const bool reload = need_lo < wbase || i0 - 1 > wbase + 7;
    const int nb = i0 >= 8 ? ((i0 - 7) & ~1) : 0;
    if (reload) {
      const FqU4 v = *(const FqU4 *)(prec + (size_t)a * (size_t)A.pstride + nb);
      pw0 = v.x; pw1 = v.y; pw2 = v.z; pw3 = v.w; wbase = nb;
    }
    const FqOccBlk *blk = (const FqOccBlk *)fq_pick2p((uint64_t)(uintptr_t)blk0, (uint64_t)(uintptr_t)blk1, a);
    const uint32_t primary = fq_pick2(primary0, primary1, a);
    FqBlkRaw bk, bl;
    fq_blk_load_pair(blk, primary, ck_ - 1, cl_, bk, bl);
    const int o1 = (i0 - 1) - wbase, o2 = need_lo - wbase;
    const uint32_t rec1 = (fq_sel4v(pw0, pw1, pw2, pw3, o1 >> 1) >> ((o1 & 1) << 4)) & 0xffffu;
    const uint32_t rec2 = (fq_sel4v(pw0, pw1, pw2, pw3, o2 >> 1) >> ((o2 & 1) << 4)) & 0xffffu;
    const int cbase = (int)(rec1 >> 12) & 7;
    const int b0 = (int)(rec1 & 31u), b1 = (int)(rec2 & 31u);
    const bool seeded = use_seed && (i0 - 1) - seed_off > 0;
    const int st = (int)(cpk >> 10) & 3, n_mm = (int)(cpk >> 12) & 31, n_gapo = (int)(cpk >> 17) & 3, n_gape = (int)(cpk >> 19) & 15;
    const int diffs = n_mm + n_gapo + (gape_mode ? n_gape : 0);
    const int m = max_diff - diffs;
    if (!tail) {
      if (m < b0) { on = false; return; }
      if (m == 0 && (st == FQ_ST_M || gape_mode || n_gape == o.max_gape)) tail = true;
    }
    uint32_t ok4[4], ol4[4];
    fq_blk_occ4(bk, ok4);
    fq_blk_occ4(bl, ol4);
    const uint32_t kk0 = L2_0 + ok4[0] + 1, kk1 = L2_1 + ok4[1] + 1, kk2 = L2_2 + ok4[2] + 1, kk3 = L2_3 + ok4[3] + 1;
    const uint32_t ll0 = L2_0 + ol4[0], ll1 = L2_1 + ol4[1], ll2 = L2_2 + ol4[2], ll3 = L2_3 + ol4[3];
    const uint32_t vmask = (kk0 <= ll0 ? 1u : 0u) | (kk1 <= ll1 ? 2u : 0u) | (kk2 <= ll2 ? 4u : 0u) | (kk3 <= ll3 ? 8u : 0u);
    const int i = i0 - 1;
    const bool mvalid = cbase < 4 && ((vmask >> (cbase & 3)) & 1u) != 0;
    const uint32_t mk = fq_sel4v(kk0, kk1, kk2, kk3, cbase & 3), ml = fq_sel4v(ll0, ll1, ll2, ll3, cbase & 3);
    const uint32_t fpk = (cpk & ~0xC00u) - 1u;
    if (tail) {
      if (cbase < 4) ch_touch += fq_touch2p(primary, seq_len, ck_ - 1, cl_, true);
      if (!mvalid) { on = false; return; }
      ck_ = mk; cl_ = ml; cpk = fpk;
      if (i == 0) { on = false; hit = true; }
      return;
    }
    ch_touch += fq_touch2p(primary, seq_len, ck_ - 1, cl_, false);
    bool allow_diff = true, allow_M = true;
    if (i > 0) {
      if (b1 > m - 1) allow_diff = false;
      else if (b1 == m - 1 && b0 == m - 1 && ((rec1 >> 5) & 1u)) allow_M = false;
      if (seeded) {
        const int m_seed = o.max_seed_diff - diffs;
        const int s1 = (int)(rec2 >> 6) & 31, s0 = (int)(rec1 >> 6) & 31;
        if (s1 > m_seed - 1) allow_diff = false;
        else if (s1 == m_seed - 1 && s0 == m_seed - 1 && ((rec1 >> 11) & 1u)) allow_M = false;
      }
    }
    const uint32_t k = ck_, l = cl_;
    if (allow_diff) {
      int tmp;
      if (o.mode & FQ_MODE_LOGGAP) { uint32_t vv = (uint32_t)(n_gape + n_gapo); int lg = 0; while (vv >>= 1) ++lg; tmp = lg / 2 + 1; }
      else tmp = n_gapo + n_gape;
      if (i >= o.indel_end_skip + tmp && len - i >= o.indel_end_skip + tmp) {
        const bool is_open = st == FQ_ST_M;
        const bool can = is_open ? n_gapo < o.max_gapo : n_gape < o.max_gape;
        const bool has_I = can && st != FQ_ST_D;
        const bool has_D = can && st != FQ_ST_I && (is_open || n_gape + n_gapo < max_diff || (l - k + 1) < (uint32_t)o.max_del_occ);
        const uint32_t cnt = (has_I ? 1u : 0u) + (has_D ? (uint32_t)FQ_POPC32(vmask) : 0u);
        if (cnt) {
          live_add += cnt;
          const int go2 = n_gapo + (is_open ? 1 : 0), ge2 = n_gape + (is_open ? 0 : 1);
          const int score = cur_b + (is_open ? o.s_gapo : o.s_gape);
          if (group_kept(score, n_mm + go2 + (gape_mode ? ge2 : 0))) {
            const int slot = is_open ? slot_open : slot_ext;
            const uint32_t common = (cpk & ((1u << 9) | (31u << 12))) | (uint32_t)go2 << 17 | (uint32_t)ge2 << 19;
            const uint32_t pkI = common | (uint32_t)i | (uint32_t)FQ_ST_I << 10 | (uint32_t)i << 23;
            const uint32_t pkD = common | (uint32_t)(i + 1) | (uint32_t)FQ_ST_D << 10 | (uint32_t)(i + 1) << 23;
            if (has_I) child(write, slot, k, l, pkI);
            if (has_D) {
              if (vmask & 1u) child(write, slot, kk0, ll0, pkD);
              if (vmask & 2u) child(write, slot, kk1, ll1, pkD);
              if (vmask & 4u) child(write, slot, kk2, ll2, pkD);
              if (vmask & 8u) child(write, slot, kk3, ll3, pkD);
            }
          }
        }
      }
      if (allow_M) {
        const uint32_t mmask = cbase < 4 ? (vmask & ~(1u << cbase)) : vmask;
        if (mmask) {
          live_add += (uint32_t)FQ_POPC32(mmask);
          if (group_kept(cur_b + o.s_mm, diffs + 1)) {
            const uint32_t pkM = ((cpk & ((1u << 9) | (31u << 12) | (3u << 17) | (15u << 19))) + (1u << 12)) | (uint32_t)i | (uint32_t)i << 23;
            for (int j = 1; j <= 4; ++j) {
              const int cc = (cbase + j) & 3;
              if ((mmask >> cc) & 1u) child(write, slot_mm, fq_sel4v(kk0, kk1, kk2, kk3, cc), fq_sel4v(ll0, ll1, ll2, ll3, cc), pkM);
            }
          }
        }
      }
    }
    if (!mvalid) { on = false; return; }
    ++ch_pushes; ++ch_pops;
    if (sequential && live_run + (int32_t)live_add + 1 > o.max_entries) { on = false; limit_hit = true; return; }   // loop-top check of the child's pop
    ck_ = mk; cl_ = ml; cpk = fpk;
    if (i == 0) { on = false; hit = true; }
  }

  FQ_HD void finish() {
    if (FQ_LANE_ID() == 0) {
      A.n_aln[w] = status ? 0u : n_aln;
      A.status[w] = status;
      if (status == 0) {
        FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS], c_pops);
        FQ_ATOMIC_ADD64(&A.counters[FQ_C_PUSHES], c_pushes);
        FQ_ATOMIC_MAX64(&A.counters[FQ_C_MAXPOPS], c_pops);
        if (c_pops > 4096) FQ_ATOMIC_ADD64(&A.counters[FQ_C_POPS_GT4K], 1);
        FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_GAP], c_touch);
      }
    }
  }

  FQ_HD void start_chains(const FqEntry &e, bool active, uint32_t c0, uint32_t c1, uint32_t c2) {
    on = active; hit = false; tail = false; limit_hit = false;
    ck_ = e.k; cl_ = e.l; cpk = e.pk; wbase = 0x7fff;
    live_add = 0; ch_pops = ch_pushes = ch_touch = 0;
    cnt0 = c0; cnt1 = c1; cnt2 = c2;
    live_run = n_live - 1;
    if (on) {   // pop-time checks of the entry itself (bwtgap.c:150-166)
      const int n_mm = (int)(cpk >> 12) & 31, n_gapo = (int)(cpk >> 17) & 3, n_gape = (int)(cpk >> 19) & 15;
      if (max_diff - (n_mm + n_gapo + (gape_mode ? n_gape : 0)) < 0) on = false;
      else if ((cpk & 511u) == 0) { on = false; hit = true; }
    }
  }
  // lane 0: a run of n entries starting at pool[first] becomes the top of bucket B (header in front of it: size, run below)
  FQ_HD void new_run(int B, uint32_t first, uint32_t n) {
    FqEntry h;
    h.k = 0; h.l = n; h.pk = 0; h.next = heads[B];
    pool[first - 1] = h;
    heads[B] = first; heads[FQ_MAX_BUCKETS + B] = n;
  }

  // the whole search of work item w_; returns the number of rounds
  FQ_HD uint32_t run(int w_) {
    const int lane = FQ_LANE_ID();
    w = w_;
    const FqGapWork gw = A.winfo[w];
    len = (int)(gw.meta & 0xffffu);
    max_diff_opt = (int)((gw.meta >> 16) & 0xffu);
    use_seed = len > o.seed_len;
    seed_off = len - o.seed_len;
    prec = A.prec + (size_t)w * 2 * (size_t)A.pstride;
    m0 = m1 = m2 = m3 = 0; status = 0; n_aln = 0;
    best_score = (max_diff_opt + 1) * o.s_mm + (o.max_gapo + 1) * o.s_gapo + (o.max_gape + 1) * o.s_gape;
    max_diff = max_diff_opt; best_cnt = 0;
    c_pops = c_pushes = c_touch = 0;
    if ((gw.meta >> 24) & 1u) { finish(); return 0; }
    for (int b = lane; b < 2 * FQ_MAX_BUCKETS; b += FQ_WAVE_SIZE) heads[b] = 0;
    // the two roots: one run of two entries in bucket 0, strand 1 on top (bwtgap.c:139-140)
    if (lane == 0) {
      FqEntry h; h.k = 0; h.l = 2; h.pk = 0; h.next = 0;          // header: l = size of the run, next = first entry of the run below (0: none)
      pool[0] = h;
      FqEntry e; e.k = 0; e.l = seq_len; e.next = 0;
      e.pk = fq_pack(len, 0, FQ_ST_M, 0, 0, 0, 0); pool[1] = e;
      e.pk = fq_pack(len, 1, FQ_ST_M, 0, 0, 0, 0); pool[2] = e;
    }
    FQ_WAVE_FENCE();
    if (lane == 0) { heads[0] = 1; heads[FQ_MAX_BUCKETS] = 2; }
    bump = 3; m0 = 1u; n_live = 2; c_pushes = 2;
    uint32_t rounds = 0;
    bool force_seq = false;
    for (;; ++rounds) {
      if ((m0 | m1 | m2 | m3) == 0) break;
      if (n_live > o.max_entries) { if (!exact) status |= FQ_SF_ENTRY_LIMIT; break; }
      const int b = m0 ? FQ_CTZ32(m0) : m1 ? 32 + FQ_CTZ32(m1) : m2 ? 64 + FQ_CTZ32(m2) : 96 + FQ_CTZ32(m3);
      if (!nonstop && b > best_score + o.s_mm) { ++c_pops; break; }                    // bwtgap.c:147
      cur_b = b;
      const uint32_t start = heads[b], left = heads[FQ_MAX_BUCKETS + b];
      const uint32_t T = force_seq ? 1u : (left < (uint32_t)FQ_WAVE_SIZE ? left : (uint32_t)FQ_WAVE_SIZE);
      const bool mine = (uint32_t)lane < T;
      FqEntry e;
      e.k = e.l = e.pk = e.next = 0;
      if (mine) e = pool[start + left - 1 - (uint32_t)lane];
      // ---- two passes over the chains (one inlined copy of the step): pass 0 counts the children each chain sends to each run and
      //      notices alignments, pass 1 runs the committed chains again and writes
      uint32_t n_commit = 0, add_all = 0;
      bool committed = false, redo = false, overflow = false;
      uint32_t c0 = 0, c1 = 0, c2 = 0;
      for (int pass = 0; pass < 2 && !redo && !overflow; ++pass) {
        start_chains(e, pass == 0 ? mine : committed, c0, c1, c2);
        while (FQ_BALLOT(on) != 0) { if (on) chain_step(pass == 1, T == 1u); }
        if (pass == 1) break;
        const uint64_t hm = FQ_BALLOT(hit || limit_hit);
        n_commit = hm ? (uint32_t)FQ_CTZ64(hm) + 1u : T;                    // lanes behind the first alignment are redone later
        committed = (uint32_t)lane < n_commit;
        // The reference checks stack->n_entries > max_entries before every pop.  A parallel round skips those checks, which is
        // sound only if the bound cannot have been crossed anywhere inside the round; otherwise redo it one entry at a time.
        add_all = fq_wave_sum(committed ? live_add : 0u);
        if (T > 1u && (int64_t)n_live + (int64_t)add_all + 1 > (int64_t)o.max_entries) { redo = true; break; }
        const uint32_t my0 = committed ? cnt0 : 0u, my1 = committed ? cnt1 : 0u, my2 = committed ? cnt2 : 0u;
        const uint32_t i0s = fq_wave_incl_scan(my0), i1s = fq_wave_incl_scan(my1), i2s = fq_wave_incl_scan(my2);
        const uint32_t tot0 = FQ_READLANE32(i0s, FQ_WAVE_SIZE - 1), tot1 = FQ_READLANE32(i1s, FQ_WAVE_SIZE - 1), tot2 = FQ_READLANE32(i2s, FQ_WAVE_SIZE - 1);
        if (bump + tot0 + tot1 + tot2 + 3u > A.tier.pool_cap) { overflow = true; break; }
        // the runs of this round: a header slot, then the children of lanes 0..n_commit-1 in lane order
        const uint32_t r0 = bump + 1, r1 = r0 + (tot0 ? tot0 + 1 : 0), r2 = r1 + (tot1 ? tot1 + 1 : 0);
        bump = r2 + (tot2 ? tot2 + 1 : 0) - 1;
        if (lane == 0) {
          if (tot0) new_run(b + o.s_gapo, r0, tot0);
          if (tot1) new_run(b + o.s_gape, r1, tot1);
          if (tot2) new_run(b + o.s_mm, r2, tot2);
        }
        if (tot0) bucket_set(b + o.s_gapo);
        if (tot1) bucket_set(b + o.s_gape);
        if (tot2) bucket_set(b + o.s_mm);
        c0 = r0 + (i0s - my0); c1 = r1 + (i1s - my1); c2 = r2 + (i2s - my2);
      }
      if (redo) { force_seq = true; continue; }
      if (overflow) { status |= FQ_SF_POOL_OVERFLOW; break; }
      FQ_WAVE_FENCE();
      // ---- commit: entries popped, counters, the source run
      {
        const uint32_t pops = fq_wave_sum(committed ? ch_pops : 0u), pushes = fq_wave_sum(committed ? ch_pushes : 0u), touch = fq_wave_sum(committed ? ch_touch : 0u);
        c_pops += n_commit + pops; c_pushes += pushes; c_touch += touch;
        n_live += (int32_t)add_all - (int32_t)n_commit;
        const uint32_t rest = left - n_commit;
        if (rest) { if (lane == 0) heads[FQ_MAX_BUCKETS + b] = rest; }
        else {
          const uint32_t below = pool[start - 1].next;
          const uint32_t below_n = below ? pool[below - 1].l : 0u;
          if (lane == 0) { heads[b] = below; heads[FQ_MAX_BUCKETS + b] = below_n; }
          if (!below) bucket_clr(b);
        }
      }
      if (T == 1u) force_seq = false;
      // ---- the last committed lane may have ended the round with an alignment or with the entry limit
      const int jl = (int)n_commit - 1;
      const bool was_limit = FQ_READLANE32(limit_hit ? 1u : 0u, jl) != 0, was_hit = FQ_READLANE32(hit ? 1u : 0u, jl) != 0;
      if (was_limit) { if (!exact) status |= FQ_SF_ENTRY_LIMIT; break; }
      if (was_hit) {
        const uint32_t hk = FQ_READLANE32(ck_, jl), hl = FQ_READLANE32(cl_, jl), hpk = FQ_READLANE32(cpk, jl);
        if (!record_hit(hk, hl, hpk, b)) break;
      }
    }
    finish();
    return rounds;
  }

  // bwtgap.c:166-198 for the alignment (hk, hl, hpk); false = the search ends
  FQ_HD bool record_hit(uint32_t hk, uint32_t hl, uint32_t hpk, int e_score) {
    const int a = (int)(hpk >> 9) & 1, n_mm = (int)(hpk >> 12) & 31, n_gapo = (int)(hpk >> 17) & 3, n_gape = (int)(hpk >> 19) & 15, ld = (int)(hpk >> 23);
    if (n_aln == 0) {
      best_score = e_score;
      const int best_diff = n_mm + n_gapo + (gape_mode ? n_gape : 0);
      if (!nonstop) max_diff = best_diff + 1 > max_diff_opt ? max_diff_opt : best_diff + 1;
    }
    if (e_score == best_score) best_cnt += (int)(hl - hk + 1);
    else if (best_cnt > o.max_top2) return false;
    FqAln *al = A.aln + (size_t)w * (size_t)A.tier.aln_cap;
    if (n_gapo) {
      bool dup = false;
      for (uint32_t j0 = 0; j0 < n_aln; j0 += FQ_WAVE_SIZE) {
        const uint32_t j = j0 + (uint32_t)FQ_LANE_ID();
        const bool d = j < n_aln && al[j].k == hk && al[j].l == hl;
        if (FQ_BALLOT(d) != 0) { dup = true; break; }
      }
      if (dup) return true;
    }
    if (ld > 0) {   // gap_shadow over width[0..ld) and the position records up to ld
      uint32_t *const ww = A.wfull + ((size_t)w * 2 + (size_t)a) * (size_t)A.wstride;
      FqPos *const pr = prec + (size_t)a * (size_t)A.pstride;
      const uint32_t xx = hl - hk + 1;
      uint32_t jj = 0, carry_w = 0;
      for (int base = 0; base <= ld; base += FQ_WAVE_SIZE) {
        const int t = base + FQ_LANE_ID();
        const bool in = t < ld, in2 = t <= ld;
        uint32_t nw = in2 ? ww[t] : 0u;
        const bool gt = in && nw > xx, eq = in && nw == xx;
        const uint64_t em = FQ_BALLOT(eq);
        if (gt) nw -= xx;
        else if (eq) nw = seq_len - (jj + (uint32_t)FQ_POPC64(em & (((uint64_t)1 << FQ_LANE_ID()) - 1)) + 1);
        if (gt || eq) ww[t] = nw;
        jj += (uint32_t)FQ_POPC64(em);
        uint32_t wleft = FQ_SHFL_UP1(nw);
        if (FQ_LANE_ID() == 0) wleft = carry_w;
        carry_w = FQ_READLANE32(nw, FQ_WAVE_SIZE - 1);
        if (in2) {
          uint32_t rec = (uint32_t)pr[t] & ~0x20u;
          if (eq) rec = (rec & ~0x1fu) | 1u;
          pr[t] = (FqPos)(rec | ((t >= 1 && wleft == nw) ? 1u << 5 : 0u));
        }
      }
      FQ_WAVE_FENCE();
    }
    if (n_aln >= A.tier.aln_cap) { status |= FQ_SF_ALN_OVERFLOW; return false; }
    if (FQ_LANE_ID() == 0) {
      FqAln h;
      h.info = (uint32_t)n_mm | (uint32_t)n_gapo << 8 | (uint32_t)n_gape << 16 | (uint32_t)a << 24;
      h.k = hk; h.l = hl; h.score = e_score;
      al[n_aln] = h;
    }
    ++n_aln;
    FQ_WAVE_FENCE();
    return true;
  }
};

// one wavefront: reads from the queue, one at a time
template <class Fetch>
FQ_HD void fq_gap_coop_wave(const FqGapArgs &A, uint32_t *heads, Fetch fetch, int wave_slot) {
  FqGapCoop C(A, heads, wave_slot);
  uint32_t rounds = 0;
  for (;;) {
    uint32_t wq = 0;
    if (FQ_LANE_ID() == 0) wq = fetch(1u);
    wq = FQ_READLANE32(wq, 0);
    if (wq >= (uint32_t)A.n_work) break;
    rounds += C.run(A.order ? A.order[wq] : (int)wq);
  }
  if (FQ_LANE_ID() == 0) { FQ_ATOMIC_MAX64(&A.counters[FQ_C_MAXTRIPS], rounds); FQ_ATOMIC_ADD64(&A.counters[FQ_C_SUMTRIPS], rounds); }
}

// ---- K_sa: bwt_sa over enumerated SA rows (src/BwtMapper.cpp:770-772, 811-853) -------------------
struct FqSaArgs {
  FqDevIndex ix;
  const FqAln *aln;          // packed hits
  const uint32_t *aln_len;   // read length per packed hit (p[j]->len at enumeration time)
  const uint64_t *row_off;   // exclusive prefix of enumerated widths per packed hit, [n_aln+1]
  uint32_t n_aln;
  uint64_t n_rows;
  uint32_t *pos;             // out [n_rows]
  uint64_t *counters;
};
FQ_HD void fq_sa_thread(const FqSaArgs &A, uint64_t q) {
  // find hit g with row_off[g] <= q < row_off[g+1]
  uint32_t lo = 0, hi = A.n_aln;
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (A.row_off[mid] <= q) lo = mid; else hi = mid;
  }
  const FqAln h = A.aln[lo];
  const uint32_t row = h.k + (uint32_t)(q - A.row_off[lo]);
  const int a = (int)(h.info >> 24) & 1;
  uint32_t steps = 0, p;
  if (a) p = fq_sa_lookup(A.ix.fm[0], row, &steps);
  else p = A.ix.fm[1].seq_len - (fq_sa_lookup(A.ix.fm[1], row, &steps) + A.aln_len[lo]);
  A.pos[q] = p;
  FQ_ATOMIC_ADD64(&A.counters[FQ_C_OCC_SA], steps);
}
// ---- dynamic programming: libbwa/stdaln.c with aln_param_bwa {26,9,5,aln_sm_maq,5,50} (:227) ------
#define FQ_GAP_O 26
#define FQ_GAP_E 9
#define FQ_GAP_END 5
#define FQ_BAND 50
FQ_HD int fq_sm_maq(int a, int b) { return (a > 3 || b > 3) ? -13 : (a == b ? 11 : -19); }  // aln_sm_maq, stdaln.c:206-212

struct FqCell { int M, I, D; };
// trace byte: Mt[0:2) It[2] (0=M,1=I) Dt[3] (0=M,1=D)
FQ_HD int fq_pick_M(const FqCell &p, int sc, uint8_t &t) {   // set_M, stdaln.c:260-276
  if (p.M >= p.I) { if (p.M >= p.D) { t = FQ_OP_M; return p.M + sc; } t = FQ_OP_D; return p.D + sc; }
  if (p.I > p.D) { t = FQ_OP_I; return p.I + sc; }
  t = FQ_OP_D; return p.D + sc;
}
FQ_HD int fq_pick_gap(int m, int g, int ext, uint8_t &from_m) {  // set_I / set_D / set_end_*, stdaln.c:277-319
  if (m - FQ_GAP_O > g) { from_m = 1; return m - FQ_GAP_O - ext; }
  from_m = 0; return g - ext;
}

// Row storage for the banded DP: one row of (M,I,D) cells updated in place.  Two layouts: a plain
// array in global memory, and lane-interleaved LDS words (element i of lane l at [i*stride + l], so a
// wavefront whose lanes sit at the same column is bank-conflict free).
struct FqRowsArr {
  FqCell *c;
  FQ_HD FqCell get(int i) const { return c[i]; }
  FQ_HD void set(int i, const FqCell &v) const { c[i] = v; }
};
struct FqRowsPlanar {
  int *M, *I, *D;
  int stride;
  FQ_HD FqCell get(int i) const { FqCell v; v.M = M[i * stride]; v.I = I[i * stride]; v.D = D[i * stride]; return v; }
  FQ_HD void set(int i, const FqCell &v) const { M[i * stride] = v.M; I[i * stride] = v.I; D[i * stride] = v.D; }
};

// Banded global alignment with end-gap penalty (aln_global_core, stdaln.c:345-525).  s1/s2 are
// 0-based code arrays; R holds len1+1 cells (single row, updated in place: the cell above is read
// before it is overwritten, the diagonal one is carried in registers); trace = (len2+1)*(len1+1)
// bytes.  ops receives the path (end -> start), returns score, *n_ops = path_len, (*fi,*fj) =
// coordinates of the last path element (path[path_len-1]).  Band geometry, edge rules (set_end_I /
// set_end_D, "I = -inf" on the band's right edge) and move precedence follow the reference exactly.
template <class Rows>
FQ_HD int fq_global_align(const uint8_t *s1, int len1, const uint8_t *s2, int len2, int band, int gap_end, const Rows &R,
                          uint8_t *trace, uint8_t *ops, int *n_ops, int *fi, int *fj) {
  if (len1 == 0 || len2 == 0) { *n_ops = 0; return 0; }
  int b1, b2;
  if (len1 > len2) { b1 = len1 - len2 + band; b2 = band; } else { b1 = band; b2 = len2 - len1 + band; }
  if (b1 > len1) b1 = len1;
  if (b2 > len2) b2 = len2;
  const int W = len1 + 1;
  const int end_ext = gap_end >= 0 ? gap_end : FQ_GAP_E;
  uint8_t fm;
  {
    FqCell left;
    left.M = 0; left.I = left.D = FQ_NEG_INF;
    R.set(0, left);
    for (int i = 1; i < b1; ++i) {
      FqCell c;
      c.M = c.I = FQ_NEG_INF;
      c.D = fq_pick_gap(left.M, left.D, end_ext, fm);
      trace[i] = (uint8_t)(fm ? 0 : 8);
      R.set(i, c);
      left = c;
    }
  }
  const int p1_end = b2 < len2 ? b2 : len2 - 1;
  for (int j = 1; j <= len2; ++j) {
    int phase;
    if (j <= p1_end) phase = 1;
    else if (j == p1_end + 1 && j == len2 && b2 != len2 - 1) phase = 5;   // "last row for part 1"
    else if (j <= len2 - b2 + 1) phase = 2;
    else if (j < len2) phase = 3;
    else phase = 4;
    const int c2 = s2[j - 1];
    uint8_t *tr = trace + (size_t)j * (size_t)W;
    const bool endD = (phase == 5 || phase == 4);
    int lo, hi;
    FqCell diag, left;
    if (phase == 1 || phase == 5) {
      lo = 1; hi = (j + b1 <= len1 + 1) ? j + b1 - 1 : len1;
      diag = R.get(0);
      left.M = left.D = FQ_NEG_INF;
      left.I = fq_pick_gap(diag.M, diag.I, end_ext, fm);
      tr[0] = (uint8_t)(fm ? 0 : 4);
      R.set(0, left);
    } else {
      lo = j - b2 + 1; hi = (phase == 2) ? j + b1 - 1 : len1;
      diag = R.get(j - b2);
      left.M = left.I = left.D = FQ_NEG_INF;
      R.set(j - b2, left);
    }
    for (int i = lo; i <= hi; ++i) {
      const FqCell up = R.get(i);
      FqCell c;
      uint8_t tM, tb;
      c.M = fq_pick_M(diag, fq_sm_maq(s1[i - 1], c2), tM);
      tb = tM;
      if (i != hi) { c.I = fq_pick_gap(up.M, up.I, FQ_GAP_E, fm); tb |= (uint8_t)(fm ? 0 : 4); }
      else if (phase == 1 || phase == 5) {
        if (j + b1 - 1 > len1) { c.I = fq_pick_gap(up.M, up.I, end_ext, fm); tb |= (uint8_t)(fm ? 0 : 4); }
        else c.I = FQ_NEG_INF;
      } else if (phase == 2) c.I = FQ_NEG_INF;
      else { c.I = fq_pick_gap(up.M, up.I, end_ext, fm); tb |= (uint8_t)(fm ? 0 : 4); }
      c.D = fq_pick_gap(left.M, left.D, endD ? end_ext : FQ_GAP_E, fm);
      tb |= (uint8_t)(fm ? 0 : 8);
      tr[i] = tb;
      R.set(i, c);
      diag = up;
      left = c;
    }
  }
  // traceback (stdaln.c:484-512)
  const FqCell last = R.get(len1);
  int i = len1, j = len2, mx = last.M;
  uint8_t tb = trace[(size_t)j * W + i];
  int type = tb & 3, ctype = FQ_OP_M;
  if (last.I > mx) { mx = last.I; type = (tb & 4) ? FQ_OP_I : FQ_OP_M; ctype = FQ_OP_I; }
  if (last.D > mx) { mx = last.D; type = (tb & 8) ? FQ_OP_D : FQ_OP_M; ctype = FQ_OP_D; }
  int n = 0, li = i, lj = j;
  ops[n++] = (uint8_t)ctype;
  do {
    if (ctype == FQ_OP_M) { --i; --j; } else if (ctype == FQ_OP_I) --j; else --i;
    ctype = type;
    tb = trace[(size_t)j * W + i];
    type = type == FQ_OP_M ? (tb & 3) : type == FQ_OP_I ? ((tb & 4) ? FQ_OP_I : FQ_OP_M) : ((tb & 8) ? FQ_OP_D : FQ_OP_M);
    if (i || j) { ops[n++] = (uint8_t)ctype; li = i; lj = j; }
  } while (i || j);
  *n_ops = n;
  *fi = li; *fj = lj;
  return mx;
}

// path -> run-length cigar in alignment order (aln_path2cigar32 stdaln.c:1010-1040 + bwa_aln_path2cigar bwtaln.c:352)
FQ_HD int fq_ops_to_cigar(const uint8_t *ops, int n_ops, uint16_t *cg, int cap) {
  int n = 0;
  for (int t = n_ops - 1; t >= 0; --t) {
    if (n && (cg[n - 1] >> 14) == ops[t]) ++cg[n - 1];
    else { if (n >= cap) return -1; cg[n++] = (uint16_t)(ops[t] << 14 | 1); }
  }
  return n;
}

// per-task global-memory scratch shared by the SW and refine kernels
struct FqDpScratch {
  uint8_t *ref, *qry, *ops, *trace;
  int *H, *E;
  FqCell *rows;
};
FQ_HD size_t fq_dp_scratch_bytes(int RL, int QL) {
  size_t b = 0;
  b += ((size_t)RL + 16) & ~(size_t)15;                       // ref
  b += ((size_t)QL + 16) & ~(size_t)15;                       // qry
  b += ((size_t)RL + QL + 16) & ~(size_t)15;                  // ops
  b += 2 * ((((size_t)RL + 2) * sizeof(int) + 15) & ~(size_t)15);   // H, E
  b += ((size_t)RL + 1) * sizeof(FqCell) + 16;                // rows
  b += ((size_t)RL + 1) * ((size_t)QL + 1) + 16;              // trace
  return (b + 63) & ~(size_t)63;
}
FQ_HD FqDpScratch fq_dp_carve(uint8_t *base, int RL, int QL) {
  FqDpScratch s;
  uint8_t *p = base;
  s.ref = p; p += ((size_t)RL + 16) & ~(size_t)15;
  s.qry = p; p += ((size_t)QL + 16) & ~(size_t)15;
  s.ops = p; p += ((size_t)RL + QL + 16) & ~(size_t)15;
  s.H = (int *)p; p += (((size_t)RL + 2) * sizeof(int) + 15) & ~(size_t)15;
  s.E = (int *)p; p += (((size_t)RL + 2) * sizeof(int) + 15) & ~(size_t)15;
  s.rows = (FqCell *)p; p += ((size_t)RL + 1) * sizeof(FqCell) + 16;
  p = (uint8_t *)(((uintptr_t)p + 15) & ~(uintptr_t)15);
  s.trace = p;
  return s;
}

// ---- K_sw: bwa_sw_core (libbwa/bwape.c:359-445) with aln_local_core (stdaln.c:529-761) ----------
// Scores stay far below the 32000 re-basing threshold of stdaln.c:247 for reads <= FQ_LMAX
// (11*500 = 5500), so the overflow arm is unreachable and omitted.
struct FqSwArgs {
  FqDevIndex ix;
  const uint8_t *seq;
  int32_t stride;
  const int32_t *len_trim;
  const FqSwTask *task;
  int32_t n_task;
  FqSwOut *out;
  uint16_t *cigar;      // [n_task][cig_cap]
  int32_t cig_cap;
  uint8_t *scratch;
  size_t scratch_stride;
  int32_t RL, QL;       // scratch dimensions
  int32_t trace_in_lds; // set by the HIP launcher: the wavefront kernel keeps the fill's trace matrix in LDS
  int32_t serial_reverse; // ... and whether aln_local_core's reverse pass runs as the serial statement on one lane (test knob)
};

// one cell of the forward pass of aln_local_core (stdaln.c:585-612); state carried by the caller
FQ_HD int fq_sw_cell(int diag, int up, int e_up, int sc, int &last_h, int &f, int &e_out) {
  const int q = FQ_GAP_O, r = FQ_GAP_E, qr = q + r;
  int h = diag + sc;
  if (h < 0) h = 0;
  if (last_h > 0) { f = f > last_h - q ? f - r : last_h - qr; if (h < f) h = f; }
  if (up >= qr + 1) { const int e = e_up > up - q ? e_up - r : up - qr; if (h < e) h = e; e_out = e; }
  else e_out = 0;
  last_h = h;
  return h;
}

// forward pass, sequential form (H[x] = H(row,x); E[x] = E for column x+1, as the packed eh[] of the reference)
FQ_HD void fq_sw_forward_seq(const uint8_t *ref, int len1, const uint8_t *qry, int len2, int *H, int *E, int *score_f, int *end_i, int *end_j) {
  for (int i = 0; i <= len1 + 1; ++i) H[i] = E[i] = 0;
  int sf = 0, ei = 0, ej = 0;
  for (int j = 1; j <= len2; ++j) {
    int last_h = 0, f = 0, diag = H[0];
    const int c2 = qry[j - 1];
    for (int i = 1; i <= len1; ++i) {
      const int up = H[i];
      int e_out;
      const int prev_h = last_h;
      const int h = fq_sw_cell(diag, up, E[i - 1], fq_sm_maq(ref[i - 1], c2), last_h, f, e_out);
      E[i - 1] = e_out;
      diag = up;
      H[i - 1] = prev_h;
      if (sf < h) { sf = h; ei = i; ej = j; }
    }
    H[len1] = last_h; E[len1] = 0;
  }
  *score_f = sf; *end_i = ei; *end_j = ej;
}

// everything after the forward pass: banded reverse pass (stdaln.c:626-679), global fill with band doubling
// (:709-727), CIGAR + clipping + mismatch/gap counts (bwape.c:389-443).  H/E: len1+2 ints of scratch.
// aln_local_core's reverse pass (stdaln.c:640-700): where the best local alignment starts.  Serial by nature: the band it
// sweeps depends on the running maximum.
// A row is walked four cells at a time: the cells' inputs from the row before (H[i], E[i + 1]) do not depend on the cells being
// computed -- a cell writes H[i + 1] / E[i + 1], behind the ones still to be read -- so they are fetched together and the chain
// through last_h / f runs out of registers; on the device the rows live in LDS, and a load per cell on the dependent chain was
// most of the kernel's time (one lane walks this pass while the wavefront waits).
FQ_HD void fq_sw_reverse(const uint8_t *ref, const uint8_t *qry, int score_f, int end_i, int end_j, int *H, int *E, int *start_i_out, int *start_j_out, int *score_r_out) {
  const int q = FQ_GAP_O, rr = FQ_GAP_E, qr = q + rr;
  for (int i = end_i; i >= 0; --i) H[i] = E[i] = 0;
  int score_r = fq_sm_maq(ref[end_i - 1], qry[end_j - 1]);
  int start_i = end_i, start_j = end_j;
  H[end_i] = qr + score_r; E[end_i] = 0;
  {
    int start = end_i - 1, end = end_i - 3;
    if (end <= 0) end = 0;
    for (int j = end_j - 1; j != 0; --j) {
      int last_h = 0, f = 0, i = start;
      bool stop = false;
      const int c2 = qry[j - 1];
      // one cell (stdaln.c:650-668): diag = H[i + 1], side = H[i], e_in = E[i + 1] of the row before
      auto cell = [&](int ii, int diag, int side, int e_in, int &e_out) {
        int h = diag + fq_sm_maq(ref[ii - 1], c2);
        if (h < 0) h = 0;
        if (last_h > 0) { f = f > last_h - q ? f - rr : last_h - qr; if (h < f) h = f; }
        int e = e_in > side - q ? e_in - rr : side - qr;
        if (e < 0) e = 0;
        if (h < e) h = e;
        e_out = e;
        return h;
      };
      while (i - end >= 4 && !stop) {
        const int d0 = H[i + 1], s0 = H[i], s1 = H[i - 1], s2 = H[i - 2], s3 = H[i - 3];
        const int e0 = E[i + 1], e1 = E[i], e2 = E[i - 1], e3 = E[i - 2];
        const int diag[4] = {d0, s0, s1, s2}, side[4] = {s0, s1, s2, s3}, ein[4] = {e0, e1, e2, e3};
        int wh[4], we[4], done = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (stop) break;
          const int ii = i - u;
          int e;
          const int h = cell(ii, diag[u], side[u], ein[u], e);
          wh[u] = last_h; we[u] = e;
          last_h = h;
          done = u + 1;
          if (score_r < h) {
            score_r = h; start_i = ii; start_j = j;
            if (score_r - qr == score_f) stop = true;
          }
        }
        for (int u = 0; u < done; ++u) { H[i - u + 1] = wh[u]; E[i - u + 1] = we[u]; }
        if (stop) { i -= done - 1; break; }      // (i = the cell that ended the pass, as the cell-by-cell loop leaves it)
        i -= 4;
      }
      if (!stop)
        for (; i != end; --i) {
          int e;
          const int h = cell(i, H[i + 1], H[i], E[i + 1], e);
          H[i + 1] = last_h; E[i + 1] = e;
          last_h = h;
          if (score_r < h) {
            score_r = h; start_i = i; start_j = j;
            if (score_r - qr == score_f) { stop = true; break; }
          }
        }
      H[i + 1] = last_h; E[i + 1] = 0;
      if (stop) break;
      if (H[start] <= qr) --start;
      if (start <= 0) start = 0;
      end = start_i - (start_j - j) - (score_r + (start_j - j) * 11) / rr - 1;
      if (end <= 0) end = 0;
    }
  }
  *start_i_out = start_i; *start_j_out = start_j; *score_r_out = score_r - qr;
}
// after the banded fill of the sub-rectangle: CIGAR with clips, position and difference counts (bwa_sw_core, bwape.c:393-445)
FQ_HD void fq_sw_post(const FqSwTask &T, const uint8_t *ref, const uint8_t *qry, int len, int end_j, int start_i, int start_j, int fi, int fj,
                      const uint8_t *ops, int n_ops, uint16_t *cg, int cig_cap, FqSwOut &O) {
  fi += start_i - 1; fj += start_j - 1;
  int n_cigar = fq_ops_to_cigar(ops, n_ops, cg + 1, cig_cap - 2);   // slot 0 reserved for a leading S
  if (n_cigar <= 0) return;
  uint32_t x = 0, y = 0;
  for (int k = 0; k < n_cigar; ++k) {
    const int op = cg[1 + k] >> 14, ln = cg[1 + k] & 0x3fff;
    if (op == FQ_OP_M) { x += ln; y += ln; } else if (op == FQ_OP_D) x += ln; else y += ln;
  }
  if (x < 20 || y < 20) return;
  const int start = (fj ? fj : 1) - 1, endq = end_j;   // path[0].j == end_j
  O.beg = T.beg + ((fi ? fi : 1) - 1);
  int off = 1;
  if (start) { cg[0] = (uint16_t)(FQ_OP_S << 14 | start); off = 0; ++n_cigar; }
  if (endq < len) { cg[off + n_cigar] = (uint16_t)(FQ_OP_S << 14 | (len - endq)); ++n_cigar; }
  if (off) for (int k = 0; k < n_cigar; ++k) cg[k] = cg[k + 1];
  int n_mm = 0, n_gapo = 0, n_gape = 0;
  x = fi ? fi - 1 : 0; y = fj ? fj - 1 : 0;
  for (int k = 0; k < n_cigar; ++k) {
    const int op = cg[k] >> 14, ln = cg[k] & 0x3fff;
    if (op == FQ_OP_M) {
      for (int u = 0; u < ln; ++u) if (ref[x + u] < 4 && qry[y + u] < 4 && ref[x + u] != qry[y + u]) ++n_mm;
      x += ln; y += ln;
    } else if (op == FQ_OP_D) { x += ln; ++n_gapo; n_gape += ln - 1; }
    else if (op == FQ_OP_I) { y += ln; ++n_gapo; n_gape += ln - 1; }
  }
  O.cnt = (uint32_t)n_mm << 16 | (uint32_t)n_gapo << 8 | (uint32_t)n_gape;
  O.n_cigar = n_cigar;
}
template <class Rows>
FQ_HD void fq_sw_finish(const FqSwTask &T, const uint8_t *ref, int len1, const uint8_t *qry, int len, int score_f, int end_i, int end_j,
                        int *H, int *E, const Rows &R, uint8_t *trace, uint8_t *ops, uint16_t *cg, int cig_cap, FqSwOut &O) {
  if (score_f < 1) return;
  int start_i, start_j, score_r;
  fq_sw_reverse(ref, qry, score_f, end_i, end_j, H, E, &start_i, &start_j, &score_r);
  int n_ops = 0, fi = 0, fj = 0, score_g;
  const int jmax = (end_i - start_i > end_j - start_j ? end_i - start_i : end_j - start_j) + 1;
  for (int b = FQ_BAND;; b <<= 1) {   // doubling band (stdaln.c:705-716)
    score_g = fq_global_align(ref + start_i - 1, end_i - start_i + 1, qry + start_j - 1, end_j - start_j + 1, b, -1, R, trace, ops, &n_ops, &fi, &fj);
    if (score_g == score_r || score_f == score_g) break;
    if (b > jmax) break;
  }
  if (score_r > score_g && score_f > score_g) return;   // "Potential bug" arm of the reference: ret < 0
  fq_sw_post(T, ref, qry, len, end_j, start_i, start_j, fi, fj, ops, n_ops, cg, cig_cap, O);
}
FQ_HD bool fq_sw_prologue(const FqSwArgs &A, const FqSwTask &T, uint8_t *ref, uint8_t *qry, int *len_out, int *len1_out) {
  const int len = A.len_trim[T.read];
  const uint8_t *row = A.seq + (size_t)T.read * (size_t)A.stride;
  const int64_t l_pac = A.ix.l_pac;
  *len_out = len;
  if (T.reglen < 20 || l_pac - T.beg < len) return false;
  int nn = 0;
  for (int k = 0; k < len; ++k) {
    const int c = T.use_rc ? fq_comp(fq_nt4(row[len - 1 - k])) : fq_nt4(row[k]);
    qry[k] = (uint8_t)c;
    nn += c >= 4;
  }
  if ((float)nn / len >= 0.25f || len - nn < 20) return false;
  int l1 = 0;
  for (int64_t k = T.beg; l1 < T.reglen && k < l_pac; ++k) ref[l1++] = (uint8_t)fq_pac_base(A.ix.pac, k);
  *len1_out = l1;
  return true;
}

// sequential form: one thread does the whole task (host-loop test backend; also the reference point for the wave kernel)
FQ_HD void fq_sw_thread(const FqSwArgs &A, int t) {
  const FqSwTask T = A.task[t];
  FqSwOut O;
  O.beg = T.beg; O.cnt = 0; O.n_cigar = 0;
  FqDpScratch S = fq_dp_carve(A.scratch + (size_t)t * A.scratch_stride, A.RL, A.QL);
  int len, len1;
  if (fq_sw_prologue(A, T, S.ref, S.qry, &len, &len1)) {
    int score_f, end_i, end_j;
    fq_sw_forward_seq(S.ref, len1, S.qry, len, S.H, S.E, &score_f, &end_i, &end_j);
    FqRowsArr R = {S.rows};
    fq_sw_finish(T, S.ref, len1, S.qry, len, score_f, end_i, end_j, S.H, S.E, R, S.trace, S.ops, A.cigar + (size_t)t * (size_t)A.cig_cap, A.cig_cap, O);
  }
  A.out[t] = O;
}

// ---- K_refine: refine_gapped_core (libbwa/bwase.c:183-232, is_end_correct=1) ---------------------
struct FqRefineArgs {
  FqDevIndex ix;
  const uint8_t *seq;
  int32_t stride;
  const int32_t *len_trim;
  const FqRefTask *task;
  int32_t n_task;
  FqRefOut *out;
  uint16_t *cigar;
  int32_t cig_cap;
  uint8_t *scratch;
  size_t scratch_stride;
  int32_t RL, QL;
};
template <class Rows>
FQ_HD void fq_refine_task(const FqRefineArgs &A, int t, const Rows &R) {
  const FqRefTask T = A.task[t];
  const int len = A.len_trim[T.read];
  const uint8_t *row = A.seq + (size_t)T.read * (size_t)A.stride;
  const int64_t l_pac = A.ix.l_pac;
  FqDpScratch S = fq_dp_carve(A.scratch + (size_t)t * A.scratch_stride, A.RL, A.QL);
  for (int k = 0; k < len; ++k) S.qry[k] = (uint8_t)(T.strand ? fq_comp(fq_nt4(row[len - 1 - k])) : fq_nt4(row[k]));
  int64_t pos = (int64_t)T.pos > l_pac ? (int64_t)(int32_t)T.pos : (int64_t)T.pos;
  const int aext = T.ext < 0 ? -T.ext : T.ext, ref_len = len + aext;
  int l = 0;
  if (T.ext > 0) { for (int64_t k = pos; k < pos + ref_len && k < l_pac; ++k) S.ref[l++] = (uint8_t)fq_pac_base(A.ix.pac, k); }
  else {
    const int64_t x = pos + len;
    for (int64_t k = x - ref_len > 0 ? x - ref_len : 0; k < x && k < l_pac; ++k) S.ref[l++] = (uint8_t)fq_pac_base(A.ix.pac, k);
  }
  int n_ops = 0, fi, fj;
  fq_global_align(S.ref, l, S.qry, len, FQ_BAND, FQ_GAP_END, R, S.trace, S.ops, &n_ops, &fi, &fj);
  uint16_t *cg = A.cigar + (size_t)t * (size_t)A.cig_cap;
  int n = fq_ops_to_cigar(S.ops, n_ops, cg, A.cig_cap);
  FqRefOut O;
  if (n <= 0) { O.pos = T.pos; O.n_cigar = 0; A.out[t] = O; return; }
  if (T.ext < 0) {
    int d = 0;
    for (int k = 0; k < n; ++k) { const int op = cg[k] >> 14, ln = cg[k] & 0x3fff; if (op == FQ_OP_D) d -= ln; else if (op == FQ_OP_I) d += ln; }
    pos += d;
  }
  if ((cg[0] >> 14) == FQ_OP_D) { pos += cg[0] & 0x3fff; for (int k = 0; k < n - 1; ++k) cg[k] = cg[k + 1]; --n; }
  if ((cg[n - 1] >> 14) == FQ_OP_D) --n;
  if ((cg[n - 1] >> 14) == FQ_OP_I) cg[n - 1] = (uint16_t)(FQ_OP_S << 14 | (cg[n - 1] & 0x3fff));
  if ((cg[0] >> 14) == FQ_OP_I) cg[0] = (uint16_t)(FQ_OP_S << 14 | (cg[0] & 0x3fff));
  O.pos = (uint32_t)pos; O.n_cigar = n;
  A.out[t] = O;
}
FQ_HD void fq_refine_thread(const FqRefineArgs &A, int t) {
  FqDpScratch S = fq_dp_carve(A.scratch + (size_t)t * A.scratch_stride, A.RL, A.QL);
  FqRowsArr R = {S.rows};
  fq_refine_task(A, t, R);
}

// ---- K_md: bwa_cal_md1 (libbwa/bwase.c:234-296) ---------------------------------------------------
FQ_HD int fq_put_int(char *dst, int at, int cap, int v) {
  char tmp[12]; int n = 0;
  if (v == 0) tmp[n++] = '0';
  while (v > 0) { tmp[n++] = (char)('0' + v % 10); v /= 10; }
  while (n > 0) { if (at < cap) dst[at] = tmp[n - 1]; ++at; --n; }
  return at;
}
// the MD string of one read into dst (at most cap characters are written; *at_out is the length it has, which may exceed cap) and its NM
FQ_HD void fq_md_core(const FqDevIndex &ix, const uint8_t *row, const FqMdTask &T, const uint16_t *cigar_arena, char *dst, int cap, int *at_out, int *nm_out) {
  const int64_t l_pac = ix.l_pac;
  // the sequence MD is computed against: s->strand ? s->rseq : s->seq, over the (trimmed) length at that time
  const int slen = T.len;
  uint32_t x = T.pos, y = 0;
  int u = 0, nm = 0, at = 0;
  if (T.n_cigar) {
    const uint16_t *cg = cigar_arena + T.cigar_off;
    for (int k = 0; k < T.n_cigar; ++k) {
      const int op = cg[k] >> 14, l = cg[k] & 0x3fff;
      if (op == FQ_OP_M) {
        for (int z = 0; z < l && (int64_t)(uint32_t)(x + z) < l_pac; ++z) {
          const int c = fq_pac_base(ix.pac, (int64_t)(uint32_t)(x + z));
          const int yy = (int)y + z;
          const int sc = T.strand ? fq_comp(fq_nt4(row[slen - 1 - yy])) : fq_nt4(row[yy]);
          if (sc > 3 || c != sc) { at = fq_put_int(dst, at, cap, u); if (at < cap) dst[at] = "ACGTN"[c]; ++at; ++nm; u = 0; }
          else ++u;
        }
        x += l; y += l;
      } else if (op == FQ_OP_I || op == FQ_OP_S) { y += l; if (op == FQ_OP_I) nm += l; }
      else {
        at = fq_put_int(dst, at, cap, u);
        if (at < cap) dst[at] = '^';
        ++at;
        for (int z = 0; z < l && (int64_t)(uint32_t)(x + z) < l_pac; ++z) { if (at < cap) dst[at] = "ACGT"[fq_pac_base(ix.pac, (int64_t)(uint32_t)(x + z))]; ++at; }
        u = 0; x += l; nm += l;
      }
    }
  } else {
    // ungapped (almost every read): eight positions per pair of loads -- 8 read bytes, 4 bytes of the 2-bit reference
    int z = 0;
    for (; z + 8 <= slen; z += 8) {
      uint64_t rb;
      uint32_t pb;
      memcpy(&rb, T.strand ? row + (slen - 8 - z) : row + z, 8);
      const uint32_t k0 = x + (uint32_t)z;
      memcpy(&pb, ix.pac + (k0 >> 2), 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t kk = k0 + (uint32_t)j;
        const int c = (int)((pb >> (8 * ((kk >> 2) - (k0 >> 2)))) >> ((~kk & 3u) << 1)) & 3;
        const int ch = (int)(rb >> (8 * (T.strand ? 7 - j : j))) & 0xff;
        const int sc = T.strand ? fq_comp(fq_nt4((uint8_t)ch)) : fq_nt4((uint8_t)ch);
        if (sc > 3 || c != sc) { at = fq_put_int(dst, at, cap, u); if (at < cap) dst[at] = "ACGTN"[c]; ++at; ++nm; u = 0; }
        else ++u;
      }
    }
    for (; z < slen; ++z) {
      const int c = fq_pac_base(ix.pac, (int64_t)(uint32_t)(x + z));
      const int sc = T.strand ? fq_comp(fq_nt4(row[slen - 1 - z])) : fq_nt4(row[z]);
      if (sc > 3 || c != sc) { at = fq_put_int(dst, at, cap, u); if (at < cap) dst[at] = "ACGTN"[c]; ++at; ++nm; u = 0; }
      else ++u;
    }
  }
  at = fq_put_int(dst, at, cap, u);
  *at_out = at; *nm_out = nm;
}

// ---- K_pair: paired-end pairing (pairing, libbwa/bwape.c:119-213; __pairing_aux / __pairing_aux2, bwape.h:55-82) --------------
// One lane per pair whose two reads are both mapped: every position of every hit of both reads (the rows k_sa resolved) becomes
// x = pos << 32 | hit index << 1 | end, the list is sorted, and one sweep keeps the last two forward hits of each end and tries
// them against every reverse hit of the other.  The insert-size term of a pair's score is (int)(-4.343 * log(.5 * erfc(...)) + .499),
// an integer-valued function of the insert size for one reference batch: the host evaluates it with its libm for every insert size
// up to high_bayesian and the kernel looks it up, so the scores are the reference's to the last bit.  The same routine is what
// the host runs for the pairs the kernel does not take (an interval of >= 1,000 rows: its positions are those of the (k,l)
// cache's first requester; more rows than a lane's scratch).
struct FqPairRead { uint32_t pos; int32_t len, full_len; uint32_t bits; };   // bits: strand | mapQ << 8 | seQ << 16
struct FqPairOut { uint32_t pos, info; int32_t score; uint32_t bits; };      // info: n_mm | n_gapo << 8 | n_gape << 16; bits: mapQ | seQ << 8 | strand << 16 | paired << 24 | moved << 25
struct FqPairIsize { uint32_t high, high_bayesian; int32_t lut_off, pad; };  // of one reference batch; lut[lut_off + l], l <= high_bayesian
FQ_HD uint64_t fq_hash_64(uint64_t key) {   // libbwa/bwape.h:42-53
  key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
  key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
  return key;
}
struct FqPairAcc { uint64_t o_score, subo_score, o_pos0, o_pos1; int o_n, subo_n; };
FQ_HD void fq_pair_try(const FqAln *aln0, const FqAln *aln1, const FqPairRead *r, const FqPairIsize &ii, const int32_t *lut, int max_isize, int max_len,
                       uint64_t u, uint64_t v, FqPairAcc &A) {
  if (u == (uint64_t)-1) return;
  const uint32_t l = (uint32_t)(v >> 32) + (uint32_t)r[v & 1].len - (uint32_t)(u >> 32);
  if (!((v >> 32) > (u >> 32) && l >= (uint32_t)max_len && ((ii.high && l <= ii.high_bayesian) || (ii.high == 0 && l <= (uint32_t)max_isize)))) return;
  const FqAln &av = ((v & 1) ? aln1 : aln0)[(uint32_t)v >> 1], &au = ((u & 1) ? aln1 : aln0)[(uint32_t)u >> 1];
  uint64_t s = (uint64_t)(int64_t)(av.score + au.score);
  s *= 10;
  if (ii.high) s += (uint64_t)(int64_t)lut[ii.lut_off + (int32_t)l];   // bwape.h:62
  s = s << 32 | (uint32_t)fq_hash_64((u >> 32 << 32) | (v >> 32));
  if (s >> 32 == A.o_score >> 32) ++A.o_n;
  else if (s >> 32 < A.o_score << 32) { A.subo_n += A.o_n; A.o_n = 1; }   // (the typo of bwape.h:65, reproduced)
  else ++A.subo_n;
  if (s < A.o_score) { A.subo_score = A.o_score; A.o_score = s; if (u & 1) A.o_pos1 = u; else A.o_pos0 = u; if (v & 1) A.o_pos1 = v; else A.o_pos0 = v; }
  else if (s < A.subo_score) A.subo_score = s;
}
// arr: the pair's sorted list.  out[e].bits bit 24 clear: no proper pair, the records stay as they are.
FQ_HD void fq_pair_sweep(const FqAln *aln0, const FqAln *aln1, const FqPairRead *r, const uint64_t *arr, uint32_t n, const FqPairIsize &ii,
                         const int32_t *lut, const int32_t *g_log_n, int max_isize, int s_mm, FqPairOut *out) {
  FqPairAcc A;
  A.o_score = A.subo_score = (uint64_t)-1; A.o_n = A.subo_n = 0; A.o_pos0 = A.o_pos1 = 0;
  uint64_t l00 = (uint64_t)-1, l01 = (uint64_t)-1, l10 = (uint64_t)-1, l11 = (uint64_t)-1;   // last[end][0..1]
  const int max_len = r[0].full_len > r[1].full_len ? r[0].full_len : r[1].full_len;
  for (uint32_t i = 0; i < n; ++i) {
    const uint64_t x = arr[i];
    const FqAln &ax = ((x & 1) ? aln1 : aln0)[(uint32_t)x >> 1];
    if (((ax.info >> 24) & 1u) == 1u) {
      const bool other1 = (x & 1) == 0;      // try against the other end's forward hits
      fq_pair_try(aln0, aln1, r, ii, lut, max_isize, max_len, other1 ? l11 : l01, x, A);
      fq_pair_try(aln0, aln1, r, ii, lut, max_isize, max_len, other1 ? l10 : l00, x, A);
    } else if (x & 1) { l10 = l11; l11 = x; }
    else { l00 = l01; l01 = x; }
  }
  out[0].bits = out[1].bits = 0; out[0].pos = out[1].pos = 0; out[0].info = out[1].info = 0; out[0].score = out[1].score = 0;
  if (A.o_score == (uint64_t)-1) return;
  int mapQ_p = 0;
  if (A.o_n == 1) {
    if (A.subo_score == (uint64_t)-1) mapQ_p = 29;
    else if ((A.subo_score >> 32) - (A.o_score >> 32) > (uint64_t)(s_mm * 10)) mapQ_p = 23;
    else {
      const int nn = A.subo_n > 255 ? 255 : A.subo_n;
      mapQ_p = (int)(((A.subo_score >> 32) - (A.o_score >> 32)) / 2) - g_log_n[nn];
      if (mapQ_p < 0) mapQ_p = 0;
    }
  }
  const FqAln &b0 = aln0[(uint32_t)A.o_pos0 >> 1], &b1 = aln1[(uint32_t)A.o_pos1 >> 1];
  const int rr0 = (int)(b0.info >> 24) & 1, rr1 = (int)(b1.info >> 24) & 1;
  int mq0 = (int)(r[0].bits >> 8) & 0xff, mq1 = (int)(r[1].bits >> 8) & 0xff, sq0 = (int)(r[0].bits >> 16) & 0xff, sq1 = (int)(r[1].bits >> 16) & 0xff;
  const bool same0 = r[0].pos == (uint32_t)(A.o_pos0 >> 32) && (int)(r[0].bits & 1u) == rr0;
  const bool same1 = r[1].pos == (uint32_t)(A.o_pos1 >> 32) && (int)(r[1].bits & 1u) == rr1;
  if (same0 && same1) {
    if (mq0 > 0 && mq1 > 0) {
      int mq = mq0 + mq1;
      if (mq > 60) mq = 60;
      mq0 = mq1 = mq;
    } else {
      if (mq0 == 0) mq0 = mapQ_p + 7 < mq1 ? mapQ_p + 7 : mq1;
      if (mq1 == 0) mq1 = mapQ_p + 7 < mq0 ? mapQ_p + 7 : mq0;
    }
  } else if (same0) { sq1 = 0; mq1 = mq0; if (mq1 > mapQ_p) mq1 = mapQ_p; }
  else if (same1) { sq0 = 0; mq0 = mq1; if (mq0 > mapQ_p) mq0 = mapQ_p; }
  else { sq0 = sq1 = 0; mapQ_p -= 20; if (mapQ_p < 0) mapQ_p = 0; mq0 = mq1 = mapQ_p; }
  out[0].pos = (uint32_t)(A.o_pos0 >> 32); out[0].info = b0.info & 0xffffffu; out[0].score = b0.score;
  out[0].bits = (uint32_t)mq0 | (uint32_t)sq0 << 8 | (uint32_t)rr0 << 16 | 1u << 24 | (same0 ? 0u : 1u << 25);
  out[1].pos = (uint32_t)(A.o_pos1 >> 32); out[1].info = b1.info & 0xffffffu; out[1].score = b1.score;
  out[1].bits = (uint32_t)mq1 | (uint32_t)sq1 << 8 | (uint32_t)rr1 << 16 | 1u << 24 | (same1 ? 0u : 1u << 25);
}
