// fq_common.h -- structures shared by host code and HIP kernels of the FASTQuick-align hot path.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FQ_HD __host__ __device__ __forceinline__
#define FQ_D __device__ __forceinline__
#else
#define FQ_HD inline
#define FQ_D inline
#endif

// ---- limits (validated in fq_ctx_create; exceeding them is FQ_EINVAL / FQ_ELIMIT, never silent) ----
#define FQ_LMIN 15             // shortest read taken: bwa_cal_maxdiff(15..37) = 2 at the default fnr, so the per-slice max_gapo clamp stays a no-op
#define FQ_LMAX 500            // read length; entry packing below gives 9 bits to i / last_diff
#define FQ_MAX_BUCKETS 128     // score buckets of the search stack (74 for 150 bp defaults)
#define FQ_SEED_MAX 64         // seed_len upper bound (default 32)
#define FQ_NIL 0xffffffffu

// record vocabulary (libbwa/bwtaln.h)
enum { FQ_ST_M = 0, FQ_ST_I = 1, FQ_ST_D = 2 };
enum { FQ_OP_M = 0, FQ_OP_I = 1, FQ_OP_D = 2, FQ_OP_S = 3 };
#define FQ_MODE_GAPE 1
#define FQ_MODE_COMPREAD 2
#define FQ_MODE_LOGGAP 4
#define FQ_MODE_NONSTOP 0x10
#define FQ_MODE_IL13 0x200   /* Illumina 1.3+ qualities (Phred+64): 31 is taken off every quality byte on input, BwtMapper.cpp:549-553 */
#define FQ_NEG_INF (-1073741823)

// ---- FM index in HBM -----------------------------------------------------------------------
// The file format interleaves 4 cumulative counts + 128 bases in 48 bytes (libbwa/bwt.h:56-63),
// which straddles cache lines.  We re-lay each strand as 32-byte blocks of 64 bases: counts of
// A/C/G/T before the block plus the low and high bit planes of its 64 symbols (base t of the
// block at bit 63-t), so one aligned 32-byte fetch answers Occ for all four bases with four
// popcounts.  Results are identical to bwt_occ/bwt_occ4 (bwt.h:98-222) by construction.
struct FqOccBlk {
  uint32_t cnt[4];
  uint64_t lo, hi;
};
struct FqFM {
  const FqOccBlk *blk;
  const uint32_t *sa;      // sampled suffix array, sa[0] = 0xffffffff (libbwa/bwtio.c:29-49)
  uint32_t primary, seq_len;
  uint32_t L2[5];
  uint32_t sa_intv, n_sa, n_blk;
};
struct FqDevIndex {
  FqFM fm[2];              // [0] forward text (.bwt/.sa), [1] reversed text (.rbwt/.rsa)
  const uint8_t *pac;      // 2 bits/base, MSB first (libbwa/bwtaln.h:26)
  int64_t l_pac;
  const uint8_t *bitmap[6];
};

// the contigs of the reduced reference (bntseq_t::anns, libbwa/bntseq.h) and its N holes, resident beside the index
struct FqDevContigs {
  int32_t n, n_holes;
  const int64_t *off;          // [n] offset in the concatenated reference
  const int32_t *len;          // [n]
  const uint32_t *name_off;    // [n + 1] into names
  const char *names;
  const int64_t *hole_off;     // [n_holes]
  const int32_t *hole_len;
};

// option block passed by value to kernels
struct FqKOpts {
  int32_t s_mm, s_gapo, s_gape, mode;
  int32_t indel_end_skip, max_del_occ, max_entries;
  int32_t max_gapo, max_gape, max_seed_diff, seed_len, max_top2;
  int32_t trim_qual, filter_thresh;
  int32_t n_buckets;
};

// bwt_aln1_t (libbwa/bwtaln.h:34-38)
struct FqAln {
  uint32_t info;   // n_mm | n_gapo<<8 | n_gape<<16 | a<<24
  uint32_t k, l;
  int32_t score;
};

// search-stack entry (gap_entry_t, libbwa/bwtgap.h:7-12) packed into 16 bytes; `next` links the
// LIFO chain of its score bucket.
struct FqEntry {
  uint32_t k, l, pk, next;
};
// pk: i[0:9) a[9] state[10:12) n_mm[12:17) n_gapo[17:19) n_gape[19:23) last_diff[23:32)
// (fq_ctx_create validates max_diff <= 30, max_gapo <= 3, max_gape <= 15, read length <= 500 accordingly)
FQ_HD uint32_t fq_pack(int i, int a, int st, int mm, int go, int ge, int ld) {
  return (uint32_t)i | (uint32_t)a << 9 | (uint32_t)st << 10 | (uint32_t)mm << 12 | (uint32_t)go << 17 | (uint32_t)ge << 19 |
         (uint32_t)ld << 23;
}

// What one search step needs to know about read position p of strand a, packed into 16 bits so that a 16-byte load
// covers eight consecutive positions (the search walks p downwards one position per step):
//   bid[0:5)   width[a][p].bid, clamped to 31 (max_diff <= 30: a clamped value compares like the exact one)
//   eq[5]      p >= 1 && width[a][p-1].w == width[a][p].w
//   sbid[6:11) seed_width[a][p - seed_off].bid (clamped), seq[11] the same equality for the seed widths
//   base[12:15) seq[a][p] (0..3, 4 = N, 5 = '-')
// Written by k_width; gap_shadow keeps bid/eq in step with the exact widths (wfull) it modifies.
typedef uint16_t FqPos;
#define FQ_POS_PAD 8   // slack behind each strand's array so that the 8-position window may start anywhere

// start-of-search record of work item w (k_width -> gap kernel)
struct FqGapWork {
  int32_t r;             // read index (row of the batch)
  uint32_t meta;         // len[0:16) max_diff[16:24) too-many-N[24]
};

// per-read search status flags
#define FQ_SF_POOL_OVERFLOW 1u   // entry pool exhausted -> rerun in a larger tier
#define FQ_SF_ALN_OVERFLOW 2u    // more hits than the aln slot holds -> rerun in a larger tier
#define FQ_SF_ENTRY_LIMIT 4u     // conservative entry count crossed max_entries -> exact tier decides
#define FQ_SF_LONG 8u            // lane-per-read kernel: more pops than tier.long_pops -> searched again by a whole wavefront
#define FQ_SF_NOHIT 32u          // with FQ_SF_NEEDGAP: the round without gap children found no hit at all (the full search will be a long one)
#define FQ_SF_NEEDGAP 16u        // search without gap children (tier.nogap): the full search could pop one -> searched again in full

struct FqGapTier {       // one launch configuration of the gap-search kernel
  uint32_t pool_cap;     // entries per read
  uint32_t aln_cap;      // hits per read
  int32_t exact;         // 1: no push-time pruning, n_entries tracked exactly (honours max_entries as the reference)
  int32_t coop;          // 1: one read per wavefront (fq_gap_coop_wave), 0: one read per lane (fq_gap_lanes)
  uint32_t long_pops;    // lane kernel: give up on a read after this many pops once the work queue is empty (0: never)
  int32_t long_always;   // test hook: give up after long_pops pops whatever the state of the queue
  int32_t nogap;         // lane kernel: 1 = first round of a large launch, the search without its gap children (see FqGapLane):
                         // a read whose result could depend on them is flagged FQ_SF_NEEDGAP and searched in full by the next round
  int32_t lane_major;    // lane kernel: 1 = a lane's pool is contiguous (slot s at [s]) instead of interleaved with its wavefront's (slot s at [s * 64])
};

// SW / refine task descriptors
struct FqSwTask {
  int32_t read;          // global read index (e*n + i) of the mate being placed
  int32_t use_rc;        // 1: align reverse complement (p->rseq), 0: forward read
  int64_t beg;
  int32_t reglen;
  int32_t pad;
};
struct FqSwOut {
  int64_t beg;
  uint32_t cnt;          // n_mm<<16 | n_gapo<<8 | n_gape
  int32_t n_cigar;       // 0 = no acceptable alignment
};
struct FqRefTask {
  int32_t read;
  int32_t strand;        // which sequence to align (1: rseq)
  uint32_t pos;
  int32_t ext;           // signed gap count, refine_gapped_core's `ext`
};
struct FqRefOut {
  uint32_t pos;
  int32_t n_cigar;
};
struct FqMdTask {
  int32_t read;
  int32_t strand;
  uint32_t pos;
  int32_t n_cigar;       // 0: ungapped
  uint32_t cigar_off;    // into the device cigar arena
  int32_t len;           // s->len at MD time
};

// work counters written by kernels (one u64 each, atomically accumulated per wave).  On the device the array is kept FQ_C_STRIPES times, a cache
// line or more apart (FQ_C_STRIDE), and a workgroup adds to the copy blockIdx picks: the atomics on one cache line are worked one after the other by
// its L2 channel (~7 ns each: 130,000 wavefronts' worth per counter were most of the 3 ms of a filter launch over 8.4 M reads).  The host folds the
// copies when it reads them (sums; maxima for the two *MAX* counters).
#define FQ_C_STRIPES 64
#define FQ_C_STRIDE 64   /* >= FQ_C_COUNT, a multiple of 16 (128 bytes) */
enum { FQ_C_OCC_WIDTH = 0, FQ_C_OCC_GAP, FQ_C_OCC_SA, FQ_C_PROBES, FQ_C_POPS, FQ_C_PUSHES, FQ_C_MAXPOPS, FQ_C_POPS_GT4K, FQ_C_MAXTRIPS, FQ_C_SUMTRIPS, FQ_C_LANETRIPS, FQ_C_BASES, FQ_C_BADLEN, FQ_C_OCC_NOGAP, FQ_C_DBG0, FQ_C_DBG_END = FQ_C_DBG0 + 16,
       // record stages (fq_records.h): pairs a lane paired, pairs with both ends unmapped, main hits resolved without enumeration, refine / MD slots that were too small
       FQ_C_PAIRS_DEV = FQ_C_DBG_END, FQ_C_UNMAPPED, FQ_C_SA_DIRECT, FQ_C_ERR_CIGAR, FQ_C_ERR_MD, FQ_C_MD_READS, FQ_C_ERR_DRAW0, FQ_C_COUNT };
static_assert(FQ_C_COUNT <= FQ_C_STRIDE && FQ_C_STRIDE % 16 == 0, "a counter stripe holds every counter and is whole cache lines");
