// fq_frontend.h -- bodies of the FASTQ front end's device kernels: BGZF members inflated in HBM, the inflated text cut
// into lines and records, the read filter's keys formed from the text, the reference's read-slot history, the surviving reads' rows.
//
// What they replace: the reader side of bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:476-613) -- gzread (libbwa/bwaseqio.c:41-52) under
// kseq_read3_fpc (libbwa/kseq.h:327-371), one record at a time on one IO thread per file (IOworkerAlt, src/BwtMapper.cpp:1973-1980) -- for
// the files sequencers write: BGZF containers of four-line records.  Anything else is refused here, record by record and member by member,
// and goes the host's byte-wise way (fq_fastq.cpp, fq_inflate.h, zlib), whose verdicts stand.
//
// The bodies are written for a wavefront of FQ_WAVE_SIZE lanes: 64 on the device; the host-loop build (tests/emu, test infrastructure) runs
// the same code with a wavefront of one.
#pragma once
#include "fq_kernels.h"

// Per-lane code is written between FQF_LANES / FQF_LANES_END: on the device the block runs once, for the lane the thread is; in the host-loop
// build (tests/emu, test infrastructure) it is a loop over the 64 lanes of the emulated wavefront, and a per-lane register (FQF_LVAR) is an
// array of 64.  Code between such blocks is the wavefront's uniform part.  Blocks are written so that no lane reads what another lane of the
// same block writes (then the order of the lanes does not matter).
#if defined(__HIP_DEVICE_COMPILE__)
#define FQF_LANES { const int lane = (int)(threadIdx.x & 63);
#define FQF_LANES_END }
#define FQF_LVAR(T, name) T name
#define FQF_LV(name) name
#define FQF_RL(name, idx) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(idx)))
#define FQF_NOINLINE __attribute__((noinline))
#else
#define FQF_LANES for (int lane = 0; lane < 64; ++lane) {
#define FQF_LANES_END }
#define FQF_LVAR(T, name) T name[64]
#define FQF_LV(name) name[lane]
#define FQF_RL(name, idx) (name[idx])
#define FQF_NOINLINE
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define FQF_UNIFORM32(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#define FQF_BREV32(x) __brev((unsigned)(x))
// Lanes of one wavefront hand bytes to each other through LDS and through the text in HBM.  The hardware keeps one wavefront's LDS operations,
// and its vector-memory operations, in issue order; what has to be pinned is the compiler: no memory operation moves across this point
// (the asm's memory clobber), and the point is not moved into or out of divergent code (wave_barrier is convergent).  No instruction is emitted.
#define FQF_WAVE_FENCE() do { __asm__ volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#define FQF_WAVE_XOR32(x) fqf_wave_xor32(x)
__device__ __forceinline__ uint32_t fqf_wave_xor32(uint32_t x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d, 64);
  return x;
}
// (a look first: atomics on one cache line are worked one after the other by its L2 channel, ~8 ns each -- 330,000 of them, five per wavefront of
//  records, were 2.5 ms of a 2.6 ms launch; the value is almost always there already.  The look goes past the CU's L1, which would keep an old value.)
#define FQF_ATOMIC_MIN32(p, v) do { const unsigned fqf_v_ = (unsigned)(v); if (fqf_v_ < __hip_atomic_load((const unsigned *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin((unsigned *)(p), fqf_v_); } while (0)
#define FQF_ATOMIC_MAX32(p, v) do { const unsigned fqf_v_ = (unsigned)(v); if (fqf_v_ > __hip_atomic_load((const unsigned *)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((unsigned *)(p), fqf_v_); } while (0)
#define FQF_ATOMIC_OR32(p, v) atomicOr((unsigned *)(p), (unsigned)(v))
#else
#define FQF_UNIFORM32(x) ((uint32_t)(x))
inline uint32_t fqf_brev32(uint32_t v) {
  v = (v >> 16) | (v << 16);
  v = ((v & 0xff00ff00u) >> 8) | ((v & 0x00ff00ffu) << 8);
  v = ((v & 0xf0f0f0f0u) >> 4) | ((v & 0x0f0f0f0fu) << 4);
  v = ((v & 0xccccccccu) >> 2) | ((v & 0x33333333u) << 2);
  v = ((v & 0xaaaaaaaau) >> 1) | ((v & 0x55555555u) << 1);
  return v;
}
#define FQF_BREV32(x) fqf_brev32((uint32_t)(x))
#define FQF_WAVE_FENCE() do { } while (0)
#define FQF_WAVE_XOR32(x) (x)
#define FQF_ATOMIC_MIN32(p, v) (*(p) = *(p) < (uint32_t)(v) ? *(p) : (uint32_t)(v))
#define FQF_ATOMIC_MAX32(p, v) (*(p) = *(p) > (uint32_t)(v) ? *(p) : (uint32_t)(v))
#define FQF_ATOMIC_OR32(p, v) (*(p) |= (uint32_t)(v))
#endif

// =====================================================================================================================================
// CRC-32 (IEEE 802.3, the gzip trailer's) of a member's output by a whole wavefront: every lane takes a contiguous piece, the pieces' values
// are put together with crc(A || B) = crc(A) * x^(8 |B|) + crc(B) in GF(2)[x] / P (the identity zlib's crc32_combine rests on; bit 31 of a
// word is the coefficient of x^0 in the reflected representation the gzip CRC uses).
// =====================================================================================================================================
#define FQZ_POLY 0xEDB88320u
FQ_HD uint32_t fqz_mulmod(uint32_t a, uint32_t b) {      // a * b mod P
  uint32_t p = 0;
  for (int i = 0; i < 32; ++i) {
    p ^= (0u - ((a >> (31 - i)) & 1u)) & b;
    b = (b >> 1) ^ ((0u - (b & 1u)) & FQZ_POLY);          // b * x
  }
  return p;
}
// tables of the host and of the device: slice-by-4 CRC tables t[4][256], and pow8[k] = x^(8 * 2^k) mod P
struct FqzCrcConst { uint32_t t[4][256]; uint32_t pow8[32]; };
inline void fqz_crc_const_make(FqzCrcConst *c) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t v = i;
    for (int k = 0; k < 8; ++k) v = (v >> 1) ^ ((0u - (v & 1u)) & FQZ_POLY);
    c->t[0][i] = v;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int s = 1; s < 4; ++s) c->t[s][i] = (c->t[s - 1][i] >> 8) ^ c->t[0][c->t[s - 1][i] & 0xff];
  uint32_t p = 0x80000000u >> 8;                          // x^8
  for (int k = 0; k < 32; ++k) { c->pow8[k] = p; p = fqz_mulmod(p, p); }
}
FQ_HD uint32_t fqz_xpow8(const uint32_t *pow8, uint32_t n) {   // x^(8 n) mod P
  uint32_t p = 0x80000000u;
  for (int k = 0; n; ++k, n >>= 1) if (n & 1u) p = fqz_mulmod(pow8[k], p);
  return p;
}

// =====================================================================================================================================
// One BGZF member (a raw DEFLATE stream, RFC 1951, of at most 64 KiB of text) per wavefront.
//   * The compressed bytes are read 256 at a time into three registers per lane (the block being read and the two behind it); the bit buffer
//     is refilled a dword at a time from them (readlane), so the symbol loop makes no memory access for its input.
//   * Decode tables in LDS: a 10-bit root table for literal / length codes and an 8-bit one for distances, one look-up per symbol; a code
//     longer than the root is found by its canonical position (first code and count per length, symbols sorted by code).  Tables are built
//     by all lanes (a lane per table slot).
//   * The symbol loop is uniform over the wavefront (every lane holds the same bit buffer): its arithmetic is scalar work for the compiler.
//     The kernel is bound by the number of instructions a symbol costs (measured: the SIMDs' issue slots are 90 % taken), so the loop has
//     a FAST form -- no change of the input registers, the table entries laid out for one bit-field extract per value, one pass of the
//     lanes per match -- and a GENERAL step (fqz_general_symbol) for what the fast form leaves: the end of an input block, codes longer than
//     the root, matches longer than 64 bytes, the member's first line of text.
//   * Output goes to an LDS ring indexed by the absolute output position; matches are copied by all lanes, from the ring while the distance
//     is inside it, from the text already in HBM otherwise; every completed 256-byte line of the ring goes out as one coalesced store.
//   * The CRC-32 of the output is computed by the wavefront afterwards and compared with the member's trailer.
// status: 0 = inflated and checked; 1 = not a stream this decoder takes (the host's decoder, then zlib, decide); 2 = CRC mismatch.
// =====================================================================================================================================
#ifndef FQZ_RING
#define FQZ_RING 2048
#endif
#ifndef FQZ_LROOT
#define FQZ_LROOT 9
#endif
static_assert((FQZ_RING & (FQZ_RING - 1)) == 0 && FQZ_RING >= 1024 && FQZ_LROOT >= 9 && FQZ_LROOT <= 11, "ring a power of two that holds a line and the longest match; root table of 9 to 11 bits");
#define FQZ_DROOT 8
enum { FQZ_OK = 0, FQZ_REFUSED = 1, FQZ_BADCRC = 2 };
#if defined(FQZ_STATS) && !defined(__HIP_DEVICE_COMPILE__)
// symbol mix of a run through the host statement (tools/inflate_mix.cpp): [0] literals [1] near matches [2] far matches [3] matches over their own bytes [4] general steps [5] refills [6] match bytes
static unsigned long long fqz_stats[16];
#define FQZ_STAT(k, n) (fqz_stats[k] += (n))
#else
#define FQZ_STAT(k, n) ((void)0)
#endif
struct FqzMember { uint64_t in_off; uint32_t out_off, in_len, out_len, crc; uint32_t pad[2]; };   // payload [in_off, in_off + in_len) of `comp`; text at out + out_off (a launch writes at most 4 GiB)
struct FqInflateArgs {
  const uint8_t *comp;          // compressed bytes, 4-byte aligned, at least 1 KiB of readable slack behind the last member
  const FqzMember *mem;
  int32_t n_mem;
  uint8_t *out;                 // 256-byte aligned
  uint32_t *status;             // [n_mem]
  const FqzCrcConst *crc;
};
struct FqzLds {
  uint32_t ring32[FQZ_RING / 4]; // first, and the struct aligned to its size: a ring address is (position & (FQZ_RING - 1)) | the struct's LDS address
  uint32_t lt[1 << FQZ_LROOT];  // literal / length root table; the CRC's slice tables afterwards
  uint32_t dt[1 << FQZ_DROOT];   // (its first 128 words hold the code-length code's table while a block's code lengths are read)
  uint16_t lsort[288], dsort[32], csort[20];
  uint16_t lfirst[16], lcount[16], loffs[16];
  uint16_t dfirst[16], dcount[16], doffs[16];
  uint16_t cfirst[16], ccount[16], coffs[16];
  uint8_t lens[320], cl[32];
};
// Table entries, laid out so that S_BFE_U32(bit buffer, entry) IS the symbol's extra bits (the instruction reads its offset from bits 0-4
// and its width from bits 16-22 of its second operand and ignores the rest):
//   bits 0-4   code length            bits 16-19 extra bits (20-22 zero)
//   bits 5-9   bits the whole symbol takes (code + extra; 0: not in the root table)
//   bit 10 literal, bit 11 length (of at most 64 bytes, extra bits included) / distance, bit 12 end of block, bit 15 a longer length
//   bits 23-31 literal / length base (<= 258);  distances: bits 13-14 k with base = (k << extra) + 1
enum { FQZ_L1_MISS = 31 };     // (a code length no entry has)
enum { FQZ_K_LIT = 1u << 10, FQZ_K_BASE = 1u << 11, FQZ_K_EOB = 1u << 12, FQZ_K_LONG = 1u << 15 };   // FQZ_K_LONG: a length whose match may exceed 64 bytes (the general form's)
FQ_HD uint32_t fqz_take_of(uint32_t e) { return (e >> 5) & 31; }
FQ_HD uint32_t fqz_extra_of(uint64_t bb, uint32_t e) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t r;
  __asm__("s_bfe_u32 %0, %1, %2" : "=s"(r) : "s"((uint32_t)bb), "s"(e) : "scc");
  return r;
#else
  return ((uint32_t)bb >> (e & 31)) & ((1u << ((e >> 16) & 15)) - 1);
#endif
}
FQ_HD uint32_t fqz_lit_entry(uint32_t s, uint32_t l) {
  if (s < 256) return FQZ_K_LIT | s << 23 | l << 5 | l;
  if (s == 256) return FQZ_K_EOB | l << 5 | l;
  if (s > 285) return 0;                                     // 286, 287: in the fixed code, never valid in data
  const uint32_t k = s - 257;
  uint32_t base, xb = 0;
  if (k < 8) base = 3 + k;
  else if (k == 28) base = 258;
  else { xb = (k >> 2) - 1; base = 3 + ((4 + (k & 3)) << xb); }
  return (base + (1u << xb) - 1 <= 64 ? FQZ_K_BASE : FQZ_K_LONG) | base << 23 | xb << 16 | (l + xb) << 5 | l;
}
FQ_HD uint32_t fqz_dist_entry(uint32_t s, uint32_t l) {
  if (s > 29) return 0;
  const uint32_t xb = s < 4 ? 0 : (s >> 1) - 1, k = s < 4 ? s : 2 + (s & 1);
  return FQZ_K_BASE | k << 13 | xb << 16 | (l + xb) << 5 | l;
}
FQ_HD uint32_t fqz_dist_base(uint32_t f) { return (((f >> 13) & 3) << ((f >> 16) & 15)) + 1; }
FQ_HD uint32_t fqz_dist_of(uint64_t bb, uint32_t f) { return (((f >> 13) & 3) << ((f >> 16) & 15)) + 1 + fqz_extra_of(bb, f); }
FQ_HD uint32_t fqz_cl_entry(uint32_t s, uint32_t l) { return FQZ_K_LIT | s << 23 | l << 5 | l; }

// Canonical Huffman code of n symbols with lengths len[] (0: unused): count / first code / offset per length, the symbols sorted by
// (length, symbol).  Returns 0: complete; 1: incomplete (code words unused); -1: over-subscribed.  *n_used: symbols in use.
FQ_HD int fqz_canon(const uint8_t *len, int n, uint16_t *sorted, uint16_t *first, uint16_t *count, uint16_t *offs, int *n_used) {
  FQF_LANES
    if (lane < 16) {                                         // (a lane per length)
      int c = 0;
      for (int s = 0; s < n; ++s) c += len[s] == lane;
      count[lane] = (uint16_t)c;
    }
  FQF_LANES_END
  FQF_WAVE_FENCE();
  int left = 1, used = 0;
  uint32_t code = 0, off = 0;
  int over = 0;
  for (int l = 1; l <= 15; ++l) {
    const int c = (int)FQF_UNIFORM32(count[l]);
    if (!over) { left = left * 2 - c; if (left < 0) over = 1; }
    first[l] = (uint16_t)code; offs[l] = (uint16_t)off;      // (every lane the same values)
    code = (code + (uint32_t)c) << 1;
    off += (uint32_t)c;
    used += c;
  }
  FQF_WAVE_FENCE();
  *n_used = used;
  if (over) return -1;
  FQF_LANES
    if (lane >= 1 && lane < 16) {
      uint32_t at = offs[lane];
      for (int s = 0; s < n; ++s) if (len[s] == lane) sorted[at++] = (uint16_t)s;
    }
  FQF_LANES_END
  FQF_WAVE_FENCE();
  return left > 0 ? 1 : 0;
}
// root table of 2^root slots: every lane looks its slots' bit patterns up in the canonical code
template <class Entry>
FQ_HD void fqz_fill_root(uint32_t *table, int root, const uint16_t *sorted, const uint16_t *first, const uint16_t *count, const uint16_t *offs, Entry entry) {
  FQF_LANES
    for (uint32_t i = (uint32_t)lane; i < (1u << root); i += 64) {
      const uint32_t r = FQF_BREV32(i);
      uint32_t e = 0;
      for (int l = 1; l <= root; ++l) {
        const uint32_t idx = (r >> (32 - l)) - first[l];
        if (idx < count[l]) { e = entry(sorted[offs[l] + idx], (uint32_t)l); break; }
      }
      table[i] = e;
    }
  FQF_LANES_END
  FQF_WAVE_FENCE();
}
// a code longer than the root: by its canonical position (0: no such code)
template <class Entry>
FQ_HD uint32_t fqz_long_code(uint64_t bb, int root, const uint16_t *sorted, const uint16_t *first, const uint16_t *count, const uint16_t *offs, Entry entry) {
  const uint32_t r = FQF_BREV32((uint32_t)bb);
  for (int l = root + 1; l <= 15; ++l) {
    const uint32_t idx = (r >> (32 - l)) - first[l];
    if (idx < count[l]) return entry(sorted[offs[l] + idx], (uint32_t)l);
  }
  return 0;
}

// the decoder's state: uniform over the wavefront but for the three input registers
struct FqzSt {
  // ---- input ----
  const uint32_t *base;      // the payload's first dword (aligned down)
  FQF_LVAR(uint32_t, wcur);  // dwords [blk * 64 + lane] of the block being read ...
  FQF_LVAR(uint32_t, wnext); // ... of the one behind it ...
  FQF_LVAR(uint32_t, wfar);  // ... and of the one behind that (asked for when `blk` began)
  // ---- the block's first-level tables: lane i holds the root tables' entry for a bit buffer whose low 6 bits are i, when its code is of 6 bits at
  //      most (FQZ_L1_MISS otherwise).  In registers, read with v_readlane: no LDS round trip in a symbol's chain of dependent steps.
  FQF_LVAR(uint32_t, l1e); FQF_LVAR(uint32_t, l1b); FQF_LVAR(uint32_t, l1t);   // literal / length: entry, its base (bits 23-31), the bits it takes
  FQF_LVAR(uint32_t, d1e); FQF_LVAR(uint32_t, d1b); FQF_LVAR(uint32_t, d1t);   // distance: entry, its base (the distance without the extra bits), the bits it takes
  FQF_LVAR(uint32_t, l1scratch);                                                // (the host statement's copy register)
  uint32_t blk, di;          // the block; the next dword of it to take (0 .. 64)
  uint32_t blk_top;          // blocks behind this one are not read (the slack the payload is promised ends there)
  uint64_t bb;               // bit buffer, next bit at bit 0
  int bc;
  uint64_t bits_end;         // bit position (from base) behind the payload
  bool overrun;              // the stream asked for bytes behind the payload's slack
  // ---- output: absolute positions in the text buffer ----
  uint8_t *out;
  uint32_t g0, g, g_end;     // the member's first byte, the next byte, the byte behind its last
  uint32_t flushed;          // whole lines below this are in HBM (a multiple of 256)
  uint8_t *ring;
};
FQ_HD void fqz_first_level(FqzSt &D, const FqzLds &S) {       // after the root tables of a block have been filled
  FQF_LANES
    const uint32_t e = S.lt[lane], f = S.dt[lane];           // (root index = lane: the bits above the low six are zero)
    const bool e_in = fqz_take_of(e) != 0 && (e & 31) <= 6, f_in = fqz_take_of(f) != 0 && (f & 31) <= 6;
    FQF_LV(D.l1e) = e_in ? e : (uint32_t)FQZ_L1_MISS; FQF_LV(D.l1b) = e >> 23; FQF_LV(D.l1t) = fqz_take_of(e);
    FQF_LV(D.d1e) = f_in ? f : (uint32_t)FQZ_L1_MISS; FQF_LV(D.d1b) = fqz_dist_base(f); FQF_LV(D.d1t) = fqz_take_of(f);
  FQF_LANES_END
}
FQ_HD void fqz_load_blocks(FqzSt &D, uint32_t blk) {         // the three input registers for reading at block blk
  D.blk = blk;
  FQF_LANES
    const uint32_t b1 = blk + 1 < D.blk_top ? blk + 1 : D.blk_top, b2 = blk + 2 < D.blk_top ? blk + 2 : D.blk_top, b0 = blk < D.blk_top ? blk : D.blk_top;
    FQF_LV(D.wcur) = D.base[(size_t)b0 * 64 + (uint32_t)lane];
    FQF_LV(D.wnext) = D.base[(size_t)b1 * 64 + (uint32_t)lane];
    FQF_LV(D.wfar) = D.base[(size_t)b2 * 64 + (uint32_t)lane];
  FQF_LANES_END
}
FQ_HD void fqz_rotate(FqzSt &D) {                            // block D.blk is used up
  ++D.blk; D.di = 0;
  if (D.blk > D.blk_top) D.overrun = true;
  FQF_LANES
    const uint32_t b2 = D.blk + 2 < D.blk_top ? D.blk + 2 : D.blk_top;
    FQF_LV(D.wcur) = FQF_LV(D.wnext); FQF_LV(D.wnext) = FQF_LV(D.wfar);
    FQF_LV(D.wfar) = D.base[(size_t)b2 * 64 + (uint32_t)lane];
  FQF_LANES_END
}
FQ_HD void fqz_refill(FqzSt &D) {                            // called with bc <= 32: 32 more bits (general form: may end a block)
  if (D.di == 64) fqz_rotate(D);
  const uint32_t v = FQF_UNIFORM32(FQF_RL(D.wcur, D.di));
  D.bb |= (uint64_t)v << D.bc;
  D.bc += 32;
  ++D.di;
}
FQ_HD void fqz_need32(FqzSt &D) { if (D.bc <= 32) fqz_refill(D); }
FQ_HD void fqz_take(FqzSt &D, int n) { D.bb >>= n; D.bc -= n; }
FQ_HD uint64_t fqz_bit_pos(const FqzSt &D) { return ((uint64_t)D.blk * 64 + D.di) * 32 - (uint64_t)D.bc; }
FQ_HD void fqz_seek(FqzSt &D, uint64_t byte_pos) {           // byte_pos from base
  const uint32_t dw = (uint32_t)(byte_pos >> 2);
  fqz_load_blocks(D, dw / 64);
  D.di = dw % 64;
  D.bb = 0; D.bc = 0;
  fqz_refill(D);
  fqz_take(D, (int)(byte_pos & 3) * 8);
}
// the whole 256-byte lines below D.g go to HBM (general form: the member's first line may begin inside a line)
FQ_HD void fqz_flush_lines(FqzSt &D) {
  while ((D.flushed >> 8) < (D.g >> 8)) {
    const uint32_t seg = D.flushed;
    if (seg >= D.g0) {
      FQF_LANES
        *(uint32_t *)(D.out + seg + 4 * (uint32_t)lane) = *(const uint32_t *)(D.ring + ((seg + 4 * (uint32_t)lane) & (FQZ_RING - 1)));
      FQF_LANES_END
    } else {
      FQF_LANES
        for (uint32_t p = D.g0 + (uint32_t)lane; p < seg + 256; p += 64) D.out[p] = D.ring[p & (FQZ_RING - 1)];
      FQF_LANES_END
    }
    D.flushed = seg + 256;
  }
  FQF_WAVE_FENCE();
}
FQ_HD void fqz_flush_rest(FqzSt &D) {
  const uint32_t lo = D.flushed > D.g0 ? D.flushed : D.g0;
  FQF_LANES
    for (uint32_t p = lo + (uint32_t)lane; p < D.g; p += 64) D.out[p] = D.ring[p & (FQZ_RING - 1)];
  FQF_LANES_END
  D.flushed = D.g;
  FQF_WAVE_FENCE();
}
// one pass of the lanes over bytes [c, c + 64) of a match of len bytes from `dist` back: byte i of a match is byte i mod dist of the dist bytes
// in front of it, which are complete -- so no lane reads what a lane of the same pass writes
FQ_HD void fqz_copy_pass(FqzSt &D, uint32_t len, uint32_t dist, uint32_t c) {
  const uint32_t g = D.g, s = g - dist;
  if (dist >= len) {
    if (dist <= FQZ_RING - 2 * 64) {                         // inside the ring (the bytes being written do not land on the ones they are copied from)
      FQF_LANES
        const uint32_t i = c + (uint32_t)lane;
        if (i < len) { const uint8_t b = D.ring[(s + i) & (FQZ_RING - 1)]; D.ring[(g + i) & (FQZ_RING - 1)] = b; }
      FQF_LANES_END
    } else {                                                 // from the text in HBM (stored when its line was completed)
      FQF_LANES
        const uint32_t i = c + (uint32_t)lane;
        if (i < len) { const uint8_t b = D.out[s + i]; D.ring[(g + i) & (FQZ_RING - 1)] = b; }
      FQF_LANES_END
    }
  } else {                                                   // the match runs into its own bytes: a period of dist
#if defined(__HIP_DEVICE_COMPILE__)
    const float rd = __builtin_amdgcn_rcpf((float)dist);
#else
    const float rd = 1.0f / (float)dist;
#endif
    const bool near = dist <= FQZ_RING - 2 * 64;
    FQF_LANES
      const uint32_t i = c + (uint32_t)lane;
      const uint32_t q = (uint32_t)((float)i * rd);          // i / dist, one too many or too few at most (i < 512)
      int32_t r = (int32_t)(i - q * dist);
      if (r < 0) r += (int32_t)dist;
      if (r >= (int32_t)dist) r -= (int32_t)dist;
      if (i < len) { const uint8_t b = near ? D.ring[(s + (uint32_t)r) & (FQZ_RING - 1)] : D.out[s + (uint32_t)r]; D.ring[(g + i) & (FQZ_RING - 1)] = b; }
    FQF_LANES_END
  }
  FQF_WAVE_FENCE();
}

// The general forms: everything the fast loop leaves.  0: go on; 1: end of block; 2: refused.
// fqz_general_dist: a length has been taken, the distance and the copy follow.
template <class DE>
FQ_HD int fqz_general_dist(FqzSt &D, FqzLds &S, uint32_t len, DE dent) {
  if (D.g > D.g_end) return 2;                               // (the fast form checks the promised size where a line is stored: it may be up to 255 bytes over)
  fqz_need32(D);
  uint32_t f = FQF_UNIFORM32(S.dt[D.bb & ((1u << FQZ_DROOT) - 1)]);
  if (!fqz_take_of(f)) f = FQF_UNIFORM32(fqz_long_code(D.bb, FQZ_DROOT, S.dsort, S.dfirst, S.dcount, S.doffs, dent));
  if (!fqz_take_of(f) || D.overrun) return 2;
  const uint32_t dist = fqz_dist_of(D.bb, f);
  fqz_take(D, (int)fqz_take_of(f));
  if (dist > D.g - D.g0 || len > D.g_end - D.g) return 2;
  for (uint32_t c = 0; c < len; c += 64) fqz_copy_pass(D, len, dist, c);
  const uint32_t g1 = D.g + len;
  const bool crossed = ((D.g ^ g1) >> 8) != 0;
  D.g = g1;
  if (crossed) fqz_flush_lines(D);
  return 0;
}
template <class LE, class DE>
FQ_HD int fqz_general_symbol(FqzSt &D, FqzLds &S, LE lent, DE dent) {
  if (D.g > D.g_end) return 2;
  fqz_need32(D);
  uint32_t e = FQF_UNIFORM32(S.lt[D.bb & ((1u << FQZ_LROOT) - 1)]);
  if (!fqz_take_of(e)) e = FQF_UNIFORM32(fqz_long_code(D.bb, FQZ_LROOT, S.lsort, S.lfirst, S.lcount, S.loffs, lent));
  if (!fqz_take_of(e) || D.overrun) return 2;
  if (e & FQZ_K_LIT) {
    fqz_take(D, (int)fqz_take_of(e));
    if (D.g >= D.g_end) return 2;
    FQF_LANES
      D.ring[D.g & (FQZ_RING - 1)] = (uint8_t)(e >> 23);     // (every lane stores the same byte at the same place)
    FQF_LANES_END
    FQF_WAVE_FENCE();
    ++D.g;
    if ((D.g & 255) == 0) fqz_flush_lines(D);
    return 0;
  }
  if (e & FQZ_K_EOB) { fqz_take(D, (int)fqz_take_of(e)); return 1; }
  if (!(e & (FQZ_K_BASE | FQZ_K_LONG))) return 2;
  const uint32_t len = (e >> 23) + fqz_extra_of(D.bb, e);
  fqz_take(D, (int)fqz_take_of(e));
  return fqz_general_dist(D, S, len, dent);
}
// len bytes of the input, byte-aligned at byte position p from base (a stored block)
FQ_HD void fqz_copy_stored(FqzSt &D, uint64_t p, uint32_t len) {
  const uint8_t *src = (const uint8_t *)D.base + p;
  for (uint32_t c = 0; c < len; c += 64) {
    FQF_LANES
      const uint32_t i = c + (uint32_t)lane;
      if (i < len) D.ring[(D.g + (uint32_t)lane) & (FQZ_RING - 1)] = src[i];
    FQF_LANES_END
    FQF_WAVE_FENCE();
    D.g += len - c < 64 ? len - c : 64;
    fqz_flush_lines(D);
  }
}

// the member's CRC-32 from the text in HBM: a piece per lane, 4 bytes per step (slice tables in LDS), the pieces put together
FQ_HD uint32_t fqz_crc_wave(const uint8_t *p, uint32_t n, const FqzCrcConst *cc, uint32_t *tab /* LDS, 1024 words */) {
  FQF_LANES
    for (int i = lane; i < 1024; i += 64) tab[i] = cc->t[i >> 8][i & 255];
  FQF_LANES_END
  FQF_WAVE_FENCE();
  const uint32_t *T0 = tab, *T1 = tab + 256, *T2 = tab + 512, *T3 = tab + 768;
  // pieces: head bytes up to a 4-byte boundary, 64 runs of whole dwords, tail bytes
  const uint32_t head = n < 4 ? n : (uint32_t)((4 - ((uintptr_t)p & 3)) & 3);
  const uint32_t n_dw = (n - head) / 4, tail = (n - head) & 3;
  const uint32_t per = (n_dw + 63) / 64;
  uint32_t sum = 0;
  FQF_LANES
    uint32_t total = 0;
    for (int v = lane; v < 66; v += 64) {                    // pieces 0 .. 65: head, 64 runs, tail (one run per lane, the first two lanes a second piece)
      uint32_t lo, len_b;                                    // byte range of the piece
      if (v == 0) { lo = 0; len_b = head; }
      else if (v == 65) { lo = head + 4 * n_dw; len_b = tail; }
      else {
        const uint32_t a = (uint32_t)(v - 1) * per < n_dw ? (uint32_t)(v - 1) * per : n_dw;
        const uint32_t b = a + per < n_dw ? a + per : n_dw;
        lo = head + 4 * a; len_b = 4 * (b - a);
      }
      uint32_t c = 0xffffffffu;
      const uint8_t *q = p + lo;
      if (v == 0 || v == 65) { for (uint32_t i = 0; i < len_b; ++i) c = T0[(c ^ q[i]) & 0xff] ^ (c >> 8); }
      else {
        const uint32_t *qw = (const uint32_t *)q;
        for (uint32_t i = 0; i < len_b / 4; ++i) {
          c ^= qw[i];
          c = T3[c & 0xff] ^ T2[(c >> 8) & 0xff] ^ T1[(c >> 16) & 0xff] ^ T0[c >> 24];
        }
      }
      c = len_b ? ~c : 0;                                    // crc32 of the piece (an empty piece: 0)
      const uint32_t after = n - lo - len_b;
      total ^= fqz_mulmod(fqz_xpow8(cc->pow8, after), c);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    sum = FQF_WAVE_XOR32(total);
#else
    sum ^= total;
#endif
  FQF_LANES_END
  return sum;
}

FQ_HD uint32_t fqz_inflate_member(const FqInflateArgs &A, int m, FqzLds &S) {
  const FqzMember M = A.mem[m];
  FqzSt D;
  D.out = A.out; D.g0 = M.out_off; D.g = M.out_off; D.g_end = M.out_off + M.out_len; D.flushed = M.out_off & ~255u; D.ring = (uint8_t *)S.ring32;
  const uint64_t a0 = M.in_off & ~(uint64_t)3;
  D.base = (const uint32_t *)(A.comp + a0);
  D.bits_end = ((M.in_off & 3) + (uint64_t)M.in_len) * 8;
  D.blk_top = (uint32_t)((D.bits_end + 31) / 32 / 64) + 1;  // (the block behind the payload's last: inside the 1 KiB of slack)
  D.overrun = false;
  fqz_seek(D, M.in_off & 3);
  int tables = 0;                                           // 0: none built, 1: the fixed code's, 2: a dynamic block's
  int status = FQZ_OK;
  auto lent = [](uint32_t s, uint32_t l) { return fqz_lit_entry(s, l); };
  auto dent = [](uint32_t s, uint32_t l) { return fqz_dist_entry(s, l); };
  auto cent = [](uint32_t s, uint32_t l) { return fqz_cl_entry(s, l); };
  for (bool last = false; !last && status == FQZ_OK;) {
    fqz_need32(D);
    if (D.overrun) { status = FQZ_REFUSED; break; }
    last = (D.bb & 1) != 0;
    const uint32_t type = (uint32_t)(D.bb >> 1) & 3;
    fqz_take(D, 3);
    if (type == 0) {                                         // stored: to the byte boundary, LEN, NLEN, bytes
      fqz_take(D, D.bc & 7);
      fqz_need32(D);
      const uint32_t len = (uint32_t)D.bb & 0xffff, nlen = (uint32_t)(D.bb >> 16) & 0xffff;
      fqz_take(D, 32);
      const uint64_t p = fqz_bit_pos(D) >> 3;
      if ((len ^ nlen) != 0xffffu || (p + len) * 8 > D.bits_end || len > D.g_end - D.g) { status = FQZ_REFUSED; break; }
      fqz_copy_stored(D, p, len);
      fqz_seek(D, p + len);
      continue;
    }
    if (type == 3) { status = FQZ_REFUSED; break; }
    int nl = 288, nd = 32;
    if (type == 1) {
      if (tables != 1) {
        FQF_LANES
          for (int s = lane; s < 320; s += 64) S.lens[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
        FQF_LANES_END
        FQF_WAVE_FENCE();
      }
    } else {
      const int hlit = (int)(D.bb & 31) + 257, hdist = (int)((D.bb >> 5) & 31) + 1, hclen = (int)((D.bb >> 10) & 15) + 4;
      fqz_take(D, 14);
      if (hlit > 286 || hdist > 30) { status = FQZ_REFUSED; break; }
      // the code-length code: up to 19 lengths of 3 bits, in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 (RFC 1951, 3.2.7)
      FQF_LANES
        if (lane < 32) S.cl[lane] = 0;
      FQF_LANES_END
      FQF_WAVE_FENCE();
      for (int i = 0; i < hclen; ++i) {
        fqz_need32(D);
        const uint32_t sym = i < 3 ? 16u + (uint32_t)i : i == 3 ? 0u : (i & 1) ? 7u - (uint32_t)((i - 5) / 2) : 8u + (uint32_t)((i - 4) / 2);
        S.cl[sym] = (uint8_t)(D.bb & 7);                     // (every lane the same byte)
        fqz_take(D, 3);
      }
      FQF_WAVE_FENCE();
      int used;
      if (FQF_UNIFORM32((uint32_t)fqz_canon(S.cl, 19, S.csort, S.cfirst, S.ccount, S.coffs, &used)) != 0) { status = FQZ_REFUSED; break; }   // (zlib: the code-length code must be complete)
      fqz_fill_root(S.dt, 7, S.csort, S.cfirst, S.ccount, S.coffs, cent);
      // the literal / length and distance code lengths, run-length coded with that code
      int i = 0;
      bool bad = false;
      while (i < hlit + hdist) {
        fqz_need32(D);
        if (D.overrun) { bad = true; break; }
        const uint32_t e = FQF_UNIFORM32(S.dt[D.bb & 127]);
        if (!fqz_take_of(e)) { bad = true; break; }
        fqz_take(D, (int)fqz_take_of(e));
        const int s = (int)(e >> 23);
        if (s < 16) { S.lens[i] = (uint8_t)s; ++i; continue; }
        int rep;
        uint32_t val = 0;
        if (s == 16) {
          if (i == 0) { bad = true; break; }
          FQF_WAVE_FENCE();
          val = FQF_UNIFORM32(S.lens[i - 1]); rep = 3 + (int)(D.bb & 3); fqz_take(D, 2);
        } else if (s == 17) { rep = 3 + (int)(D.bb & 7); fqz_take(D, 3); }
        else { rep = 11 + (int)(D.bb & 127); fqz_take(D, 7); }
        if (i + rep > hlit + hdist) { bad = true; break; }
        for (int k = 0; k < rep; ++k) S.lens[i + k] = (uint8_t)val;      // (every lane the same bytes)
        i += rep;
      }
      FQF_WAVE_FENCE();
      if (bad) { status = FQZ_REFUSED; break; }
      nl = hlit; nd = hdist;
      if (FQF_UNIFORM32(S.lens[256]) == 0) { status = FQZ_REFUSED; break; }   // no end-of-block code
    }
    if (type == 2 || tables != 1) {
      const uint8_t *dl = S.lens + (type == 1 ? 288 : nl);
      int used;
      int rc = (int)FQF_UNIFORM32((uint32_t)fqz_canon(S.lens, nl, S.lsort, S.lfirst, S.lcount, S.loffs, &used));
      if (rc != 0) { status = FQZ_REFUSED; break; }          // an incomplete literal / length code: zlib takes one case of it (a single one-bit code); left to it
      fqz_fill_root(S.lt, FQZ_LROOT, S.lsort, S.lfirst, S.lcount, S.loffs, lent);
      rc = (int)FQF_UNIFORM32((uint32_t)fqz_canon(dl, nd, S.dsort, S.dfirst, S.dcount, S.doffs, &used));
      used = (int)FQF_UNIFORM32((uint32_t)used);
      if (rc < 0 || (rc > 0 && !(used == 0 || (used == 1 && FQF_UNIFORM32(S.dcount[1]) == 1)))) { status = FQZ_REFUSED; break; }   // zlib's rule: incomplete only as no code at all or one one-bit code
      fqz_fill_root(S.dt, FQZ_DROOT, S.dsort, S.dfirst, S.dcount, S.doffs, dent);
      fqz_first_level(D, S);
      tables = type == 1 ? 1 : 2;
    }
    // ---- the block's symbols ----
    // The fast form runs while it meets nothing but root-table codes, matches of up to 64 bytes and input left in the block being read; its
    // state is the real state at every point, so whatever it meets it leaves the loop and the general form goes on from there:
    //   why 0: at a symbol's start (the block of input is used up, a code longer than the root, the end of the block, the promised size reached)
    //   why 1: a length has been taken (`len`); the distance needs the general form (input block used up, longer code, long match, bad distance)
    for (;;) {
      int why = 0;
      uint32_t len = 0;
#if defined(__HIP_DEVICE_COMPILE__)
      if (D.flushed > D.g0 || (D.g0 & 255) == 0) {           // (the member's first line of text, which may begin inside a line, is out)
        // The fast form, hand-written.  The wavefronts of a CU share ONE scalar unit, and a symbol's decoding is scalar work: with 41 scalar
        // instructions per match (of 70) the kernel ran at 80 % of that unit's one instruction per cycle whatever else was done
        // (profiles/round5_inflate_sq_counters_v3.txt).  This form spends about 20: table entries come out of registers by v_readlane (three
        // words per entry -- flags + extra-bits operand, base, bits taken -- so that no field is shifted out of another), positions are kept as a
        // vector (position + lane) and as the two scalars the checks need (bytes written, position in the line), every lane copies (the bytes a
        // match's copy leaves behind the match are overwritten by the next symbols before anything reads them: a ring position ahead of g is
        // 2048 - 63 behind at least, which no near source reaches), and the promised size is checked where a line is stored.  Registers:
        //   s36 ring mask  s[40:41] bit buffer  s42 bits in it  s43 next dword of the input block  s44 bytes written (g - g0)  s45 g mod 256 (+ len)
        //   s46 the member's size  s47 why  s48 len  s49 / s54 / s55 literal-length entry, base, bits  s50 / s53 / s52 distance entry, base -> distance, bits
        //   s51 scratch  s[56:57] the text buffer  s58 flushed  s59 g0  s[60:61] refill
        //   v40 the input block's dwords  v41 lane  v42 4 * lane  v43 the LDS address of the ring (and of the tables behind it)  v44-v49 scratch
        //   v50 float(lane)  v51-v53 / v54-v56 first-level tables  v57 g + lane  v45 / v58 the pending write's bytes / address
        uint32_t bb_lo = (uint32_t)D.bb, bb_hi = (uint32_t)(D.bb >> 32), a_bc = (uint32_t)D.bc, a_di = D.di, a_avail = D.g - D.g0, a_gl = D.g & 255, a_why = 0, a_len = 0, a_flushed = D.flushed;
        const uint32_t out_lo = (uint32_t)(uintptr_t)D.out, out_hi = (uint32_t)((uintptr_t)D.out >> 32);
        const uint32_t v_lane = threadIdx.x & 63, v_lane4 = 4 * (threadIdx.x & 63);
        const uint32_t v_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) FqzLds *)&S;
        const float v_lanef = (float)(threadIdx.x & 63);
        uint32_t v_pos = D.g + v_lane;
        __asm__ volatile(
            "s_mov_b32 s47, 0\n"
            "s_mov_b32 s48, 0\n"
            "v_and_or_b32 v58, v57, s36, v43\n"             // (the write that is always pending: at first a harmless one, ahead of g)
            "v_mov_b32_e32 v45, 0\n"
            ".Lfqz_top_%=:\n"
            "  s_cmp_gt_i32 s42, 32\n"
            "  s_cbranch_scc1 .Lfqz_have1_%=\n"
            "  s_cmp_eq_u32 s43, 64\n"
            "  s_cbranch_scc1 .Lfqz_exit_%=\n"
            "  s_mov_b32 s61, 0\n"
            "  v_readlane_b32 s60, v40, s43\n"
            "  s_add_i32 s43, s43, 1\n"
            "  s_lshl_b64 s[60:61], s[60:61], s42\n"
            "  s_or_b64 s[40:41], s[40:41], s[60:61]\n"
            "  s_add_i32 s42, s42, 32\n"
            ".Lfqz_have1_%=:\n"
            "  s_and_b32 s51, s40, 63\n"
            "  v_readlane_b32 s49, v51, s51\n"
            "  v_readlane_b32 s54, v52, s51\n"
            "  v_readlane_b32 s55, v53, s51\n"
            ".Lfqz_have_e_%=:\n"
            "  s_bitcmp1_b32 s49, 11\n"
            "  s_cbranch_scc0 .Lfqz_notmatch_%=\n"
            // ---- a match: the length, then its distance
            "  s_bfe_u32 s48, s40, s49\n"
            "  s_add_i32 s48, s48, s54\n"
            "  s_lshr_b64 s[40:41], s[40:41], s55\n"
            "  s_sub_i32 s42, s42, s55\n"
            "  s_cmp_gt_i32 s42, 32\n"
            "  s_cbranch_scc1 .Lfqz_have2_%=\n"
            "  s_cmp_eq_u32 s43, 64\n"
            "  s_cbranch_scc1 .Lfqz_exit_len_%=\n"
            "  s_mov_b32 s61, 0\n"
            "  v_readlane_b32 s60, v40, s43\n"
            "  s_add_i32 s43, s43, 1\n"
            "  s_lshl_b64 s[60:61], s[60:61], s42\n"
            "  s_or_b64 s[40:41], s[40:41], s[60:61]\n"
            "  s_add_i32 s42, s42, 32\n"
            ".Lfqz_have2_%=:\n"
            "  s_and_b32 s51, s40, 63\n"
            "  v_readlane_b32 s50, v54, s51\n"
            "  v_readlane_b32 s53, v55, s51\n"
            "  v_readlane_b32 s52, v56, s51\n"
            ".Lfqz_have_f_%=:\n"
            "  s_bitcmp1_b32 s50, 11\n"
            "  s_cbranch_scc0 .Lfqz_dist_other_%=\n"
            "  s_bfe_u32 s51, s40, s50\n"
            "  s_add_i32 s53, s53, s51\n"
            "  s_cmp_gt_u32 s53, s44\n"                     // a distance in front of the member's first byte
            "  s_cbranch_scc1 .Lfqz_exit_len_%=\n"
            "  s_lshr_b64 s[40:41], s[40:41], s52\n"
            "  s_sub_i32 s42, s42, s52\n"
            // ---- the copy: byte lane of the match = byte (lane mod dist) of the dist bytes in front of it
            "  s_cmp_lt_u32 s53, s48\n"
            "  s_cbranch_scc1 .Lfqz_period_%=\n"
            "  v_subrev_u32_e32 v44, s53, v57\n"
            ".Lfqz_src_%=:\n"
            // The copy is one symbol behind: the source bytes are asked for here and written when the NEXT symbol has been decoded (LDS works
            // a wavefront's instructions in order, so that symbol's own source -- which may be these bytes -- is read after them).
            "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "  ds_write_b8 v58, v45\n"
            "  v_and_or_b32 v58, v57, s36, v43\n"
            "  s_cmpk_gt_u32 s53, %[near_max]\n"
            "  s_cbranch_scc1 .Lfqz_far_%=\n"
            "  v_and_or_b32 v44, v44, s36, v43\n"
            "  ds_read_u8 v45, v44\n"
            ".Lfqz_copied_%=:\n"
            "  s_add_i32 s44, s44, s48\n"
            "  v_add_u32_e32 v57, s48, v57\n"
            "  s_add_i32 s45, s45, s48\n"
            "  s_cmp_lt_u32 s45, 0x100\n"
            "  s_cbranch_scc1 .Lfqz_top_%=\n"
            ".Lfqz_flush_%=:\n"                               // the line below the position is complete: one coalesced store -- unless more than the promised size has been written
            "  s_sub_i32 s45, s45, 0x100\n"
            "  s_cmp_gt_u32 s44, s46\n"
            "  s_cbranch_scc1 .Lfqz_exit_%=\n"
            "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "  ds_write_b8 v58, v45\n"                       // (the symbol that completed the line; written once more behind the next symbol, to no effect)
            "  s_add_i32 s51, s59, s44\n"
            "  s_and_b32 s51, s51, 0xffffff00\n"
            "  s_add_u32 s51, s51, 0xffffff00\n"
            "  s_and_b32 s52, s51, %[line_mask]\n"
            "  v_add_u32_e32 v44, s52, v42\n"
            "  v_add_u32_e32 v44, v44, v43\n"
            "  ds_read_b32 v47, v44\n"
            "  v_add_u32_e32 v46, s51, v42\n"
            "  s_add_u32 s58, s51, 0x100\n"
            "  s_waitcnt lgkmcnt(0)\n"
            "  global_store_dword v46, v47, s[56:57]\n"
            "  s_branch .Lfqz_top_%=\n"
            ".Lfqz_far_%=:\n"                                 // a source further back than the ring keeps: from the text in HBM (its lines were stored when they were completed)
            "  global_load_ubyte v45, v44, s[56:57]\n"
            "  s_branch .Lfqz_copied_%=\n"
            ".Lfqz_period_%=:\n"                              // dist < len: lane mod dist (lane, dist < 64: the quotient is exact to one either way)
            "  v_cvt_f32_u32_e32 v47, s53\n"
            "  v_rcp_iflag_f32_e32 v47, v47\n"
            "  s_nop 1\n"
            "  v_mul_f32_e32 v47, v47, v50\n"
            "  v_cvt_u32_f32_e32 v47, v47\n"
            "  v_mul_lo_u32 v47, v47, s53\n"
            "  v_sub_u32_e32 v47, v41, v47\n"
            "  v_ashrrev_i32_e32 v48, 31, v47\n"
            "  v_and_b32_e32 v48, s53, v48\n"
            "  v_add_u32_e32 v47, v47, v48\n"
            "  v_mov_b32_e32 v48, s53\n"
            "  v_cmp_le_u32_e32 vcc, s53, v47\n"
            "  v_cndmask_b32_e32 v48, 0, v48, vcc\n"
            "  v_sub_u32_e32 v47, v47, v48\n"
            "  s_add_i32 s51, s59, s44\n"
            "  s_sub_i32 s51, s51, s53\n"
            "  v_add_u32_e32 v44, s51, v47\n"
            "  s_branch .Lfqz_src_%=\n"
            ".Lfqz_notmatch_%=:\n"
            "  s_bitcmp1_b32 s49, 10\n"
            "  s_cbranch_scc0 .Lfqz_lt_other_%=\n"
            "  s_lshr_b64 s[40:41], s[40:41], s55\n"       // a literal
            "  s_sub_i32 s42, s42, s55\n"
            "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "  ds_write_b8 v58, v45\n"
            "  v_and_or_b32 v58, v57, s36, v43\n"
            "  v_mov_b32_e32 v45, s54\n"
            "  s_add_i32 s44, s44, 1\n"
            "  v_add_u32_e32 v57, 1, v57\n"
            "  s_add_i32 s45, s45, 1\n"
            "  s_cmp_lt_u32 s45, 0x100\n"
            "  s_cbranch_scc1 .Lfqz_top_%=\n"
            "  s_branch .Lfqz_flush_%=\n"
            ".Lfqz_lt_other_%=:\n"
            "  s_cmp_eq_u32 s49, %[miss]\n"
            "  s_cbranch_scc0 .Lfqz_exit_%=\n"                // the end of the block, a code longer than the root table, a long match
            "  s_and_b32 s51, s40, %[lt_mask]\n"             // not in the first-level table: the root table in LDS
            "  v_lshl_add_u32 v44, s51, 2, v43\n"
            "  ds_read_b32 v44, v44 offset:%[lt_off]\n"
            "  s_waitcnt lgkmcnt(0)\n"
            "  v_readfirstlane_b32 s49, v44\n"
            "  s_lshr_b32 s54, s49, 23\n"
            "  s_bfe_u32 s55, s49, 0x50005\n"
            "  s_branch .Lfqz_have_e_%=\n"
            ".Lfqz_dist_other_%=:\n"
            "  s_cmp_eq_u32 s50, %[miss]\n"
            "  s_cbranch_scc0 .Lfqz_exit_len_%=\n"
            "  s_and_b32 s51, s40, %[dt_mask]\n"
            "  v_lshl_add_u32 v44, s51, 2, v43\n"
            "  ds_read_b32 v44, v44 offset:%[dt_off]\n"
            "  s_waitcnt lgkmcnt(0)\n"
            "  v_readfirstlane_b32 s50, v44\n"
            "  s_bfe_u32 s53, s50, 0x2000d\n"
            "  s_bfe_u32 s52, s50, 0x40010\n"
            "  s_lshl_b32 s53, s53, s52\n"
            "  s_add_i32 s53, s53, 1\n"
            "  s_bfe_u32 s52, s50, 0x50005\n"
            "  s_branch .Lfqz_have_f_%=\n"
            ".Lfqz_exit_len_%=:\n"
            "  s_mov_b32 s47, 1\n"
            ".Lfqz_exit_%=:\n"
            "  s_waitcnt vmcnt(0) lgkmcnt(0)\n"
            "  ds_write_b8 v58, v45\n"
            "  s_waitcnt lgkmcnt(0)\n"
            : "+{s40}"(bb_lo), "+{s41}"(bb_hi), "+{s42}"(a_bc), "+{s43}"(a_di), "+{s44}"(a_avail), "+{s45}"(a_gl), "+{s47}"(a_why), "+{s48}"(a_len), "+{s58}"(a_flushed), "+{v57}"(v_pos)
            : "{s36}"((uint32_t)(FQZ_RING - 1)), "{s46}"(D.g_end - D.g0), "{s59}"(D.g0), "{s56}"(out_lo), "{s57}"(out_hi), "{v40}"(D.wcur), "{v41}"(v_lane), "{v42}"(v_lane4), "{v43}"(v_lds), "{v50}"(v_lanef),
              "{v51}"(D.l1e), "{v52}"(D.l1b), "{v53}"(D.l1t), "{v54}"(D.d1e), "{v55}"(D.d1b), "{v56}"(D.d1t),
              [lt_off] "i"(offsetof(FqzLds, lt)), [dt_off] "i"(offsetof(FqzLds, dt)), [miss] "i"(FQZ_L1_MISS),
              [lt_mask] "i"((1 << FQZ_LROOT) - 1), [dt_mask] "i"((1 << FQZ_DROOT) - 1), [near_max] "i"(FQZ_RING - 2 * 64), [line_mask] "i"((FQZ_RING - 1) & ~255)
            : "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s60", "s61", "v44", "v45", "v46", "v47", "v48", "v49", "v58", "vcc", "scc", "memory");
        D.bb = (uint64_t)bb_hi << 32 | bb_lo; D.bc = (int)a_bc; D.di = a_di; D.g = D.g0 + a_avail; D.flushed = a_flushed;
        why = (int)a_why; len = a_len;
      }
#else
      if (D.flushed > D.g0 || (D.g0 & 255) == 0) {           // (the member's first line of text, which may begin inside a line, is out)
        uint64_t bb = D.bb;
        int bc = D.bc;
        uint32_t di = D.di, g = D.g;
        const uint32_t g0 = D.g0, g_end = D.g_end;
        uint8_t *const ring = D.ring, *const out = D.out;
        for (;;) {
          if (bc <= 32) {
            if (di == 64) break;
            bb |= (uint64_t)FQF_UNIFORM32(FQF_RL(D.wcur, di)) << bc; bc += 32; ++di;
          }
          uint32_t e = FQF_RL(D.l1e, bb & 63);               // the first-level table, then the root table
          if (e == FQZ_L1_MISS) e = FQF_UNIFORM32(S.lt[bb & ((1u << FQZ_LROOT) - 1)]);
          FQZ_STAT(8 + ((e & 31) <= 6 ? 0 : (e & 31) <= 7 ? 1 : 2), 1);
          uint32_t g1;
          if (e & FQZ_K_BASE) {
            len = (e >> 23) + fqz_extra_of(bb, e);
            const uint32_t t1 = fqz_take_of(e);
            bb >>= t1; bc -= (int)t1;
            why = 1;
            if (bc <= 32) {
              if (di == 64) break;
              bb |= (uint64_t)FQF_UNIFORM32(FQF_RL(D.wcur, di)) << bc; bc += 32; ++di;
            }
            uint32_t f = FQF_RL(D.d1e, bb & 63);
            if (f == FQZ_L1_MISS) f = FQF_UNIFORM32(S.dt[bb & ((1u << FQZ_DROOT) - 1)]);
            FQZ_STAT(11 + ((f & 31) <= 6 ? 0 : (f & 31) <= 7 ? 1 : 2), 1);
            if (!(f & FQZ_K_BASE)) break;
            const uint32_t dist = fqz_dist_of(bb, f);
            if (dist > g - g0) break;
            g1 = g + len;
            const uint32_t t2 = fqz_take_of(f);
            bb >>= t2; bc -= (int)t2;
            why = 0;
            const uint32_t s = g - dist;
            FQZ_STAT(6, len); FQZ_STAT(dist < len ? 3 : dist <= FQZ_RING - 2 * 64 ? 1 : 2, 1);
            // (every lane copies: what lands behind the match is ahead of g, where nothing is read before the next symbols have written it)
            if (dist >= len) {
              if (dist <= FQZ_RING - 2 * 64) {
                FQF_LANES
                  const uint8_t b = ring[(s + (uint32_t)lane) & (FQZ_RING - 1)];
                  FQF_LV(D.l1scratch) = b;
                FQF_LANES_END
              } else {
                FQF_LANES
                  FQF_LV(D.l1scratch) = out[s + (uint32_t)lane];
                FQF_LANES_END
              }
            } else {
              FQF_LANES
                const uint8_t b = ring[(s + (uint32_t)lane % dist) & (FQZ_RING - 1)];   // (dist < len <= 64: inside the ring)
                FQF_LV(D.l1scratch) = b;
              FQF_LANES_END
            }
            FQF_LANES
              ring[(g + (uint32_t)lane) & (FQZ_RING - 1)] = (uint8_t)FQF_LV(D.l1scratch);
            FQF_LANES_END
            FQF_WAVE_FENCE();
          } else {
            if (!(e & FQZ_K_LIT)) break;                     // the end of the block, a longer code, a long match
            const uint32_t t1 = fqz_take_of(e);
            bb >>= t1; bc -= (int)t1;
            FQZ_STAT(0, 1);
            FQF_LANES
              ring[(g + (uint32_t)lane) & (FQZ_RING - 1)] = (uint8_t)(e >> 23);
            FQF_LANES_END
            FQF_WAVE_FENCE();
            g1 = g + 1;
          }
          const bool line = ((g ^ g1) >> 8) != 0;
          g = g1;
          if (line) {                                        // a line is complete: stored, unless more than the promised size has been written
            if (g > g_end) break;
            const uint32_t seg = (g & ~255u) - 256;
            FQF_LANES
              *(uint32_t *)(out + seg + 4 * (uint32_t)lane) = *(const uint32_t *)(ring + ((seg + 4 * (uint32_t)lane) & (FQZ_RING - 1)));
            FQF_LANES_END
            FQF_WAVE_FENCE();
            D.flushed = seg + 256;
          }
        }
        D.bb = bb; D.bc = bc; D.di = di; D.g = g;
      }
#endif
      FQZ_STAT(4, 1);
      const int r = why ? fqz_general_dist(D, S, len, dent) : fqz_general_symbol(D, S, lent, dent);
      if (r == 1) break;
      if (r == 2) { status = FQZ_REFUSED; break; }
    }
  }
  // the stream has ended: inside the payload, after exactly the promised output
  if (status == FQZ_OK && (D.overrun || fqz_bit_pos(D) > D.bits_end || D.g != D.g_end)) status = FQZ_REFUSED;
  if (status == FQZ_OK) {
    fqz_flush_rest(D);
    const uint32_t crc = fqz_crc_wave(A.out + M.out_off, M.out_len, A.crc, S.lt);
    if (crc != M.crc) status = FQZ_BADCRC;
  }
  return (uint32_t)status;
}

// =====================================================================================================================================
// Inflated text in HBM -> what the alignment call takes: the read filter's keys, lengths, the reference's read-slot history, names.
// kseq_read3_fpc (libbwa/kseq.h:327-371) as bwa_read_seq_with_hash_dev consumes it (src/BwtMapper.cpp:526-588), for text that is what
// sequencers write: four lines per record.  Every record is held to exactly the conditions under which kseq_read3_fpc returns the same
// tokens -- '@' first; the name up to the first white space; a base line of printable characters without '+', '>', '@'; a '+' line; a quality
// line as long as the base line -- the same conditions fq_fastq.cpp's record loop checks on the host; the first record of a file's text that
// fails them is reported (stat[FQT_FIRST_BAD]) and everything from there on goes the host's byte-wise way.
// =====================================================================================================================================
struct FqTextRec { uint32_t name_off, seq_off, qual_off; uint16_t len, name_len; };   // offsets into the file's text buffer
enum { FQT_FIRST_BAD = 0, FQT_MAX_NAME = 1, FQT_MIN_LEN = 2, FQT_MAX_LEN = 3, FQT_SHORTER_AFTER_LONGER = 4, FQT_MIN_NAME = 5, FQT_N_STAT = 8 };
FQ_HD bool fqt_is_space(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }
FQ_HD bool fqt_bad_base(uint32_t c) { return !(c > 32 && c < 127) || c == '+' || c == '>' || c == '@'; }   // what kseq would not read as one token of bases
FQ_HD uint32_t fqt_load32(const uint8_t *p) { uint32_t w; __builtin_memcpy(&w, p, 4); return w; }            // (any alignment: gfx950 reads unaligned dwords)

struct FqTokArgs {
  const uint8_t *text;       // one file's text (a whole number of records from its start on; readable 64 bytes behind n_text)
  const uint32_t *nl;        // positions of its line ends, ascending
  int32_t n_rec;             // records of this chunk (4 n_rec line ends are there)
  int32_t row0;              // the file's first row in the batch's arrays: end * n_pairs
  int32_t n_rows;            // rows of the batch (2 n_pairs; n for a single-end batch)
  int32_t max_len;           // a read longer than this is not taken (the rows the aligner sizes)
  uint32_t text0;            // where the first record begins (0..15: `text` is 16-byte aligned, the text proper need not be)
  FqTextRec *rec;            // [n_rows]
  uint64_t *head;            // [3][n_rows]: the filter's 32-mers (src/BwtIndexer.cpp:441-456), bases behind a short read as 'A' (fqt_slot_bases_thread puts the slot's there)
  uint16_t *hlen;            // [n_rows]
  uint32_t *stat;            // [FQT_N_STAT] of this file: first bad record (min), longest name (max), shortest / longest read
};
// one thread per record: the four lines, the name, the lengths.  What the record adds to the file's statistics comes back in st (the launcher folds
// a wavefront's into one atomic each: 4 M records' worth of same-address atomics were most of this kernel's time).
struct FqTokStat { uint32_t first_bad, max_name, min_name, max_len, min_len; };
FQ_HD FqTokStat fqt_stat_none() { FqTokStat t; t.first_bad = 0xffffffffu; t.max_name = 0; t.min_name = 0xffffffffu; t.max_len = 0; t.min_len = 0xffffffffu; return t; }
FQ_HD void fqt_stat_commit(const FqTokArgs &A, const FqTokStat &t) {
  if (t.first_bad != 0xffffffffu) FQF_ATOMIC_MIN32(&A.stat[FQT_FIRST_BAD], t.first_bad);
  if (t.min_len == 0xffffffffu) return;                      // (no good record)
  FQF_ATOMIC_MAX32(&A.stat[FQT_MAX_NAME], t.max_name);
  FQF_ATOMIC_MIN32(&A.stat[FQT_MIN_NAME], t.min_name);
  FQF_ATOMIC_MAX32(&A.stat[FQT_MAX_LEN], t.max_len);
  FQF_ATOMIC_MIN32(&A.stat[FQT_MIN_LEN], t.min_len);
}
FQ_HD FqTokStat fqt_rec_thread(const FqTokArgs &A, int i) {
  const uint8_t *T = A.text;
  const uint32_t l0 = i ? A.nl[4 * (size_t)i - 1] + 1 : A.text0;
  const uint32_t e0 = A.nl[4 * (size_t)i], e1 = A.nl[4 * (size_t)i + 1], e2 = A.nl[4 * (size_t)i + 2], e3 = A.nl[4 * (size_t)i + 3];
  const uint32_t l1 = e0 + 1, l2 = e1 + 1, l3 = e2 + 1;
  bool ok = l0 < e0 && T[l0] == '@' && l2 < e2 && T[l2] == '+' && (e1 - l1) == (e3 - l3) && (e1 - l1) <= (uint32_t)A.max_len;
  uint32_t ne = l0 + 1;
  if (ok) {                                                  // the name ends at the first white space (four bytes a step; the line end at e0 is one)
    for (;;) {
      const uint32_t w = fqt_load32(T + ne);
      uint32_t k = 0;
      while (k < 4 && !fqt_is_space((w >> (8 * k)) & 0xff)) ++k;
      ne += k;
      if (k < 4) break;
    }
    if (ne > e0) ne = e0;
  }
  uint32_t name_len = ok ? ne - (l0 + 1) : 0;
  if (name_len > 301) name_len = 301;                        // the reference's name buffer holds 2 * read_len = 302 bytes (libbwa/bwaseqio.c:233)
  const uint32_t L = ok ? e1 - l1 : 0;
  FqTextRec r;
  r.name_off = l0 + 1; r.seq_off = l1; r.qual_off = l3; r.len = (uint16_t)L; r.name_len = (uint16_t)name_len;
  A.rec[A.row0 + i] = r;
  A.hlen[A.row0 + i] = (uint16_t)L;
  FqTokStat t = fqt_stat_none();
  if (!ok) { t.first_bad = (uint32_t)i; return t; }
  t.max_name = t.min_name = name_len; t.max_len = t.min_len = L;
  return t;
}
// one thread per (record, piece of 32 bases): the base line's characters checked; pieces 0 .. 2 are the filter's three 32-mers
FQ_HD void fqt_piece_thread(const FqTokArgs &A, int64_t idx) {
  const int P = (A.max_len + 31) >> 5;
  const int i = (int)(idx / P), p = (int)(idx - (int64_t)i * P);
  const FqTextRec r = A.rec[A.row0 + i];
  const int L = r.len, p0 = 32 * p;
  if (p0 >= L && p >= 3) return;
  const uint8_t *s = A.text + r.seq_off;
  uint64_t k = 0;
  bool bad = false;
  for (int q = 0; q < 8; ++q) {
    const int at = p0 + 4 * q;
    uint32_t w = at < L ? fqt_load32(s + at) : 0;            // (reads up to three bytes behind the read: the line end and the '+' line are there)
    for (int j = 0; j < 4; ++j) {
      const uint32_t c = (w >> (8 * j)) & 0xff;
      const bool in = at + j < L;
      bad |= in && fqt_bad_base(c);
      k = (k << 2) | (in ? (uint64_t)fq_nt4_fast(c) : 0);    // (a non-ACGT code is OR-ed in unmasked and smears into the base before it, as in the reference)
    }
  }
  if (p < 3) A.head[(size_t)p * (size_t)A.n_rows + (size_t)(A.row0 + i)] = k;
  if (bad) FQF_ATOMIC_MIN32(&A.stat[FQT_FIRST_BAD], (uint32_t)i);
}

// ---- the reference's reused read slots (fq_fastq.cpp slot_apply) -----------------------------------------------------
// bwa_read_seq_with_hash_dev fills two sets of batch_pairs read slots in turn and never clears them: the filter sees, behind a read shorter
// than 96 bp, the bases earlier reads of its slot left (src/BwtIndexer.cpp:441-456 reads 96 bytes whatever the length), and a name keeps the
// tail of a longer earlier name of its slot (strncpy without terminator, src/BwtMapper.cpp:564).  The slots' state lives in HBM; a thread
// per slot walks the chunk's records of its slot in order (record g of the file has slot g mod 2 batch_pairs).
struct FqSlotArgs {
  const uint8_t *text;
  const FqTextRec *rec;      // the batch's records; this file's begin at row0
  int32_t n_rec, row0, n_rows;
  int64_t g0;                // records of the file before this chunk
  int32_t n_slots;           // 2 * batch_pairs
  int32_t mode;              // FQ_FASTQ_SLOTS_*: 0 reused (names and bases), 1 clean names (bases), 2 fresh (nothing lingers)
  uint8_t *slot_base;        // [n_slots][96], zero = never written
  uint16_t *slot_len;        // [n_slots] longest read the slot has held
  uint8_t *slot_name;        // [n_slots][304], NUL padded
  uint64_t *head;            // [3][n_rows]
  char *names;               // out: [n_rows][name_stride], NUL padded: the name each record prints under
  int32_t name_stride;
  uint32_t *stat;
  int32_t all_long;          // every read of the chunk has at least 96 bases: a slot keeps the bases of its last record only
  int32_t plain_names;       // every name of the stream so far has one length: a record prints under its own name, a slot keeps its last record's
};
FQ_HD void fqt_slot_bases_thread(const FqSlotArgs &A, int slot) {
  int64_t first = (int64_t)slot - A.g0 % A.n_slots;
  if (first < 0) first += A.n_slots;
  if (first >= A.n_rec) return;
  uint8_t *h = A.slot_base + (size_t)slot * 96;
  uint32_t longest = A.slot_len[slot];
  if (A.all_long) {
    // no read of the chunk is short: nothing reads the slot, and what it holds afterwards is its last record's first 96 bases
    int64_t lastrec = first;
    for (int64_t i = first; i < A.n_rec; i += A.n_slots) {
      const uint32_t n = A.rec[A.row0 + i].len;
      if (n < longest) A.stat[FQT_SHORTER_AFTER_LONGER] = 1; else longest = n;
      lastrec = i;
    }
    const uint8_t *s = A.text + A.rec[A.row0 + lastrec].seq_off;
    for (int q = 0; q < 24; ++q) { const uint32_t w = fqt_load32(s + 4 * q); __builtin_memcpy(h + 4 * q, &w, 4); }
    A.slot_len[slot] = (uint16_t)(longest > 65535 ? 65535 : longest);
    return;
  }
  for (int64_t i = first; i < A.n_rec; i += A.n_slots) {
    const FqTextRec r = A.rec[A.row0 + i];
    const uint32_t n = r.len;
    if (n < longest) A.stat[FQT_SHORTER_AFTER_LONGER] = 1; else longest = n;
    const uint8_t *s = A.text + r.seq_off;
    if (n < 96) {                                            // the slot's earlier bases stand behind the read: the filter's keys again, with them
      for (int ch = 0; ch < 3; ++ch) {
        uint64_t k = 0;
        for (int j = 0; j < 32; ++j) {
          const uint32_t p = 32 * (uint32_t)ch + (uint32_t)j;
          const uint32_t c = p < n ? s[p] : (h[p] ? h[p] : (uint32_t)'A');
          k = (k << 2) | (uint64_t)fq_nt4_fast(c);
        }
        A.head[(size_t)ch * (size_t)A.n_rows + (size_t)(A.row0 + i)] = k;
      }
    }
    const uint32_t m = n < 96 ? n : 96;
    uint32_t p = 0;                                          // the read's first 96 bases over the slot's, by words (h is aligned, the read is not)
    for (; p + 4 <= m; p += 4) { const uint32_t w = fqt_load32(s + p); __builtin_memcpy(h + p, &w, 4); }
    if (p < m) {
      uint32_t old_w;
      __builtin_memcpy(&old_w, h + p, 4);
      const uint32_t mask = (1u << (8 * (m - p))) - 1u;
      const uint32_t w = (old_w & ~mask) | (fqt_load32(s + p) & mask);
      __builtin_memcpy(h + p, &w, 4);
    }
  }
  A.slot_len[slot] = (uint16_t)(longest > 65535 ? 65535 : longest);
}
// names of one length so far (or slots that keep nothing): a thread per RECORD writes the name it prints under -- its own, without a mate suffix
FQ_HD void fqt_names_plain_thread(const FqSlotArgs &A, int i) {
  const FqTextRec r = A.rec[A.row0 + i];
  const uint8_t *nm = A.text + r.name_off;
  const uint32_t l = r.name_len;
  const bool mate_suffix = l > 2 && nm[l - 2] == '/' && (nm[l - 1] == '1' || nm[l - 1] == '2');   // src/BwtMapper.cpp:565-570
  char *o = A.names + (size_t)(A.row0 + i) * (size_t)A.name_stride;
  uint32_t keep = mate_suffix ? l - 2 : l;
  if (keep > (uint32_t)A.name_stride - 1) keep = (uint32_t)A.name_stride - 1;
  for (uint32_t p = 0; p < (uint32_t)A.name_stride; p += 4) {    // (name_stride is a multiple of 16; the output rows are aligned)
    uint32_t w = 0;
    for (uint32_t j = 0; j < 4; ++j) if (p + j < keep) w |= (uint32_t)nm[p + j] << (8 * j);
    __builtin_memcpy(o + p, &w, 4);
  }
}
FQ_HD void fqt_slot_names_thread(const FqSlotArgs &A, int slot) {
  int64_t first = (int64_t)slot - A.g0 % A.n_slots;
  if (first < 0) first += A.n_slots;
  uint8_t *b = A.slot_name + (size_t)slot * 304;
  if (A.plain_names) {
    // (the records' names were written by fqt_names_plain_thread) the slot keeps its last record's name, as the general walk would leave it
    if (A.mode != 0 || first >= A.n_rec) return;
    int64_t lastrec = first;
    for (int64_t i = first; i < A.n_rec; i += A.n_slots) lastrec = i;
    const FqTextRec r = A.rec[A.row0 + lastrec];
    const uint8_t *nm = A.text + r.name_off;
    const uint32_t l = r.name_len;
    for (uint32_t p = 0; p < l; ++p) b[p] = nm[p];
    if (l > 2 && nm[l - 2] == '/' && (nm[l - 1] == '1' || nm[l - 1] == '2')) b[l - 2] = 0;
    return;
  }
  for (int64_t i = first; i < A.n_rec; i += A.n_slots) {
    const FqTextRec r = A.rec[A.row0 + i];
    const uint8_t *nm = A.text + r.name_off;
    const uint32_t l = r.name_len;
    const bool mate_suffix = l > 2 && nm[l - 2] == '/' && (nm[l - 1] == '1' || nm[l - 1] == '2');   // src/BwtMapper.cpp:565-570
    char *o = A.names + (size_t)(A.row0 + i) * (size_t)A.name_stride;
    uint32_t keep;
    if (A.mode != 0) {                                       // fresh buffers / clean names: the record's own name
      keep = mate_suffix ? l - 2 : l;
      if (keep > (uint32_t)A.name_stride - 1) keep = (uint32_t)A.name_stride - 1;
      for (uint32_t p = 0; p < keep; ++p) o[p] = (char)nm[p];
    } else {
      // the name over the slot's buffer (strncpy without terminator), the printed name = the buffer as a C string: by 4-byte words -- the buffer
      // and the output row are aligned, the name is read at any alignment (byte by byte this loop was 5 ms per million pairs of Illumina-style names)
      uint32_t p = 0, pl = 0xffffffffu;                       // pl: the first zero byte of the buffer
      for (; p + 4 <= l; p += 4) {
        const uint32_t w = fqt_load32(nm + p);
        __builtin_memcpy(b + p, &w, 4);
        if (pl == 0xffffffffu && ((w - 0x01010101u) & ~w & 0x80808080u)) for (uint32_t j = 0; j < 4; ++j) if (((w >> (8 * j)) & 0xff) == 0) { pl = p + j; break; }
      }
      if (p < l) {
        uint32_t old_w;
        __builtin_memcpy(&old_w, b + p, 4);
        const uint32_t mask = (1u << (8 * (l - p))) - 1u;
        const uint32_t w = (old_w & ~mask) | (fqt_load32(nm + p) & mask);    // (reads up to three bytes behind the name: the rest of its line is there)
        __builtin_memcpy(b + p, &w, 4);
      }
      if (mate_suffix) { b[l - 2] = 0; if (pl > l - 2) pl = l - 2; }
      for (uint32_t q = l & ~3u; pl == 0xffffffffu && q < 304; q += 4) {     // the buffer behind the name (and the name's last, partial word)
        uint32_t w;
        __builtin_memcpy(&w, b + q, 4);
        if ((w - 0x01010101u) & ~w & 0x80808080u) for (uint32_t j = 0; j < 4; ++j) if (((w >> (8 * j)) & 0xff) == 0 && q + j >= (l & ~3u)) { pl = q + j; break; }
      }
      if (pl > 303) pl = 303;
      keep = pl < (uint32_t)A.name_stride - 1 ? pl : (uint32_t)A.name_stride - 1;
      for (uint32_t q = 0; q < (uint32_t)A.name_stride; q += 4) {            // (name_stride is a multiple of 16: whole words, zero behind the name)
        uint32_t w = 0;
        if (q < keep) { __builtin_memcpy(&w, b + q, 4); if (q + 4 > keep) w &= (1u << (8 * (keep - q))) - 1u; }
        __builtin_memcpy(o + q, &w, 4);
      }
      continue;
    }
    for (uint32_t p = keep; p < (uint32_t)A.name_stride; ++p) o[p] = 0;
  }
}

// ---- the reads of surviving pairs: rows gathered from the text ------------------------------------------------------------------------
// Compact row t = 2 * survivor pair + end.  The bases go out as the packed path leaves them (fq_unpack_piece + fq_patch_thread): "ACGT" by
// code, 'N' for any other character, '-' kept, zero behind the read -- so every later kernel sees the same rows whichever way a batch came.
struct FqTextGatherArgs {
  const uint8_t *text[2];
  const FqTextRec *rec;      // [n_rows] batch rows: end * n_pairs + pair
  const char *names;         // [n_rows][name_stride]
  int32_t name_stride, n_pairs, single_end;
  const int32_t *pair_list;  // survivor pair -> pair of the batch
  int32_t n_out;             // compact rows (2 * survivors)
  uint8_t *seq, *qual;       // out: [n_out][stride]
  int32_t stride;            // a multiple of 16
  int32_t *len_out, *len_trim;
  char *names_out;           // out: [n_out][name_stride]
};
FQ_HD void fqt_gather_piece(const FqTextGatherArgs &A, int64_t g) {
  const int per_row = A.stride >> 4;
  const int t = (int)(g / per_row), c = (int)(g - (int64_t)t * per_row);
  const int e = t & 1, p0 = 16 * c;
  uint8_t *so = A.seq + (size_t)t * (size_t)A.stride + (size_t)p0, *qo = A.qual + (size_t)t * (size_t)A.stride + (size_t)p0;
  if (A.single_end && e) {                                   // the absent mate of a single-end read: an empty row
    for (int j = 0; j < 16; ++j) { so[j] = 0; qo[j] = 0; }
    if (c == 0) { A.len_out[t] = 0; A.len_trim[t] = 0; for (int j = 0; j < A.name_stride; ++j) A.names_out[(size_t)t * A.name_stride + j] = 0; }
    return;
  }
  const size_t row = (size_t)e * (size_t)A.n_pairs + (size_t)A.pair_list[t >> 1];
  const FqTextRec r = A.rec[row];
  const uint8_t *T = A.text[e];
  const int L = r.len;
  for (int j = 0; j < 16; ++j) {
    const int p = p0 + j;
    uint8_t b = 0, q = 0;
    if (p < L) {
      const uint32_t code = fq_nt4_fast(T[r.seq_off + p]);
      b = code < 4 ? (uint8_t)((0x54474341u >> (8 * code)) & 0xffu) : code == 5 ? (uint8_t)'-' : (uint8_t)'N';
      q = T[r.qual_off + p];
    }
    so[j] = b; qo[j] = q;
  }
  if (c == 0) {
    A.len_out[t] = L; A.len_trim[t] = L;
    const char *nm = A.names + row * (size_t)A.name_stride;
    char *no = A.names_out + (size_t)t * (size_t)A.name_stride;
    for (int j = 0; j < A.name_stride; ++j) no[j] = nm[j];
  }
}
// bwa_trim_read (libbwa/bwaseqio.c:75-88) over every read of the batch, the qualities read where they lie in the text: len_trim[row], and the
// longest trimmed read per reference batch (infer_isize's max_len counts filtered reads too, libbwa/bwape.c:60-61)
struct FqTextTrimArgs {
  FqKOpts o;
  const uint8_t *text[2];
  const FqTextRec *rec;
  int32_t n_rows, n_pairs, batch_pairs;
  int32_t *len_trim;         // out [n_rows]
  int32_t *sub_max;          // out (zeroed by the caller)
};
FQ_HD void fqt_trim_all_thread(const FqTextTrimArgs &A, int r) {
  const int e = r >= A.n_pairs ? 1 : 0;
  const FqTextRec R = A.rec[r];
  const int lt = fq_trim_len(A.o, A.text[e] + R.qual_off, R.len);
  A.len_trim[r] = lt;
  FQ_ATOMIC_MAX32(&A.sub_max[(r >= A.n_pairs ? r - A.n_pairs : r) / A.batch_pairs], lt);
}
