// fq_inflate.h -- raw DEFLATE (RFC 1951) decoder and CRC-32 for the FASTQ front end's BGZF members.
//
// The reference reads its FASTQ files through zlib's gzread (libbwa/bwaseqio.c:41-52, kseq.h:327-371); a member-parallel reader spends
// most of its time in inflate() and crc32().  This is a from-scratch decoder for the case the front end has -- the whole member in
// memory, the output size known from the member's trailer -- built the way fast software decoders are: a 64-bit bit buffer refilled
// without branches, one table look-up per literal/length symbol (11 bits, second-level tables behind longer codes), one per distance
// (8 bits), match copies in 8-byte words.  It decodes exactly what inflate() decodes or fails; the caller checks CRC-32 and ISIZE as
// gzread does, and hands any member this decoder refuses to zlib, so what is accepted and what is reported as corrupt does not change.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define FQZ_HAVE_CLMUL 1
#endif

namespace fqz {

// ---- CRC-32 (IEEE 802.3, the gzip trailer's), sixteen bytes per step -------------------------------------------------------------
struct CrcTables {
  uint32_t t[16][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int s = 1; s < 16; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
  }
};
inline const CrcTables &crc_tables() { static const CrcTables T; return T; }
#if defined(FQZ_HAVE_CLMUL)
// Carry-less multiplication folds 64 bytes per step: the method of V. Gopal, E. Ozturk et al., "Fast CRC Computation for Generic Polynomials
// Using PCLMULQDQ Instruction" (Intel white paper 323102, 2009).  The folding / reduction constants for the reflected polynomial 0xEDB88320
// (k1..k5 = x^(4*128+32), x^(4*128-32), x^(128+32), x^(128-32), x^64 mod P, bit-reflected; Barrett mu and P) and the four-accumulator
// schedule are those of the widely published routine that follows the paper -- zlib's x86 forks (Intel's crc_folding.c of 2013, Chromium's
// crc32_simd.c, zlib-ng's crc32_pclmulqdq) all carry the same values, as any implementation of the paper for this polynomial must.
// State in, state out, len >= 64 and a multiple of 16.  Chosen at run time where the CPU has it; tests/test_inflate.py holds both
// paths to zlib's crc32.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_clmul(const uint8_t *buf, size_t len, uint32_t state) {
  const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596LL, 0x0154442bd4LL), k3k4 = _mm_set_epi64x(0x00ccaa009eLL, 0x01751997d0LL);
  const __m128i k5k0 = _mm_set_epi64x(0, 0x0163cd6124LL), poly = _mm_set_epi64x(0x01f7011641LL, 0x01db710641LL);
  __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0)), x2 = _mm_loadu_si128((const __m128i *)(buf + 16));
  __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 32)), x4 = _mm_loadu_si128((const __m128i *)(buf + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
  __m128i x0 = k1k2;
  buf += 64; len -= 64;
  while (len >= 64) {
    const __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00), x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
    const __m128i x7 = _mm_clmulepi64_si128(x3, x0, 0x00), x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
    x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i *)(buf + 0)));
    x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i *)(buf + 16)));
    x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i *)(buf + 32)));
    x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i *)(buf + 48)));
    buf += 64; len -= 64;
  }
  x0 = k3k4;   // four lanes -> one
  __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
  x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
  while (len >= 16) {
    x2 = _mm_loadu_si128((const __m128i *)buf);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    buf += 16; len -= 16;
  }
  // 128 -> 64 bits, then Barrett reduction to 32
  x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
  const __m128i m32 = _mm_setr_epi32(~0, 0, ~0, 0);
  x1 = _mm_srli_si128(x1, 8); x1 = _mm_xor_si128(x1, x2);
  x0 = k5k0;
  x2 = _mm_srli_si128(x1, 4); x1 = _mm_and_si128(x1, m32);
  x1 = _mm_clmulepi64_si128(x1, x0, 0x00); x1 = _mm_xor_si128(x1, x2);
  x0 = poly;
  x2 = _mm_and_si128(x1, m32); x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
  x2 = _mm_and_si128(x2, m32); x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
  x1 = _mm_xor_si128(x1, x2);
  return (uint32_t)_mm_extract_epi32(x1, 1);
}
inline bool have_clmul() { static const bool h = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1"); return h; }
#endif
inline uint32_t crc32(const uint8_t *p, size_t n, uint32_t crc = 0, bool tables_only = false) {
  const CrcTables &T = crc_tables();
#if defined(FQZ_HAVE_CLMUL)
  if (n >= 64 && !tables_only && have_clmul()) {
    const size_t body = n & ~(size_t)15;
    const uint32_t st = crc32_clmul(p, body, ~crc);
    crc = ~st; p += body; n -= body;
    if (!n) return crc;
  }
#endif
  uint32_t c = ~crc;
  while (n && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8); --n; }
  while (n >= 16) {
    uint64_t a, b;
    memcpy(&a, p, 8); memcpy(&b, p + 8, 8);
    a ^= c;
    c = T.t[15][a & 0xff] ^ T.t[14][(a >> 8) & 0xff] ^ T.t[13][(a >> 16) & 0xff] ^ T.t[12][(a >> 24) & 0xff] ^
        T.t[11][(a >> 32) & 0xff] ^ T.t[10][(a >> 40) & 0xff] ^ T.t[9][(a >> 48) & 0xff] ^ T.t[8][a >> 56] ^
        T.t[7][b & 0xff] ^ T.t[6][(b >> 8) & 0xff] ^ T.t[5][(b >> 16) & 0xff] ^ T.t[4][(b >> 24) & 0xff] ^
        T.t[3][(b >> 32) & 0xff] ^ T.t[2][(b >> 40) & 0xff] ^ T.t[1][(b >> 48) & 0xff] ^ T.t[0][b >> 56];
    p += 16; n -= 16;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return ~c;
}

// ---- decode tables -----------------------------------------------------------------------------------------------------------------
// entry: bits 0-7 the bits to consume -- code length (second-level entries: the length beyond the first level's bits) plus, for lengths and
// distances, their extra bits, so that a symbol costs the bit buffer one shift; bits 8-12 the code length alone (lengths, distances: the extra
// bits are read out of a copy of the buffer) or the second level's index bits (pointers); bits 13-15 kind; bits 16-31 literal / base value /
// second-level offset
// (literal entries: bit 8 set = two literals, the first in bits 16-23, the second in bits 24-31, the length that of both codes)
enum : uint32_t { K_LIT2 = 1u << 8, K_LIT = 1u << 13, K_BASE = 1u << 14, K_EOB = 1u << 15, K_SUB = 3u << 14, K_MASK = 7u << 13 };   // (K_LIT is a bit of literals only)
constexpr int LIT_BITS = 11, DIST_BITS = 8, MAX_CODE = 15;
constexpr int LIT_TABLE = (1 << LIT_BITS) + 1024, DIST_TABLE = (1 << DIST_BITS) + 512;   // room for the second-level tables (enough: checked while building)

struct Inflater {
  uint32_t lit[LIT_TABLE], dist[DIST_TABLE];
  uint8_t lens[288 + 32];
  bool fixed_built = false;
  uint32_t fixed_lit[LIT_TABLE], fixed_dist[DIST_TABLE];
};

inline uint32_t rev_bits(uint32_t v, int n) {
  uint32_t r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
  return r;
}

// canonical Huffman code of `n` symbols with lengths `len` -> table (first level `tb` bits).  `sym_entry(s)` = the entry of symbol s
// without its code length.  False: over-subscribed or incomplete code (zlib accepts an incomplete distance code of one symbol, and an
// incomplete literal/length code never; a lone distance symbol is filled in, its unused codeword left invalid).
template <class F>
inline bool build_table(const uint8_t *len, int n, int tb, uint32_t *table, int cap, bool allow_incomplete_single, F sym_entry) {
  int count[MAX_CODE + 1] = {0};
  for (int s = 0; s < n; ++s) ++count[len[s]];
  if (count[0] == n) {   // no symbol at all: every look-up is invalid (a block without distance codes may still be all literals)
    for (int i = 0; i < (1 << tb); ++i) table[i] = 0;
    return allow_incomplete_single;
  }
  int left = 1;
  for (int l = 1; l <= MAX_CODE; ++l) { left = (left << 1) - count[l]; if (left < 0) return false; }
  if (left > 0 && !(allow_incomplete_single && n - count[0] == 1 && count[1] == 1)) return false;
  uint32_t next[MAX_CODE + 2];
  next[1] = 0;
  for (int l = 1; l <= MAX_CODE; ++l) next[l + 1] = (next[l] + (uint32_t)count[l]) << 1;
  for (int i = 0; i < (1 << tb); ++i) table[i] = 0;
  // second-level tables: one per first-level prefix of the codes longer than tb bits, sized for the longest code under that prefix
  int used = 1 << tb;
  // (pass 1: the longest code under each long prefix)
  uint8_t sub_bits[1 << LIT_BITS];
  bool any_long = false;
  for (int l = tb + 1; l <= MAX_CODE; ++l) if (count[l]) any_long = true;
  if (any_long) {
    memset(sub_bits, 0, (size_t)1 << tb);
    uint32_t nx[MAX_CODE + 2];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
      const int l = len[s];
      if (l == 0) continue;
      const uint32_t code = nx[l]++;
      if (l > tb) { const uint32_t pre = rev_bits(code >> (l - tb), tb); if (l - tb > sub_bits[pre]) sub_bits[pre] = (uint8_t)(l - tb); }
    }
    for (int pre = 0; pre < (1 << tb); ++pre) {
      if (!sub_bits[pre]) continue;
      if (used + (1 << sub_bits[pre]) > cap) return false;
      table[pre] = K_SUB | (uint32_t)used << 16 | (uint32_t)sub_bits[pre] << 8 | (uint32_t)tb;
      for (int i = 0; i < (1 << sub_bits[pre]); ++i) table[used + i] = 0;
      used += 1 << sub_bits[pre];
    }
  }
  for (int s = 0; s < n; ++s) {
    const int l = len[s];
    if (l == 0) continue;
    const uint32_t code = next[l]++;
    const uint32_t e = sym_entry(s);
    if (l <= tb) {
      const uint32_t r = rev_bits(code, l);
      const uint32_t ent = (e & K_MASK) == K_BASE ? ((e & ~(31u << 8)) | (uint32_t)l << 8 | ((uint32_t)l + ((e >> 8) & 31))) : (e | (uint32_t)l);
      for (uint32_t i = r; i < (1u << tb); i += 1u << l) table[i] = ent;
    } else {
      const uint32_t pre = rev_bits(code >> (l - tb), tb);
      const uint32_t p = table[pre];
      const int sb = (int)(p >> 8) & 31;
      const uint32_t off = p >> 16;
      const uint32_t r = rev_bits(code & ((1u << (l - tb)) - 1), l - tb);
      const uint32_t l2 = (uint32_t)(l - tb);
      const uint32_t ent = (e & K_MASK) == K_BASE ? ((e & ~(31u << 8)) | l2 << 8 | (l2 + ((e >> 8) & 31))) : (e | l2);
      for (uint32_t i = r; i < (1u << sb); i += 1u << (l - tb)) table[off + i] = ent;
    }
  }
  return true;
}

// Two literals per look-up where both codes fit the first level's bits (sequence lines that the compressor left as literals cost two to
// five bits a base; zlib level 1 leaves few -- there the text is matches of 4.6 bytes on average -- higher levels and other writers more).
inline void pair_literals(uint32_t *table, int tb) {
  static thread_local uint32_t base[1 << LIT_BITS];
  memcpy(base, table, sizeof(uint32_t) << tb);
  for (uint32_t i = 0; i < (1u << tb); ++i) {
    const uint32_t e = base[i];
    if (!(e & K_LIT)) continue;
    const int l1 = (int)(e & 0xff);
    if (l1 >= tb) continue;
    const uint32_t f = base[i >> l1];
    if (!(f & K_LIT) || l1 + (int)(f & 0xff) > tb) continue;
    table[i] = K_LIT | K_LIT2 | ((e >> 16) & 0xff) << 16 | ((f >> 16) & 0xff) << 24 | (uint32_t)(l1 + (int)(f & 0xff));
  }
}

inline uint32_t lit_entry(int s) {
  static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  if (s < 256) return K_LIT | (uint32_t)s << 16;
  if (s == 256) return K_EOB;
  if (s > 285) return 0;   // 286, 287: in the fixed code, never valid in data
  return K_BASE | (uint32_t)base[s - 257] << 16 | (uint32_t)extra[s - 257] << 8;
}
inline uint32_t dist_entry(int s) {
  static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  if (s > 29) return 0;
  return K_BASE | (uint32_t)base[s] << 16 | (uint32_t)extra[s] << 8;
}

// ---- the decoder ---------------------------------------------------------------------------------------------------------------------
// One member being decoded.  (Two members decoded side by side by one thread -- two independent bit-buffer chains -- were measured: + 5 %;
// the loop is bound by its ~60 instructions per match, not by the latency of its look-ups.)
struct Dec {
  const uint8_t *in, *in_end;
  uint8_t *out, *out_end, *dst;
  uint64_t bb;      // bit buffer, next bit at bit 0
  int bc;           // valid bits in bb
  size_t over;      // zero bytes supplied beyond in_end (a stream that needs them is truncated: caught when it ends)
  const uint32_t *LT, *DT;
  bool last;
  Inflater *Z;
  void begin(Inflater &z, const uint8_t *src, size_t n, uint8_t *d, size_t out_len) {
    in = src; in_end = src + n; out = d; out_end = d + out_len; dst = d; bb = 0; bc = 0; over = 0; LT = DT = nullptr; last = false; Z = &z;
  }
};
#define FQZ_TAKE(D, nb) ((D).bb >>= (nb), (D).bc -= (nb))
inline void dec_refill(Dec &d) {
  if (d.in_end - d.in >= 8) {
    uint64_t w;
    memcpy(&w, d.in, 8);
    d.bb |= w << d.bc;
    d.in += (63 - d.bc) >> 3;
    d.bc |= 56;
  } else {
    while (d.bc <= 56) {
      if (d.in < d.in_end) d.bb |= (uint64_t)*d.in++ << d.bc; else ++d.over;
      d.bc += 8;
    }
  }
}
// Block headers up to the next Huffman block (stored blocks are copied here).  0: a Huffman block begins (LT / DT set); 1: the stream has
// ended; -1: not a stream this decoder accepts.
inline int dec_next_block(Dec &d, bool one_stored = false) {   // one_stored: return 2 after ONE stored block (a caller that looks after the room on both sides between blocks)
  Inflater &Z = *d.Z;
  for (;;) {
    if (d.last) return 1;
    dec_refill(d);
    d.last = (d.bb & 1) != 0;
    const int type = (int)(d.bb >> 1) & 3;
    FQZ_TAKE(d, 3);
    if (type == 0) {   // stored: skip to the byte boundary, LEN, NLEN, bytes
      FQZ_TAKE(d, d.bc & 7);
      const size_t back = (size_t)d.bc >> 3;     // the whole bytes of the bit buffer go back to the input
      if (d.over > back) return -1;
      d.in -= back - d.over; d.over = 0; d.bb = 0; d.bc = 0;
      if (d.in_end - d.in < 4) return -1;
      const uint32_t len = d.in[0] | (uint32_t)d.in[1] << 8, nlen = d.in[2] | (uint32_t)d.in[3] << 8;
      d.in += 4;
      if ((len ^ nlen) != 0xffffu || (size_t)(d.in_end - d.in) < len || (size_t)(d.out_end - d.out) < len) return -1;
      memcpy(d.out, d.in, len);
      d.in += len; d.out += len;
      if (one_stored) return 2;
      continue;
    }
    if (type == 1) {
      if (!Z.fixed_built) {
        uint8_t l[288 + 32];
        for (int s = 0; s < 144; ++s) l[s] = 8;
        for (int s = 144; s < 256; ++s) l[s] = 9;
        for (int s = 256; s < 280; ++s) l[s] = 7;
        for (int s = 280; s < 288; ++s) l[s] = 8;
        for (int s = 0; s < 32; ++s) l[288 + s] = 5;
        if (!build_table(l, 288, LIT_BITS, Z.fixed_lit, LIT_TABLE, false, lit_entry)) return -1;
        pair_literals(Z.fixed_lit, LIT_BITS);
        if (!build_table(l + 288, 32, DIST_BITS, Z.fixed_dist, DIST_TABLE, false, dist_entry)) return -1;
        Z.fixed_built = true;
      }
      d.LT = Z.fixed_lit; d.DT = Z.fixed_dist;
      return 0;
    }
    if (type != 2) return -1;
    const int hlit = (int)(d.bb & 31) + 257, hdist = (int)((d.bb >> 5) & 31) + 1, hclen = (int)((d.bb >> 10) & 15) + 4;
    FQZ_TAKE(d, 14);
    if (hlit > 286 || hdist > 30) return -1;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
      if (d.bc < 3) dec_refill(d);
      cl[order[i]] = (uint8_t)(d.bb & 7);
      FQZ_TAKE(d, 3);
    }
    uint32_t ct[1 << 7];
    if (!build_table(cl, 19, 7, ct, 1 << 7, false, [](int s) { return (uint32_t)s << 16 | K_LIT; })) return -1;
    int i = 0;
    while (i < hlit + hdist) {
      dec_refill(d);
      const uint32_t e = ct[d.bb & 127];
      if (!(e & K_MASK)) return -1;
      FQZ_TAKE(d, (int)(e & 0xff));
      const int s = (int)(e >> 16);
      if (s < 16) { Z.lens[i++] = (uint8_t)s; continue; }
      int rep, val = 0;
      if (s == 16) { if (i == 0) return -1; val = Z.lens[i - 1]; rep = 3 + (int)(d.bb & 3); FQZ_TAKE(d, 2); }
      else if (s == 17) { rep = 3 + (int)(d.bb & 7); FQZ_TAKE(d, 3); }
      else { rep = 11 + (int)(d.bb & 127); FQZ_TAKE(d, 7); }
      if (i + rep > hlit + hdist) return -1;
      while (rep--) Z.lens[i++] = (uint8_t)val;
    }
    if (Z.lens[256] == 0) return -1;   // no end-of-block code
    if (!build_table(Z.lens, hlit, LIT_BITS, Z.lit, LIT_TABLE, false, lit_entry)) return -1;
    pair_literals(Z.lit, LIT_BITS);
    if (!build_table(Z.lens + hlit, hdist, DIST_BITS, Z.dist, DIST_TABLE, true, dist_entry)) return -1;
    d.LT = Z.lit; d.DT = Z.dist;
    return 0;
  }
}
// The fast loop of a Huffman block: runs of literals and matches while there is room for the longest match plus a run of literals on the
// output side and sixteen readable bytes on the input side.  The state lives in locals here (a decoder that kept it in the struct, behind
// a reference, lost a third of its speed under clang: every byte stored could have been a field).  0: the margins ended (the careful
// loop takes over); 1: end of block; -1: refused.
inline int dec_fast_loop(Dec &d) {
  const uint8_t *in = d.in, *const in_end = d.in_end;
  uint8_t *out = d.out, *const out_end = d.out_end, *const dst = d.dst;
  uint64_t bb = d.bb;
  int bc = d.bc;
  const uint32_t *const LT = d.LT, *const DT = d.DT;
  int r = 0;
#define FQZ_T(nb) (bb >>= (nb), bc -= (nb))
  while (in_end - in >= 16 && out_end - out >= 258 + 72) {
    {   // refill: >= 56 bits
      uint64_t w;
      memcpy(&w, in, 8);
      bb |= w << bc;
      in += (63 - bc) >> 3;
      bc |= 56;
    }
    uint32_t e = LT[bb & ((1u << LIT_BITS) - 1)];
    if (e & K_LIT) {                            // literals for as long as the buffer holds a whole first-level index (<= 2 bytes per >= 2 bits: <= 56 bytes)
      do {
        FQZ_T((int)(e & 0xff));
        out[0] = (uint8_t)(e >> 16); out[1] = (uint8_t)(e >> 24);
        out += 1 + ((e >> 8) & 1);
        e = LT[bb & ((1u << LIT_BITS) - 1)];
      } while ((e & K_LIT) && bc >= LIT_BITS);
      continue;                                 // (what follows is decoded after the refill)
    }
    if ((e & K_MASK) == K_SUB) { FQZ_T(LIT_BITS); e = LT[(e >> 16) + (bb & ((1u << ((e >> 8) & 31)) - 1))]; }
    uint64_t saved = bb;
    FQZ_T((int)(e & 0xff));                     // code and extra bits at once: <= 15 + 5 = 20 of >= 56 bits
    if (e & K_LIT) { *out++ = (uint8_t)(e >> 16); continue; }   // (a literal with a long code)
    if ((e & K_MASK) != K_BASE) { r = (e & K_MASK) == K_EOB ? 1 : -1; break; }
    const uint32_t length = (e >> 16) + (uint32_t)((saved >> ((e >> 8) & 31)) & ((1u << ((e & 0xff) - ((e >> 8) & 31))) - 1));
    uint32_t dd = DT[bb & ((1u << DIST_BITS) - 1)];
    if ((dd & K_MASK) == K_SUB) { FQZ_T(DIST_BITS); dd = DT[(dd >> 16) + (bb & ((1u << ((dd >> 8) & 31)) - 1))]; }
    if ((dd & K_MASK) != K_BASE) { r = -1; break; }
    saved = bb;
    FQZ_T((int)(dd & 0xff));                    // <= 15 + 13 bits: <= 48 of >= 56
    const uint32_t distance = (dd >> 16) + (uint32_t)((saved >> ((dd >> 8) & 31)) & ((1u << ((dd & 0xff) - ((dd >> 8) & 31))) - 1));
    if (distance > (size_t)(out - dst)) { r = -1; break; }
    const uint8_t *from = out - distance;
    uint8_t *const stop = out + length;
    if (distance >= 8) {
      do { uint64_t w; memcpy(&w, from, 8); memcpy(out, &w, 8); from += 8; out += 8; } while (out < stop);
    } else if (distance == 1) {
      memset(out, *from, length);
    } else {
      do { *out++ = *from++; } while (out < stop);
    }
    out = stop;
  }
#undef FQZ_T
  d.in = in; d.out = out; d.bb = bb; d.bc = bc;
  return r;
}
// One symbol with every bound checked (the ends of the input and of the output).  Returns as dec_fast_loop, for one symbol.
inline int dec_careful_turn(Dec &d) {
  dec_refill(d);
  const uint32_t *const LT = d.LT, *const DT = d.DT;
  uint32_t e = LT[d.bb & ((1u << LIT_BITS) - 1)];
  if ((e & K_MASK) == K_SUB) { FQZ_TAKE(d, LIT_BITS); e = LT[(e >> 16) + (d.bb & ((1u << ((e >> 8) & 31)) - 1))]; }
  const uint64_t saved = d.bb;
  FQZ_TAKE(d, (int)(e & 0xff));
  if (e & K_LIT) {
    const size_t nl = 1 + ((e >> 8) & 1);
    if ((size_t)(d.out_end - d.out) < nl) return -1;
    *d.out++ = (uint8_t)(e >> 16);
    if (nl == 2) *d.out++ = (uint8_t)(e >> 24);
    return 0;
  }
  if ((e & K_MASK) == K_EOB) return 1;
  if ((e & K_MASK) != K_BASE) return -1;
  const uint32_t length = (e >> 16) + (uint32_t)((saved >> ((e >> 8) & 31)) & ((1u << ((e & 0xff) - ((e >> 8) & 31))) - 1));
  uint32_t dd = DT[d.bb & ((1u << DIST_BITS) - 1)];
  if ((dd & K_MASK) == K_SUB) { FQZ_TAKE(d, DIST_BITS); dd = DT[(dd >> 16) + (d.bb & ((1u << ((dd >> 8) & 31)) - 1))]; }
  if ((dd & K_MASK) != K_BASE) return -1;
  const uint64_t saved_d = d.bb;
  FQZ_TAKE(d, (int)(dd & 0xff));
  const uint32_t distance = (dd >> 16) + (uint32_t)((saved_d >> ((dd >> 8) & 31)) & ((1u << ((dd & 0xff) - ((dd >> 8) & 31))) - 1));
  if (distance > (size_t)(d.out - d.dst) || length > (size_t)(d.out_end - d.out)) return -1;
  const uint8_t *from = d.out - distance;
  for (uint32_t i = 0; i < length; ++i) d.out[i] = from[i];
  d.out += length;
  return 0;
}
// the stream has ended: inside the input (bits left in the buffer belong to the last bytes read), after exactly the promised output
inline bool dec_finished_well(const Dec &d) { return d.over <= ((size_t)d.bc >> 3) && d.out == d.out_end; }
// the symbols of the current Huffman block.  1: end of block; -1: refused.
inline int dec_block(Dec &d) {
  for (;;) {
    int r = dec_fast_loop(d);
    if (r) return r;
    r = dec_careful_turn(d);
    if (r) return r;
  }
}

// Decodes one complete raw DEFLATE stream of n bytes at src into dst; true when the stream ends exactly after out_len bytes of output
// (the BGZF trailer's ISIZE) without reading past src + n.  False for anything else -- the caller asks zlib what it thinks of such a
// member.
inline bool inflate_raw(Inflater &Z, const uint8_t *src, size_t n, uint8_t *dst, size_t out_len) {
  Dec d;
  d.begin(Z, src, n, dst, out_len);
  for (;;) {
    const int b = dec_next_block(d);
    if (b) return b == 1 && dec_finished_well(d);
    if (dec_block(d) < 0) return false;
  }
}
#undef FQZ_TAKE

}  // namespace fqz
