// fq_deflate.h -- BGZF members written ON THE DEVICE: one wavefront compresses one block of the BAM record stream into a complete gzip member
// (SAM/BAM specification 4.1: BC extra field, raw DEFLATE payload, CRC-32, ISIZE).
//
// What it replaces: the reference hands every SamRecord to SamFile::WriteRecord, whose BGZF layer (htslib's bgzf_write, VerifyBamID/statgen/
// BgzfFileType.h) deflates 64 KiB blocks with zlib on the one thread that also aligns.  A BAM file is defined by what it inflates to; which valid
// DEFLATE stream carries a block is the writer's choice -- so the blocks are compressed where the records are formatted (fq_emit.h), and the host
// only appends members to the file.
//
// The stream of a block is ONE fixed-Huffman block (RFC 1951, 3.2.6) of greedy LZ77 tokens:
//   * 64 positions per step, a lane each: hash of the 4 bytes there -> the last position of an EARLIER step with that hash (16-bit table in LDS),
//     match length by comparing words;
//   * the greedy parse of the step is the only serial part: a uniform loop over its tokens (a token per iteration: one readlane of the match length at
//     the parse position), which also hands every token its bit offset;
//   * every chosen lane ORs its code (Huffman codes bit-reversed: DEFLATE packs them from the most significant bit) into a small LDS buffer, whole
//     bytes of which leave for HBM once per step.
// Blocks are FQD_BLOCK bytes of input, so that a block of nothing but 9-bit literals still fits a 64 KiB member.
//
// Written in the wavefront idiom of fq_frontend.h (FQF_LANES: once per lane on the device, a loop over 64 lanes in the host-loop build).
#pragma once
#include "fq_frontend.h"

#define FQD_BLOCK 0xd000u          // input bytes per member: 53,248 x 9 / 8 + 27 < 65,536
#define FQD_SLOT 65536u            // staging bytes per member
#define FQD_HBITS 11              // 2,048 hash slots: 4 KB of LDS per wavefront (with 4,096 twelve wavefronts fit a CU, and the kernel is bound by the latency of a step's dependent loads)
#define FQD_MIN_MATCH 4
#define FQD_MAX_MATCH 258

struct FqdLds {
  union {
    uint16_t htab[1 << FQD_HBITS]; // position + 1 of the last occurrence of a hash (0: none)
    uint32_t crc_tab[1024];        // ... and, when the block is compressed, the CRC's slice tables in the same bytes
  };
  uint32_t obuf[96];               // the step's bits
};
struct FqDeflateArgs {
  const uint8_t *in; uint64_t n;   // the record stream
  uint8_t *stage;                  // [n_blocks][FQD_SLOT]
  uint32_t *bsize;                 // [n_blocks] bytes of member b
  const FqzCrcConst *crc;
  uint32_t n_blocks;
};
struct FqDeflatePackArgs { const uint8_t *stage; const uint32_t *bsize; const uint64_t *off; uint8_t *out; uint32_t n_blocks; };

FQ_HD uint32_t fqd_rev(uint32_t code, int len) { return FQF_BREV32(code) >> (32 - len); }
// (bits, nbits) of a literal in the fixed code
FQ_HD void fqd_literal(uint32_t b, uint64_t *bits, int *nbits) {
  if (b < 144) { *bits = fqd_rev(0x30 + b, 8); *nbits = 8; }
  else { *bits = fqd_rev(0x190 + (b - 144), 9); *nbits = 9; }
}
// ... of a match: length code + extra bits, distance code + extra bits
FQ_HD void fqd_match(uint32_t len, uint32_t dist, uint64_t *bits, int *nbits) {
  uint32_t l = len - 3, lcode, leb, lex;
  if (l < 8) { lcode = 257 + l; leb = 0; lex = 0; }
  else if (len == 258) { lcode = 285; leb = 0; lex = 0; }
  else { const int n = 31 - __builtin_clz(l); leb = (uint32_t)n - 2; lcode = 257 + 4 * leb + 4 + ((l >> leb) & 3); lex = l & ((1u << leb) - 1); }
  uint64_t v; int nb;
  if (lcode < 280) { v = fqd_rev(lcode - 256, 7); nb = 7; }
  else { v = fqd_rev(0xC0 + (lcode - 280), 8); nb = 8; }
  v |= (uint64_t)lex << nb; nb += (int)leb;
  const uint32_t d = dist - 1;
  uint32_t dcode, deb, dex;
  if (d < 4) { dcode = d; deb = 0; dex = 0; }
  else { const int n = 31 - __builtin_clz(d); deb = (uint32_t)n - 1; dcode = 2 * (uint32_t)n + ((d >> (n - 1)) & 1); dex = d & ((1u << deb) - 1); }
  v |= (uint64_t)fqd_rev(dcode, 5) << nb; nb += 5;
  v |= (uint64_t)dex << nb; nb += (int)deb;
  *bits = v; *nbits = nb;
}
FQ_HD uint32_t fqd_load32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// member b of the record stream: returns its size (header and trailer included)
FQ_HD uint32_t fqd_member(const FqDeflateArgs &A, uint32_t b, FqdLds &S) {
  const uint64_t lo = (uint64_t)b * FQD_BLOCK;
  const uint32_t n = (uint32_t)(A.n - lo < FQD_BLOCK ? A.n - lo : FQD_BLOCK);
  const uint8_t *in = A.in + lo;
  uint8_t *out = A.stage + (size_t)b * FQD_SLOT;
  FQF_LANES
    for (int i = lane; i < (1 << FQD_HBITS); i += 64) S.htab[i] = 0;
    for (int i = lane; i < 96; i += 64) S.obuf[i] = 0;
  FQF_LANES_END
  FQF_WAVE_FENCE();
  // the block header: BFINAL = 1, BTYPE = 01 (fixed Huffman) -- bits 1, 1, 0
  uint32_t byte_base = 18, bit_in = 3;
  FQF_LANES
    if (lane == 0) S.obuf[0] = 3;
  FQF_LANES_END
  FQF_WAVE_FENCE();
  FQF_LVAR(uint32_t, mlen); FQF_LVAR(uint32_t, tok_lo); FQF_LVAR(uint32_t, tok_hi); FQF_LVAR(uint32_t, tok_nb); FQF_LVAR(uint32_t, sel); FQF_LVAR(uint32_t, off); FQF_LVAR(uint32_t, hsh);
  uint32_t carry = 0;               // positions at the head of the step that the last match of the step before covers
  for (uint32_t base = 0; base < n; base += 64) {
    // ---- candidates of the step: the table holds positions of earlier steps only
    FQF_LANES
      const uint32_t i = base + (uint32_t)lane;
      uint32_t L = 0, dist = 0, h = 0xffffffffu;
      if (i + FQD_MIN_MATCH <= n) {
        const uint32_t v = fqd_load32(in + i);
        h = (v * 2654435761u) >> (32 - FQD_HBITS);
        const uint32_t c1 = S.htab[h];
        if (c1 && i - (c1 - 1) <= 32768u) {            // (DEFLATE's window)
          const uint32_t c = c1 - 1;
          if (fqd_load32(in + c) == v) {
            const uint32_t cap = n - i < FQD_MAX_MATCH ? n - i : FQD_MAX_MATCH;
            L = 4;
            while (L + 4 <= cap && fqd_load32(in + c + L) == fqd_load32(in + i + L)) L += 4;
            while (L < cap && in[c + L] == in[i + L]) ++L;
            dist = i - c;
          }
        }
      }
      uint64_t bits; int nb;
      if (L >= FQD_MIN_MATCH) fqd_match(L, dist, &bits, &nb);
      else { L = 0; if (i < n) fqd_literal(in[i], &bits, &nb); else { bits = 0; nb = 0; } }
      FQF_LV(mlen) = L; FQF_LV(tok_lo) = (uint32_t)bits; FQF_LV(tok_hi) = (uint32_t)(bits >> 32); FQF_LV(tok_nb) = (uint32_t)nb; FQF_LV(sel) = 0; FQF_LV(off) = 0; FQF_LV(hsh) = h;
    FQF_LANES_END
    FQF_WAVE_FENCE();
    FQF_LANES
      if (FQF_LV(hsh) != 0xffffffffu) S.htab[FQF_LV(hsh)] = (uint16_t)(base + (uint32_t)lane + 1);      // (several lanes, one slot: any of them is a valid candidate)
    FQF_LANES_END
    // ---- the greedy parse of the step.  Only matches make it serial: between two chosen matches every position is a literal, so the loop runs once
    //      per chosen MATCH (find the next candidate at or behind the parse position in the ballot of the lanes that found one; everything in front
    //      of it is literals) -- a handful of iterations per step on BAM records, where a token per iteration was sixty.  The tokens' bit offsets are
    //      a prefix sum over the chosen lanes afterwards.
    const uint32_t step_n = n - base < 64 ? n - base : 64;
    uint64_t cand = 0;                         // lanes whose position starts a match
#if defined(__HIP_DEVICE_COMPILE__)
    cand = __ballot(mlen != 0);
#else
    for (int l = 0; l < 64; ++l) if (mlen[l]) cand |= 1ull << l;
#endif
    uint64_t chosen = 0;                       // positions that emit a token
    uint32_t pos = carry;
    while (pos < step_n) {
      const uint64_t ahead = cand >> pos;
      if (!ahead) { chosen |= (~0ull >> (64 - step_n)) & (~0ull << pos); pos = step_n; break; }      // literals to the end of the step
#if defined(__HIP_DEVICE_COMPILE__)
      const uint32_t c = pos + (uint32_t)__builtin_ctzll(ahead);
#else
      const uint32_t c = pos + (uint32_t)__builtin_ctzll(ahead);
#endif
      if (c >= step_n) { chosen |= (~0ull >> (64 - step_n)) & (~0ull << pos); pos = step_n; break; }
      if (c > pos) chosen |= (~0ull >> (64 - c)) & (~0ull << pos);      // literals in front of the match
      chosen |= 1ull << c;
      pos = c + FQF_RL(mlen, c);
    }
    carry = pos - step_n;
    uint32_t bit = bit_in;
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const int lane = (int)(threadIdx.x & 63);
      sel = (uint32_t)((chosen >> lane) & 1ull);
      const uint32_t nb = sel ? tok_nb : 0u;
      uint32_t inc = nb;                       // inclusive prefix sum over the wavefront
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t up = (uint32_t)__shfl_up((int)inc, d, 64); if (lane >= d) inc += up; }
      off = bit_in + inc - nb;
      bit = bit_in + (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
#else
    for (int l = 0; l < 64; ++l) { sel[l] = (uint32_t)((chosen >> l) & 1ull); off[l] = bit; if (sel[l]) bit += tok_nb[l]; }
#endif
    // ---- the chosen tokens' bits into the step's buffer
    FQF_LANES
      if (FQF_LV(sel)) {
        const uint64_t v = (uint64_t)FQF_LV(tok_lo) | (uint64_t)FQF_LV(tok_hi) << 32;
        const uint32_t o = FQF_LV(off), w = o >> 5, s = o & 31;
        FQF_ATOMIC_OR32(&S.obuf[w], (uint32_t)(v << s));
        const uint64_t rest = s ? v >> (32 - s) : v >> 32;
        if ((uint32_t)rest) FQF_ATOMIC_OR32(&S.obuf[w + 1], (uint32_t)rest);
        if (rest >> 32) FQF_ATOMIC_OR32(&S.obuf[w + 2], (uint32_t)(rest >> 32));
      }
    FQF_LANES_END
    FQF_WAVE_FENCE();
    // ---- whole bytes leave for HBM; the partial byte stays as the head of the next step's buffer
    const uint32_t full = bit >> 3;
    FQF_LANES
      for (uint32_t j = (uint32_t)lane; j < full; j += 64) out[byte_base + j] = (uint8_t)(S.obuf[j >> 2] >> (8 * (j & 3)));
    FQF_LANES_END
    FQF_WAVE_FENCE();
    const uint32_t tail = (S.obuf[full >> 2] >> (8 * (full & 3))) & 0xffu;
    FQF_WAVE_FENCE();
    FQF_LANES
      for (int i = lane; i < 96; i += 64) S.obuf[i] = i == 0 ? tail : 0u;
    FQF_LANES_END
    FQF_WAVE_FENCE();
    byte_base += full; bit_in = bit & 7;
  }
  // end of block (symbol 256: seven zero bits), the last byte
  bit_in += 7;
  const uint32_t last = (bit_in + 7) >> 3;
  FQF_LANES
    for (uint32_t j = (uint32_t)lane; j < last; j += 64) out[byte_base + j] = (uint8_t)(S.obuf[j >> 2] >> (8 * (j & 3)));
  FQF_LANES_END
  byte_base += last;
  const uint32_t crc = fqz_crc_wave(in, n, A.crc, S.crc_tab);
  const uint32_t bs = byte_base + 8;
  FQF_LANES
    if (lane == 0) {
      const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bs - 1) & 0xff), (uint8_t)((bs - 1) >> 8)};
      for (int k = 0; k < 18; ++k) out[k] = hdr[k];
      for (int k = 0; k < 4; ++k) { out[byte_base + k] = (uint8_t)(crc >> (8 * k)); out[byte_base + 4 + k] = (uint8_t)(n >> (8 * k)); }
    }
  FQF_LANES_END
  return bs;
}
// members packed behind each other: thread t copies 16 bytes of member b = t / (FQD_SLOT / 16)
FQ_HD void fqd_pack_piece(const FqDeflatePackArgs &A, uint64_t t) {
  const uint32_t b = (uint32_t)(t / (FQD_SLOT / 16)), k = (uint32_t)(t % (FQD_SLOT / 16)) * 16;
  if (b >= A.n_blocks || k >= A.bsize[b]) return;
  const uint8_t *src = A.stage + (size_t)b * FQD_SLOT + k;
  uint8_t *dst = A.out + A.off[b] + k;
  const uint32_t m = A.bsize[b] - k < 16 ? A.bsize[b] - k : 16;
  for (uint32_t j = 0; j < m; ++j) dst[j] = src[j];
}
