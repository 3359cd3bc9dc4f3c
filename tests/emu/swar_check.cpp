// TEST INFRASTRUCTURE ONLY: the sixteen-bytes-at-a-time forms of csrc/fq_emit.h against the byte statements they stand for -- every byte value in every position of a word,
// then whole records' runs (all forms: hit on either strand, no hit, borrowed strand; lengths around the piece boundaries; bytes of every kind) piece by piece against
// fq_sam_body_char / fq_bam_body_byte.  Exit code 0 and "ok" when nothing differs.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "fastquick_amd.h"
#include "fq_kernels.h"
#include "fq_emit.h"

int main() {
  long bad = 0;
  for (int pos = 0; pos < 4; ++pos)
    for (int v = 0; v < 256; ++v)
      for (int fill = 0; fill < 256; fill += 51) {
        uint8_t b[4] = {(uint8_t)fill, (uint8_t)fill, (uint8_t)fill, (uint8_t)fill};
        b[pos] = (uint8_t)v;
        uint32_t w; memcpy(&w, b, 4);
        for (int comp = 0; comp < 2; ++comp) {
          const uint32_t L = fq_swar_letters(w, comp != 0), C = fq_swar_codes(w, comp != 0);
          for (int k = 0; k < 4; ++k) {
            int cc = fq_nt4(b[k]); if (comp) cc = fq_comp(cc);
            const char want = "ACGTN"[cc > 4 ? 4 : cc];
            const int code = cc > 3 ? 15 : 1 << cc;
            if ((char)(L >> (8 * k)) != want) ++bad;
            if ((int)((C >> (8 * k)) & 0xff) != code) ++bad;
          }
        }
        const uint32_t S = fq_swar_sub33(w);
        for (int k = 0; k < 4; ++k) if (((S >> (8 * k)) & 0xff) != (uint32_t)((b[k] - 33) & 0xff)) ++bad;
        if (fq_swar_eq(w, 0x41414141u) != ((b[0] == 0x41 ? 0xffu : 0u) | (b[1] == 0x41 ? 0xff00u : 0u) | (b[2] == 0x41 ? 0xff0000u : 0u) | (b[3] == 0x41 ? 0xff000000u : 0u))) ++bad;
      }
  for (uint32_t c = 0; c < 65536; ++c) {      // four codes -> two packed bytes
    const uint32_t codes = (c & 15) | ((c >> 4) & 15) << 8 | ((c >> 8) & 15) << 16 | ((c >> 12) & 15) << 24;
    const uint32_t want = (((c & 15) << 4) | ((c >> 4) & 15)) | ((((c >> 8) & 15) << 4) | ((c >> 12) & 15)) << 8;
    if (fq_swar_pack2(codes) != want) ++bad;
  }
  // whole runs, piece by piece
  std::mt19937 rng(12345);
  const char alphabet[] = "ACGTNacgtn-.XRYKM*\x00\x7f\x80\xff";
  const int stride = 272;
  std::vector<uint8_t> row(stride + 64), qual(stride + 64), text(4096), want(4096);
  fq_result_t rec; memset(&rec, 0, sizeof rec);
  for (int iter = 0; iter < 60000; ++iter) {
    const int full_len = 1 + (int)(rng() % 260), len = 1 + (int)(rng() % full_len), clip_len = len;
    for (int i = 0; i < stride; ++i) { row[i] = (uint8_t)alphabet[rng() % (sizeof alphabet - 1)]; qual[i] = (uint8_t)((iter & 7) == 0 ? rng() & 0xff : 33 + rng() % 60); }
    const int form = (int)(rng() & 3), qsub = (iter % 11 == 0) ? 31 : 0;
    {   // SAM
      FqSamBody B; B.nomatch = form & 1; B.strand = (form >> 1) & 1; B.len = len; B.full_len = full_len; B.clip_len = clip_len; B.qsub = qsub; B.row = row.data(); B.qual = qual.data();
      const int n = fq_sam_body_len(B);
      for (int b = 0; b < n; ++b) want[b] = (uint8_t)fq_sam_body_char(B, b);
      FqSamArgs A; memset(&A, 0, sizeof A);
      uint32_t ln = 1, meta = (uint32_t)form << 16; uint64_t off = 0; int32_t pl = 0;
      rec.len = len; rec.full_len = full_len; rec.clip_len = clip_len;
      A.len = &ln; A.meta = &meta; A.off = &off; A.rec = &rec; A.packed = 1; A.seq = row.data(); A.stride = stride; A.qual = qual.data(); A.qual_stride = stride; A.text = (char *)text.data();
      A.mode = qsub ? FQ_MODE_IL13 : 0; A.pair_list = &pl; A.n_pairs = 1;
      memset(text.data(), 0xEE, text.size());
      for (int c = 0; c * FQ_SAM_PIECE < 2 * stride + 1; ++c) fq_sam_body_piece(A, 0, c);
      if (memcmp(text.data(), want.data(), (size_t)n) != 0 || text[n] != 0xEE) { if (bad < 5) fprintf(stderr, "SAM run differs: form %d len %d full %d qsub %d\n", form, len, full_len, qsub); ++bad; }
    }
    {   // BAM
      FqBamBody B; B.any = form & 1; B.strand = (form >> 1) & 1; B.len = len; B.full_len = full_len; B.clip_len = clip_len; B.qsub = qsub; B.l_seq = B.any ? full_len : len; B.row = row.data(); B.qual = qual.data();
      const int n = fq_bam_body_len(B);
      for (int b = 0; b < n; ++b) want[b] = (uint8_t)fq_bam_body_byte(B, b);
      FqBamArgs A; memset(&A, 0, sizeof A);
      uint32_t ln = 1, meta = (uint32_t)form << 16; uint64_t off = 0; int32_t pl = 0;
      rec.len = len; rec.full_len = full_len; rec.clip_len = clip_len;
      A.len = &ln; A.meta = &meta; A.off = &off; A.s.rec = &rec; A.s.packed = 1; A.s.seq = row.data(); A.s.stride = stride; A.s.qual = qual.data(); A.s.qual_stride = stride; A.out = text.data();
      A.s.mode = qsub ? FQ_MODE_IL13 : 0; A.s.pair_list = &pl; A.s.n_pairs = 1;
      memset(text.data(), 0xEE, text.size());
      for (int c = 0; c * FQ_SAM_PIECE < stride + stride / 2 + 1; ++c) fq_bam_body_piece(A, 0, c);
      if (memcmp(text.data(), want.data(), (size_t)n) != 0 || text[n] != 0xEE) { if (bad < 5) fprintf(stderr, "BAM run differs: form %d len %d full %d qsub %d\n", form, len, full_len, qsub); ++bad; }
    }
  }
  if (bad) { fprintf(stderr, "%ld differences\n", bad); return 1; }
  puts("ok");
  return 0;
}
