// Symbol mix of a BGZF file as the device's member decoder sees it (the decoder's host statement, -DFQZ_STATS): literals, matches by where
// their source is, steps that leave the fast loop.  g++ -O2 -DFQZ_STATS -I fastquick_amd/csrc -I include tools/inflate_mix.cpp -lz
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fq_frontend.h"
int main(int argc, char **argv) {
  if (argc < 2) return 2;
  FILE *fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  std::vector<uint8_t> f;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = fread(buf, 1, sizeof buf, fp)) > 0) f.insert(f.end(), buf, buf + n);
  fclose(fp);
  const size_t limit = argc > 2 ? (size_t)atoll(argv[2]) : 2000;      // members looked at
  FqzCrcConst *cc = new FqzCrcConst; fqz_crc_const_make(cc);
  FqzLds *lds = new FqzLds;
  size_t at = 0, members = 0; unsigned long long text = 0, comp = 0;
  std::vector<uint8_t> cbuf, obuf;
  while (at + 18 <= f.size() && members < limit) {
    const uint8_t *h = f.data() + at;
    const size_t xlen = h[10] | h[11] << 8, sz = (size_t)(h[16] | h[17] << 8) + 1, hdr = 12 + xlen;
    const uint32_t isize = h[sz - 4] | h[sz - 3] << 8 | h[sz - 2] << 16 | (uint32_t)h[sz - 1] << 24, crc = h[sz - 8] | h[sz - 7] << 8 | h[sz - 6] << 16 | (uint32_t)h[sz - 5] << 24;
    if (isize) {
      cbuf.assign(sz - hdr - 8 + 2048, 0); memcpy(cbuf.data(), h + hdr, sz - hdr - 8);
      obuf.assign(isize + 512, 0);
      uint8_t *o = (uint8_t *)(((uintptr_t)obuf.data() + 255) & ~(uintptr_t)255);
      FqzMember m{}; m.in_off = 0; m.in_len = (uint32_t)(sz - hdr - 8); m.out_off = 0; m.out_len = isize; m.crc = crc;
      uint32_t st = 9;
      FqInflateArgs A{}; A.comp = cbuf.data(); A.mem = &m; A.n_mem = 1; A.out = o; A.status = &st; A.crc = cc;
      st = fqz_inflate_member(A, 0, *lds);
      if (st) { fprintf(stderr, "member %zu: status %u\n", members, st); return 1; }
      text += isize; comp += sz; ++members;
    }
    at += sz;
  }
  const double sym = (double)(fqz_stats[0] + fqz_stats[1] + fqz_stats[2] + fqz_stats[3]);
  printf("members %zu text %llu comp %llu (%.3f)  symbols %.0f (%.2f bytes, %.2f bits each)\n literals %.1f%%  near matches %.1f%%  far %.1f%%  over own bytes %.1f%%  mean match %.1f bytes  general steps per 1000 symbols %.2f\n",
         members, text, comp, (double)comp / text, sym, text / sym, 8.0 * comp / sym, 100 * fqz_stats[0] / sym, 100 * fqz_stats[1] / sym, 100 * fqz_stats[2] / sym, 100 * fqz_stats[3] / sym,
         (double)fqz_stats[6] / (double)(fqz_stats[1] + fqz_stats[2] + fqz_stats[3] + 1e-9), 1000.0 * fqz_stats[4] / sym);
  printf(" literal/length codes: <= 6 bits %.1f%%, 7 bits %.1f%%, longer %.1f%%;  distance codes: <= 6 bits %.1f%%, 7 bits %.1f%%, longer %.1f%%\n",
         100.0 * fqz_stats[8] / (fqz_stats[8] + fqz_stats[9] + fqz_stats[10]), 100.0 * fqz_stats[9] / (fqz_stats[8] + fqz_stats[9] + fqz_stats[10]), 100.0 * fqz_stats[10] / (fqz_stats[8] + fqz_stats[9] + fqz_stats[10]),
         100.0 * fqz_stats[11] / (fqz_stats[11] + fqz_stats[12] + fqz_stats[13]), 100.0 * fqz_stats[12] / (fqz_stats[11] + fqz_stats[12] + fqz_stats[13]), 100.0 * fqz_stats[13] / (fqz_stats[11] + fqz_stats[12] + fqz_stats[13]));
  return 0;
}
