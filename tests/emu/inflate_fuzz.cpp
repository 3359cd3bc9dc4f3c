#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <cstring>
#include <zlib.h>
#include "fq_inflate.h"
// TEST INFRASTRUCTURE: the front end's DEFLATE decoder under AddressSanitizer / UBSan with exact-size heap buffers -- valid streams of every
// kind and damaged ones (bit flips, truncation, trailing bytes, wrong promised sizes).  tests/test_inflate.py builds and runs it.
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 3000;
  std::mt19937 rng(12345);
  fqz::Inflater *Z = new fqz::Inflater;
  long ok = 0, refused = 0, mism = 0;
  for (int it = 0; it < iters; ++it) {
    const size_t n = 1 + rng() % (it % 7 == 0 ? 70000 : 3000);
    std::vector<uint8_t> data(n);
    const int kind = rng() % 4;
    for (size_t i = 0; i < n; ++i) data[i] = kind == 0 ? "ACGT"[rng() & 3] : kind == 1 ? (uint8_t)rng() : kind == 2 ? "ACGTN\n@+F:,#"[rng() % 12] : (i > 10 && (rng() % 3) ? data[i - 1 - rng() % 10] : "AB"[rng() & 1]);
    uLongf cl = compressBound(n) + 64;
    std::vector<uint8_t> comp(cl);
    z_stream zs{}; deflateInit2(&zs, 1 + rng() % 9, Z_DEFLATED, -15, 8, (rng() % 5 == 0) ? Z_FIXED : Z_DEFAULT_STRATEGY);
    zs.next_in = data.data(); zs.avail_in = n; zs.next_out = comp.data(); zs.avail_out = cl; deflate(&zs, Z_FINISH); cl = zs.total_out; deflateEnd(&zs);
    // damage
    size_t cn = cl;
    const int dmg = rng() % 5;
    std::vector<uint8_t> c2(comp.begin(), comp.begin() + cl);
    if (dmg == 1) { for (int k = 0; k < 1 + (int)(rng() % 4); ++k) c2[rng() % cn] ^= 1u << (rng() & 7); }
    else if (dmg == 2) { cn = rng() % cn; c2.resize(cn); }
    else if (dmg == 3) { for (int k = 0; k < 8; ++k) c2.push_back((uint8_t)rng()); cn = c2.size(); }
    size_t out_len = n;
    if (dmg == 4) out_len = rng() % 2 ? n + 1 + rng() % 50 : (n > 1 ? n - 1 - rng() % std::min<size_t>(n - 1, 50) : 0);
    // exact-size heap buffers: the sanitizer sees every overrun
    uint8_t *src = (uint8_t *)malloc(cn ? cn : 1); memcpy(src, c2.data(), cn);
    uint8_t *dst = (uint8_t *)malloc(out_len ? out_len : 1);
    {   // the checksum, both ways (carry-less multiplication where the CPU has it; the tables), against zlib's
      const uint32_t want = (uint32_t)crc32(0L, data.data(), (uInt)n);
      if (fqz::crc32(data.data(), n) != want || fqz::crc32(data.data(), n, 0, true) != want) ++mism;
      const size_t o = n > 3 ? 1 + rng() % 3 : 0;   // (an unaligned start)
      if (fqz::crc32(data.data() + o, n - o) != (uint32_t)crc32(0L, data.data() + o, (uInt)(n - o))) ++mism;
    }
    const bool r = fqz::inflate_raw(*Z, src, cn, dst, out_len);
    if (r) { ++ok; if (dmg == 0 || dmg == 3) { if (out_len != n || memcmp(dst, data.data(), n)) ++mism; } } else { ++refused; if (dmg == 0 || dmg == 3) ++mism; }
    free(src); free(dst);
  }
  printf("ok %ld refused %ld mismatches %ld\n", ok, refused, mism);
  return mism != 0;
}
