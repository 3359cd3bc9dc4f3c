#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <zlib.h>
#include "fq_frontend.h"
// TEST INFRASTRUCTURE: the DEVICE's BGZF member decoder (fq_frontend.h, fqz_inflate_member) run as a wavefront of one lane on the host, under
// AddressSanitizer / UBSan: valid streams of every kind (every level and strategy, stored blocks, flushed streams, members packed back to
// back at every alignment) must come out as zlib's bytes with status 0; damaged ones (bit flips, truncation, wrong promised sizes, wrong
// CRC) must be refused or -- where the damage leaves a valid stream -- decode to exactly what zlib's inflate() returns.
// tests/test_device_frontend.py builds and runs it.
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &data, int level, int strategy, int flush_every) {
  z_stream zs{};
  deflateInit2(&zs, level, Z_DEFLATED, -15, 8, strategy);
  std::vector<uint8_t> comp(compressBound(data.size()) + 64 + (flush_every ? 16 * (data.size() / flush_every + 2) : 0));
  zs.next_out = comp.data(); zs.avail_out = (uInt)comp.size();
  size_t at = 0;
  if (flush_every)
    for (; at + flush_every < data.size(); at += flush_every) {
      zs.next_in = const_cast<Bytef *>(data.data() + at); zs.avail_in = (uInt)flush_every;
      deflate(&zs, (at / flush_every) & 1 ? Z_SYNC_FLUSH : Z_FULL_FLUSH);
    }
  zs.next_in = const_cast<Bytef *>(data.data() + at); zs.avail_in = (uInt)(data.size() - at);
  deflate(&zs, Z_FINISH);
  comp.resize(zs.total_out);
  deflateEnd(&zs);
  return comp;
}
int main(int argc, char **argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  std::mt19937 rng(4242);
  FqzCrcConst *cc = new FqzCrcConst;
  fqz_crc_const_make(cc);
  FqzLds *lds = new FqzLds;
  long ok = 0, refused = 0, badcrc = 0, mism = 0;
  {   // the CRC pieces and their combination, against zlib's, at every alignment and length class
    std::vector<uint8_t> buf(70000 + 16);
    for (auto &b : buf) b = (uint8_t)rng();
    for (int o = 0; o < 9; ++o)
      for (uint32_t n : {0u, 1u, 2u, 3u, 4u, 5u, 7u, 8u, 63u, 64u, 255u, 256u, 257u, 1000u, 4096u, 65279u, 65280u, 65535u, 65536u}) {
        const uint32_t got = fqz_crc_wave(buf.data() + o, n, cc, lds->lt);
        if (got != (uint32_t)crc32(0L, buf.data() + o, n)) { ++mism; fprintf(stderr, "crc mismatch off %d n %u\n", o, n); }
      }
  }
  for (int it = 0; it < iters; ++it) {
    // several members back to back in one compressed buffer and one output buffer, at arbitrary alignments
    const int n_mem = 1 + rng() % 4;
    std::vector<std::vector<uint8_t>> datas, comps;
    std::vector<FqzMember> mem;
    std::vector<int> dmg_of;
    std::vector<uint8_t> comp_all((size_t)(rng() % 4), 0xAA);
    uint64_t out_at = rng() % 300;
    for (int k = 0; k < n_mem; ++k) {
      const size_t n = (it % 11 == 0 && k == 0) ? 60000 + rng() % 5536 : rng() % (it % 5 == 0 ? 20000 : 2500);
      std::vector<uint8_t> data(n);
      const int kind = rng() % 6;
      for (size_t i = 0; i < n; ++i)
        data[i] = kind == 0 ? "ACGT"[rng() & 3] : kind == 1 ? (uint8_t)rng() : kind == 2 ? "ACGTN\n@+F:,#"[rng() % 12] : kind == 3 ? (i > 10 && (rng() % 3) ? data[i - 1 - rng() % 10] : "AB"[rng() & 1])
                  : kind == 4 ? (i >= 316 && (rng() % 8) ? data[i - 316] : "ACGTF:,#\n@r0123456789+"[rng() % 23]) : (i > 5000 && (rng() % 16) ? data[i - 4000 - rng() % 900] : (uint8_t)(rng() % 7 + 'a'));
      const int level = rng() % 10;
      const int strat = (rng() % 5 == 0) ? Z_FIXED : (rng() % 7 == 0) ? Z_HUFFMAN_ONLY : (rng() % 9 == 0) ? Z_RLE : Z_DEFAULT_STRATEGY;
      std::vector<uint8_t> comp = deflate_raw(data, level, strat, (rng() % 6 == 0 && n > 600) ? 1 + (int)(rng() % 500) + 100 : 0);
      int dmg = rng() % 7;
      uint32_t out_len = (uint32_t)n, crc = (uint32_t)crc32(0L, data.data(), (uInt)n);
      if (dmg == 1 && !comp.empty()) { for (int q = 0; q < 1 + (int)(rng() % 3); ++q) comp[rng() % comp.size()] ^= 1u << (rng() & 7); }
      else if (dmg == 2 && !comp.empty()) comp.resize(rng() % comp.size());
      else if (dmg == 3) out_len = rng() % 2 ? (uint32_t)n + 1 + rng() % 40 : (uint32_t)(n > 1 ? n - 1 - rng() % std::min<size_t>(n - 1, 40) : 0);
      else if (dmg == 4) crc ^= 1u << (rng() & 31);
      else dmg = 0;
      FqzMember m{};
      m.in_off = comp_all.size(); m.in_len = (uint32_t)comp.size(); m.out_off = (uint32_t)out_at; m.out_len = out_len; m.crc = crc;
      comp_all.insert(comp_all.end(), comp.begin(), comp.end());
      for (int q = rng() % 3; q > 0; --q) comp_all.push_back((uint8_t)rng());     // (a trailer's worth of other bytes between members)
      out_at += out_len;
      mem.push_back(m); datas.push_back(std::move(data)); comps.push_back(std::move(comp)); dmg_of.push_back(dmg);
    }
    // exact-size heap buffers (+ the slack the kernel's contract names behind the compressed bytes)
    const size_t cpad = comp_all.size() + 1024 + 8;
    uint8_t *craw = (uint8_t *)malloc(cpad + 4);
    uint8_t *cbuf = (uint8_t *)(((uintptr_t)craw + 3) & ~(uintptr_t)3);
    memset(cbuf, 0, cpad); memcpy(cbuf, comp_all.data(), comp_all.size());
    uint8_t *oraw = (uint8_t *)malloc(out_at + 256 + 1);
    uint8_t *obuf = (uint8_t *)(((uintptr_t)oraw + 255) & ~(uintptr_t)255);
    memset(obuf, 0x5A, out_at ? out_at : 1);
    std::vector<uint32_t> status(n_mem, 99);
    FqInflateArgs A{};
    A.comp = cbuf; A.mem = mem.data(); A.n_mem = n_mem; A.out = obuf; A.status = status.data(); A.crc = cc;
    for (int k = 0; k < n_mem; ++k) status[k] = fqz_inflate_member(A, k, *lds);
    for (int k = 0; k < n_mem; ++k) {
      const int dmg = dmg_of[k];
      const std::vector<uint8_t> &data = datas[k];
      if (status[k] == FQZ_OK) {
        ++ok;
        // accepted: the bytes must be what zlib's inflate() makes of the same payload, and its CRC the member's
        std::vector<uint8_t> ref(mem[k].out_len + 1);
        z_stream zs{}; inflateInit2(&zs, -15);
        zs.next_in = cbuf + mem[k].in_off; zs.avail_in = mem[k].in_len; zs.next_out = ref.data(); zs.avail_out = (uInt)ref.size();
        const int rc = inflate(&zs, Z_FINISH);
        const bool zok = rc == Z_STREAM_END && zs.total_out == mem[k].out_len;
        inflateEnd(&zs);
        if (!zok || memcmp(ref.data(), obuf + mem[k].out_off, mem[k].out_len) || (uint32_t)crc32(0L, ref.data(), mem[k].out_len) != mem[k].crc) { ++mism; fprintf(stderr, "it %d member %d: accepted, but differs from zlib (dmg %d)\n", it, k, dmg); }
        if (dmg == 0 && (mem[k].out_len != data.size() || memcmp(data.data(), obuf + mem[k].out_off, data.size()))) { ++mism; fprintf(stderr, "it %d member %d: wrong bytes\n", it, k); }
      } else {
        if (status[k] == FQZ_BADCRC) ++badcrc; else ++refused;
        if (dmg == 0) { ++mism; fprintf(stderr, "it %d member %d: a valid stream was refused (status %u, n %zu)\n", it, k, status[k], data.size()); }
        if (dmg == 4 && status[k] != FQZ_BADCRC) { ++mism; fprintf(stderr, "it %d member %d: wrong CRC not reported as such\n", it, k); }
      }
      // whatever happened, nothing outside the member's own output range may have been written
    }
    uint64_t covered = 0;
    for (int k = 0; k < n_mem; ++k) covered += mem[k].out_len;
    for (uint64_t p = 0; p < mem[0].out_off; ++p) if (obuf[p] != 0x5A) { ++mism; fprintf(stderr, "it %d: wrote in front of the first member\n", it); break; }
    free(craw); free(oraw);
  }
  printf("ok %ld refused %ld badcrc %ld mismatches %ld\n", ok, refused, badcrc, mism);
  return mism != 0;
}
