// fq_frontend.cpp -- host side of the FASTQ front end on the device (kernels: fq_frontend.h).
//
// Replaces, for BGZF files of four-line records, the reader side of bwa_read_seq_with_hash_dev (src/BwtMapper.cpp:476-613): gzread
// (libbwa/bwaseqio.c:41-52) under kseq_read3_fpc (libbwa/kseq.h:327-371) on one IO thread per file (IOworkerAlt, src/BwtMapper.cpp:1973-1980).
// The host reads compressed bytes and walks the members' headers (18 bytes per 64 KiB of text); everything per byte of text happens in HBM.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_backend.h"
#include "fq_inflate.h"

namespace {
// ---- BGZF members of a compressed image (SAM spec 4.1: gzip members with a 'BC' extra subfield that states the member's size) ----
struct HostMember { size_t at, size, hdr; uint32_t isize, crc; };
// size of the member that starts at h (>= 18 readable bytes), 0 if it is not a BGZF member
size_t bgzf_member(const uint8_t *h, size_t n, size_t *hdr_len) {
  if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return 0;
  const size_t xlen = h[10] | (size_t)h[11] << 8;
  if (12 + xlen > n) return 0;
  size_t p = 12;
  while (p + 4 <= 12 + xlen) {
    const size_t sl = h[p + 2] | (size_t)h[p + 3] << 8;
    if (h[p] == 'B' && h[p + 1] == 'C' && sl == 2 && p + 6 <= 12 + xlen) { *hdr_len = 12 + xlen; return (size_t)(h[p + 4] | (size_t)h[p + 5] << 8) + 1; }
    p += 4 + sl;
  }
  return 0;
}
inline uint32_t le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
// the whole members of [p, p + n): false when something that is not a member stands at a member boundary
bool walk_members(const uint8_t *p, size_t n, std::vector<HostMember> &ms, size_t *consumed) {
  size_t at = 0;
  while (n - at >= 18) {
    size_t hdr = 0;
    const size_t sz = bgzf_member(p + at, n - at, &hdr);
    if (sz == 0) { *consumed = at; return false; }
    if (sz < hdr + 8) { *consumed = at; return false; }
    if (n - at < sz) break;
    ms.push_back({at, sz, hdr, le32(p + at + sz - 4), le32(p + at + sz - 8)});
    at += sz;
  }
  *consumed = at;
  return true;
}

struct DevScope {
  fqdev::State *s;
  explicit DevScope(int dev) : s(fqdev::state_create(dev)) {}
  ~DevScope() { fqdev::state_destroy(s); }
};
struct DevMem {
  void *p = nullptr;
  ~DevMem() { fqdev::dfree(p); }
  bool alloc(size_t n) { p = fqdev::dmalloc(n); return p != nullptr; }
};
}  // namespace

// ---- the member decoder on its own (tests, measurement) ----------------------------------------------------------------------------
// n_streams raw DEFLATE streams (RFC 1951) src[k] of n[k] bytes, each promising out_len[k] bytes with CRC-32 crc[k], inflated by the
// device's decoder -- one wavefront per stream -- into dst[k] (out_len[k] bytes each); status[k] = 0 inflated and checked, 1 refused
// (a stream the device does not take: the reader hands such a member to fq_inflate.h, then zlib), 2 CRC mismatch.
extern "C" int fq_inflate_device(int device, int n_streams, const uint8_t *const *src, const size_t *n, uint8_t *const *dst, const uint32_t *out_len, const uint32_t *crc,
                                 uint32_t *status, int repeats, double *kernel_ms) {
  if (n_streams < 0 || (n_streams && (!src || !n || !dst || !out_len || !crc || !status))) return FQ_EINVAL;
  DevScope scope(device);
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  std::vector<FqzMember> mem((size_t)n_streams);
  size_t c_total = 0, o_total = 0;
  for (int k = 0; k < n_streams; ++k) {
    if (n[k] > 0xffffffffull) return FQ_ELIMIT;
    if (o_total + out_len[k] > 0xfffffff0ull) return FQ_ELIMIT;      // (32-bit text positions inside one launch)
    mem[k].in_off = c_total; mem[k].in_len = (uint32_t)n[k]; mem[k].out_off = (uint32_t)o_total; mem[k].out_len = out_len[k]; mem[k].crc = crc[k]; mem[k].pad[0] = mem[k].pad[1] = 0;
    c_total += n[k] + (k % 3);       // (members of a file follow each other at any alignment)
    o_total += out_len[k];
  }
  std::vector<uint8_t> comp(c_total + 1024 + 16, 0);
  for (int k = 0; k < n_streams; ++k) if (n[k]) memcpy(comp.data() + mem[k].in_off, src[k], n[k]);
  DevMem d_comp, d_out, d_mem, d_status;
  if (!d_comp.alloc(comp.size()) || !d_out.alloc(o_total + 512) || !d_mem.alloc(sizeof(FqzMember) * (size_t)std::max(1, n_streams)) || !d_status.alloc(4 * (size_t)std::max(1, n_streams))) return FQ_ENOMEM;
  const FqzCrcConst *cc = fqdev::crc_const();
  if (!cc) return FQ_ENODEV;
  if (fqdev::h2d(d_comp.p, comp.data(), comp.size()) || fqdev::h2d(d_mem.p, mem.data(), sizeof(FqzMember) * mem.size()) || fqdev::dzero(d_out.p, o_total + 512) || fqdev::sync()) return FQ_ENODEV;
  FqInflateArgs a{};
  a.comp = (const uint8_t *)d_comp.p; a.mem = (const FqzMember *)d_mem.p; a.n_mem = n_streams; a.out = (uint8_t *)d_out.p; a.status = (uint32_t *)d_status.p; a.crc = cc;
  double ms[FQ_K_COUNT] = {0};
  uint64_t launches[FQ_K_COUNT] = {0};
  for (int r = 0; r < std::max(1, repeats); ++r) {
    fqdev::time_begin(0);
    if (fqdev::launch_inflate(a)) return FQ_ENODEV;
    fqdev::time_end(0);
  }
  if (fqdev::sync()) return FQ_ENODEV;
  fqdev::time_collect(ms, launches, FQ_K_COUNT);
  if (kernel_ms) *kernel_ms = launches[0] ? ms[0] / (double)launches[0] : 0.0;
  std::vector<uint8_t> out(o_total + 1);
  if (fqdev::d2h(out.data(), d_out.p, o_total) || fqdev::d2h(status, d_status.p, 4 * (size_t)n_streams) || fqdev::sync()) return FQ_ENODEV;
  for (int k = 0; k < n_streams; ++k) if (out_len[k]) memcpy(dst[k], out.data() + mem[k].out_off, out_len[k]);
  return FQ_OK;
}

// The members of a BGZF image (a whole file, or any run of whole members) inflated on the device into `out` (their text back to back).
// status[k] per member as above (NULL: not wanted); *n_members, *text_len: what was found.  FQ_EIO: not a run of BGZF members;
// FQ_ELIMIT: out_cap too small.
extern "C" int fq_bgzf_inflate_device(int device, const uint8_t *file, size_t n, uint8_t *out, size_t out_cap, int64_t *n_members, int64_t *text_len, uint32_t *status, int64_t status_cap,
                                      int repeats, double *kernel_ms) {
  if (!file || !out) return FQ_EINVAL;
  std::vector<HostMember> ms;
  size_t used = 0;
  if (!walk_members(file, n, ms, &used) || used != n) return FQ_EIO;
  std::vector<const uint8_t *> src;
  std::vector<size_t> len;
  std::vector<uint8_t *> dst;
  std::vector<uint32_t> olen, crc;
  size_t o = 0;
  for (const auto &m : ms) {
    if (m.isize == 0) continue;          // (the empty end-of-file member)
    src.push_back(file + m.at + m.hdr); len.push_back(m.size - m.hdr - 8); dst.push_back(out + o); olen.push_back(m.isize); crc.push_back(m.crc);
    o += m.isize;
    if (o > out_cap) return FQ_ELIMIT;
  }
  if (n_members) *n_members = (int64_t)src.size();
  if (text_len) *text_len = (int64_t)o;
  std::vector<uint32_t> st(src.size() + 1);
  const int rc = fq_inflate_device(device, (int)src.size(), src.data(), len.data(), dst.data(), olen.data(), crc.data(), st.data(), repeats, kernel_ms);
  if (rc) return rc;
  if (status) memcpy(status, st.data(), 4 * (size_t)std::min<int64_t>(status_cap, (int64_t)src.size()));
  return FQ_OK;
}
