// fq_bam.cpp -- the BAM consumer of the alignment records: BwtMapper::SetSamRecord (src/BwtMapper.cpp:977-1264, the BAM_DEBUG
// branches that are compiled in: :946) and SetSamFileHeader (:947-975), written as BAM through an own BGZF layer (zlib raw deflate
// in 64 KB blocks).  Unlike the --sam_out dialect the records carry GENOME coordinates: a reduced-reference contig is named
// CHR:POS@REF/ALT[|L], the record's RNAME is CHR and its position POS - flank + offset-in-contig - 1 (:1026-1043), the header's
// @SQ lines are the original reference's (.fai), and every record carries RG:Z.  Records are built after StatCollector's
// contig-end un-mapping (AddAlignment runs first, :2075-2079).
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_index.h"
#include "fq_kernels.h"
#include "fq_pipeline.h"
#include "fq_backend.h"
#include <mutex>

// What the reference's translation unit sees as PACKAGE_VERSION when SetSamFileHeader is compiled: libbwa's 0.0.1 (bwase.h, included
// first), not src/Version.h's 1.0.6 -- the header the reference writes says VN:0.0.1 (oracle/_ref/fq_ref_driver --bam_dump).
#define FQ_PACKAGE_VERSION "0.0.1"

namespace {
// ---- BGZF (SAM/BAM specification 4.1): gzip members with a BC extra field, at most 64 KB of payload each ----------------------
struct Bgzf {
  // BGZF blocks are compressed independently (a fresh deflate stream each), so a run of them goes through zlib on several threads and
  // comes out byte for byte what one thread would write: the block boundaries (every 0xff00 bytes of the record stream) do not move.
  FILE *fp = nullptr;
  std::vector<uint8_t> buf;
  bool ok = true;
  static const size_t kBlock = 0xff00;
  static const int kThreads = 8;
  struct Out { uint8_t d[0x10000 + 64]; size_t n = 0; };
  static bool compress_block(const uint8_t *data, size_t n, Out &o) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(data); zs.avail_in = (uInt)n;
    zs.next_out = o.d + 18; zs.avail_out = sizeof o.d - 18 - 8;
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { deflateEnd(&zs); return false; }
    const size_t clen = zs.total_out;
    deflateEnd(&zs);
    const size_t bsize = clen + 18 + 8;
    const uint8_t hdr[18] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0, (uint8_t)((bsize - 1) & 0xff), (uint8_t)((bsize - 1) >> 8)};
    memcpy(o.d, hdr, 18);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n), isz = (uint32_t)n;
    memcpy(o.d + 18 + clen, &crc, 4);
    memcpy(o.d + 18 + clen + 4, &isz, 4);
    o.n = bsize;
    return true;
  }
  void flush_blocks(size_t n_blocks, size_t last_len) {   // the first n_blocks - 1 blocks are full; the last holds last_len bytes
    if (!n_blocks) return;
    std::vector<Out> outs(n_blocks);
    std::vector<char> good(n_blocks, 0);
    auto work = [&](size_t lo, size_t hi) {
      for (size_t b = lo; b < hi; ++b) good[b] = compress_block(buf.data() + b * kBlock, b + 1 == n_blocks ? last_len : kBlock, outs[b]);
    };
    const size_t T = std::min<size_t>(kThreads, n_blocks);
    if (T <= 1) work(0, n_blocks);
    else {
      std::vector<std::thread> th;
      const size_t per = (n_blocks + T - 1) / T;
      for (size_t t = 0; t < T; ++t) { const size_t lo = t * per, hi = std::min(n_blocks, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
      for (auto &x : th) x.join();
    }
    for (size_t b = 0; b < n_blocks; ++b) { if (!good[b] || fwrite(outs[b].d, 1, outs[b].n, fp) != outs[b].n) ok = false; }
  }
  void write(const void *p, size_t n) {
    const uint8_t *s = (const uint8_t *)p;
    buf.insert(buf.end(), s, s + n);
    if (buf.size() >= kBlock * 64) {                    // a few MB at a time
      const size_t nb = buf.size() / kBlock;
      flush_blocks(nb, kBlock);
      buf.erase(buf.begin(), buf.begin() + nb * kBlock);
    }
  }
  void flush_all() {                 // what is buffered goes out as blocks now (members made elsewhere follow)
    const size_t nb = (buf.size() + kBlock - 1) / kBlock;
    if (nb) flush_blocks(nb, buf.size() - (nb - 1) * kBlock);
    buf.clear();
  }
  void write_members(const void *p, size_t n) { if (n && fwrite(p, 1, n, fp) != n) ok = false; }
  void close() {
    if (!fp) return;
    const size_t nb = (buf.size() + kBlock - 1) / kBlock;
    if (nb) flush_blocks(nb, buf.size() - (nb - 1) * kBlock);
    buf.clear();
    Out eof;
    if (!compress_block(nullptr, 0, eof) || fwrite(eof.d, 1, eof.n, fp) != eof.n) ok = false;   // the empty end-of-file block
    fclose(fp);
    fp = nullptr;
  }
};

int64_t ref_end(const FqRead &p) {   // pos_end, libbwa/bwase.c:420-432
  if (!p.cigar.empty()) { int64_t x = p.pos; for (uint16_t g : p.cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += g & 0x3fff; } return x; }
  return (int64_t)p.pos + p.len;
}
int64_t ref_end_multi(const FqMulti &q, int len) {
  if (!q.cigar.empty()) { int64_t x = q.pos; for (uint16_t g : q.cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += g & 0x3fff; } return x; }
  return (int64_t)q.pos + len;
}
int64_t five_prime(const FqRead &p) { return p.type != FQ_TYPE_NO_MATCH ? (p.strand ? ref_end(p) : (int64_t)p.pos) : -1; }
int reg2bin(int64_t beg, int64_t end) {   // SAM specification 5.3
  --end;
  if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
// The optional fields of a record.  The reference keeps them in SamRecord's hash of tags (LongHash<int> extras, 32 slots, slot =
// key & mask with linear probing, key = type << 16 | tag[1] << 8 | tag[0]: misc/bam/SamRecord.h:629, VerifyBamID/statgen/LongHash.h)
// and writes them to the record in slot order (setTagsInBuffer, misc/bam/SamRecord.cpp:3308-3340), not in the order SetSamRecord
// adds them: AM MD NM RG SM, then the X? tags in the order they were added.  bytes() replays the insertions and walks the slots.
struct Tags {
  struct Entry { uint8_t t0, t1; std::vector<uint8_t> b; };
  std::vector<Entry> list;           // in the order SetSamRecord adds them
  std::vector<uint8_t> *cur = nullptr;
  void key(const char *t, char type) {
    list.push_back(Entry{(uint8_t)t[0], (uint8_t)t[1], {}});
    cur = &list.back().b;
    cur->push_back((uint8_t)t[0]); cur->push_back((uint8_t)t[1]); cur->push_back((uint8_t)type);
  }
  void z(const char *t, const std::string &v) { key(t, 'Z'); cur->insert(cur->end(), v.begin(), v.end()); cur->push_back(0); }
  void a(const char *t, char v) { key(t, 'A'); cur->push_back((uint8_t)v); }
  void i(const char *t, long long v) {   // smallest integer type that holds the value
    if (v >= 0 && v <= 255) { key(t, 'C'); cur->push_back((uint8_t)v); }
    else if (v >= -128 && v <= 127) { key(t, 'c'); cur->push_back((uint8_t)(int8_t)v); }
    else if (v >= 0 && v <= 65535) { key(t, 'S'); const uint16_t x = (uint16_t)v; cur->insert(cur->end(), (const uint8_t *)&x, (const uint8_t *)&x + 2); }
    else if (v >= -32768 && v <= 32767) { key(t, 's'); const int16_t x = (int16_t)v; cur->insert(cur->end(), (const uint8_t *)&x, (const uint8_t *)&x + 2); }
    else { key(t, 'i'); const int32_t x = (int32_t)v; cur->insert(cur->end(), (const uint8_t *)&x, (const uint8_t *)&x + 4); }
  }
  std::vector<uint8_t> bytes() const {
    std::vector<int> slot(32, -1);
    auto place = [&](std::vector<int> &tab, int e) {
      const size_t mask = tab.size() - 1;
      size_t h = list[(size_t)e].t0 & mask;      // (the table never outgrows 256 slots here: the key's low byte is the tag's first letter)
      while (tab[h] >= 0) h = (h + 1) & mask;
      tab[h] = e;
    };
    size_t count = 0;
    for (size_t e = 0; e < list.size(); ++e) {
      if (count * 2 > slot.size()) {               // LongHash::Add grows before it inserts; SetSize re-inserts in slot order
        std::vector<int> bigger(slot.size() * 2, -1);
        for (int old : slot) if (old >= 0) place(bigger, old);
        slot.swap(bigger);
      }
      place(slot, (int)e);
      ++count;
    }
    std::vector<uint8_t> out;
    for (int e : slot) if (e >= 0) out.insert(out.end(), list[(size_t)e].b.begin(), list[(size_t)e].b.end());
    return out;
  }
};
}  // namespace

struct fq_bam {
  // the formatter's tables on the device (fq_emit.h): per contig the BAM reference id of its chromosome and where it lies in the genome
  std::mutex dev_mu;
  bool dev_on = false;
  bool host_deflate = [] { const char *e = getenv("FASTQUICK_BAM_HOST_DEFLATE"); return e && *e && *e != '0'; }();   // A/B: zlib on the host's threads for device-formatted records too
  std::vector<void *> d_bufs;
  const int32_t *d_rid = nullptr, *d_g0 = nullptr;
  const char *d_rg = nullptr;
  ~fq_bam() { for (void *p : d_bufs) fqdev::dfree(p); }
  const fq_index *ix = nullptr;
  fq_qc_opts_t o{};
  Bgzf z;
  std::vector<uint8_t> last;                             // fq_bam_format_last: the records of the last batch it formatted
  std::string err, rg_id, header_text;
  std::vector<std::pair<std::string, int>> contigs;      // BwtIndexer::contigSize
  std::map<std::string, int> ref_id;

  // genome coordinate of offset `pos1` (1-based) in reduced contig `seqid` (:1026-1043)
  void genome_coord(int seqid, int pos1, std::string *chrom, int *start) const {
    const std::string &name = ix->contigs[seqid].name;
    const size_t at = name.find('@'), colon = name.find(':');
    *chrom = name.substr(0, colon);
    const int refCoord = (int)strtol(name.substr(colon + 1, at - colon + 1).c_str(), nullptr, 10);
    *start = refCoord - (name.back() == 'L' ? o.flank_long_len : o.flank_len) + pos1 - 1;
  }
  int id_of(const std::string &chrom) const { auto it = ref_id.find(chrom); return it == ref_id.end() ? -1 : it->second; }
  // appends one BAM record to `dst`; se: SetSamRecord(p, mate = 0)
  void record(std::vector<uint8_t> &dst, const fq_opts_t *ao, const FqHostReads &hb, int n_pairs, FqRead p, const FqRead &mate, bool se = false) const;
};

// SetSamRecord, src/BwtMapper.cpp:977-1264
void fq_bam::record(std::vector<uint8_t> &dst, const fq_opts_t *ao, const FqHostReads &hb, int n_pairs, FqRead p, const FqRead &mate, bool se) const {
  const std::string name = fq_read_name(&hb, p.r % n_pairs, p.r / n_pairs, p.revived);
  uint8_t codes[FQ_LMAX + 8];
  hb.codes((size_t)p.r, p.full_len, codes);
  const uint8_t *hq = hb.qual((size_t)p.r);
  const int qsub = (ao->mode & FQ_MODE_IL13) ? 31 : 0;
  int flag, rid = -1, pos1 = 0, mrid = -1, mpos1 = 0, mapq = 0;
  long long isize = 0;
  std::vector<uint16_t> cigar;
  std::string seq, qual;
  Tags T;
  bool mate_same = false;
  if (p.type != FQ_TYPE_NO_MATCH || (!se && mate.type != FQ_TYPE_NO_MATCH)) {
    int seqid, nn, am = 0, j, readRealStart = 0;
    flag = p.extra_flag;
    if (p.type == FQ_TYPE_NO_MATCH) { p.pos = mate.pos; p.strand = mate.strand; flag |= 4; j = 1; }
    else j = (int)(ref_end(p) - p.pos);
    nn = fq_coor_pac2real(ix, p.pos, j, &seqid);
    if (p.type != FQ_TYPE_NO_MATCH && (int64_t)p.pos + j - ix->contigs[seqid].offset > ix->contigs[seqid].len) flag |= 4;
    if (p.strand) flag |= 16;
    if (!se) { if (mate.type != FQ_TYPE_NO_MATCH) { if (mate.strand) flag |= 32; } else flag |= 8; }
    std::string chrom;
    if (p.type == FQ_TYPE_NO_MATCH) { rid = -1; pos1 = 0; }
    else { genome_coord(seqid, (int)((int64_t)p.pos - ix->contigs[seqid].offset + 1), &chrom, &readRealStart); rid = id_of(chrom); pos1 = readRealStart; }
    mapq = p.mapQ;
    if (p.type != FQ_TYPE_NO_MATCH) { if (!p.cigar.empty()) cigar = p.cigar; else cigar.push_back((uint16_t)(FQ_OP_M << 14 | p.len)); }
    if (se) { mrid = -1; mpos1 = 0; isize = 0; }      // no mate: "*", 0, 0 (:1118-1122)
    else if (mate.type != FQ_TYPE_NO_MATCH) {
      int m_seqid, mstart;
      am = mate.seQ < p.seQ ? mate.seQ : p.seQ;
      fq_coor_pac2real(ix, mate.pos, mate.len, &m_seqid);
      std::string mchrom;
      genome_coord(m_seqid, (int)((int64_t)mate.pos - ix->contigs[m_seqid].offset + 1), &mchrom, &mstart);
      mate_same = seqid == m_seqid;
      mrid = mate_same ? rid : id_of(mchrom);     // "=" is resolved against this record's own reference
      isize = mate_same ? five_prime(mate) - five_prime(p) : 0;
      if (p.type == FQ_TYPE_NO_MATCH) isize = 0;
      mpos1 = mstart;
    } else { mate_same = true; mrid = rid; mpos1 = readRealStart; isize = 0; }
    if (p.strand == 0) for (j = 0; j != p.full_len; ++j) seq += "ACGTN"[codes[j] > 4 ? 4 : codes[j]];
    else for (j = 0; j != p.full_len; ++j) { const int c = codes[p.full_len - 1 - j]; seq += "TGCAN"[c > 4 ? 4 : c]; }
    // qualities: 31 back on the first len bytes of Phred+64 input, the first len bytes reversed for a reverse-strand record
    for (j = 0; j < p.full_len; ++j) {
      const int src = (p.strand && j < p.len) ? p.len - 1 - j : j;
      qual += (char)(j < p.len ? hq[src] : hq[src] - qsub);
    }
    if (!rg_id.empty()) T.z("RG", rg_id);
    if (p.clip_len < p.full_len) T.i("XC", p.clip_len);
    if (p.type != FQ_TYPE_NO_MATCH) {
      char XT = "NURM"[p.type];
      if (nn > 10) XT = 'N';
      T.a("XT", XT);
      T.i((ao->mode & FQ_MODE_COMPREAD) ? "NM" : "CM", p.nm);
      if (nn) T.i("XN", nn);
      if (!se) { T.i("SM", p.seQ); T.i("AM", am); }
      if (p.type != FQ_TYPE_MATESW) { T.i("X0", p.c1); if ((int)p.c1 <= ao->max_top2) T.i("X1", p.c2); }
      T.i("XM", p.n_mm); T.i("XO", p.n_gapo); T.i("XG", p.n_gapo + p.n_gape);
      if (p.has_md) T.z("MD", p.md);
      if (!p.multi.empty()) {
        std::ostringstream ss;
        for (const FqMulti &q : p.multi) {
          int sid;
          fq_coor_pac2real(ix, q.pos, (int)(ref_end_multi(q, p.len) - q.pos), &sid);
          ss << ix->contigs[sid].name << "," << (q.strand ? '-' : '+') << (int)((int64_t)q.pos - ix->contigs[sid].offset + 1) << ",";
          if (!q.cigar.empty()) for (uint16_t g : q.cigar) ss << (g & 0x3fff) << "MIDS"[g >> 14]; else ss << p.len << "M";
          ss << "," << q.gap + q.mm << ";";
        }
        T.z("XA", ss.str());
      }
    }
  } else {   // no match on either mate (:1225-1257)
    flag = p.extra_flag | 4 | (se ? 0 : 8);
    for (int j = 0; j != p.len; ++j) {
      int cc = codes[j];
      if (p.strand) { cc = j < p.clip_len ? codes[p.clip_len - 1 - j] : 3; cc = cc < 4 ? 3 - cc : cc; }
      seq += "ACGTN"[cc > 4 ? 4 : cc];
    }
    for (int j = 0; j < p.full_len; ++j) qual += (char)(hq[(p.strand && j < p.len) ? p.len - 1 - j : j] - qsub);
    if (!rg_id.empty()) T.z("RG", rg_id);
    if (p.clip_len < p.full_len) T.i("XC", p.clip_len);
  }
  // ---- BAM record (SAM specification 4.2) ----
  int64_t end0 = pos1 > 0 ? pos1 - 1 : 0;
  for (uint16_t g : cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) end0 += g & 0x3fff; }
  const int bin = pos1 > 0 ? reg2bin(pos1 - 1, cigar.empty() ? pos1 : end0) : 4680;
  std::vector<uint8_t> rec;
  auto put32 = [&](int32_t v) { rec.insert(rec.end(), (const uint8_t *)&v, (const uint8_t *)&v + 4); };
  auto put16 = [&](uint16_t v) { rec.insert(rec.end(), (const uint8_t *)&v, (const uint8_t *)&v + 2); };
  put32(rid); put32(pos1 - 1);
  rec.push_back((uint8_t)(name.size() + 1)); rec.push_back((uint8_t)mapq); put16((uint16_t)bin);
  put16((uint16_t)cigar.size()); put16((uint16_t)flag); put32((int32_t)seq.size());
  put32(mrid); put32(mpos1 - 1); put32((int32_t)isize);
  rec.insert(rec.end(), name.begin(), name.end()); rec.push_back(0);
  static const int bam_op[4] = {0, 1, 2, 4};   // M I D S
  for (uint16_t g : cigar) put32((int32_t)((uint32_t)(g & 0x3fff) << 4 | (uint32_t)bam_op[g >> 14]));
  for (size_t j = 0; j < seq.size(); j += 2) {
    auto nib = [](char c) { return c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 4 : c == 'T' ? 8 : 15; };
    rec.push_back((uint8_t)(nib(seq[j]) << 4 | (j + 1 < seq.size() ? nib(seq[j + 1]) : 0)));
  }
  for (size_t j = 0; j < seq.size(); ++j) rec.push_back((uint8_t)(j < qual.size() ? qual[j] - 33 : 0xff));
  { const std::vector<uint8_t> tb = T.bytes(); rec.insert(rec.end(), tb.begin(), tb.end()); }
  const int32_t bs = (int32_t)rec.size();
  dst.insert(dst.end(), (const uint8_t *)&bs, (const uint8_t *)&bs + 4);
  dst.insert(dst.end(), rec.begin(), rec.end());
  (void)mate_same;
}

extern "C" int fq_bam_create(const fq_index_t *ix, const char *fai_path, const char *bam_path, const char *rg_line, const fq_qc_opts_t *o, fq_bam_t **out) {
  if (!ix || !fai_path || !o || !out) return FQ_EINVAL;
  *out = nullptr;
  fq_bam *b = new fq_bam;
  b->ix = ix; b->o = *o;
  std::ifstream fai(fai_path);
  if (!fai.is_open()) { delete b; return FQ_EIO; }
  std::string line;
  while (std::getline(fai, line)) {   // BwtIndexer::LoadContigSize, src/BwtIndexer.cpp:771-781
    std::stringstream ss(line);
    std::string chr, length;
    ss >> chr;
    if (chr.empty()) continue;
    if (chr.find("chr") != std::string::npos || chr.find("CHR") != std::string::npos) chr = chr.substr(3);
    ss >> length;
    b->contigs.emplace_back(chr, atoi(length.c_str()));
  }
  // SetSamFileHeader (:947-975): @PG, the @RG line as given (tags in the order of the line), one @SQ per contig of the .fai
  std::ostringstream h;
  h << "@PG\tID:FASTQuick\tVN:" << FQ_PACKAGE_VERSION << "\n";
  const std::string rg = rg_line ? rg_line : "";
  if (rg.compare(0, 3, "@RG") == 0) {
    std::string esc;   // bwa_escape: "\t" written as two characters becomes a tab
    for (size_t i = 0; i < rg.size(); ++i) { if (rg[i] == '\\' && i + 1 < rg.size() && rg[i + 1] == 't') { esc += '\t'; ++i; } else esc += rg[i]; }
    const size_t idp = esc.find("\tID:");
    if (idp != std::string::npos) { size_t e = idp + 4; while (e < esc.size() && esc[e] != '\t' && esc[e] != '\n') ++e; b->rg_id = esc.substr(idp + 4, e - idp - 4); }
    std::stringstream toks(esc);
    std::string tok, id_field, rest;
    while (toks >> tok) {
      if (tok == "@RG") continue;
      if (tok.compare(0, 3, "ID:") == 0) id_field = tok; else rest += "\t" + tok;
    }
    if (!b->rg_id.empty()) h << "@RG\t" << (id_field.empty() ? "ID:" + b->rg_id : id_field) << rest << "\n";
  }
  for (size_t i = 0; i < b->contigs.size(); ++i) {
    h << "@SQ\tSN:" << b->contigs[i].first << "\tLN:" << b->contigs[i].second << "\n";
    b->ref_id.emplace(b->contigs[i].first, (int)i);   // (a repeated name keeps its first id)
  }
  b->header_text = h.str();
  if (!bam_path) { *out = b; return FQ_OK; }          // a formatter without a file (fq_bam_format_last)
  b->z.fp = fopen(bam_path, "wb");
  if (!b->z.fp) { delete b; return FQ_EIO; }
  const int32_t l_text = (int32_t)b->header_text.size(), n_ref = (int32_t)b->contigs.size();
  b->z.write("BAM\1", 4);
  b->z.write(&l_text, 4);
  b->z.write(b->header_text.data(), b->header_text.size());
  b->z.write(&n_ref, 4);
  for (const auto &cg : b->contigs) {
    const int32_t l_name = (int32_t)cg.first.size() + 1, l_ref = cg.second;
    b->z.write(&l_name, 4);
    b->z.write(cg.first.c_str(), (size_t)l_name);
    b->z.write(&l_ref, 4);
  }
  *out = b;
  return FQ_OK;
}

bool fq_bam_wants_members(const fq_bam *b) { return b->z.fp != nullptr && !b->host_deflate; }
// the formatter's part of a call's kernel arguments (on the calling context's bound state)
int fq_bam_device_prepare(fq_bam *b, FqBamArgs *a) {
  std::lock_guard<std::mutex> lk(b->dev_mu);
  if (!b->dev_on) {
    const size_t nc = b->ix->contigs.size();
    std::vector<int32_t> rid(nc + 1, -1), g0(nc + 1, 0);
    for (size_t i = 0; i < nc; ++i) {
      std::string chrom; int start;
      b->genome_coord((int)i, 1, &chrom, &start);        // start = refCoord - flank + 1 - 1
      rid[i] = b->id_of(chrom); g0[i] = start;
    }
    bool ok = true;
    auto up = [&](const void *src, size_t bytes) -> void * {
      void *d = fqdev::dmalloc(bytes ? bytes : 16);
      if (!d) { ok = false; return nullptr; }
      b->d_bufs.push_back(d);
      if (bytes && fqdev::h2d(d, src, bytes)) ok = false;
      return d;
    };
    b->d_rid = (const int32_t *)up(rid.data(), rid.size() * 4); b->d_g0 = (const int32_t *)up(g0.data(), g0.size() * 4);
    b->d_rg = (const char *)up(b->rg_id.c_str(), b->rg_id.size() + 1);
    if (!ok || fqdev::sync()) return FQ_ENODEV;
    b->dev_on = true;
  }
  a->ctg_rid = b->d_rid; a->ctg_g0 = b->d_g0; a->rg = b->d_rg; a->rg_len = (int32_t)b->rg_id.size();
  return FQ_OK;
}

// the BAM branch of PairEndMapper's consumer loop over one batch (src/BwtMapper.cpp:2054-2085): the batch's records, in input order
static int format_last(fq_bam_t *b, fq_ctx_t *c, std::vector<std::vector<uint8_t>> &parts) {
  const FqBatchState *S = fq_ctx_state(c);
  const FqHostReads hb = fq_ctx_host_reads(c);
  const fq_opts_t *ao = fq_ctx_opts(c);
  if (S->n_surv > 0 && !S->rec) { b->err = "the call's result arrays were left on the device (FQ_EMIT_DEVICE_ONLY)"; return FQ_EINVAL; }
  if (S->n_surv > 0 && !hb.has_qual()) { b->err = "the batch carries no qualities"; return FQ_EINVAL; }
  // records are independent of each other: ranges of pairs are formatted on several threads and handed to the BGZF layer in order
  auto format_range = [&](int lo, int hi, std::vector<uint8_t> &dst) {
  for (int sp = lo; sp < hi; ++sp) {
    if (S->rec[2 * (size_t)sp].type == FQ_TYPE_NO_MATCH && S->rec[2 * (size_t)sp + 1].type == FQ_TYPE_NO_MATCH) continue;
    if (ao->single_end) {   // SingleEndMapper's BAM branch (src/BwtMapper.cpp:1372-1387): AddAlignment(p, 0), SetSamRecord(p, 0)
      FqRead p = S->read(2 * (size_t)sp);
      int seqid;
      const int j = (int)(ref_end(p) - p.pos);
      fq_coor_pac2real(b->ix, p.pos, j, &seqid);
      if ((int64_t)p.pos + j - b->ix->contigs[seqid].offset > b->ix->contigs[seqid].len) p.type = FQ_TYPE_NO_MATCH;
      b->record(dst, ao, hb, S->n_pairs, p, p, true);
      continue;
    }
    FqRead p = S->read(2 * (size_t)sp), q = S->read(2 * (size_t)sp + 1);
    for (FqRead *r : {&p, &q})   // StatCollector::AddAlignment first (src/StatCollector.cpp:955-971)
      if (r->type != FQ_TYPE_NO_MATCH) {
        int seqid;
        const int j = (int)(ref_end(*r) - r->pos);
        fq_coor_pac2real(b->ix, r->pos, j, &seqid);
        if ((int64_t)r->pos + j - b->ix->contigs[seqid].offset > b->ix->contigs[seqid].len) r->type = FQ_TYPE_NO_MATCH;
      }
    b->record(dst, ao, hb, S->n_pairs, p, q);
    if (p.type == FQ_TYPE_NO_MATCH && q.type != FQ_TYPE_NO_MATCH) { p.pos = q.pos; p.strand = q.strand; }   // what the first call left in p (:991-994)
    b->record(dst, ao, hb, S->n_pairs, q, p);
  }
  };
  const int T = S->n_surv >= 256 ? 8 : 1;
  parts.assign((size_t)T, std::vector<uint8_t>());
  if (T == 1) format_range(0, S->n_surv, parts[0]);
  else {
    std::vector<std::thread> th;
    const int per = (S->n_surv + T - 1) / T;
    for (int t = 0; t < T; ++t) { const int lo = t * per, hi = std::min(S->n_surv, lo + per); if (lo < hi) th.emplace_back(format_range, lo, hi, std::ref(parts[(size_t)t])); }
    for (auto &x : th) x.join();
  }
  return FQ_OK;
}
extern "C" int fq_bam_add_last(fq_bam_t *b, fq_ctx_t *c) {
  if (!b || !c || !b->z.fp) return FQ_EINVAL;
  if (const FqBamCallOut *D = fq_ctx_bam_out(c)) {        // the records were formatted by the call's kernels (fq_ctx_attach_bam): they only leave the device here
    if (D->owner != b || !D->ready) { b->err = "fq_bam_add_last: the context's last call formatted its records for another writer, or failed"; return FQ_EINVAL; }
    if (fq_ctx_emit_wait(c)) { b->err = "fq_bam_add_last: waiting for the call's kernels failed"; return FQ_ENODEV; }      // (z_bytes comes back with them)
    int64_t n;
    if (D->z_bytes) {       // finished BGZF members (fq_deflate.h): appended behind whatever the host's layer still holds
      b->z.flush_all();
      n = fq_ctx_bam_stream(c, [](void *user, const void *data, int64_t len) -> int { ((fq_bam *)user)->z.write_members(data, (size_t)len); return ((fq_bam *)user)->z.ok ? 0 : 1; }, b, 1);
    } else n = fq_ctx_bam_stream(c, [](void *user, const void *data, int64_t len) -> int { ((fq_bam *)user)->z.write(data, (size_t)len); return ((fq_bam *)user)->z.ok ? 0 : 1; }, b, 0);
    if (n < 0) { b->err = "fq_bam_add_last: fetching the records from the device failed"; return (int)n; }
    return b->z.ok ? FQ_OK : FQ_EIO;
  }
  std::vector<std::vector<uint8_t>> parts;
  const int rc = format_last(b, c, parts);
  if (rc) return rc;
  for (auto &part : parts) if (!part.empty()) b->z.write(part.data(), part.size());
  return b->z.ok ? FQ_OK : FQ_EIO;
}
// The same records as bytes (uncompressed BAM records, block_size first), for a caller that writes them itself or elsewhere: several
// devices format the batches of their FASTQ pairs at once and one writer appends them in input order (fq_bam_write_records).
extern "C" int fq_bam_format_last(fq_bam_t *b, fq_ctx_t *c, const void **data, int64_t *len) {
  if (!b || !c || !data || !len) return FQ_EINVAL;
  if (const FqBamCallOut *D = fq_ctx_bam_out(c)) {
    if (D->owner != b || !D->ready) { b->err = "fq_bam_format_last: the context's last call formatted its records for another writer, or failed"; return FQ_EINVAL; }
    b->last.clear();
    b->last.reserve((size_t)D->bytes);
    const int64_t n = fq_ctx_bam_stream(c, [](void *user, const void *d, int64_t l) -> int { auto *v = (std::vector<uint8_t> *)user; v->insert(v->end(), (const uint8_t *)d, (const uint8_t *)d + l); return 0; }, &b->last, 0);
    if (n < 0) return (int)n;
    *data = b->last.data(); *len = (int64_t)b->last.size();
    return FQ_OK;
  }
  std::vector<std::vector<uint8_t>> parts;
  const int rc = format_last(b, c, parts);
  if (rc) return rc;
  b->last.clear();
  for (auto &part : parts) b->last.insert(b->last.end(), part.begin(), part.end());
  *data = b->last.data(); *len = (int64_t)b->last.size();
  return FQ_OK;
}
extern "C" int fq_bam_write_records(fq_bam_t *b, const void *data, int64_t len) {
  if (!b || !b->z.fp || len < 0 || (len > 0 && !data)) return FQ_EINVAL;
  if (len) b->z.write(data, (size_t)len);
  return b->z.ok ? FQ_OK : FQ_EIO;
}
// (tests, tools) n bytes as BGZF members written by the device's compressor (fq_deflate.h): the members behind each other into out (capacity cap);
// *out_len their size, *kernel_ms the compressor kernel's time.  FQ_ELIMIT when cap is too small.
extern "C" int fq_bgzf_deflate_device(int device, const uint8_t *in, int64_t n, uint8_t *out, int64_t cap, int64_t *out_len, double *kernel_ms) {
  if (!in || !out || !out_len || n < 0) return FQ_EINVAL;
  struct DevScope { fqdev::State *s; ~DevScope() { fqdev::state_destroy(s); } } scope{fqdev::state_create(device)};
  if (!scope.s || fqdev::bind(scope.s)) return FQ_ENODEV;
  const uint32_t nb = (uint32_t)((n + FQD_BLOCK - 1) / FQD_BLOCK);
  *out_len = 0;
  if (kernel_ms) *kernel_ms = 0;
  if (!nb) return FQ_OK;
  uint8_t *d_in = (uint8_t *)fqdev::dmalloc((size_t)n + 64), *d_stage = (uint8_t *)fqdev::dmalloc((size_t)nb * FQD_SLOT), *d_out = (uint8_t *)fqdev::dmalloc((size_t)nb * FQD_SLOT);
  uint32_t *d_bs = (uint32_t *)fqdev::dmalloc(((size_t)nb + 1) * 4);
  uint64_t *d_off = (uint64_t *)fqdev::dmalloc(((size_t)nb + 2) * 8);
  int rc = FQ_OK;
  uint64_t total = 0;
  if (!d_in || !d_stage || !d_out || !d_bs || !d_off) rc = FQ_ENOMEM;
  else {
    FqDeflateArgs a{d_in, (uint64_t)n, d_stage, d_bs, fqdev::crc_const(), nb};
    FqDeflatePackArgs pk{d_stage, d_bs, d_off, d_out, nb};
    double ms[FQ_K_COUNT] = {0}; uint64_t ln[FQ_K_COUNT] = {0};
    if (!a.crc || fqdev::h2d(d_in, in, (size_t)n) || fqdev::launch_deflate(a) || fqdev::launch_scan(d_bs, d_off, nb) || fqdev::launch_deflate_pack(pk) ||
        fqdev::d2h(&total, d_off + nb, 8) || fqdev::sync()) rc = FQ_ENODEV;
    else if ((int64_t)total > cap) rc = FQ_ELIMIT;
    else if (fqdev::d2h(out, d_out, (size_t)total) || fqdev::sync()) rc = FQ_ENODEV;
    fqdev::time_collect(ms, ln, FQ_K_COUNT);
    if (kernel_ms) *kernel_ms = ms[FQ_K_EMIT];
  }
  for (void *p : {(void *)d_in, (void *)d_stage, (void *)d_out, (void *)d_bs, (void *)d_off}) fqdev::dfree(p);
  *out_len = (int64_t)total;
  return rc;
}

extern "C" int fq_bam_close(fq_bam_t *b) {
  if (!b) return FQ_EINVAL;
  b->z.close();
  const bool ok = b->z.ok;
  delete b;
  return ok ? FQ_OK : FQ_EIO;
}
