// fq_emit.h -- the consumers of a call's records ON THE DEVICE: the SAM text of bwa_print_sam1 (libbwa/bwase.c:455-581) and what StatCollector
// does with every pair (src/StatCollector.cpp:424-1101) as kernels over the result arrays of the call, which are resident when stage F has
// written them (fq_records.h: fq_flat_fill_thread).  Round 5 formatted and counted on the host: 2.8 us per surviving pair for the SAM text and 3.8 us
// for the statistics, against 0.03 us per pair of device time for the alignment itself -- on on-target input the command line ran at a hundredth
// of the kernels' rate.
//
//   SAM text      a thread per record measures its line, a prefix sum places the lines, a thread per record writes its line: one D2H of text
//                 (fq_sam_line: ONE routine for both passes, so that a line is as long as it was measured)
//   statistics    a thread per pair decides what AddAlignment decides (contig-end un-mapping, which mates are added, the .InsertSizeTable line,
//                 the duplicate key, sex-chromosome counts); a wavefront per added read adds its bases to the depth / Q20 / Q30 tables of the
//                 flank regions (64 consecutive positions per atomic instruction) and to the quality / cycle histograms; the order-dependent
//                 outputs -- .InsertSizeTable lines, the markers' pileup entries -- are measured, placed by prefix sums and written in input order,
//                 so that the host only appends them
//
// Every body is a FQ_HD function of (args, index); fq_device.hip wraps each in a __global__ kernel, tests/emu loops over it.
// Reference citations are paths under the Griffan/FASTQuick tree.
#pragma once
#include "fq_kernels.h"
#include "../../include/fastquick_amd.h"

// bns_coor_pac2real (libbwa/bntseq.c:268-302): the contig of pos and the N bases inside [pos, pos + len)
FQ_HD int fq_dev_pac2real(const FqDevContigs &C, int64_t pos, int len, int *seqid) {
  int left = 0, mid = 0, right = C.n, nn = 0;
  const int ns = right;
  while (left < right) {
    mid = (left + right) >> 1;
    if (pos >= C.off[mid]) {
      if (mid == ns - 1) break;
      if (pos < C.off[mid + 1]) break;
      left = mid + 1;
    } else right = mid;
  }
  *seqid = mid;
  left = 0; right = C.n_holes;
  while (left < right) {
    const int m = (left + right) >> 1;
    const int64_t ho = C.hole_off[m];
    const int hl = C.hole_len[m];
    if (pos >= ho + hl) left = m + 1;
    else if (pos + len <= ho) right = m;
    else {
      if (pos >= ho) nn += ho + hl < pos + len ? (int)(ho + hl - pos) : len;
      else nn += ho + hl < pos + len ? hl : (int)(len - (ho - pos));
      break;
    }
  }
  return nn;
}

// text that is either measured (dst == nullptr) or written
struct FqTxt {
  char *dst;
  int64_t at;
  int32_t body_at = -1;          // where the line's SEQ column begins (fq_sam_line notes it, and which form and strand the run has)
  int32_t body_form = 0;         // bit 0: the no-match form; bit 1: the strand the columns are printed in
  bool body = true;              // false: the SEQ / tab / QUAL run of a line is stepped over, not produced (k_sam_body writes it, sixteen bytes per thread)
  FQ_HD void ch(char c) { if (dst) dst[at] = c; ++at; }
  FQ_HD void str(const char *s) { while (*s) ch(*s++); }
  FQ_HD void bytes(const char *s, int n) { for (int k = 0; k < n; ++k) ch(s[k]); }
  FQ_HD void num(long long v) {     // %lld
    char tmp[24]; int n = 0;
    unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
    if (v < 0) ch('-');
    do { tmp[n++] = (char)('0' + (int)(u % 10)); u /= 10; } while (u);
    while (n > 0) ch(tmp[--n]);
  }
  FQ_HD void cigar(const uint16_t *cg, int n) { for (int k = 0; k < n; ++k) { num(cg[k] & 0x3fff); ch("MIDS"[cg[k] >> 14]); } }
};

struct FqSamArgs {
  FqDevContigs cg;
  int32_t n_surv, n_pairs, packed, single_end, mode, max_top2;
  const int32_t *pair_list;      // [n_surv] survivor pair -> pair of the batch
  // the result arrays of the call on the device (what fq_flat_fill_thread wrote)
  const fq_result_t *rec; const uint16_t *cigar; const char *md; const fq_multi_t *multi;
  // the reads: rows of bases as the record stages read them (row of record idx: packed ? idx : end * n_pairs + pair), qualities and names compact
  // (row idx = 2 * survivor + end)
  const uint8_t *seq; int32_t stride;
  const uint8_t *qual; int32_t qual_stride;
  const char *names; int32_t name_stride;
  uint32_t *len;                 // [2 n_surv] length of the record's line (0: not printed)
  const uint64_t *off;           // ... exclusive prefix sums
  uint32_t *meta;                // [2 n_surv] where the line's SEQ column begins [0:16), the no-match form [16], the strand the columns are printed in [17]
  int32_t split;                 // 1: k_sam_fill leaves the SEQ / QUAL runs to k_sam_body (0: it writes whole lines; A/B, FASTQUICK_SAM_BODY=0)
  char *text;
};
// ---- sixteen bytes at a time ------------------------------------------------------------------------------------------------------------------
// The SEQ / QUAL runs of a line (and a BAM record's packed bases and qualities) are stated byte by byte below (fq_sam_body_char, fq_bam_body_byte); the kernels that
// write them take a piece of sixteen bytes that lies wholly inside one run in four 32-bit words: the same bytes, decided four at a time.
struct FqB16 { uint32_t w[4]; };
FQ_HD FqB16 fq_load16(const uint8_t *p) { FqB16 v; memcpy(&v, p, 16); return v; }
FQ_HD void fq_store16(void *p, const FqB16 &v) { memcpy(p, &v, 16); }
FQ_HD uint32_t fq_bswap32(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); }
FQ_HD FqB16 fq_rev16(const FqB16 &v) { FqB16 r; r.w[0] = fq_bswap32(v.w[3]); r.w[1] = fq_bswap32(v.w[2]); r.w[2] = fq_bswap32(v.w[1]); r.w[3] = fq_bswap32(v.w[0]); return r; }
// 0xff in every byte of w that equals the byte of pat there
FQ_HD uint32_t fq_swar_eq(uint32_t w, uint32_t pat) {
  const uint32_t t = w ^ pat;
  const uint32_t nz = ((((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t) >> 7) & 0x01010101u;      // 1 where the byte differs
  return (nz ^ 0x01010101u) * 0xffu;
}
// "ACGTN"[min(nt4(x), 4)] of four bytes (comp: "TGCAN"): nt4 takes A/a C/c G/g T/t and nothing else to 0..3 (fq_nt4), and x & 0xdf is 'A' for 'A' and 'a' alone
struct FqBaseMasks { uint32_t a, c, g, t; };
FQ_HD FqBaseMasks fq_swar_bases(uint32_t w) {
  const uint32_t up = w & 0xdfdfdfdfu;
  FqBaseMasks m;
  m.a = fq_swar_eq(up, 0x41414141u); m.c = fq_swar_eq(up, 0x43434343u); m.g = fq_swar_eq(up, 0x47474747u); m.t = fq_swar_eq(up, 0x54545454u);
  return m;
}
FQ_HD uint32_t fq_swar_letters(uint32_t w, bool comp) {
  const FqBaseMasks m = fq_swar_bases(w);
  const uint32_t la = comp ? 0x54545454u : 0x41414141u, lc = comp ? 0x47474747u : 0x43434343u, lg = comp ? 0x43434343u : 0x47474747u, lt = comp ? 0x41414141u : 0x54545454u;
  return (m.a & la) | (m.c & lc) | (m.g & lg) | (m.t & lt) | (~(m.a | m.c | m.g | m.t) & 0x4e4e4e4eu);
}
// the BAM codes of four bases (A 1, C 2, G 4, T 8, anything else 15; comp: of their complements), one per byte
FQ_HD uint32_t fq_swar_codes(uint32_t w, bool comp) {
  const FqBaseMasks m = fq_swar_bases(w);
  const uint32_t ca = comp ? 0x08080808u : 0x01010101u, cc = comp ? 0x04040404u : 0x02020202u, cg = comp ? 0x02020202u : 0x04040404u, ct = comp ? 0x01010101u : 0x08080808u;
  return (m.a & ca) | (m.c & cc) | (m.g & cg) | (m.t & ct) | (~(m.a | m.c | m.g | m.t) & 0x0f0f0f0fu);
}
// four codes (bytes b0 b1 b2 b3, b0 at the lowest address) -> the two packed bytes b0 << 4 | b1, b2 << 4 | b3, in bits 0..15
FQ_HD uint32_t fq_swar_pack2(uint32_t codes) {
  const uint32_t x = (codes << 4) | (codes >> 8);
  return (x & 0xffu) | ((x >> 8) & 0xff00u);
}
// every byte minus 33, modulo 256
FQ_HD uint32_t fq_swar_sub33(uint32_t x) { return ((x | 0x80808080u) - 0x21212121u) ^ (~x & 0x80808080u); }

// The SEQ column, a tab and the QUAL column of a line: three quarters of its bytes.  One statement of every character, used by the line routine and by the
// kernel that writes these runs sixteen bytes per thread (coalesced, where a thread per record puts its lanes' bytes 430 apart).
struct FqSamBody { int nomatch, strand, len, full_len, clip_len, qsub; const uint8_t *row, *qual; };
FQ_HD int fq_sam_body_len(const FqSamBody &B) { return (B.nomatch ? B.len : B.full_len) + 1 + B.full_len; }
FQ_HD char fq_sam_body_char(const FqSamBody &B, int b) {
  const int slen = B.nomatch ? B.len : B.full_len;
  if (b < slen) {
    const int j = b;
    if (B.nomatch) {          // the record of a read without a match (bwase.c:563-579): len bases of seq, or of rseq when the lost hit was on the reverse strand
      int cc = fq_nt4(B.row[j]);
      if (B.strand) { cc = j < B.clip_len ? fq_nt4(B.row[B.clip_len - 1 - j]) : 3; cc = cc < 4 ? 3 - cc : cc; }
      return "ACGTN"[cc > 4 ? 4 : cc];
    }
    if (B.strand == 0) { const int cc = fq_nt4(B.row[j]); return "ACGTN"[cc > 4 ? 4 : cc]; }
    const int cc = fq_nt4(B.row[B.full_len - 1 - j]);
    return "TGCAN"[cc > 4 ? 4 : cc];
  }
  if (b == slen) return '\t';
  const int j = b - slen - 1;
  if (B.nomatch) return (char)(B.qual[(B.strand && j < B.len) ? B.len - 1 - j : j] - B.qsub);
  // Phred+64 input: 31 comes off every quality byte on input (src/BwtMapper.cpp:549-553) and goes back on the first len bytes only when they are printed (bwase.c:516-519)
  if (j < B.len) return (char)B.qual[B.strand ? B.len - 1 - j : j];
  return (char)(B.qual[j] - B.qsub);
}
FQ_HD void fq_sam_body(const FqSamBody &B, FqTxt &o) {
  o.body_at = (int32_t)o.at;
  o.body_form = (B.nomatch ? 1 : 0) | (B.strand ? 2 : 0);
  const int n = fq_sam_body_len(B);
  if (!o.body) { o.at += n; return; }
  for (int b = 0; b < n; ++b) o.ch(fq_sam_body_char(B, b));
}

FQ_HD int fq_emit_row(int packed, int n_pairs, const int32_t *pair_list, int idx) { return packed ? idx : (idx & 1) * n_pairs + pair_list[idx >> 1]; }
FQ_HD int64_t fq_emit_ref_end(const fq_result_t &p, const uint16_t *cigar) {   // pos_end, libbwa/bwase.c:420-432
  if (p.n_cigar) {
    int64_t x = p.pos;
    const uint16_t *cg = cigar + p.cigar_off;
    for (int k = 0; k < p.n_cigar; ++k) { const int op = cg[k] >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += cg[k] & 0x3fff; }
    return x;
  }
  return (int64_t)p.pos + p.len;
}
// StatCollector::AddAlignment runs before a record is printed and turns a hit that hangs over the end of its contig into NO_MATCH
// (src/StatCollector.cpp:955-971): the type every consumer sees
FQ_HD int fq_emit_bridged_type(const FqDevContigs &C, const fq_result_t &p, const uint16_t *cigar, int *seqid) {
  *seqid = 0;
  if (p.type == FQ_TYPE_NO_MATCH) return FQ_TYPE_NO_MATCH;
  const int j = (int)(fq_emit_ref_end(p, cigar) - p.pos);
  fq_dev_pac2real(C, p.pos, j, seqid);
  return (int64_t)p.pos + j - C.off[*seqid] > C.len[*seqid] ? FQ_TYPE_NO_MATCH : p.type;
}
// the name a record prints under: `/1` `/2` stripped (src/BwtMapper.cpp:565-570), a revived mate under its partner's name written over its own
// without a terminator (expand_seq, libbwa/bwape.c:456)
FQ_HD void fq_emit_name(const char *names, int name_stride, int idx, bool revived, FqTxt &o) {
  const char *s = names + (size_t)idx * (size_t)name_stride;
  int sl = 0;
  while (sl < name_stride && s[sl]) ++sl;
  const char *q = s;
  int ql = 0;
  if (revived) {
    q = names + (size_t)(idx ^ 1) * (size_t)name_stride;
    while (ql < name_stride && q[ql]) ++ql;
  }
  const int t = ql >= sl ? (revived ? ql : sl) : sl;       // length of the overlaid name
  // character k of it: the partner's while it lasts, then this read's own
  int end = t;
  if (t > 2) {
    const char c2 = (t - 2 < ql) ? q[t - 2] : s[t - 2], c1 = (t - 1 < ql) ? q[t - 1] : s[t - 1];
    if (c2 == '/' && (c1 == '1' || c1 == '2')) end = t - 2;
  }
  for (int k = 0; k < end; ++k) o.ch(k < ql ? q[k] : s[k]);
}

// bwa_print_sam1(p, mate) for record idx (se: mate = 0, the single-end mapper's call, src/BwtMapper.cpp:1369)
FQ_HD void fq_sam_line(const FqSamArgs &A, int idx, FqTxt &o) {
  const int sp = idx >> 1;
  const bool se = A.single_end != 0;
  if (se && (idx & 1)) return;
  fq_result_t p = A.rec[idx];
  const fq_result_t mate = se ? p : A.rec[idx ^ 1];
  if (p.type == FQ_TYPE_NO_MATCH && mate.type == FQ_TYPE_NO_MATCH) return;     // src/BwtMapper.cpp:2038-2042 (before AddAlignment's un-mapping)
  int seqid = 0, m_seqid0 = 0;
  p.type = (uint8_t)fq_emit_bridged_type(A.cg, p, A.cigar, &seqid);
  const int mate_type = se ? p.type : fq_emit_bridged_type(A.cg, mate, A.cigar, &m_seqid0);
  const uint8_t *row = A.seq + (size_t)fq_emit_row(A.packed, A.n_pairs, A.pair_list, idx) * (size_t)A.stride;
  const uint8_t *qual = A.qual + (size_t)idx * (size_t)A.qual_stride;
  const int qsub = (A.mode & FQ_MODE_IL13) ? 31 : 0;
  fq_emit_name(A.names, A.name_stride, idx, p.revived != 0, o);
  if (p.type == FQ_TYPE_NO_MATCH && (se || mate_type == FQ_TYPE_NO_MATCH)) {
    // both hits of the pair hung over a contig end: the record of a read without a match (bwase.c:563-579)
    o.ch('\t'); o.num(p.extra_flag | 4 | (se ? 0 : 8)); o.str("\t*\t0\t0\t*\t*\t0\t0\t");
    const FqSamBody B = {1, p.strand, p.len, p.full_len, p.clip_len, qsub, row, qual};
    fq_sam_body(B, o);
    if (p.clip_len < p.full_len) { o.str("\tXC:i:"); o.num(p.clip_len); }
    o.ch('\n');
    return;
  }
  int nn, am = 0, flag = p.extra_flag, j;
  if (p.type == FQ_TYPE_NO_MATCH) { p.pos = mate.pos; p.strand = mate.strand; flag |= 4; j = 1; }
  else j = (int)(fq_emit_ref_end(p, A.cigar) - p.pos);
  nn = fq_dev_pac2real(A.cg, p.pos, j, &seqid);
  if (p.type != FQ_TYPE_NO_MATCH && (int64_t)p.pos + j - A.cg.off[seqid] > A.cg.len[seqid]) flag |= 4;
  if (p.strand) flag |= 16;
  if (!se) { if (mate_type != FQ_TYPE_NO_MATCH) { if (mate.strand) flag |= 32; } else flag |= 8; }
  o.ch('\t'); o.num(flag); o.ch('\t');
  o.bytes(A.cg.names + A.cg.name_off[seqid], (int)(A.cg.name_off[seqid + 1] - A.cg.name_off[seqid]));
  o.ch('\t'); o.num((int)(p.pos - A.cg.off[seqid] + 1)); o.ch('\t'); o.num(p.mapQ); o.ch('\t');
  if (p.n_cigar) o.cigar(A.cigar + p.cigar_off, p.n_cigar);
  else if (p.type == FQ_TYPE_NO_MATCH) o.ch('*');
  else { o.num(p.len); o.ch('M'); }
  if (se) o.str("\t*\t0\t0\t");
  else if (mate_type != FQ_TYPE_NO_MATCH) {
    int m_seqid;
    am = mate.seQ < p.seQ ? mate.seQ : p.seQ;
    fq_dev_pac2real(A.cg, mate.pos, mate.len, &m_seqid);
    o.ch('\t');
    if (seqid == m_seqid) o.ch('='); else o.bytes(A.cg.names + A.cg.name_off[m_seqid], (int)(A.cg.name_off[m_seqid + 1] - A.cg.name_off[m_seqid]));
    o.ch('\t');
    // 5' ends: the mate's (mapped here) minus this read's; a read that borrowed its mate's position prints 0
    const long long m5 = mate.strand ? (long long)fq_emit_ref_end(mate, A.cigar) : (long long)mate.pos;
    const long long p5 = p.strand ? (long long)fq_emit_ref_end(p, A.cigar) : (long long)p.pos;
    long long isize = seqid == m_seqid ? m5 - p5 : 0;
    if (p.type == FQ_TYPE_NO_MATCH) isize = 0;
    o.num((int)(mate.pos - A.cg.off[m_seqid] + 1)); o.ch('\t'); o.num(isize); o.ch('\t');
  } else { o.str("\t=\t"); o.num((int)(p.pos - A.cg.off[seqid] + 1)); o.str("\t0\t"); }
  {
    const FqSamBody B = {0, p.strand, p.len, p.full_len, p.clip_len, qsub, row, qual};
    fq_sam_body(B, o);
  }
  if (p.clip_len < p.full_len) { o.str("\tXC:i:"); o.num(p.clip_len); }
  if (p.type != FQ_TYPE_NO_MATCH) {
    char XT = "NURM"[p.type];
    if (nn > 10) XT = 'N';
    o.str("\tXT:A:"); o.ch(XT); o.str((A.mode & FQ_MODE_COMPREAD) ? "\tNM:i:" : "\tCM:i:"); o.num(p.nm);
    if (nn) { o.str("\tXN:i:"); o.num(nn); }
    if (!se) { o.str("\tSM:i:"); o.num(p.seQ); o.str("\tAM:i:"); o.num(am); }
    if (p.type != FQ_TYPE_MATESW) { o.str("\tX0:i:"); o.num((int)p.c1); if ((int)p.c1 <= A.max_top2) { o.str("\tX1:i:"); o.num((int)p.c2); } }
    o.str("\tXM:i:"); o.num(p.n_mm); o.str("\tXO:i:"); o.num(p.n_gapo); o.str("\tXG:i:"); o.num(p.n_gapo + p.n_gape);
    if (p.md_off != 0xffffffffu) { o.str("\tMD:Z:"); const char *m = A.md + p.md_off; while (*m) o.ch(*m++); }
    if (p.n_multi) {
      o.str("\tXA:Z:");
      for (int t = 0; t < p.n_multi; ++t) {
        const fq_multi_t q = A.multi[p.multi_off + t];
        int64_t qe = (int64_t)q.pos + p.len;
        if (q.n_cigar) { qe = q.pos; const uint16_t *cg = A.cigar + q.cigar_off; for (int k = 0; k < q.n_cigar; ++k) { const int op = cg[k] >> 14; if (op == FQ_OP_M || op == FQ_OP_D) qe += cg[k] & 0x3fff; } }
        int qs;
        fq_dev_pac2real(A.cg, q.pos, (int)(qe - q.pos), &qs);
        o.bytes(A.cg.names + A.cg.name_off[qs], (int)(A.cg.name_off[qs + 1] - A.cg.name_off[qs]));
        o.ch(','); o.ch(q.strand ? '-' : '+'); o.num((int)(q.pos - A.cg.off[qs] + 1)); o.ch(',');
        if (q.n_cigar) o.cigar(A.cigar + q.cigar_off, q.n_cigar); else { o.num(p.len); o.ch('M'); }
        o.ch(','); o.num(q.gap + q.mm); o.ch(';');
      }
    }
  }
  o.ch('\n');
}
// (the strand the columns are printed in: a read without a match of its own that borrowed its mate's position prints in the mate's)
FQ_HD void fq_sam_len_thread(const FqSamArgs &A, int idx) {
  FqTxt o; o.dst = nullptr; o.at = 0; o.body = false;
  fq_sam_line(A, idx, o);
  A.len[idx] = (uint32_t)o.at;
  A.meta[idx] = (uint32_t)(o.body_at < 0 ? 0 : o.body_at) | (uint32_t)o.body_form << 16;
}
// everything of a line but its SEQ / QUAL run
FQ_HD void fq_sam_fill_thread(const FqSamArgs &A, int idx) {
  if (!A.len[idx]) return;
  FqTxt o; o.dst = A.text + A.off[idx]; o.at = 0; o.body = A.split == 0;
  fq_sam_line(A, idx, o);
}
#define FQ_SAM_PIECE 16
// piece c of record idx's SEQ / tab / QUAL run: sixteen consecutive bytes of the text per thread.  Which form the run has and in which strand it is printed
// was decided by the line routine when it measured the line (meta).
FQ_HD void fq_sam_body_piece(const FqSamArgs &A, int idx, int c) {
  if (!A.len[idx]) return;
  const fq_result_t p = A.rec[idx];
  const int b0 = c * FQ_SAM_PIECE;
  if (b0 >= 2 * p.full_len + 1) return;              // (no run is longer)
  const uint32_t meta = A.meta[idx];
  FqSamBody B;
  B.nomatch = (int)((meta >> 16) & 1u); B.strand = (int)((meta >> 17) & 1u);
  B.len = p.len; B.full_len = p.full_len; B.clip_len = p.clip_len; B.qsub = (A.mode & FQ_MODE_IL13) ? 31 : 0;
  B.row = A.seq + (size_t)fq_emit_row(A.packed, A.n_pairs, A.pair_list, idx) * (size_t)A.stride;
  B.qual = A.qual + (size_t)idx * (size_t)A.qual_stride;
  const int n = fq_sam_body_len(B);
  char *dst = A.text + A.off[idx] + (meta & 0xffffu) + b0;
  const int slen = B.nomatch ? B.len : B.full_len;
  if (b0 + FQ_SAM_PIECE <= slen && !(B.nomatch && B.strand)) {      // wholly inside SEQ, read off the row forwards or (a hit on the reverse strand) backwards and complemented
    FqB16 v = B.strand ? fq_rev16(fq_load16(B.row + (B.full_len - FQ_SAM_PIECE - b0))) : fq_load16(B.row + b0);
    for (int k = 0; k < 4; ++k) v.w[k] = fq_swar_letters(v.w[k], B.strand != 0);
    fq_store16(dst, v);
    return;
  }
  if (b0 > slen && b0 + FQ_SAM_PIECE <= n && B.qsub == 0) {          // wholly inside QUAL: bytes as they came, the first len of them backwards on the reverse strand
    const int j0 = b0 - slen - 1;
    if (!B.strand || j0 >= B.len) { fq_store16(dst, fq_load16(B.qual + j0)); return; }
    if (j0 + FQ_SAM_PIECE <= B.len) { fq_store16(dst, fq_rev16(fq_load16(B.qual + (B.len - FQ_SAM_PIECE - j0)))); return; }
  }
  for (int t = 0; t < FQ_SAM_PIECE && b0 + t < n; ++t) dst[t] = fq_sam_body_char(B, b0 + t);
}

// ---- BAM records: SetSamRecord (src/BwtMapper.cpp:977-1264), restated field by field as fq_bam.cpp does on the host -----------------------
struct FqBamArgs {
  FqSamArgs s;
  const int32_t *ctg_rid;        // [n contigs] id of the contig's chromosome among the BAM header's references (-1: not there)
  const int32_t *ctg_g0;         // refCoord - flank: the genome coordinate of offset x of the contig is g0 + x + 1 (1-based)
  const char *rg; int32_t rg_len;   // the read group's ID ("": none)
  uint32_t *len; const uint64_t *off; uint8_t *out;
  uint32_t *meta;                // [2 n_surv] where the record's packed bases begin [0:16), the placed arm [16], the strand they are written in [17]
  int32_t split;                 // 1: k_bam_fill leaves bases and qualities to k_bam_body
};
struct FqBin {                   // bytes that are either measured (dst == nullptr) or written
  uint8_t *dst;
  int64_t at;
  int32_t body_at = -1, body_form = 0;   // where the record's packed bases begin; bit 0: a mate is placed (SetSamRecord's first arm), bit 1: the strand they are written in
  bool body = true;              // false: the packed bases and the qualities are stepped over (k_bam_body writes them, sixteen bytes per thread)
  FQ_HD void u8(uint32_t v) { if (dst) dst[at] = (uint8_t)v; ++at; }
  FQ_HD void u16(uint32_t v) { u8(v & 0xff); u8((v >> 8) & 0xff); }
  FQ_HD void u32(uint32_t v) { u8(v & 0xff); u8((v >> 8) & 0xff); u8((v >> 16) & 0xff); u8(v >> 24); }
  FQ_HD void tag(char a, char b, char t) { u8((uint8_t)a); u8((uint8_t)b); u8((uint8_t)t); }
  FQ_HD void tag_int(char a, char b, long long v) {      // the smallest integer type that holds the value (SamRecord::addIntTag)
    if (v >= 0 && v <= 255) { tag(a, b, 'C'); u8((uint32_t)v); }
    else if (v >= -128 && v <= 127) { tag(a, b, 'c'); u8((uint32_t)(int8_t)v); }
    else if (v >= 0 && v <= 65535) { tag(a, b, 'S'); u16((uint32_t)v); }
    else if (v >= -32768 && v <= 32767) { tag(a, b, 's'); u16((uint32_t)(uint16_t)(int16_t)v); }
    else { tag(a, b, 'i'); u32((uint32_t)(int32_t)v); }
  }
};
FQ_HD int fq_bam_reg2bin(int64_t beg, int64_t end) {   // SAM specification 5.3
  --end;
  if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
struct FqBamBody { int any, strand, len, full_len, clip_len, qsub, l_seq; const uint8_t *row, *qual; };
FQ_HD int fq_bam_body_len(const FqBamBody &B) { return (B.l_seq + 1) / 2 + B.l_seq; }
FQ_HD int fq_bam_base_code(const FqBamBody &B, int jj) {      // nt4 code of base jj of the SEQ column
  if (B.any) return B.strand == 0 ? fq_nt4(B.row[jj]) : fq_comp(fq_nt4(B.row[B.full_len - 1 - jj]));
  int cc = fq_nt4(B.row[jj]);
  if (B.strand) { cc = jj < B.clip_len ? fq_nt4(B.row[B.clip_len - 1 - jj]) : 3; cc = cc < 4 ? 3 - cc : cc; }
  return cc;
}
FQ_HD uint32_t fq_bam_body_byte(const FqBamBody &B, int b) {
  const int nseq = (B.l_seq + 1) / 2;
  if (b < nseq) {
    const int j = 2 * b;
    const int c0 = fq_bam_base_code(B, j), c1 = j + 1 < B.l_seq ? fq_bam_base_code(B, j + 1) : -1;
    const int n0 = c0 > 3 ? 15 : (1 << c0), n1 = c1 < 0 ? 0 : (c1 > 3 ? 15 : (1 << c1));
    return (uint32_t)(n0 << 4 | n1);
  }
  const int j = b - nseq;
  int q;
  if (j >= B.full_len) q = 0xff + 33;                // (a no-match record prints len bases and full_len qualities: the rest of the column is absent)
  else if (B.any) { const int src = (B.strand && j < B.len) ? B.len - 1 - j : j; q = j < B.len ? B.qual[src] : B.qual[src] - B.qsub; }
  else q = B.qual[(B.strand && j < B.len) ? B.len - 1 - j : j] - B.qsub;
  return (uint32_t)(q - 33) & 0xffu;
}
// the optional fields in the order the reference WRITES them: SamRecord keeps its tags in a 32-slot hash keyed by the tag's first letter (linear
// probing) and walks the slots (misc/bam/SamRecord.cpp:3308-3340); SetSamRecord adds them in the order of the list below
enum { FQ_BT_RG = 0, FQ_BT_XC, FQ_BT_XT, FQ_BT_NM, FQ_BT_XN, FQ_BT_SM, FQ_BT_AM, FQ_BT_X0, FQ_BT_X1, FQ_BT_XM, FQ_BT_XO, FQ_BT_XG, FQ_BT_MD, FQ_BT_XA, FQ_BT_COUNT };
FQ_HD void fq_bam_record(const FqBamArgs &A, int idx, FqBin &o) {
  const FqSamArgs &S = A.s;
  const bool se = S.single_end != 0;
  if (se && (idx & 1)) return;
  fq_result_t p = S.rec[idx];
  const fq_result_t mate = se ? p : S.rec[idx ^ 1];
  if (p.type == FQ_TYPE_NO_MATCH && mate.type == FQ_TYPE_NO_MATCH) return;     // src/BwtMapper.cpp:2038-2042
  int seqid = 0, m_seqid0 = 0;
  p.type = (uint8_t)fq_emit_bridged_type(S.cg, p, S.cigar, &seqid);
  const int mate_type = se ? p.type : fq_emit_bridged_type(S.cg, mate, S.cigar, &m_seqid0);
  const uint8_t *row = S.seq + (size_t)fq_emit_row(S.packed, S.n_pairs, S.pair_list, idx) * (size_t)S.stride;
  const uint8_t *hq = S.qual + (size_t)idx * (size_t)S.qual_stride;
  const int qsub = (S.mode & FQ_MODE_IL13) ? 31 : 0;
  const int64_t start = o.at;
  o.u32(0);                                          // block_size: patched below
  int flag, rid = -1, pos1 = 0, mrid = -1, mpos1 = 0, mapq = 0, n_cig = 0, nn = 0, am = 0;
  long long isize = 0;
  const bool any = p.type != FQ_TYPE_NO_MATCH || (!se && mate_type != FQ_TYPE_NO_MATCH);
  const uint8_t ptype = p.type;
  if (any) {
    int j, readRealStart = 0;
    flag = p.extra_flag;
    if (p.type == FQ_TYPE_NO_MATCH) { p.pos = mate.pos; p.strand = mate.strand; flag |= 4; j = 1; }
    else j = (int)(fq_emit_ref_end(p, S.cigar) - p.pos);
    nn = fq_dev_pac2real(S.cg, p.pos, j, &seqid);
    if (ptype != FQ_TYPE_NO_MATCH && (int64_t)p.pos + j - S.cg.off[seqid] > S.cg.len[seqid]) flag |= 4;
    if (p.strand) flag |= 16;
    if (!se) { if (mate_type != FQ_TYPE_NO_MATCH) { if (mate.strand) flag |= 32; } else flag |= 8; }
    if (ptype == FQ_TYPE_NO_MATCH) { rid = -1; pos1 = 0; }
    else { readRealStart = A.ctg_g0[seqid] + (int)((int64_t)p.pos - S.cg.off[seqid] + 1) - 1; rid = A.ctg_rid[seqid]; pos1 = readRealStart; }
    mapq = p.mapQ;
    if (ptype != FQ_TYPE_NO_MATCH) n_cig = p.n_cigar ? p.n_cigar : 1;
    if (se) { mrid = -1; mpos1 = 0; isize = 0; }
    else if (mate_type != FQ_TYPE_NO_MATCH) {
      int m_seqid;
      am = mate.seQ < p.seQ ? mate.seQ : p.seQ;
      fq_dev_pac2real(S.cg, mate.pos, mate.len, &m_seqid);
      const int mstart = A.ctg_g0[m_seqid] + (int)((int64_t)mate.pos - S.cg.off[m_seqid] + 1) - 1;
      const bool same = seqid == m_seqid;
      mrid = same ? rid : A.ctg_rid[m_seqid];
      const long long m5 = mate.strand ? (long long)fq_emit_ref_end(mate, S.cigar) : (long long)mate.pos;
      const long long p5 = ptype != FQ_TYPE_NO_MATCH ? (p.strand ? (long long)fq_emit_ref_end(p, S.cigar) : (long long)p.pos) : -1;
      isize = same ? m5 - p5 : 0;
      if (ptype == FQ_TYPE_NO_MATCH) isize = 0;
      mpos1 = mstart;
    } else { mrid = rid; mpos1 = readRealStart; isize = 0; }
  } else flag = p.extra_flag | 4 | (se ? 0 : 8);
  const int l_seq = any ? p.full_len : p.len;
  // name
  FqTxt nm; nm.dst = nullptr; nm.at = 0;
  fq_emit_name(S.names, S.name_stride, idx, p.revived != 0, nm);
  const int l_name = (int)nm.at + 1;
  int64_t end0 = pos1 > 0 ? pos1 - 1 : 0;
  if (n_cig) {
    if (p.n_cigar) { const uint16_t *cg = S.cigar + p.cigar_off; for (int k = 0; k < p.n_cigar; ++k) { const int op = cg[k] >> 14; if (op == FQ_OP_M || op == FQ_OP_D) end0 += cg[k] & 0x3fff; } }
    else end0 += p.len;
  }
  const int bin = pos1 > 0 ? fq_bam_reg2bin(pos1 - 1, n_cig == 0 ? pos1 : end0) : 4680;
  o.u32((uint32_t)rid); o.u32((uint32_t)(pos1 - 1));
  o.u8((uint32_t)l_name & 0xff); o.u8((uint32_t)mapq); o.u16((uint32_t)bin);
  o.u16((uint32_t)n_cig); o.u16((uint32_t)flag); o.u32((uint32_t)l_seq);
  o.u32((uint32_t)mrid); o.u32((uint32_t)(mpos1 - 1)); o.u32((uint32_t)(int32_t)isize);
  { FqTxt w; w.dst = o.dst ? (char *)o.dst + o.at : nullptr; w.at = 0; fq_emit_name(S.names, S.name_stride, idx, p.revived != 0, w); o.at += w.at; o.u8(0); }
  if (n_cig) {
    if (p.n_cigar) { const uint16_t *cg = S.cigar + p.cigar_off; for (int k = 0; k < p.n_cigar; ++k) { const int op = cg[k] >> 14; o.u32((uint32_t)(cg[k] & 0x3fff) << 4 | (uint32_t)(op == 3 ? 4 : op)); } }
    else o.u32((uint32_t)p.len << 4);
  }
  // bases, two per byte (=ACMGRSVTWYHKDBN: A 1, C 2, G 4, T 8, N 15), then qualities: two thirds of a record's bytes (fq_bam_body_byte states every one of them)
  {
    const FqBamBody B = {any ? 1 : 0, p.strand, p.len, p.full_len, p.clip_len, qsub, l_seq, row, hq};
    o.body_at = (int32_t)(o.at - start);
    o.body_form = (B.any ? 1 : 0) | (B.strand ? 2 : 0);
    const int nb = fq_bam_body_len(B);
    if (!o.body) o.at += nb;
    else for (int b = 0; b < nb; ++b) o.u8(fq_bam_body_byte(B, b));
  }
  // ---- tags, in the order of the reference's 32-slot hash
  bool have[FQ_BT_COUNT];
  for (int t = 0; t < FQ_BT_COUNT; ++t) have[t] = false;
  have[FQ_BT_RG] = A.rg_len > 0;
  have[FQ_BT_XC] = p.clip_len < p.full_len;
  if (any && ptype != FQ_TYPE_NO_MATCH) {
    have[FQ_BT_XT] = have[FQ_BT_NM] = true;
    have[FQ_BT_XN] = nn != 0;
    have[FQ_BT_SM] = have[FQ_BT_AM] = !se;
    have[FQ_BT_X0] = ptype != FQ_TYPE_MATESW;
    have[FQ_BT_X1] = have[FQ_BT_X0] && (int)p.c1 <= S.max_top2;
    have[FQ_BT_XM] = have[FQ_BT_XO] = have[FQ_BT_XG] = true;
    have[FQ_BT_MD] = p.md_off != 0xffffffffu;
    have[FQ_BT_XA] = p.n_multi != 0;
  }
  const char first_letter[FQ_BT_COUNT] = {'R', 'X', 'X', (S.mode & FQ_MODE_COMPREAD) ? 'N' : 'C', 'X', 'S', 'A', 'X', 'X', 'X', 'X', 'X', 'M', 'X'};
  int8_t slot[32];
  for (int h = 0; h < 32; ++h) slot[h] = -1;
  for (int t = 0; t < FQ_BT_COUNT; ++t) {
    if (!have[t]) continue;
    int h = first_letter[t] & 31;
    while (slot[h] >= 0) h = (h + 1) & 31;
    slot[h] = (int8_t)t;
  }
  for (int h = 0; h < 32; ++h) {
    switch (slot[h]) {
      case FQ_BT_RG: o.tag('R', 'G', 'Z'); for (int k = 0; k < A.rg_len; ++k) o.u8((uint8_t)A.rg[k]); o.u8(0); break;
      case FQ_BT_XC: o.tag_int('X', 'C', p.clip_len); break;
      case FQ_BT_XT: { char XT = "NURM"[ptype]; if (nn > 10) XT = 'N'; o.tag('X', 'T', 'A'); o.u8((uint8_t)XT); break; }
      case FQ_BT_NM: o.tag_int((S.mode & FQ_MODE_COMPREAD) ? 'N' : 'C', 'M', p.nm); break;
      case FQ_BT_XN: o.tag_int('X', 'N', nn); break;
      case FQ_BT_SM: o.tag_int('S', 'M', p.seQ); break;
      case FQ_BT_AM: o.tag_int('A', 'M', am); break;
      case FQ_BT_X0: o.tag_int('X', '0', (long long)p.c1); break;
      case FQ_BT_X1: o.tag_int('X', '1', (long long)p.c2); break;
      case FQ_BT_XM: o.tag_int('X', 'M', p.n_mm); break;
      case FQ_BT_XO: o.tag_int('X', 'O', p.n_gapo); break;
      case FQ_BT_XG: o.tag_int('X', 'G', p.n_gapo + p.n_gape); break;
      case FQ_BT_MD: { o.tag('M', 'D', 'Z'); const char *m = S.md + p.md_off; while (*m) o.u8((uint8_t)*m++); o.u8(0); break; }
      case FQ_BT_XA: {
        o.tag('X', 'A', 'Z');
        FqTxt w; w.dst = o.dst ? (char *)o.dst + o.at : nullptr; w.at = 0;
        for (int t = 0; t < p.n_multi; ++t) {
          const fq_multi_t q = S.multi[p.multi_off + t];
          int64_t qe = (int64_t)q.pos + p.len;
          if (q.n_cigar) { qe = q.pos; const uint16_t *cg = S.cigar + q.cigar_off; for (int k = 0; k < q.n_cigar; ++k) { const int op = cg[k] >> 14; if (op == FQ_OP_M || op == FQ_OP_D) qe += cg[k] & 0x3fff; } }
          int qs;
          fq_dev_pac2real(S.cg, q.pos, (int)(qe - q.pos), &qs);
          w.bytes(S.cg.names + S.cg.name_off[qs], (int)(S.cg.name_off[qs + 1] - S.cg.name_off[qs]));
          w.ch(','); w.ch(q.strand ? '-' : '+'); w.num((int)((int64_t)q.pos - S.cg.off[qs] + 1)); w.ch(',');
          if (q.n_cigar) w.cigar(S.cigar + q.cigar_off, q.n_cigar); else { w.num(p.len); w.ch('M'); }
          w.ch(','); w.num(q.gap + q.mm); w.ch(';');
        }
        o.at += w.at; o.u8(0);
        break;
      }
      default: break;
    }
  }
  if (o.dst) { const uint32_t bs = (uint32_t)(o.at - start - 4); o.dst[start] = (uint8_t)bs; o.dst[start + 1] = (uint8_t)(bs >> 8); o.dst[start + 2] = (uint8_t)(bs >> 16); o.dst[start + 3] = (uint8_t)(bs >> 24); }
}
FQ_HD void fq_bam_len_thread(const FqBamArgs &A, int idx) {
  FqBin o; o.dst = nullptr; o.at = 0; o.body = false;
  fq_bam_record(A, idx, o);
  A.len[idx] = (uint32_t)o.at;
  A.meta[idx] = (uint32_t)(o.body_at < 0 ? 0 : o.body_at) | (uint32_t)o.body_form << 16;
}
FQ_HD void fq_bam_fill_thread(const FqBamArgs &A, int idx) {
  if (!A.len[idx]) return;
  FqBin o; o.dst = A.out + A.off[idx]; o.at = 0; o.body = A.split == 0;
  fq_bam_record(A, idx, o);
}
// piece c of record idx's packed bases and qualities: sixteen consecutive bytes per thread
FQ_HD void fq_bam_body_piece(const FqBamArgs &A, int idx, int c) {
  if (!A.len[idx]) return;
  const fq_result_t p = A.s.rec[idx];
  const uint32_t meta = A.meta[idx];
  FqBamBody B;
  B.any = (int)((meta >> 16) & 1u); B.strand = (int)((meta >> 17) & 1u);
  B.len = p.len; B.full_len = p.full_len; B.clip_len = p.clip_len; B.qsub = (A.s.mode & FQ_MODE_IL13) ? 31 : 0;
  B.l_seq = B.any ? p.full_len : p.len;
  B.row = A.s.seq + (size_t)fq_emit_row(A.s.packed, A.s.n_pairs, A.s.pair_list, idx) * (size_t)A.s.stride;
  B.qual = A.s.qual + (size_t)idx * (size_t)A.s.qual_stride;
  const int n = fq_bam_body_len(B), b0 = c * FQ_SAM_PIECE;
  if (b0 >= n) return;
  uint8_t *dst = A.out + A.off[idx] + (meta & 0xffffu) + b0;
  const int nseq = (B.l_seq + 1) / 2;
  if (B.any && 2 * (b0 + FQ_SAM_PIECE) <= B.l_seq) {                 // sixteen bytes of packed bases = 32 bases of the row, forwards or backwards and complemented
    const uint8_t *src = B.strand ? B.row + (B.full_len - 2 * FQ_SAM_PIECE - 2 * b0) : B.row + 2 * b0;
    FqB16 lo = fq_load16(src), hi = fq_load16(src + 16);
    if (B.strand) { const FqB16 t = fq_rev16(hi); hi = fq_rev16(lo); lo = t; }
    FqB16 o;
    o.w[0] = fq_swar_pack2(fq_swar_codes(lo.w[0], B.strand != 0)) | fq_swar_pack2(fq_swar_codes(lo.w[1], B.strand != 0)) << 16;
    o.w[1] = fq_swar_pack2(fq_swar_codes(lo.w[2], B.strand != 0)) | fq_swar_pack2(fq_swar_codes(lo.w[3], B.strand != 0)) << 16;
    o.w[2] = fq_swar_pack2(fq_swar_codes(hi.w[0], B.strand != 0)) | fq_swar_pack2(fq_swar_codes(hi.w[1], B.strand != 0)) << 16;
    o.w[3] = fq_swar_pack2(fq_swar_codes(hi.w[2], B.strand != 0)) | fq_swar_pack2(fq_swar_codes(hi.w[3], B.strand != 0)) << 16;
    fq_store16(dst, o);
    return;
  }
  if (b0 >= nseq && B.qsub == 0) {                                    // wholly inside the qualities that are there: the bytes minus 33
    const int j0 = b0 - nseq;
    if (j0 + FQ_SAM_PIECE <= (B.l_seq < B.full_len ? B.l_seq : B.full_len)) {
      bool ok = true;
      FqB16 v;
      if (!B.strand || j0 >= B.len) v = fq_load16(B.qual + j0);
      else if (j0 + FQ_SAM_PIECE <= B.len) v = fq_rev16(fq_load16(B.qual + (B.len - FQ_SAM_PIECE - j0)));
      else ok = false;
      if (ok) { for (int k = 0; k < 4; ++k) v.w[k] = fq_swar_sub33(v.w[k]); fq_store16(dst, v); return; }
    }
  }
  for (int t = 0; t < FQ_SAM_PIECE && b0 + t < n; ++t) dst[t] = (uint8_t)fq_bam_body_byte(B, b0 + t);
}

enum { FQ_EOP_SAM_LEN = 0, FQ_EOP_SAM_FILL, FQ_EOP_BAM_LEN, FQ_EOP_BAM_FILL, FQ_EOP_SAM_BODY, FQ_EOP_BAM_BODY, FQ_EOP_COUNT };

// =====================================================================================================================================
// StatCollector on the device: AddAlignment (src/StatCollector.cpp:950-1101), AddSingleAlignment (:424-620), ProcessPairStatus (:623-921)
// =====================================================================================================================================
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_ATOMIC_INC32(p) atomicAdd((unsigned int *)(p), 1u)
#define FQ_ATOMIC_DEC32(p) atomicAdd((unsigned int *)(p), 0xffffffffu)
#define FQ_ATOMIC_ADD64_PLAIN(p, v) atomicAdd((unsigned long long *)(p), (unsigned long long)(v))
#define FQ_ATOMIC_MIN64_PLAIN(p, v) atomicMin((unsigned long long *)(p), (unsigned long long)(v))
#define FQ_ATOMIC_CAS64(p, cmp, v) atomicCAS((unsigned long long *)(p), (unsigned long long)(cmp), (unsigned long long)(v))
#else
#define FQ_ATOMIC_INC32(p) (++*(p))
#define FQ_ATOMIC_DEC32(p) (--*(p))
#define FQ_ATOMIC_ADD64_PLAIN(p, v) (*(p) += (v))
#define FQ_ATOMIC_MIN64_PLAIN(p, v) (*(p) = *(p) < (uint64_t)(v) ? *(p) : (uint64_t)(v))
static inline uint64_t fq_host_cas64(uint64_t *p, uint64_t cmp, uint64_t v) { const uint64_t old = *p; if (old == cmp) *p = v; return old; }
#define FQ_ATOMIC_CAS64(p, cmp, v) fq_host_cas64((uint64_t *)(p), (uint64_t)(cmp), (uint64_t)(v))
#endif

#define FQ_QC_INSERT_LIMIT 4096            // INSERT_SIZE_LIMIT
#define FQ_QC_DUP_EMPTY 0xffffffffffffffffull
// what RestoreVcfSites leaves (src/StatCollector.cpp:1742-1839), flattened: flank regions and markers per chromosome, the contigs' place in the genome
struct FqQcGeom {
  const int32_t *ctg_chrom;      // [n contigs] chromosome of the lists below; -1: none of them; -2: the contig's name has no ':' (AddSingleAlignment does not take it)
  const int32_t *ctg_g0;         // refCoord - flank: the genome coordinate of a read at offset x of the contig is g0 + x
  const int32_t *ctg_reg_lo;     // the first flank region of the contig's chromosome that ends at or behind the contig's first base (where a read's walk over the regions starts)
  const uint8_t *ctg_sex;        // the name holds an 'X' or a 'Y'
  const int32_t *chr_reg0;       // [n_chrom + 1] first flank region of a chromosome; regions sorted by start, disjoint (RegionList::Collapse)
  const int32_t *reg_start, *reg_end;
  const uint32_t *reg_base;      // first index of the region in the depth tables
  const int32_t *chr_mk0;        // [n_chrom + 1] first marker of a chromosome; markers sorted by position
  const int32_t *mk_pos;
  const uint32_t *mk_idx;        // index into the markers' pileups
  const uint8_t *dbsnp;          // [table] 1: a known variant site (its mismatches are not counted)
};
enum { FQ_QC_C_RETAINED1 = 0, FQ_QC_C_RETAINED2, FQ_QC_C_FAILED1, FQ_QC_C_FAILED2, FQ_QC_C_UNMAPPED, FQ_QC_C_DUP, FQ_QC_C_PROPER, FQ_QC_C_COUNT };
struct FqPileEntry { uint32_t k; int32_t cyc; uint8_t base; int8_t qual; uint8_t maq, strand; };   // one read base over a marker (UpdateInfoVecAtMarker, :339-360)
struct FqQcArgs {
  FqSamArgs s;                   // the call's records, reads and names (text / len / off unused)
  FqDevIndex ix;                 // the 2-bit reference
  FqQcGeom g;
  int32_t cal_dup, shard;
  uint64_t ord_base;             // pairs the consumer has seen before this call (orders the first counts of the sex-chromosome contigs)
  // the consumer's tables
  uint32_t *depth, *q20, *q30;   // over the flank positions.  depth: a DIFFERENCE table (+1 where a run of counted positions begins, -1 behind its end; modulo 2^32); q20 / q30: the counted bases BELOW the threshold
  uint64_t *hist;                // [4][256] EmpRep, misEmpRep, EmpCycle, misEmpCycle
  uint64_t *insert_dist;         // [FQ_QC_INSERT_LIMIT]
  uint64_t *est_hist;            // [4][FQ_QC_INSERT_LIMIT] what InsertSizeEstimator reads back from the .InsertSizeTable lines (src/InsertSizeEstimator.cpp:43-143), counted as
                                 // the lines are decided: observed inserts of PropPair lines; censored ones of FwdOnly, of RevOnly, of half-clipped PartialPair lines
  uint64_t *counters;            // striped like the work counters (FQ_C_STRIPES x FQ_C_STRIDE)
  uint32_t *sex_cnt;             // [n contigs][4] overlapped, fully, pair_overlapped, fully_paired
  uint64_t *sex_first;           // [n contigs] key of the first count (2 * pair ordinal + which), ~0: never
  uint64_t *dup_tab; uint64_t dup_mask;   // open-addressing set of the proper pairs' keys (start << 32 | end); a shard consumer lists its keys instead:
  uint64_t *dup_key;             // [n_surv] key or ~0 (shard)
  // per call
  int32_t *added;                // [2 n_surv] the contig of a record that went through AddSingleAlignment, or -1
  uint32_t *ist_len; const uint64_t *ist_off; char *ist_text;      // .InsertSizeTable lines per pair
  uint32_t *pt_cnt; const uint64_t *pt_off; FqPileEntry *pt;       // pileup entries per record
};

FQ_HD bool fq_dupset_insert(uint64_t *tab, uint64_t mask, uint64_t key) {   // true: the key was there
  uint64_t h = key * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29;
  for (uint64_t at = h & mask;; at = (at + 1) & mask) {
    const uint64_t cur = FQ_ATOMIC_CAS64(&tab[at], FQ_QC_DUP_EMPTY, key);
    if (cur == FQ_QC_DUP_EMPTY) return false;
    if (cur == key) return true;
  }
}
FQ_HD void fq_dupset_rehash_thread(const uint64_t *old, uint64_t *tab, uint64_t mask, int64_t i) {
  if (old[i] != FQ_QC_DUP_EMPTY) (void)fq_dupset_insert(tab, mask, old[i]);
}

// a mate as the .InsertSizeTable sees it (MateSpan of fq_qc.cpp)
struct FqSpan {
  fq_result_t r;
  int idx, type, present, placed;
  int contig, flag, clip_left, clip_right;
  int64_t start, stop, contig_lo, contig_hi;
};
FQ_HD bool fq_span_reverse(const FqSpan &m) { return m.r.strand != 0; }
FQ_HD int fq_span_room(const FqSpan &m) {
  return fq_span_reverse(m) ? (m.contig_hi >= m.stop ? (int)(m.stop - m.contig_lo) : -1) : (m.start >= m.contig_lo ? (int)(m.contig_hi - m.start) : -1);
}
FQ_HD FqSpan fq_span_of(const FqQcArgs &A, int idx, int type, bool present, bool placed) {
  FqSpan m;
  m.idx = idx; m.type = type; m.present = present ? 1 : 0; m.placed = placed ? 1 : 0;
  m.contig = -1; m.flag = 0; m.clip_left = m.clip_right = 0; m.start = m.stop = m.contig_lo = m.contig_hi = 0;
  if (!present) { m.r = fq_result_t(); return m; }
  m.r = A.s.rec[idx];
  m.flag = m.r.extra_flag | (type == FQ_TYPE_NO_MATCH ? 4 : 0) | (m.r.strand ? 16 : 0);
  if (!placed) return m;
  fq_dev_pac2real(A.s.cg, m.r.pos, (int)(fq_emit_ref_end(m.r, A.s.cigar) - m.r.pos), &m.contig);
  m.contig_lo = A.s.cg.off[m.contig]; m.contig_hi = m.contig_lo + (int64_t)A.s.cg.len[m.contig];
  if (m.r.n_cigar) {
    const uint16_t *cg = A.s.cigar + m.r.cigar_off;
    if ((cg[0] >> 14) == FQ_OP_S) m.clip_left = cg[0] & 0x3fff;
    if ((cg[m.r.n_cigar - 1] >> 14) == FQ_OP_S) m.clip_right = cg[m.r.n_cigar - 1] & 0x3fff;
  }
  m.start = (int64_t)(uint32_t)(m.r.pos - (uint32_t)m.clip_left);
  m.stop = (int64_t)(uint32_t)(m.r.pos - (uint32_t)m.clip_left + (uint32_t)m.r.len);
  return m;
}
FQ_HD void fq_span_columns(const FqQcArgs &A, const FqSpan &m, FqTxt &o) {
  if (m.placed) {
    o.ch('\t'); o.bytes(A.s.cg.names + A.s.cg.name_off[m.contig], (int)(A.s.cg.name_off[m.contig + 1] - A.s.cg.name_off[m.contig]));
    o.ch('\t'); o.num((long long)((int64_t)m.r.pos - m.contig_lo + 1)); o.ch('\t'); o.num(m.flag); o.ch('\t'); o.num(m.r.len); o.ch('\t');
    if (m.r.n_cigar) o.cigar(A.s.cigar + m.r.cigar_off, m.r.n_cigar); else { o.num(m.r.len); o.ch('M'); }
  } else { o.str("\t*\t*\t"); o.num(m.flag); o.str("\t0\t*"); }
}
// what the estimator's reader makes of the line that is being written (InputInsertSizeTable): kind 0 PropPair, 1 FwdOnly, 2 RevOnly, 3 PartialPair; other lines count nothing
FQ_HD void fq_est_count(const FqQcArgs &A, const FqSpan &a, const FqSpan &b, int lim_fwd, int lim_rev, int insert, int kind) {
  const int L = FQ_QC_INSERT_LIMIT;
  int Max = lim_fwd, Max2 = lim_rev, Obs = insert;
  if (Max >= L || Max == -1) Max = L - 1;
  if (Max2 >= L || Max2 == -1) Max2 = L - 1;
  if (Obs >= L || Obs == -1) Obs = L - 1;
  if (Max < 0 || Max2 < 0 || Obs < 0) return;          // (the reference indexes unchecked; its own writer produces no such line)
  if (kind == 0) FQ_ATOMIC_ADD64_PLAIN(&A.est_hist[0 * L + Obs], 1);
  else if (kind == 1) FQ_ATOMIC_ADD64_PLAIN(&A.est_hist[1 * L + Max], 1);
  else if (kind == 2) FQ_ATOMIC_ADD64_PLAIN(&A.est_hist[2 * L + Max2], 1);
  else {
    auto clipped = [&](const FqSpan &m) FQ_LAMBDA_INLINE {     // the CIGAR column holds an 'S'
      if (!m.placed || !m.r.n_cigar) return false;
      const uint16_t *cg = A.s.cigar + m.r.cigar_off;
      for (int k = 0; k < m.r.n_cigar; ++k) if ((cg[k] >> 14) == FQ_OP_S) return true;
      return false;
    };
    const bool s1 = clipped(a), s2 = clipped(b);
    if (!s1 && s2) FQ_ATOMIC_ADD64_PLAIN(&A.est_hist[3 * L + ((a.flag & 16) ? Max2 : Max)], 1);
    else if (s1 && !s2) FQ_ATOMIC_ADD64_PLAIN(&A.est_hist[3 * L + ((b.flag & 16) ? Max2 : Max)], 1);
  }
}
FQ_HD void fq_ist_line(const FqQcArgs &A, const FqSpan &a, const FqSpan &b, int name_idx, bool name_revived, int lim_fwd, int lim_rev, int insert, const char *outcome, FqTxt &o) {
  fq_emit_name(A.s.names, A.s.name_stride, name_idx, name_revived, o);
  o.ch('\t'); o.num(lim_fwd); o.ch('\t'); o.num(lim_rev); o.ch('\t'); o.num(insert);
  fq_span_columns(A, a, o);
  fq_span_columns(A, b, o);
  o.ch('\t'); o.str(outcome); o.ch('\n');
}
// ProcessPairStatus: type 0 only the first mate is placed, 1 both, 2 only the second; q_present = 0: the single-end mapper's call (no mate at all).
// effects: count (insert-size histogram, duplicate key); the line goes to o either way.
FQ_HD void fq_pair_status(const FqQcArgs &A, int sp, int tP, int tQ, int type, bool q_present, bool effects, FqTxt &o) {
  const FqSpan a = fq_span_of(A, 2 * sp, tP, true, type != 2), b = fq_span_of(A, 2 * sp + 1, tQ, q_present, q_present && type != 0);
  if (type != 1) {
    const FqSpan &m = type == 0 ? a : b;
    if (m.r.mapQ == 0) { fq_ist_line(A, a, b, m.idx, m.r.revived != 0, -1, -1, -1, "LowQual", o); return; }
    const int room = fq_span_room(m);
    if (room < 0) return;
    const bool rv = fq_span_reverse(m);
    fq_ist_line(A, a, b, m.idx, m.r.revived != 0, rv ? -1 : room, rv ? room : -1, -1, rv ? "RevOnly" : "FwdOnly", o);
    if (effects) fq_est_count(A, a, b, rv ? -1 : room, rv ? room : -1, -1, rv ? 2 : 1);
    return;
  }
  const FqSpan *fwd = nullptr, *rev = nullptr;
  if (!fq_span_reverse(a) && fq_span_reverse(b) && a.r.pos < b.r.pos) { fwd = &a; rev = &b; }
  else if (!fq_span_reverse(b) && fq_span_reverse(a) && b.r.pos < a.r.pos) { fwd = &b; rev = &a; }
  if (!fwd) { fq_ist_line(A, a, b, a.idx, a.r.revived != 0, -1, -1, -1, "NotPair", o); return; }
  const int rf = fq_span_room(*fwd), rr = fq_span_room(*rev);
  const int lim_fwd = rf < FQ_QC_INSERT_LIMIT - 1 ? rf : FQ_QC_INSERT_LIMIT - 1, lim_rev = rr < FQ_QC_INSERT_LIMIT - 1 ? rr : FQ_QC_INSERT_LIMIT - 1;
  if (a.contig != b.contig) { if (effects) FQ_ATOMIC_ADD64_PLAIN(&A.insert_dist[0], 1); fq_ist_line(A, a, b, a.idx, a.r.revived != 0, lim_fwd, lim_rev, -1, "NotPair", o); return; }
  if (a.r.mapQ == 0 || b.r.mapQ == 0) { fq_ist_line(A, a, b, a.idx, a.r.revived != 0, lim_fwd, lim_rev, -1, "LowQual", o); return; }
  const int start = (int)(uint32_t)fwd->start, end = (int)(uint32_t)rev->stop, insert = end - start;
  const bool proper = lim_fwd != -1 && lim_rev != -1, unclipped = fwd->clip_left == 0 && rev->clip_right == 0;
  if (effects && insert >= 0 && insert < FQ_QC_INSERT_LIMIT) FQ_ATOMIC_ADD64_PLAIN(&A.insert_dist[insert], 1);
  fq_ist_line(A, a, b, a.idx, a.r.revived != 0, lim_fwd, lim_rev, insert, proper ? "PropPair" : "PartialPair", o);
  if (effects) {
    fq_est_count(A, a, b, lim_fwd, lim_rev, insert, proper ? 0 : 3);
    bool dup = false;
    const bool keyed = proper && unclipped;
    // the duplicate key: contig and both outer ends ("%d:%d:%d" in the reference).  A proper pair's ends lie inside its contig, so the ends alone name it.
    const uint64_t key = (uint64_t)(uint32_t)start << 32 | (uint64_t)(uint32_t)end;
    if (keyed) {
      if (A.shard) A.dup_key[sp] = key;
      else dup = fq_dupset_insert(A.dup_tab, A.dup_mask, key);
    }
    FQ_WAVE_COUNT(&A.counters[FQ_QC_C_PROPER], keyed);
    FQ_WAVE_COUNT(&A.counters[FQ_QC_C_DUP], dup);
  }
}

// the reference bases a read covers, block by block (for_match_blocks of fq_qc.cpp): f(absoluteSite, length, cycle, onRead, onRef) per M block
struct FqBlockIter {
  const uint16_t *cg; int n_cigar, k;
  int strand, sign, absolute, cyc, on_read, on_ref, len;
  bool done_single;
};
FQ_HD FqBlockIter fq_blocks_begin(const fq_result_t &p, const uint16_t *cigar, int readRealStart) {
  FqBlockIter it;
  it.cg = cigar + p.cigar_off; it.n_cigar = p.n_cigar; it.k = 0;
  it.strand = p.strand != 0; it.sign = p.strand ? -1 : 1;
  it.absolute = readRealStart; it.cyc = p.strand != 0 ? p.full_len - 1 : 0; it.on_read = 0; it.on_ref = 0; it.len = p.len;
  it.done_single = false;
  return it;
}
// the next M block: its (absoluteSite, length, cycle, onRead, onRef); false at the end
FQ_HD bool fq_blocks_next(FqBlockIter &it, int *abs0, int *cl, int *cyc, int *on_read, int *on_ref) {
  if (!it.n_cigar) {
    if (it.done_single) return false;
    it.done_single = true;
    *abs0 = it.absolute; *cl = it.len; *cyc = it.cyc; *on_read = 0; *on_ref = 0;
    return true;
  }
  while (it.k < it.n_cigar) {
    const int l = it.cg[it.k] & 0x3fff, op = it.cg[it.k] >> 14;
    ++it.k;
    if (op == FQ_OP_M) {
      *abs0 = it.absolute; *cl = l; *cyc = it.cyc; *on_read = it.on_read; *on_ref = it.on_ref;
      it.absolute += l; it.cyc += l * it.sign; it.on_read += l; it.on_ref += l;
      return true;
    }
    if (op == FQ_OP_D) { it.absolute += l; it.on_ref += l; }
    else { it.cyc += l * it.sign; it.on_read += l; }      // S, I
  }
  return false;
}
// AddSingleAlignment's conditions for record idx of bridged type `type`: mapped, mapQ >= 20, a contig of this path
FQ_HD bool fq_qc_takes(const FqQcArgs &A, const fq_result_t &p, int type, int *seqid, int *readRealStart) {
  if (type == FQ_TYPE_NO_MATCH || p.mapQ < 20) return false;
  fq_dev_pac2real(A.s.cg, p.pos, (int)(fq_emit_ref_end(p, A.s.cigar) - p.pos), seqid);
  if (A.g.ctg_chrom[*seqid] == -2) return false;
  *readRealStart = A.g.ctg_g0[*seqid] + (int)((int64_t)p.pos - A.s.cg.off[*seqid]);
  return true;
}
// the pileup entries of an added record (UpdateInfoVecAtMarker): counted, or written to out
FQ_HD uint32_t fq_qc_pile(const FqQcArgs &A, int idx, const fq_result_t &p, int seqid, int readRealStart, FqPileEntry *out) {
  const int chrom = A.g.ctg_chrom[seqid];
  if (chrom < 0) return 0;
  const int m0 = A.g.chr_mk0[chrom], m1 = A.g.chr_mk0[chrom + 1];
  if (m0 == m1) return 0;
  const uint8_t *row = A.s.seq + (size_t)fq_emit_row(A.s.packed, A.s.n_pairs, A.s.pair_list, idx) * (size_t)A.s.stride;
  const uint8_t *hq = A.s.qual + (size_t)idx * (size_t)A.s.qual_stride;
  const int qsub = (A.s.mode & FQ_MODE_IL13) ? 31 : 0;
  uint32_t n = 0;
  FqBlockIter it = fq_blocks_begin(p, A.s.cigar, readRealStart);
  int abs0, cl, cyc0, on_read, on_ref;
  while (fq_blocks_next(it, &abs0, &cl, &cyc0, &on_read, &on_ref)) {
    int lo = m0, hi = m1;                      // first marker at or behind abs0
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (A.g.mk_pos[mid] < abs0) lo = mid + 1; else hi = mid; }
    for (int m = lo; m < m1 && A.g.mk_pos[m] < abs0 + cl; ++m) {
      if (out) {
        const int d = A.g.mk_pos[m] - abs0, rr = on_read + d;
        FqPileEntry e;
        e.k = A.g.mk_idx[m]; e.cyc = cyc0 + d * it.sign; e.maq = (uint8_t)(p.mapQ + 33); e.strand = (uint8_t)(p.strand != 0);
        if (p.strand == 0) { const int cc = fq_nt4(row[rr]); e.base = (uint8_t)"ACGTN"[cc > 4 ? 4 : cc]; e.qual = (int8_t)(hq[rr] - qsub - 33); }
        else { const int cc = fq_nt4(row[p.full_len - 1 - rr]); e.base = (uint8_t)"TGCAN"[cc > 4 ? 4 : cc]; e.qual = (int8_t)(hq[p.full_len - 1 - rr] - qsub - 33); }
        out[n] = e;
      }
      ++n;
    }
  }
  return n;
}

// AddAlignment for one surviving pair (the consumer loop of PairEndMapper, src/BwtMapper.cpp:2026-2052; SingleEndMapper's :1355-1370)
FQ_HD void fq_qc_pair_core(const FqQcArgs &A, int sp, bool effects, FqTxt &o) {
  const bool se = A.s.single_end != 0;
  const fq_result_t P = A.s.rec[2 * sp], Q = A.s.rec[2 * sp + 1];
  bool unmapped = false;
  int retained = 0, failed = 0;
  bool addP = false, addQ = false;
  if (P.type == FQ_TYPE_NO_MATCH && Q.type == FQ_TYPE_NO_MATCH) unmapped = true;
  else {
    int seqid = 0, seqid2 = 0, rs, sd;
    const int tP = fq_emit_bridged_type(A.s.cg, P, A.s.cigar, &seqid);
    const int tQ = se ? FQ_TYPE_NO_MATCH : fq_emit_bridged_type(A.s.cg, Q, A.s.cigar, &seqid2);
    auto partial = [&](const fq_result_t &r) FQ_LAMBDA_INLINE { const uint16_t *cg = A.s.cigar + r.cigar_off; for (int k = 0; k < r.n_cigar; ++k) if ((cg[k] >> 14) == FQ_OP_S) return true; return false; };
    // cs(name): counts of a sex-chromosome contig, and when it was first counted (the file lists the contigs in the order of an unordered_map filled in that order)
    auto cs = [&](int contig, int field, int which) FQ_LAMBDA_INLINE {
      if (!effects) return;
      FQ_ATOMIC_INC32(&A.sex_cnt[(size_t)contig * 4 + field]);
      FQ_ATOMIC_MIN64_PLAIN(&A.sex_first[contig], (A.ord_base + (uint64_t)sp) * 2 + (uint64_t)which);
    };
    if (se) {
      if (tP == FQ_TYPE_NO_MATCH) failed = 2;
      else if (fq_qc_takes(A, P, tP, &sd, &rs)) {
        addP = true;
        if (A.g.ctg_sex[seqid]) { cs(seqid, 0, 0); if (!partial(P)) cs(seqid, 1, 0); }
        fq_pair_status(A, sp, tP, FQ_TYPE_NO_MATCH, 0, false, effects, o);
        failed = 1; retained = 1;
      } else failed = 2;
    } else if (tP == FQ_TYPE_NO_MATCH) {
      if (fq_qc_takes(A, Q, tQ, &sd, &rs)) {
        addQ = true;
        if (A.g.ctg_sex[seqid2]) { cs(seqid2, 0, 0); if (!partial(Q)) cs(seqid2, 1, 0); }
        fq_pair_status(A, sp, tP, tQ, 2, true, effects, o);
        failed = 1; retained = 1;
      } else failed = 2;
    } else if (tQ == FQ_TYPE_NO_MATCH) {
      if (fq_qc_takes(A, P, tP, &sd, &rs)) {
        addP = true;
        if (A.g.ctg_sex[seqid]) { cs(seqid, 0, 0); if (!partial(P)) cs(seqid, 1, 0); }
        fq_pair_status(A, sp, tP, tQ, 0, true, effects, o);
        failed = 1; retained = 1;
      } else failed = 2;
    } else {
      const bool same = seqid == seqid2;       // (contig names are distinct: pname == qname)
      if (partial(P)) {
        if (A.g.ctg_sex[seqid2]) {
          if (partial(Q)) cs(seqid2, 0, 0);
          else { cs(seqid2, 0, 0); cs(seqid2, 1, 0); }
          if (same) cs(seqid2, 2, 0);
          cs(seqid, 0, 1);
        }
      } else if (A.g.ctg_sex[seqid2]) {
        if (partial(Q)) { cs(seqid2, 0, 0); if (same) cs(seqid2, 2, 0); }
        else {
          cs(seqid2, 0, 0); cs(seqid2, 1, 0);
          if (same) { cs(seqid2, 2, 0); cs(seqid2, 3, 0); }
        }
        cs(seqid, 0, 1); cs(seqid, 1, 1);
      }
      fq_pair_status(A, sp, tP, tQ, 1, true, effects, o);
      addP = fq_qc_takes(A, P, tP, &sd, &rs);
      addQ = fq_qc_takes(A, Q, tQ, &sd, &rs);
      retained = (addP ? 1 : 0) + (addQ ? 1 : 0);
      failed = 2 - retained;
    }
  }
  if (!effects) return;
  A.ist_len[sp] = (uint32_t)o.at;
  for (int e = 0; e < 2; ++e) {
    uint32_t n = 0;
    int seqid = -1;
    if (e ? addQ : addP) {
      const fq_result_t &r = e ? Q : P;
      int rs;
      fq_dev_pac2real(A.s.cg, r.pos, (int)(fq_emit_ref_end(r, A.s.cigar) - r.pos), &seqid);
      rs = A.g.ctg_g0[seqid] + (int)((int64_t)r.pos - A.s.cg.off[seqid]);
      n = fq_qc_pile(A, 2 * sp + e, r, seqid, rs, nullptr);
    }
    A.added[2 * sp + e] = seqid;           // (the per-base kernel starts from the contig: no search of its own)
    A.pt_cnt[2 * sp + e] = n;
  }
  if (A.shard && A.dup_key && A.ist_len[sp] == 0) { /* (no line: no key either) */ }
  FQ_WAVE_COUNT(&A.counters[FQ_QC_C_UNMAPPED], unmapped);
  FQ_WAVE_COUNT(&A.counters[FQ_QC_C_RETAINED1], retained == 1);
  FQ_WAVE_COUNT(&A.counters[FQ_QC_C_RETAINED2], retained == 2);
  FQ_WAVE_COUNT(&A.counters[FQ_QC_C_FAILED1], failed == 1);
  FQ_WAVE_COUNT(&A.counters[FQ_QC_C_FAILED2], failed == 2);
}
FQ_HD void fq_qc_pair_thread(const FqQcArgs &A, int sp) {
  if (A.shard) A.dup_key[sp] = FQ_QC_DUP_EMPTY;
  FqTxt o; o.dst = nullptr; o.at = 0;
  fq_qc_pair_core(A, sp, true, o);
}
FQ_HD void fq_qc_ist_fill_thread(const FqQcArgs &A, int sp) {
  if (!A.ist_len[sp]) return;
  FqTxt o; o.dst = A.ist_text + A.ist_off[sp]; o.at = 0;
  fq_qc_pair_core(A, sp, false, o);
}
FQ_HD void fq_qc_pile_fill_thread(const FqQcArgs &A, int idx) {
  if (!A.pt_cnt[idx]) return;
  const fq_result_t r = A.s.rec[idx];
  int seqid;
  fq_dev_pac2real(A.s.cg, r.pos, (int)(fq_emit_ref_end(r, A.s.cigar) - r.pos), &seqid);
  const int rs = A.g.ctg_g0[seqid] + (int)((int64_t)r.pos - A.s.cg.off[seqid]);
  (void)fq_qc_pile(A, idx, r, seqid, rs, A.pt + A.pt_off[idx]);
}

// The per-base statistics of an added record (UpdateInfoVecAtRegularSite / StatVecDistUpdate, :381-422, :304-317): `nl` lanes share the record,
// lane `lane` takes every nl-th base of a block -- on the device a wavefront, so that 64 consecutive positions of the depth tables are one
// atomic instruction; hist: [4][256] counts of this workgroup (LDS), added to the consumer's histograms when the workgroup ends.
#if defined(__HIP_DEVICE_COMPILE__)
#define FQ_HIST_INC(p) atomicAdd((unsigned int *)(p), 1u)
#else
#define FQ_HIST_INC(p) (++*(p))
#endif
FQ_HD void fq_qc_base_record(const FqQcArgs &A, int idx, int lane, int nl, uint32_t *hist) {
  const int seqid = A.added[idx];
  if (seqid < 0) return;
  const int chrom = A.g.ctg_chrom[seqid];
  if (chrom < 0) return;
  const int r1 = A.g.chr_reg0[chrom + 1];
  int lo = A.g.ctg_reg_lo[seqid];
  if (lo >= r1) return;
  const fq_result_t p = A.s.rec[idx];
  const int rs = A.g.ctg_g0[seqid] + (int)((int64_t)p.pos - A.s.cg.off[seqid]);
  const uint8_t *row = A.s.seq + (size_t)fq_emit_row(A.s.packed, A.s.n_pairs, A.s.pair_list, idx) * (size_t)A.s.stride;
  const uint8_t *hq = A.s.qual + (size_t)idx * (size_t)A.s.qual_stride;
  const int qsub = (A.s.mode & FQ_MODE_IL13) ? 31 : 0;
  FqBlockIter it = fq_blocks_begin(p, A.s.cigar, rs);
  int abs0, cl, cyc0, on_read, on_ref;
  while (fq_blocks_next(it, &abs0, &cl, &cyc0, &on_read, &on_ref)) {
    while (lo < r1 && A.g.reg_end[lo] < abs0) ++lo;            // first region that ends at or behind the block's first base (blocks come in increasing position)
    // the regions the block touches, in registers (a block of a few hundred bases meets one region, seldom two): what is left is arithmetic per base
    int rs_[3], re_[3], nr = 0;
    uint32_t rb_[3];
    for (int rg = lo; rg < r1 && nr < 3 && A.g.reg_start[rg] < abs0 + cl; ++rg, ++nr) { rs_[nr] = A.g.reg_start[rg]; re_[nr] = A.g.reg_end[rg]; rb_[nr] = A.g.reg_base[rg]; }
    const bool more = nr == 3 && lo + 3 < r1 && A.g.reg_start[lo + 3] < abs0 + cl;      // (more than three: the general walk below)
    // The depth is kept as a DIFFERENCE table (the consumer's pull() sums it up): a read's aligned block inside a flank region is a run of consecutive positions
    // that all count, and costs two atomic adds -- +1 where it begins, -1 behind its end -- instead of one per position.  The Q20 and Q30 depths are kept as what
    // is MISSING from the depth: a base below the threshold counts one in q20 / q30 at its position, and pull() takes depth minus that -- bases of low quality are
    // the few, and an atomic add on a table of millions of positions is what this kernel is bound by.
    auto inreg = [&](int t, uint32_t *k) FQ_LAMBDA_INLINE -> bool {      // is base t of the block inside a flank region?  its table index
      if (t < 0 || t >= cl) return false;
      const int i = abs0 + t;
      bool in = false;
      for (int j = 0; j < nr; ++j) if (i >= rs_[j] && i <= re_[j]) { *k = rb_[j] + (uint32_t)(i - rs_[j]); in = true; }      // RegionList::IsOverlapped
      if (!in && more) {
        int rg = lo + 3;
        while (rg < r1 && A.g.reg_end[rg] < i) ++rg;
        if (rg < r1 && A.g.reg_start[rg] <= i) { *k = A.g.reg_base[rg] + (uint32_t)(i - A.g.reg_start[rg]); in = true; }
      }
      return in;
    };
    for (int t = lane; t < cl; t += nl) {
      uint32_t k = 0, kn = 0;
      if (!inreg(t, &k)) continue;
      const int rr0 = on_read + t;
      const int q = p.strand == 0 ? (int8_t)(hq[rr0] - qsub - 33) : (int8_t)(hq[p.full_len - 1 - rr0] - qsub - 33);
      if (!inreg(t - 1, &kn)) FQ_ATOMIC_INC32(&A.depth[k]);
      if (!inreg(t + 1, &kn)) FQ_ATOMIC_DEC32(&A.depth[k + 1]);
      if (q < 30) {
        FQ_ATOMIC_INC32(&A.q30[k]);
        if (q < 20) FQ_ATOMIC_INC32(&A.q20[k]);
      }
      const int rr = on_read + t, cyc = cyc0 + t * it.sign;
      const int code = p.strand == 0 ? fq_nt4(row[rr]) : fq_comp(fq_nt4(row[p.full_len - 1 - rr]));
      const int rc = fq_pac_base(A.ix.pac, (int64_t)(uint32_t)(p.pos + (uint32_t)(on_ref + t)));   // the reference base under it (what the MD tag spells out)
      FQ_HIST_INC(&hist[0 * 256 + (uint8_t)q]);
      FQ_HIST_INC(&hist[2 * 256 + (uint8_t)cyc]);
      if (code < 4 && rc != code && !A.g.dbsnp[k]) {
        FQ_HIST_INC(&hist[1 * 256 + (uint8_t)q]);
        FQ_HIST_INC(&hist[3 * 256 + (uint8_t)cyc]);
      }
    }
  }
}
enum { FQ_QOP_PAIR = 0, FQ_QOP_IST_FILL, FQ_QOP_PILE_FILL, FQ_QOP_BASE, FQ_QOP_COUNT };
