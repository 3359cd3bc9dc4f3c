// fq_sam.cpp -- consumers of the alignment records: SAM text in the --sam_out dialect of
// bwa_print_sam1 (libbwa/bwase.c:455-581; header bwase.c:593-599 + bwase.h:27-30) and the canonical
// per-stage dump used by the parity tests (same text as oracle/ref_driver.cpp).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_index.h"
#include "fq_kernels.h"
#include "fq_pipeline.h"

void fq_ctx_all_reads(const fq_ctx_t *c, const uint8_t **filtered, const int32_t **len_trim);

namespace {
struct Out {
  std::string s;
  void printf(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n < (int)sizeof buf) s.append(buf, (size_t)n);
    else {
      std::vector<char> big((size_t)n + 1);
      va_start(ap, fmt);
      vsnprintf(big.data(), big.size(), fmt, ap);
      va_end(ap);
      s.append(big.data(), (size_t)n);
    }
  }
  void putc(char ch) { s.push_back(ch); }
};
int64_t emit(const std::string &s, char *buf, int64_t cap) {
  if (buf && cap > (int64_t)s.size()) { memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
  return (int64_t)s.size();
}
void put_cigar(Out &o, const std::vector<uint16_t> &cg) { for (uint16_t x : cg) o.printf("%d%c", x & 0x3fff, "MIDS"[x >> 14]); }

int64_t ref_end(const FqRead &p) {   // pos_end, bwase.c:420-432
  if (!p.cigar.empty()) {
    int64_t x = p.pos;
    for (uint16_t g : p.cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += g & 0x3fff; }
    return x;
  }
  return (int64_t)p.pos + p.len;
}
int64_t ref_end_multi(const FqMulti &q, int len) {
  if (!q.cigar.empty()) {
    int64_t x = q.pos;
    for (uint16_t g : q.cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += g & 0x3fff; }
    return x;
  }
  return (int64_t)q.pos + len;
}
int64_t five_prime(const FqRead &p) { return p.type != FQ_TYPE_NO_MATCH ? (p.strand ? ref_end(p) : (int64_t)p.pos) : -1; }

std::string read_name(const FqHostReads *hb, int pair, int end, bool revived) {
  if (!hb->has_names()) return "*";
  const char *nm = hb->name_of(pair, end);
  std::string s(nm, strnlen(nm, (size_t)hb->name_stride));
  if (revived && hb->mates_named()) {   // expand_seq writes the mate's name over this read's, without a terminator (bwape.c:456)
    const char *qn = hb->name_of(pair, end ^ 1);
    const std::string q(qn, strnlen(qn, (size_t)hb->name_stride));
    s = q.size() >= s.size() ? q : q + s.substr(q.size());
  }
  const size_t t = s.size();
  if (t > 2 && s[t - 2] == '/' && (s[t - 1] == '1' || s[t - 1] == '2')) s.resize(t - 2);   // BwtMapper.cpp:565-570
  return s;
}

// StatCollector::AddAlignment runs before the records are printed (src/BwtMapper.cpp:2047-2050, 2075-2079) and turns a hit that hangs
// over the end of its contig into NO_MATCH (src/StatCollector.cpp:955-971); every consumer sees the record after that.
void bridge_mutation(const fq_index *ix, FqRead &p) {
  if (p.type == FQ_TYPE_NO_MATCH) return;
  int seqid;
  const int j = (int)(ref_end(p) - p.pos);
  fq_coor_pac2real(ix, p.pos, j, &seqid);
  if ((int64_t)p.pos + j - ix->contigs[seqid].offset > ix->contigs[seqid].len) p.type = FQ_TYPE_NO_MATCH;
}

// se: bwa_print_sam1(p, mate = 0), the single-end mapper's call (src/BwtMapper.cpp:1369)
void print_sam(const fq_index *ix, const fq_opts_t *o, const FqHostReads *hb, int n_pairs, Out &out, FqRead p, const FqRead &mate, bool se = false) {
  const int pair = p.r % n_pairs;
  uint8_t seq[FQ_LMAX + 8];
  hb->codes((size_t)p.r, p.full_len, seq);
  const uint8_t *qual = hb->qual((size_t)p.r);
  const std::string name = read_name(hb, pair, p.r / n_pairs, p.revived);
  if (p.type == FQ_TYPE_NO_MATCH && (se || mate.type == FQ_TYPE_NO_MATCH)) {
    // both hits of the pair hung over a contig end: the record of a read without a match (bwase.c:563-579).  It prints p->len bases
    // of p->seq, or of p->rseq when the lost hit was on the reverse strand: the reverse complement of the (trimmed) read, and past a
    // trimmed read's end whatever the reference's slot buffer holds -- code 0 here (not modelled, like the other slot leftovers).
    out.printf("%s\t%d\t*\t0\t0\t*\t*\t0\t0\t", name.c_str(), p.extra_flag | 4 | (se ? 0 : 8));
    for (int j = 0; j < p.len; ++j) {
      int cc = seq[j];
      if (p.strand) { cc = j < p.clip_len ? seq[p.clip_len - 1 - j] : 3; cc = cc < 4 ? 3 - cc : cc; }
      out.putc("ACGTN"[cc > 4 ? 4 : cc]);
    }
    out.putc('\t');
    const int qsub0 = (o->mode & FQ_MODE_IL13) ? 31 : 0;
    for (int j = 0; j < p.full_len; ++j) out.putc((char)(qual[(p.strand && j < p.len) ? p.len - 1 - j : j] - qsub0));   // (no +31 on this branch)
    if (p.clip_len < p.full_len) out.printf("\tXC:i:%d", p.clip_len);
    out.putc('\n');
    return;
  }
  // only called when at least one mate is mapped (both-unmapped pairs are dropped before, BwtMapper.cpp:2038)
  int seqid, nn, am = 0, flag = p.extra_flag, j;
  if (p.type == FQ_TYPE_NO_MATCH) { p.pos = mate.pos; p.strand = mate.strand; flag |= 4; j = 1; }
  else j = (int)(ref_end(p) - p.pos);
  nn = fq_coor_pac2real(ix, p.pos, j, &seqid);
  if (p.type != FQ_TYPE_NO_MATCH && (int64_t)p.pos + j - ix->contigs[seqid].offset > ix->contigs[seqid].len) flag |= 4;
  if (p.strand) flag |= 16;
  if (!se) { if (mate.type != FQ_TYPE_NO_MATCH) { if (mate.strand) flag |= 32; } else flag |= 8; }
  out.printf("%s\t%d\t%s\t%d\t%d\t", name.c_str(), flag, ix->contigs[seqid].name.c_str(), (int)(p.pos - ix->contigs[seqid].offset + 1), p.mapQ);
  if (!p.cigar.empty()) put_cigar(out, p.cigar);
  else if (p.type == FQ_TYPE_NO_MATCH) out.putc('*');
  else out.printf("%dM", p.len);
  if (se) out.s.append("\t*\t0\t0\t");
  else if (mate.type != FQ_TYPE_NO_MATCH) {
    int m_seqid;
    am = mate.seQ < p.seQ ? mate.seQ : p.seQ;
    fq_coor_pac2real(ix, mate.pos, mate.len, &m_seqid);
    out.printf("\t%s\t", seqid == m_seqid ? "=" : ix->contigs[m_seqid].name.c_str());
    long long isize = seqid == m_seqid ? five_prime(mate) - five_prime(p) : 0;
    if (p.type == FQ_TYPE_NO_MATCH) isize = 0;
    out.printf("%d\t%lld\t", (int)(mate.pos - ix->contigs[m_seqid].offset + 1), isize);
  } else out.printf("\t=\t%d\t0\t", (int)(p.pos - ix->contigs[seqid].offset + 1));
  if (p.strand == 0) for (j = 0; j < p.full_len; ++j) out.putc("ACGTN"[seq[j] > 4 ? 4 : seq[j]]);
  else for (j = 0; j < p.full_len; ++j) { const int cc = seq[p.full_len - 1 - j]; out.putc("TGCAN"[cc > 4 ? 4 : cc]); }
  out.putc('\t');
  // Phred+64 input: the reference takes 31 off every quality byte on input (BwtMapper.cpp:549-553) and puts it back on the first
  // len bytes only when it prints (bwase.c:516-519), so the clipped tail of a trimmed read comes out 31 lower than it went in
  const int qsub = (o->mode & FQ_MODE_IL13) ? 31 : 0;
  if (p.strand) { for (j = 0; j < p.len; ++j) out.putc((char)qual[p.len - 1 - j]); for (; j < p.full_len; ++j) out.putc((char)(qual[j] - qsub)); }
  else { for (j = 0; j < p.len; ++j) out.putc((char)qual[j]); for (; j < p.full_len; ++j) out.putc((char)(qual[j] - qsub)); }
  if (p.clip_len < p.full_len) out.printf("\tXC:i:%d", p.clip_len);
  if (p.type != FQ_TYPE_NO_MATCH) {
    char XT = "NURM"[p.type];
    if (nn > 10) XT = 'N';
    out.printf("\tXT:A:%c\t%s:i:%d", XT, (o->mode & FQ_MODE_COMPREAD) ? "NM" : "CM", p.nm);
    if (nn) out.printf("\tXN:i:%d", nn);
    if (!se) out.printf("\tSM:i:%d\tAM:i:%d", p.seQ, am);
    if (p.type != FQ_TYPE_MATESW) { out.printf("\tX0:i:%d", (int)p.c1); if ((int)p.c1 <= o->max_top2) out.printf("\tX1:i:%d", (int)p.c2); }
    out.printf("\tXM:i:%d\tXO:i:%d\tXG:i:%d", p.n_mm, p.n_gapo, p.n_gapo + p.n_gape);
    if (p.has_md) { out.s.append("\tMD:Z:"); out.s.append(p.md); }
    if (!p.multi.empty()) {
      out.s.append("\tXA:Z:");
      for (const FqMulti &q : p.multi) {
        j = (int)(ref_end_multi(q, p.len) - q.pos);
        fq_coor_pac2real(ix, q.pos, j, &seqid);
        out.printf("%s,%c%d,", ix->contigs[seqid].name.c_str(), q.strand ? '-' : '+', (int)(q.pos - ix->contigs[seqid].offset + 1));
        if (!q.cigar.empty()) put_cigar(out, q.cigar); else out.printf("%dM", p.len);
        out.printf(",%d;", q.gap + q.mm);
      }
    }
  }
  out.putc('\n');
}

}  // namespace
std::string fq_read_name(const FqHostReads *hb, int pair, int end, bool revived) { return read_name(hb, pair, end, revived); }
namespace {
void dump_cigar(Out &o, const std::vector<uint16_t> &cg) { if (cg.empty()) o.putc('*'); else put_cigar(o, cg); }
void dump_rec(Out &o, char tag, int end, int idx, const FqRead &p, bool fin) {
  o.printf("%c %d %d type=%d strand=%d pos=%u sa=%u mapQ=%d seQ=%d c1=%d c2=%d flag=%d mm=%d go=%d ge=%d score=%d filt=%d len=%d", tag, end, idx,
           p.type, p.strand, p.pos, p.sa, p.mapQ, p.seQ, (int)p.c1, (int)p.c2, p.extra_flag, p.n_mm, p.n_gapo, p.n_gape, p.score, p.filtered, p.len);
  o.s.append(" cigar=");
  dump_cigar(o, p.cigar);
  if (fin) o.printf(" nm=%d md=%s", p.nm, p.has_md ? p.md.c_str() : "*");
  o.printf(" multi=%d", (int)p.multi.size());
  for (const FqMulti &q : p.multi) { o.printf(" [%u,%d,%d,%d,", q.pos, q.gap, q.mm, q.strand); dump_cigar(o, q.cigar); o.putc(']'); }
  o.putc('\n');
}
}  // namespace

extern "C" int64_t fq_sam_header(const fq_index_t *ix, char *buf, int64_t cap) {
  if (!ix) return FQ_EINVAL;
  Out o;
  for (const auto &cg : ix->contigs) o.printf("@SQ\tSN:%s\tLN:%d\n", cg.name.c_str(), cg.len);
  o.s.append("@PG\tID:FastqA\tPN:FastqA\tVN:0.0.1\n");
  return emit(o.s, buf, cap);
}

extern "C" int64_t fq_sam_format_last(fq_ctx_t *c, char *buf, int64_t cap) {
  if (!c) return FQ_EINVAL;
  const FqBatchState *S = fq_ctx_state(c);
  const fq_index *ix = fq_ctx_index(c);
  const FqHostReads hbv = fq_ctx_host_reads(c), *hb = &hbv;
  const fq_opts_t *o = fq_ctx_opts(c);
  if (S->n_surv > 0 && (!S->rec || !hb->has_qual())) return FQ_EINVAL;     // (no result arrays on the host: FQ_EMIT_DEVICE_ONLY -- fq_sam_device_last has the text)
  // records are independent of each other: ranges of pairs are formatted on several threads and concatenated in order
  auto format_range = [&](int lo, int hi, Out &out) {
  out.s.reserve((size_t)(hi - lo) * 900);
  for (int sp = lo; sp < hi; ++sp) {
    if (S->rec[2 * (size_t)sp].type == FQ_TYPE_NO_MATCH && S->rec[2 * (size_t)sp + 1].type == FQ_TYPE_NO_MATCH) continue;   // src/BwtMapper.cpp:2038-2042
    if (o->single_end) {   // src/BwtMapper.cpp:1355-1370: AddAlignment(p, 0), then bwa_print_sam1(p, 0)
      FqRead a = S->read(2 * (size_t)sp);
      bridge_mutation(ix, a);
      print_sam(ix, o, hb, S->n_pairs, out, a, a, true);
      continue;
    }
    FqRead a = S->read(2 * (size_t)sp), b = S->read(2 * (size_t)sp + 1);
    bridge_mutation(ix, a);
    bridge_mutation(ix, b);
    print_sam(ix, o, hb, S->n_pairs, out, a, b);
    print_sam(ix, o, hb, S->n_pairs, out, b, a);
  }
  };
  const int T = S->n_surv >= 256 ? 8 : 1;
  std::vector<Out> parts((size_t)T);
  if (T == 1) format_range(0, S->n_surv, parts[0]);
  else {
    std::vector<std::thread> th;
    const int per = (S->n_surv + T - 1) / T;
    for (int t = 0; t < T; ++t) { const int lo = t * per, hi = std::min(S->n_surv, lo + per); if (lo < hi) th.emplace_back(format_range, lo, hi, std::ref(parts[(size_t)t])); }
    for (auto &x : th) x.join();
  }
  if (T == 1) return emit(parts[0].s, buf, cap);
  std::string all;
  size_t total = 0;
  for (auto &p : parts) total += p.s.size();
  all.reserve(total);
  for (auto &p : parts) all += p.s;
  return emit(all, buf, cap);
}

extern "C" int64_t fq_stage_dump_last(fq_ctx_t *c, char *buf, int64_t cap) {
  if (!c) return FQ_EINVAL;
  const FqBatchState *S = fq_ctx_state(c);
  const FqHostReads hbv = fq_ctx_host_reads(c), *hb = &hbv;
  const uint8_t *filt; const int32_t *ltrim;
  fq_ctx_all_reads(c, &filt, &ltrim);
  if (S->n_pairs > 0 && (!filt || !ltrim)) return FQ_EINVAL;   // per-read arrays of the whole batch are only fetched in debug mode
  if (S->n_surv > 0 && !S->rec) return FQ_EINVAL;
  const int n = S->n_pairs;
  const fq_opts_t *opts = fq_ctx_opts(c);
  const int n_ends = opts->single_end ? 1 : 2;   // the single-end mapper's dump has one end, no insert-size line and no mate-rescue stage
  Out o;
  std::vector<int> surv_of(n, -1);
  for (int sp = 0; sp < S->n_surv; ++sp) surv_of[S->pair_idx[sp]] = sp;
  const int Bp = S->batch_pairs > 0 ? S->batch_pairs : n;
  const int n_sub = n ? (n + Bp - 1) / Bp : 0;
  for (int sb = 0; sb < n_sub; ++sb) {
    const int i0 = sb * Bp, i1 = std::min(n, i0 + Bp);
    o.printf("B %d %d\n", sb, i1 - i0);
    for (int e = 0; e < n_ends; ++e)
      for (int i = i0; i < i1; ++i) o.printf("F %d %d filt=%d len=%d clip=%d full=%d\n", e, i - i0, filt[e * n + i], ltrim[e * n + i], ltrim[e * n + i], hb->len((size_t)e * n + i));
    for (int e = 0; e < n_ends; ++e)
      for (int i = i0; i < i1; ++i) {
        const int sp = surv_of[i];
        const int s = sp < 0 ? -1 : S->s_of(2 * (size_t)sp + e);
        const int na = s < 0 ? 0 : (int)S->aln_n[s];
        o.printf("A %d %d n=%d", e, i - i0, na);
        for (int k = 0; k < na; ++k) {
          const FqAln &a = S->aln[S->aln_off[s] + k];
          o.printf(" %d,%d,%d,%d,%u,%u,%d", a.info & 0xff, (a.info >> 8) & 0xff, (a.info >> 16) & 0xff, (a.info >> 24) & 1, a.k, a.l, a.score);
        }
        o.putc('\n');
      }
    const fq_isize_t &ii = S->isize_sub[sb];
    uint64_t a, s, p;
    memcpy(&a, &ii.avg, 8); memcpy(&s, &ii.std, 8); memcpy(&p, &ii.ap_prior, 8);
    if (n_ends == 2)
      o.printf("I avg=%016llx std=%016llx ap=%016llx low=%u high=%u hb=%u\n", (unsigned long long)a, (unsigned long long)s, (unsigned long long)p,
               ii.low, ii.high, ii.high_bayesian);
    const std::vector<FqRead> *stages[3] = {&S->stage_P, &S->stage_S, nullptr};   // (the final records: from the C-ABI arrays)
    const char tags[3] = {'P', 'S', 'R'};
    for (int st = 0; st < 3; ++st) {
      if (stages[st] && stages[st]->size() != (size_t)S->n_surv * 2) continue;   // snapshots are only kept in debug mode
      if (n_ends == 1 && st == 1) continue;
      for (int e = 0; e < n_ends; ++e)
        for (int i = i0; i < i1; ++i) {
          const int sp = surv_of[i];
          if (sp >= 0) { dump_rec(o, tags[st], e, i - i0, stages[st] ? (*stages[st])[2 * (size_t)sp + e] : S->read(2 * (size_t)sp + e), st == 2); continue; }
          FqRead d;   // both mates filtered: untouched record (bwa_clean_read_seq state + flags of BwtMapper.cpp:749)
          d.filtered = 1; d.extra_flag = n_ends == 1 ? 0 : (1 | (e == 0 ? 64 : 128)); d.len = ltrim[e * n + i]; d.full_len = hb->len((size_t)e * n + i);
          if (st == 2 && d.len != d.full_len) {   // bwa_correct_trimmed touches every record
            d.cigar.push_back((uint16_t)(FQ_OP_M << 14 | d.len)); d.cigar.push_back((uint16_t)(FQ_OP_S << 14 | (d.full_len - d.len))); d.len = d.full_len;
          }
          dump_rec(o, tags[st], e, i - i0, d, st == 2);
        }
    }
  }
  return emit(o.s, buf, cap);
}
