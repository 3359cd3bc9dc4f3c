// fq_qc.cpp -- the QC consumer of the alignment records: what StatCollector does with every pair that
// BwtMapper::PairEndMapper hands it (src/BwtMapper.cpp:2047-2050 / 2075-2079) and the files its ProcessCore writes
// (src/StatCollector.cpp:2012-2028).  Restated from the reference's behaviour, function by function:
//   RestoreVcfSites      src/StatCollector.cpp:1742-1839   marker / dbSNP / GC tables, flank regions
//   AddAlignment         :950-1101     contig-end un-mapping (Q10), which mates are added, sex-chromosome counters
//   AddSingleAlignment   :424-620      per-base depth / quality / cycle statistics and marker pileups
//   ProcessPairStatus    :623-921      one .InsertSizeTable line per pair, insert sizes, PCR duplicates
//   GetDepthDist .. SummaryOutput  :1858-2028, 2343-2483   the output files
// Containers whose iteration order reaches an output (.SexChromInfo: an unordered_map walked in bucket order) are the same
// standard containers, filled in the same order, so the files come out byte for byte; the others are order-free sums.
//   GetInsertSizeDist    :1969-2002 + src/InsertSizeEstimator.cpp:43-173   the censoring-adjusted insert-size density
//   GetVCF               :2113-2155, 2185-2275   genotype likelihoods of the marker pileups (float arithmetic as there)
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/fastquick_amd.h"
#include "fq_index.h"
#include "fq_kernels.h"
#include "fq_pipeline.h"
#include "fq_pool.h"
#include "fq_backend.h"
#include <mutex>
#include <condition_variable>

namespace {
const int kInsertLimit = 4096;                 // INSERT_SIZE_LIMIT
const char kSign[2] = {1, -1};

std::string chrom_key(std::string chr) {       // upper case, a name containing "CHR" loses its first three characters (:1760-1764)
  std::transform(chr.begin(), chr.end(), chr.begin(), ::toupper);
  if (chr.find("CHR") != std::string::npos) chr = chr.substr(3);
  return chr;
}

// RegionList (src/RegionList.cpp): start -> end per chromosome, inclusive
struct Regions {
  std::map<std::string, std::map<int, int>> list;
  uint64_t length = 0;
  void add(const std::string &chr, int start, int end) { list[chrom_key(chr)][start] = end; }
  // Union of the regions of each chromosome (what RegionList::Collapse leaves, src/RegionList.cpp:73-113): one sweep over the
  // regions in order of their start, a region that begins inside or at the end of the current one extends it (regions that merely
  // touch -- end + 1 == next start -- stay apart, as there)
  void collapse() {
    length = 0;
    for (auto &chr : list) {
      std::map<int, int> merged;
      bool open = false;
      int lo = 0, hi = 0;
      for (const auto &reg : chr.second) {
        if (open && reg.first <= hi) { hi = std::max(hi, reg.second); continue; }
        if (open) merged[lo] = hi;
        lo = reg.first; hi = reg.second; open = true;
      }
      if (open) merged[lo] = hi;
      chr.second.swap(merged);
      for (const auto &reg : chr.second) length += (uint64_t)(reg.second - reg.first) + 1;
    }
  }
  bool overlapped(const std::string &chr, int pos) const {   // RegionList::IsOverlapped, :48-66
    auto it = list.find(chr);
    if (it == list.end()) return false;
    auto lo = it->second.lower_bound(pos);
    if (lo != it->second.end() && lo->first <= pos && lo->second >= pos) return true;
    if (lo != it->second.begin()) { --lo; if (lo->first <= pos && lo->second >= pos) return true; }
    return false;
  }
};

struct ContigStatus { int overlapped = 0, fully = 0, pair_overlapped = 0, fully_paired = 0; };

struct FileStat {                               // FileStatCollector, src/StatCollector.h:46-62
  long long NumRead = 0, NumBase = 0, TotalFiltered = 0, BwaUnmapped = 0, TotalMAPQ = 0, TotalRetained = 0;
  std::string f1, f2;
};

// a record as AddAlignment sees it: the bwa_seq_t fields it reads
struct Rec {
  const FqRead *r = nullptr;
  int type = 0;
  std::string name;
  int64_t end() const {                         // pos_end, libbwa/bwase.c:420-432
    if (!r->cigar.empty()) { int64_t x = r->pos; for (uint16_t g : r->cigar) { const int op = g >> 14; if (op == FQ_OP_M || op == FQ_OP_D) x += g & 0x3fff; } return x; }
    return (int64_t)r->pos + r->len;
  }
};

std::string cigar_string(const FqRead &p) {     // Cigar2String, :58-72
  std::string s;
  for (uint16_t g : p.cigar) { s += std::to_string(g & 0x3fff); s.push_back("MIDS"[g >> 14]); }
  if (p.cigar.empty()) s = std::to_string(p.len) + "M";
  return s;
}

// The reference bases under a read's aligned blocks, from the read, its CIGAR and its MD tag (what RecoverRefseqByMDandCigar
// returns, src/StatCollector.cpp:98-205): the bases of the M blocks in order, then the MD tag walked left to right -- a number
// skips that many matching bases, a letter replaces the base it stands for, '^' + letters puts deleted reference bases back in.
// An MD tag that is one number equal to the read's length means "the read is the reference".
std::string recover_ref(const std::string &read, std::string md, const std::vector<uint16_t> &cigar) {
  for (char &ch : md) ch = (char)toupper((unsigned char)ch);
  if (md.find_first_of("ATCGN") == std::string::npos && atol(md.c_str()) == (long)read.size()) return read;
  std::string ref;
  if (cigar.empty()) ref = read;
  else {
    size_t on_read = 0;
    for (uint16_t g : cigar) {
      const size_t n = g & 0x3fff;
      const int op = g >> 14;
      if (op == FQ_OP_M) ref.append(read, on_read, n);
      if (op != FQ_OP_D) on_read += n;
    }
  }
  size_t at = 0;                       // bases of `ref` the tag has accounted for
  for (size_t i = 0; i < md.size();) {
    size_t run = 0;
    while (i < md.size() && isdigit((unsigned char)md[i])) run = run * 10 + (size_t)(md[i++] - '0');
    if (i >= md.size()) break;
    at += run;
    if (md[i] == '^') {
      size_t j = ++i;
      while (j < md.size() && !isdigit((unsigned char)md[j])) ++j;
      ref.insert(std::min(at, ref.size()), md, i, j - i);
      at += j - i;
      i = j;
    } else {
      if (at < ref.size()) ref[at] = md[i];
      ++at; ++i;
    }
  }
  return ref;
}
}  // namespace

struct fq_qc {
  const fq_index *ix = nullptr;
  fq_qc_opts_t o{};
  std::string err;
  // RestoreVcfSites
  struct Marker { std::string chrom_raw, id, ref, alt, qual, filter, info; int pos = 0; };
  std::vector<Marker> markers;
  std::map<std::string, std::map<int, unsigned>> vcf_table;
  std::unordered_map<std::string, std::unordered_map<int, unsigned>> dbsnp;
  // GC content per position: the reference keeps a hash of every position of every marker's window (a later marker's window writes over an
  // earlier one's where they overlap) and reads it at the positions of the flank regions only (GetDepthDist; a position no window covers
  // reads 0): kept here as one byte per flank-region position, laid out like the depth tables
  std::vector<uint8_t> gc_flat;
  Regions flank;
  uint64_t NumXorY = 0, NumShort = 0, NumLong = 0;
  // statistics
  uint64_t NumPCRDup = 0, NumPairReads = 0, NumBaseMapped = 0, NumCov = 0, NumCov2 = 0, NumCov5 = 0, NumCov10 = 0, total_region_size = 0;
  // Depth tables are dense over the flank regions (PositionTable of the reference: a hash of the positions touched so far, whose
  // members all have depth >= 1 -- so the untouched entries of a dense table are exactly its non-members, and every use is a sum)
  std::map<std::string, std::map<int, std::pair<int, size_t>>> flank_idx;   // chrom -> region start -> (region end, first table index)
  std::vector<uint32_t> depth, q20, q30;
  std::vector<std::string> seq_vec, qual_vec;
  std::vector<std::vector<int>> cycle_vec;
  std::vector<std::vector<unsigned char>> maq_vec;
  std::vector<std::vector<bool>> strand_vec;
  std::vector<size_t> DepthDist = std::vector<size_t>(1024, 0), CycleDist = std::vector<size_t>(512, 0), GCDist = std::vector<size_t>(256, 0),
                      PosNum = std::vector<size_t>(101, 0), EmpRep = std::vector<size_t>(256, 0), misEmpRep = std::vector<size_t>(256, 0),
                      EmpCycle = std::vector<size_t>(256, 0), misEmpCycle = std::vector<size_t>(256, 0), InsertDist = std::vector<size_t>(kInsertLimit, 0);
  std::unordered_set<std::string> dup_table;
  std::unordered_map<std::string, ContigStatus> contig_status;
  std::vector<FileStat> files;
  FileStat cur;
  bool file_open = false;
  // ---- shard consumers (fq_qc_state_reset / _export / fq_qc_merge): what the order-dependent outputs need to be put together again
  bool shard = false;                              // logs below are kept since the last reset
  bool cur_continues = false;                      // the open file was begun before the last reset (its counters here are a part of it)
  std::vector<char> files_continue;                // per closed file of this segment: the same
  std::vector<std::string> contig_order;           // sex-chromosome contigs in the order they were first counted (the file walks an unordered_map)
  std::vector<std::string> dup_log;                // key of every proper pair, in order (the duplicate count needs the keys of all shards)
  std::streamoff table_mark = 0;                   // .InsertSizeTable bytes written before the last reset
  ContigStatus &cs(const std::string &name) {
    auto it = contig_status.find(name);
    if (it == contig_status.end()) { if (shard) contig_order.push_back(name); it = contig_status.emplace(name, ContigStatus()).first; }
    return it->second;
  }
  std::string out_prefix;
  std::ofstream table;
  // ---- the consumer's tables on the device (fq_emit.h): set up by the first call of a context this consumer is attached to (fq_ctx_attach_qc);
  //      the sums stay there until pull() adds them to the tables above (fq_qc_write, fq_qc_state_export)
  std::mutex dev_mu;
  bool dev_on = false;
  bool device_adds = false, host_adds = false;    // a consumer counts its records on one side only (the duplicate set lives there)
  fqdev::State *dev = nullptr;                     // this object's own streams: pull / reset from any thread
  std::vector<void *> d_bufs;                      // geometry
  FqQcGeom geom{};
  size_t table_size = 0, n_contigs = 0;
  uint32_t *d_depth = nullptr, *d_q20 = nullptr, *d_q30 = nullptr, *d_sex_cnt = nullptr;
  uint64_t *d_hist = nullptr, *d_insert = nullptr, *d_sex_first = nullptr, *d_dup = nullptr, *d_est = nullptr;
  // what InsertSizeEstimator would read back from the .InsertSizeTable (counted by the device as the lines were decided): valid while every line of
  // the table came from the device path of THIS consumer (no merged segment, no host-side batch)
  std::vector<uint64_t> est_hist = std::vector<uint64_t>(4 * (size_t)kInsertLimit, 0);
  bool est_valid = true;
  uint64_t dup_cap = 0, ord_next = 0;
  // calls of several contexts that share this consumer may overlap (a stream pipelined over two contexts): their counting steps -- which place
  // the call's pairs in the input's order and may grow the duplicate set -- run one at a time, in the order of the calls' order-dependent parts
  std::mutex gate_mu;
  std::condition_variable gate_cv;
  uint64_t tickets = 0, serving = 0;
  int device_setup();
  int pull();                                      // device sums -> host tables; the device tables start again from zero
  int bind_own();
  ~fq_qc() { if (dev_on || dev) { if (!bind_own()) { for (void *b : d_bufs) fqdev::dfree(b); for (void *b : {(void *)d_depth, (void *)d_q20, (void *)d_q30, (void *)d_sex_cnt, (void *)d_hist, (void *)d_insert, (void *)d_sex_first, (void *)d_dup, (void *)d_est}) fqdev::dfree(b); } if (dev) fqdev::state_destroy(dev); } }

  int restore(const std::string &ref_prefix);
  bool add_single(const Rec &p, const FqHostReads &hb);
  struct MateSpan;
  MateSpan mate_span(const Rec *R, bool placed) const;
  int pair_status(const Rec *p, const Rec *q, int type);
  struct ReadGeom;
  struct StatHist { size_t EmpRep[256] = {}, EmpCycle[256] = {}, misEmpRep[256] = {}, misEmpCycle[256] = {}; };
  std::vector<const FqRead *> stat_jobs;   // reads whose per-base statistics are still to be added (valid until the batch's records go away)
  FqWorkPool pool;                         // ... and the workers they are spread over
  bool read_geom(const Rec &P, ReadGeom &g) const;
  void read_stats(const FqRead &p, const FqHostReads &hb, StatHist &h);
  void run_stat_jobs(const FqHostReads &hb, int threads);
  int add_alignment(Rec &p, Rec &q, const FqHostReads &hb, long long &total_add_failed);
  int add_alignment_se(Rec &p, const FqHostReads &hb, long long &total_add_failed);   // AddAlignment(p, q = 0): the single-end mapper's call
  const char *contig_name(int seqid) const { return ix->contigs[seqid].name.c_str(); }
};

int fq_qc::restore(const std::string &ref_prefix) {
  std::ifstream vcf(ref_prefix + ".SelectedSite.vcf"), gcf(ref_prefix + ".gc", std::ios_base::binary), db(ref_prefix + ".dbSNP.subset.vcf");
  if (!vcf.is_open() || !gcf.is_open() || !db.is_open()) { err = "cannot open " + ref_prefix + ".SelectedSite.vcf / .gc / .dbSNP.subset.vcf"; return FQ_EIO; }
  const int chopped = (int)std::floor(o.read_len * 0.65f + 0.5);   // FLANK_EDGE, :28, :1754
  struct GcWin { std::string chr; int start; uint32_t len; size_t at; };
  std::vector<GcWin> gc_win;
  std::vector<uint8_t> gc_bytes;
  std::string line;
  while (std::getline(vcf, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::stringstream ss(line);
    Marker m;
    std::string pos;
    std::getline(ss, m.chrom_raw, '\t'); std::getline(ss, pos, '\t'); std::getline(ss, m.id, '\t'); std::getline(ss, m.ref, '\t');
    std::getline(ss, m.alt, '\t'); std::getline(ss, m.qual, '\t'); std::getline(ss, m.filter, '\t'); std::getline(ss, m.info, '\t');
    m.pos = atoi(pos.c_str());
    markers.push_back(m);
    const std::string chr = chrom_key(m.chrom_raw);
    vcf_table[chr][m.pos] = (unsigned)markers.size() - 1;
    uint32_t glen = 0;
    gcf.read(reinterpret_cast<char *>(&glen), 4);
    const size_t g_at = gc_bytes.size();
    gc_bytes.resize(g_at + glen);
    gcf.read(reinterpret_cast<char *>(gc_bytes.data() + g_at), glen);
    if (!gcf) { err = "short .gc file"; return FQ_EIO; }
    gc_win.push_back(GcWin{chr, m.pos - ((int)glen - 1) / 2, glen, g_at});
    if (chr == "X" || chr == "Y") { ++NumXorY; flank.add(chr, m.pos - o.flank_len + chopped, m.pos + o.flank_len - chopped); }
    else if (!m.id.empty() && m.id.back() == 'L') { ++NumLong; flank.add(chr, m.pos - o.flank_long_len + chopped, m.pos + o.flank_long_len - chopped); }
    else { ++NumShort; flank.add(chr, m.pos - o.flank_len + chopped, m.pos + o.flank_len - chopped); }
    seq_vec.emplace_back(""); qual_vec.emplace_back(""); cycle_vec.emplace_back(0); maq_vec.emplace_back(0); strand_vec.emplace_back(0);
  }
  flank.collapse();
  {
    size_t at = 0;
    for (const auto &kv : flank.list) for (const auto &r : kv.second) { flank_idx[kv.first][r.first] = std::make_pair(r.second, at); at += (size_t)(r.second - r.first) + 1; }
    depth.assign(at, 0); q20.assign(at, 0); q30.assign(at, 0);
    gc_flat.assign(at, 0);
  }
  for (const GcWin &w : gc_win) {           // in marker order: the later window's values stand where two overlap
    auto fc = flank_idx.find(w.chr);
    if (fc == flank_idx.end() || w.len == 0) continue;
    const int64_t lo = w.start, hi = (int64_t)w.start + w.len - 1;
    auto it = fc->second.upper_bound((int)std::min<int64_t>(lo, INT32_MAX));
    if (it != fc->second.begin()) --it;
    for (; it != fc->second.end() && it->first <= hi; ++it) {
      const int64_t a = std::max<int64_t>(lo, it->first), b = std::min<int64_t>(hi, it->second.first);
      if (a > b) continue;
      memcpy(gc_flat.data() + it->second.second + (size_t)(a - it->first), gc_bytes.data() + w.at + (size_t)(a - lo), (size_t)(b - a + 1));
    }
  }
  while (std::getline(db, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::stringstream ss(line);
    std::string c, pos;
    std::getline(ss, c, '\t'); std::getline(ss, pos, '\t');
    dbsnp[chrom_key(c)][atoi(pos.c_str())] = 1;
  }
  return FQ_OK;
}

// AddSingleAlignment, :424-620 (the reduced-reference branch: contig names `CHR:POS@REF/ALT[|L]`), in two parts:
//  * what depends on the order of the reads -- the pileup strings of the markers a read covers (UpdateInfoVecAtMarker) -- here,
//    in input order;
//  * the per-base statistics (UpdateInfoVecAtRegularSite / StatVecDistUpdate: depth, Q20 / Q30 depth, quality and cycle
//    histograms, mismatch counts) are sums, so the reads are queued (stat_jobs) and fq_qc_add_last spreads them over threads once
//    the batch has been walked.  A StatCollector pass over an on-target batch was 40x the batch's alignment time.
struct fq_qc::ReadGeom { int seqid, readRealStart; std::string chrom; };
bool fq_qc::read_geom(const Rec &P, ReadGeom &g) const {
  const FqRead &p = *P.r;
  g.seqid = 0;
  fq_coor_pac2real(ix, p.pos, (int)(P.end() - p.pos), &g.seqid);
  const std::string &chrName = ix->contigs[g.seqid].name;
  const int pos = (int)((int64_t)p.pos - ix->contigs[g.seqid].offset + 1);
  const size_t colon = chrName.find(':');
  if (colon == std::string::npos) return false;          // (external alignments are not this path)
  const size_t at = chrName.find('@');
  const int refCoord = (int)strtol(chrName.substr(colon + 1, at - colon + 1).c_str(), nullptr, 10);
  const int fl = chrName[chrName.size() - 1] == 'L' ? o.flank_long_len : o.flank_len;
  g.readRealStart = refCoord - fl + pos - 1;
  g.chrom = chrom_key(chrName.substr(0, colon));         // AddMatchBaseInfo, :362-379
  return true;
}
template <class F> static void for_match_blocks(const FqRead &p, int readRealStart, F f) {   // f(absoluteSite, length, cycle, onRead, onRef) per M block
  int absoluteSite = readRealStart, tmpCycle = p.strand != 0 ? p.full_len - 1 : 0, onRead = 0, onRef = 0;
  if (!p.cigar.empty()) {
    for (uint16_t g : p.cigar) {
      const int cl = g & 0x3fff, op = g >> 14;
      if (op == FQ_OP_M) { f(absoluteSite, cl, tmpCycle, onRead, onRef); absoluteSite += cl; tmpCycle += cl * kSign[p.strand]; onRead += cl; onRef += cl; }
      else if (op == FQ_OP_S) { tmpCycle += cl * kSign[p.strand]; onRead += cl; }
      else if (op == FQ_OP_D) { absoluteSite += cl; onRef += cl; }
      else { tmpCycle += cl * kSign[p.strand]; onRead += cl; }
    }
  } else f(absoluteSite, (int)p.len, tmpCycle, onRead, onRef);
}
bool fq_qc::add_single(const Rec &P, const FqHostReads &hb) {
  const FqRead &p = *P.r;
  if (P.type == FQ_TYPE_NO_MATCH || p.mapQ < 20) return false;
  ReadGeom g;
  if (!read_geom(P, g)) return false;
  const auto vt = vcf_table.find(g.chrom);
  if (vt != vcf_table.end()) {
    uint8_t codes[FQ_LMAX + 8];
    bool have = false;
    const uint8_t *hq = nullptr;
    const int qsub = (o.mode & FQ_MODE_IL13) ? 31 : 0;    // qualities are kept 31 lower in --I mode (src/BwtMapper.cpp:549-553)
    for_match_blocks(p, g.readRealStart, [&](int absoluteSite, int cl, int tmpCycle, int onRead, int) {
      // UpdateInfoVecAtMarker, :339-360: the markers inside [absoluteSite, absoluteSite + cl), in increasing position
      for (auto hit = vt->second.lower_bound(absoluteSite); hit != vt->second.end() && hit->first < absoluteSite + cl; ++hit) {
        if (!have) { hb.codes((size_t)p.r, p.full_len, codes); hq = hb.qual((size_t)p.r); have = true; }
        const int d = hit->first - absoluteSite, cyc = tmpCycle + d * kSign[p.strand], rr = onRead + d;
        const unsigned k = hit->second;
        char base, ql;      // the read in the orientation of the reference (SetSamRecord's strings)
        if (p.strand == 0) { const int cc = codes[rr]; base = "ACGTN"[cc > 4 ? 4 : cc]; ql = (char)(hq[rr] - qsub - 33); }
        else { const int cc = codes[p.full_len - 1 - rr]; base = "TGCAN"[cc > 4 ? 4 : cc]; ql = (char)(hq[p.full_len - 1 - rr] - qsub - 33); }
        seq_vec[k] += base; qual_vec[k] += ql;
        cycle_vec[k].push_back(cyc); maq_vec[k].push_back((unsigned char)(p.mapQ + 33)); strand_vec[k].push_back(p.strand != 0);
      }
    });
  }
  stat_jobs.push_back(P.r);
  return true;
}
// the per-base statistics of one queued read; `h` = this thread's quality / cycle histograms (summed afterwards), the depth tables
// are shared and updated atomically
void fq_qc::read_stats(const FqRead &p, const FqHostReads &hb, StatHist &h) {
  Rec P; P.r = &p; P.type = p.type;
  ReadGeom g;
  if (!read_geom(P, g)) return;
  uint8_t codes[FQ_LMAX + 8];
  hb.codes((size_t)p.r, p.full_len, codes);
  const uint8_t *hq = hb.qual((size_t)p.r);
  const int qsub = (o.mode & FQ_MODE_IL13) ? 31 : 0;
  std::string seq, qual;
  seq.reserve((size_t)p.full_len); qual.reserve((size_t)p.full_len);
  if (p.strand == 0) for (int j = 0; j != p.full_len; ++j) { seq += "ACGTN"[codes[j] > 4 ? 4 : codes[j]]; qual += (char)(hq[j] - qsub - 33); }
  else for (int j = 0; j != p.full_len; ++j) { const int c = codes[p.full_len - 1 - j]; seq += "TGCAN"[c > 4 ? 4 : c]; qual += (char)(hq[p.full_len - 1 - j] - qsub - 33); }
  const std::string refSeq = recover_ref(seq, p.md, p.cigar);
  const auto fl_it = flank_idx.find(g.chrom);
  const auto db_it = dbsnp.find(g.chrom);
  int f_lo = 1, f_hi = 0;          // the region the previous position fell into (consecutive positions mostly share it)
  size_t f_base = 0;
  auto in_flank = [&](int pos) {   // RegionList::IsOverlapped, :48-66
    if (pos >= f_lo && pos <= f_hi) return true;
    if (fl_it == flank_idx.end()) return false;
    auto lo = fl_it->second.lower_bound(pos);
    if (lo != fl_it->second.end() && lo->first <= pos && lo->second.first >= pos) { f_lo = lo->first; f_hi = lo->second.first; f_base = lo->second.second; return true; }
    if (lo != fl_it->second.begin()) { --lo; if (lo->first <= pos && lo->second.first >= pos) { f_lo = lo->first; f_hi = lo->second.first; f_base = lo->second.second; return true; } }
    return false;
  };
  for_match_blocks(p, g.readRealStart, [&](int absoluteSite, int cl, int tmpCycle, int onRead, int onRef) {
    // UpdateInfoVecAtRegularSite, :381-422
    int cyc = tmpCycle, rr = onRead, rf = onRef;
    for (int i = absoluteSite; i != absoluteSite + cl; ++i, cyc += kSign[p.strand], ++rr, ++rf) {
      if (!in_flank(i)) continue;
      const char refBase = rf >= 0 && (size_t)rf < refSeq.size() ? refSeq[rf] : 0, readBase = seq[rr], baseQual = qual[rr];
      const size_t k = f_base + (size_t)(i - f_lo);
      __atomic_fetch_add(&depth[k], 1u, __ATOMIC_RELAXED);
      if (baseQual >= 20) { __atomic_fetch_add(&q20[k], 1u, __ATOMIC_RELAXED); if (baseQual >= 30) __atomic_fetch_add(&q30[k], 1u, __ATOMIC_RELAXED); }
      // StatVecDistUpdate, :304-317
      ++h.EmpRep[(unsigned char)baseQual];
      ++h.EmpCycle[(unsigned char)cyc];
      if (readBase != 'N' && refBase != readBase && refBase != 'N' && (db_it == dbsnp.end() || db_it->second.find(i) == db_it->second.end())) {
        ++h.misEmpRep[(unsigned char)baseQual];
        ++h.misEmpCycle[(unsigned char)cyc];
      }
    }
  });
}
// the queued reads of a batch over `threads` threads
void fq_qc::run_stat_jobs(const FqHostReads &hb, int threads) {
  const size_t n = stat_jobs.size();
  if (!n) return;
  const int T = n >= 512 ? std::max(1, threads) : 1;
  std::vector<StatHist> hist((size_t)T);
  auto work = [&](size_t lo, size_t hi, int t) { for (size_t j = lo; j < hi; ++j) read_stats(*stat_jobs[j], hb, hist[(size_t)t]); };
  if (T == 1) work(0, n, 0);
  else {
    const size_t per = (n + (size_t)T - 1) / (size_t)T;
    pool.run(T, [&](int t) { const size_t lo = (size_t)t * per, hi = std::min(n, lo + per); if (lo < hi) work(lo, hi, t); });
  }
  for (const StatHist &h : hist)
    for (int v = 0; v < 256; ++v) { EmpRep[v] += h.EmpRep[v]; EmpCycle[v] += h.EmpCycle[v]; misEmpRep[v] += h.misEmpRep[v]; misEmpCycle[v] += h.misEmpCycle[v]; }
  stat_jobs.clear();
}

// ---- one .InsertSizeTable line per pair, and what it feeds: the insert-size histogram and the duplicate set -----------------------
// (the semantics are ProcessPairStatus', src/StatCollector.cpp:623-921: the file format, the histogram and the duplicate key are fixed
// by byte-for-byte output; the arrangement below is this file's)
//
// A mate as the table sees it: where its alignment starts and stops on the concatenated reference once the left soft clip is taken
// back (32-bit arithmetic, as the reference's bwtint_t: it wraps for a hit hanging over the start), on which contig, and how much
// room that contig leaves on the side the insert grows to.
struct fq_qc::MateSpan {
  const FqRead *r = nullptr;
  const std::string *name = nullptr;
  int contig = -1, flag = 0, clip_left = 0, clip_right = 0;
  int64_t start = 0, stop = 0, contig_lo = 0, contig_hi = 0;
  std::string cigar;
  bool present() const { return r != nullptr; }
  bool reverse() const { return r->strand != 0; }
  // upper bound of an insert that starts at this forward mate / ends at this reverse mate; -1: the mate itself leaves the contig
  int room() const { return reverse() ? (contig_hi >= stop ? (int)(stop - contig_lo) : -1) : (start >= contig_lo ? (int)(contig_hi - start) : -1); }
  void columns(std::ostream &o, bool placed) const {        // contig, 1-based position, flag, length, CIGAR -- or the row of an absent / unplaced mate
    if (placed) o << "\t" << contig_name << "\t" << (int64_t)r->pos - contig_lo + 1 << "\t" << flag << "\t" << r->len << "\t" << cigar;
    else o << "\t*\t*\t" << flag << "\t" << 0 << "\t*";
  }
  const char *contig_name = "";
};
fq_qc::MateSpan fq_qc::mate_span(const Rec *R, bool placed) const {
  MateSpan m;
  if (!R) return m;
  m.r = R->r; m.name = &R->name;
  m.flag = m.r->extra_flag | (R->type == FQ_TYPE_NO_MATCH ? 4 : 0) | (m.r->strand ? 16 : 0);
  if (!placed) return m;
  fq_coor_pac2real(ix, m.r->pos, (int)(R->end() - m.r->pos), &m.contig);
  m.contig_name = contig_name(m.contig);
  m.contig_lo = ix->contigs[m.contig].offset; m.contig_hi = m.contig_lo + (int64_t)ix->contigs[m.contig].len;
  if (!m.r->cigar.empty()) {
    if ((m.r->cigar.front() >> 14) == FQ_OP_S) m.clip_left = m.r->cigar.front() & 0x3fff;
    if ((m.r->cigar.back() >> 14) == FQ_OP_S) m.clip_right = m.r->cigar.back() & 0x3fff;
  }
  m.start = (int64_t)(uint32_t)(m.r->pos - (uint32_t)m.clip_left);
  m.stop = (int64_t)(uint32_t)(m.r->pos - (uint32_t)m.clip_left + (uint32_t)m.r->len);
  m.cigar = cigar_string(*m.r);
  return m;
}
// type: 0 only the first mate is placed, 1 both, 2 only the second.  Returns 2 when the pair counts as low quality (TotalMAPQ), else 0.
int fq_qc::pair_status(const Rec *P, const Rec *Q, int type) {
  const MateSpan a = mate_span(P, type != 2), b = mate_span(Q, type != 0);
  auto line = [&](const std::string &name, int lim_fwd, int lim_rev, int insert, const char *outcome) {
    table << name << "\t" << lim_fwd << "\t" << lim_rev << "\t" << insert;
    a.columns(table, type != 2);
    b.columns(table, type != 0);
    table << "\t" << outcome << std::endl;
  };
  if (type != 1) {                                  // one mate placed: its own room on its contig is all the table can say
    const MateSpan &m = type == 0 ? a : b;
    if (m.r->mapQ == 0) { line(*m.name, -1, -1, -1, "LowQual"); return 2; }
    const int room = m.room();
    if (room < 0) return 2;                         // (no line either: the reference returns before it writes one)
    line(*m.name, m.reverse() ? -1 : room, m.reverse() ? room : -1, -1, m.reverse() ? "RevOnly" : "FwdOnly");
    return 0;
  }
  // both placed: a pair is a forward mate followed by a reverse mate
  const MateSpan *fwd = nullptr, *rev = nullptr;
  if (!a.reverse() && b.reverse() && a.r->pos < b.r->pos) { fwd = &a; rev = &b; }
  else if (!b.reverse() && a.reverse() && b.r->pos < a.r->pos) { fwd = &b; rev = &a; }
  if (!fwd) { line(*a.name, -1, -1, -1, "NotPair"); return 0; }
  const int lim_fwd = std::min(fwd->room(), kInsertLimit - 1), lim_rev = std::min(rev->room(), kInsertLimit - 1);
  if (a.contig != b.contig) { ++InsertDist[0]; line(*a.name, lim_fwd, lim_rev, -1, "NotPair"); return 0; }
  if (a.r->mapQ == 0 || b.r->mapQ == 0) { line(*a.name, lim_fwd, lim_rev, -1, "LowQual"); return 2; }
  const int start = (int)(uint32_t)fwd->start, end = (int)(uint32_t)rev->stop, insert = end - start;
  const bool proper = lim_fwd != -1 && lim_rev != -1, unclipped = fwd->clip_left == 0 && rev->clip_right == 0;
  if (insert >= 0 && insert < kInsertLimit) ++InsertDist[insert];   // (the reference indexes unchecked; inserts beyond the table are out of its bounds)
  line(*a.name, lim_fwd, lim_rev, insert, proper ? "PropPair" : "PartialPair");
  if (proper && unclipped) {                        // the duplicate key: contig and both outer ends
    char key[1024];
    snprintf(key, sizeof key, "%d:%d:%d", a.contig, start, end);
    if (!dup_table.insert(std::string(key)).second) NumPCRDup += 2;
    NumPairReads += 2;
    if (shard) dup_log.emplace_back(key);
  }
  return 0;
}

// AddAlignment, :950-1101
int fq_qc::add_alignment(Rec &P, Rec &Q, const FqHostReads &hb, long long &failed) {
  int seqid = 0, seqid2 = 0;
  auto bridge = [&](Rec &R, int &id) {
    if (R.type == FQ_TYPE_NO_MATCH) return;
    const int j = (int)(R.end() - R.r->pos);
    fq_coor_pac2real(ix, R.r->pos, j, &id);
    if ((int64_t)R.r->pos + j - ix->contigs[id].offset > ix->contigs[id].len) R.type = FQ_TYPE_NO_MATCH;
  };
  bridge(P, seqid);
  bridge(Q, seqid2);
  auto partial = [](const Rec &R) { for (uint16_t g : R.r->cigar) if ((g >> 14) == FQ_OP_S) return true; return false; };
  auto sex = [](const std::string &n) { return n.find('Y') != std::string::npos || n.find('X') != std::string::npos; };
  const std::string qname = contig_name(seqid2);
  if (P.type == FQ_TYPE_NO_MATCH) {
    if (add_single(Q, hb)) {
      if (sex(qname)) { ++cs(qname).overlapped; if (!partial(Q)) ++cs(qname).fully; }
      pair_status(&P, &Q, 2);
      failed += 1;
      return 1;
    }
    failed += 2;
    return 0;
  }
  const std::string pname = contig_name(seqid);
  if (Q.type == FQ_TYPE_NO_MATCH) {
    if (add_single(P, hb)) {
      if (sex(pname)) { ++cs(pname).overlapped; if (!partial(P)) ++cs(pname).fully; }
      pair_status(&P, &Q, 0);
      failed += 1;
      return 1;
    }
    failed += 2;
    return 0;
  }
  if (partial(P)) {
    if (sex(qname)) {
      if (partial(Q)) ++cs(qname).overlapped;
      else { ++cs(qname).overlapped; ++cs(qname).fully; }
      if (pname == qname) ++cs(qname).pair_overlapped;
      ++cs(pname).overlapped;
    }
  } else if (sex(qname)) {
    if (partial(Q)) { ++cs(qname).overlapped; if (pname == qname) ++cs(qname).pair_overlapped; }
    else {
      ++cs(qname).overlapped; ++cs(qname).fully;
      if (pname == qname) { ++cs(qname).pair_overlapped; ++cs(qname).fully_paired; }
    }
    ++cs(pname).overlapped; ++cs(pname).fully;
  }
  if (pair_status(&P, &Q, 1) != 1 || o.cal_dup) {
    if (add_single(P, hb)) {
      if (add_single(Q, hb)) return 2;
      failed += 1;
      return 1;
    }
    if (add_single(Q, hb)) { failed += 1; return 1; }
    failed += 2;
    return 0;
  }
  failed += 2;
  return 0;
}

// ---- the device side ---------------------------------------------------------------------------------------------------------
int fq_qc::bind_own() {
  if (!dev) dev = fqdev::state_create(ix->device);
  if (!dev || fqdev::bind(dev)) { err = std::string("QC consumer: no device state: ") + fqdev::last_error(); return FQ_ENODEV; }
  return FQ_OK;
}
// RestoreVcfSites' tables flattened for the kernels (FqQcGeom), the sums zeroed.  Runs on the calling context's bound state.
int fq_qc::device_setup() {
  if (dev_on) return FQ_OK;
  const size_t nc = ix->contigs.size();
  n_contigs = nc;
  std::vector<std::string> chroms;                  // the chromosomes of the flank regions (the markers' chromosomes: the same keys)
  std::map<std::string, int> chrom_id;
  for (const auto &kv : flank_idx) { chrom_id[kv.first] = (int)chroms.size(); chroms.push_back(kv.first); }
  for (const auto &kv : vcf_table) if (!chrom_id.count(kv.first)) { chrom_id[kv.first] = (int)chroms.size(); chroms.push_back(kv.first); }
  std::vector<int32_t> ctg_chrom(nc), ctg_g0(nc), ctg_reg_lo(nc, 0), chr_reg0(chroms.size() + 1, 0), reg_start, reg_end, chr_mk0(chroms.size() + 1, 0), mk_pos;
  std::vector<uint8_t> ctg_sex(nc);
  std::vector<uint32_t> reg_base, mk_idx;
  for (size_t i = 0; i < nc; ++i) {
    const std::string &nm = ix->contigs[i].name;
    ctg_sex[i] = nm.find('Y') != std::string::npos || nm.find('X') != std::string::npos;
    const size_t colon = nm.find(':');
    if (colon == std::string::npos) { ctg_chrom[i] = -2; ctg_g0[i] = 0; continue; }
    const size_t at = nm.find('@');
    const int refCoord = (int)strtol(nm.substr(colon + 1, at - colon + 1).c_str(), nullptr, 10);     // read_geom
    const int fl = nm[nm.size() - 1] == 'L' ? o.flank_long_len : o.flank_len;
    ctg_g0[i] = refCoord - fl;
    auto it = chrom_id.find(chrom_key(nm.substr(0, colon)));
    ctg_chrom[i] = it == chrom_id.end() ? -1 : it->second;
  }
  for (size_t c = 0; c < chroms.size(); ++c) {
    chr_reg0[c] = (int32_t)reg_start.size();
    auto fi = flank_idx.find(chroms[c]);
    if (fi != flank_idx.end()) for (const auto &r : fi->second) { reg_start.push_back(r.first); reg_end.push_back(r.second.first); reg_base.push_back((uint32_t)r.second.second); }
    chr_mk0[c] = (int32_t)mk_pos.size();
    auto vi = vcf_table.find(chroms[c]);
    if (vi != vcf_table.end()) for (const auto &m : vi->second) { mk_pos.push_back(m.first); mk_idx.push_back(m.second); }
  }
  chr_reg0[chroms.size()] = (int32_t)reg_start.size(); chr_mk0[chroms.size()] = (int32_t)mk_pos.size();
  for (size_t i = 0; i < nc; ++i) {          // where a read on contig i starts its walk over the flank regions: the first region of its chromosome that ends at or behind the contig's first base
    if (ctg_chrom[i] < 0) continue;
    const int r0 = chr_reg0[(size_t)ctg_chrom[i]], r1 = chr_reg0[(size_t)ctg_chrom[i] + 1];
    const int first = ctg_g0[i];               // (genome coordinate of offset 0 of the contig; a read's bases lie at or behind it)
    ctg_reg_lo[i] = (int32_t)(std::lower_bound(reg_end.begin() + r0, reg_end.begin() + r1, first) - reg_end.begin());
  }
  table_size = depth.size();
  std::vector<uint8_t> db(table_size + 1, 0);
  for (const auto &kv : flank_idx) {                 // the known variant sites that lie in a flank region (the only positions the statistics look at)
    auto di = dbsnp.find(kv.first);
    if (di == dbsnp.end()) continue;
    for (const auto &site : di->second) {
      auto r = kv.second.upper_bound(site.first);
      if (r == kv.second.begin()) continue;
      --r;
      if (site.first >= r->first && site.first <= r->second.first) db[r->second.second + (size_t)(site.first - r->first)] = 1;
    }
  }
  bool ok = true;
  auto up = [&](const void *src, size_t bytes) -> void * {
    void *d = fqdev::dmalloc(bytes ? bytes : 16);
    if (!d) { ok = false; return nullptr; }
    d_bufs.push_back(d);
    if (bytes && fqdev::h2d(d, src, bytes)) ok = false;
    return d;
  };
  geom.ctg_chrom = (const int32_t *)up(ctg_chrom.data(), nc * 4); geom.ctg_g0 = (const int32_t *)up(ctg_g0.data(), nc * 4); geom.ctg_sex = (const uint8_t *)up(ctg_sex.data(), nc);
  geom.ctg_reg_lo = (const int32_t *)up(ctg_reg_lo.data(), nc * 4);
  geom.chr_reg0 = (const int32_t *)up(chr_reg0.data(), chr_reg0.size() * 4); geom.reg_start = (const int32_t *)up(reg_start.data(), reg_start.size() * 4);
  geom.reg_end = (const int32_t *)up(reg_end.data(), reg_end.size() * 4); geom.reg_base = (const uint32_t *)up(reg_base.data(), reg_base.size() * 4);
  geom.chr_mk0 = (const int32_t *)up(chr_mk0.data(), chr_mk0.size() * 4); geom.mk_pos = (const int32_t *)up(mk_pos.data(), mk_pos.size() * 4);
  geom.mk_idx = (const uint32_t *)up(mk_idx.data(), mk_idx.size() * 4); geom.dbsnp = (const uint8_t *)up(db.data(), db.size());
  auto zeroed = [&](size_t bytes, int fill) -> void * {
    void *d = fqdev::dmalloc(bytes ? bytes : 16);
    if (!d) { ok = false; return nullptr; }
    if (fqdev::dfill(d, fill, bytes ? bytes : 16)) ok = false;
    return d;
  };
  d_depth = (uint32_t *)zeroed((table_size + 1) * 4, 0); d_q20 = (uint32_t *)zeroed((table_size + 1) * 4, 0); d_q30 = (uint32_t *)zeroed((table_size + 1) * 4, 0);
  d_hist = (uint64_t *)zeroed(4 * 256 * 8, 0); d_insert = (uint64_t *)zeroed((size_t)kInsertLimit * 8, 0);
  d_sex_cnt = (uint32_t *)zeroed((nc + 1) * 4 * 4, 0); d_sex_first = (uint64_t *)zeroed((nc + 1) * 8, 0xff);
  d_est = (uint64_t *)zeroed(4 * (size_t)kInsertLimit * 8, 0);
  if (!ok || fqdev::sync()) { err = std::string("QC consumer: staging its tables on the device failed: ") + fqdev::last_error(); return FQ_ENODEV; }
  dev_on = true;
  return FQ_OK;
}
// The consumer's part of a call's kernel arguments, on the calling context's bound state: tables, geometry, where this call's pairs stand in the
// input; the duplicate set is grown to hold the call's pairs (calls on contexts that share a consumer must not overlap: the command line's do not).
int fq_qc_device_prepare(fq_qc *q, FqQcArgs *a, int n_surv) {
  std::lock_guard<std::mutex> lk(q->dev_mu);
  if (q->host_adds) { q->err = "QC consumer: records were added on the host before (fq_qc_add_last without fq_ctx_attach_qc): a consumer counts on one side only"; return FQ_EINVAL; }
  int rc = q->device_setup();
  if (rc) return rc;
  q->device_adds = true;
  if (!q->shard) {
    const uint64_t need = 2 * (q->ord_next + (uint64_t)n_surv) + 1024;
    if (need > q->dup_cap) {
      uint64_t cap = 1 << 16;
      while (cap < 2 * need) cap <<= 1;
      uint64_t *tab = (uint64_t *)fqdev::dmalloc(cap * 8);
      if (!tab) { q->err = "QC consumer: out of device memory for the duplicate set"; return FQ_ENOMEM; }
      if (fqdev::dfill(tab, 0xff, cap * 8) || fqdev::launch_dup_rehash(q->d_dup, q->dup_cap, tab, cap - 1) || fqdev::sync()) { fqdev::dfree(tab); q->err = std::string("QC consumer: growing the duplicate set failed: ") + fqdev::last_error(); return FQ_ENODEV; }
      fqdev::dfree(q->d_dup);
      q->d_dup = tab; q->dup_cap = cap;
    }
  }
  a->g = q->geom;
  a->cal_dup = q->o.cal_dup; a->shard = q->shard ? 1 : 0;
  a->ord_base = q->ord_next;
  q->ord_next += (uint64_t)n_surv;
  a->depth = q->d_depth; a->q20 = q->d_q20; a->q30 = q->d_q30; a->hist = q->d_hist; a->insert_dist = q->d_insert;
  a->sex_cnt = q->d_sex_cnt; a->sex_first = q->d_sex_first; a->est_hist = q->d_est;
  a->dup_tab = q->d_dup; a->dup_mask = q->dup_cap ? q->dup_cap - 1 : 0;
  return FQ_OK;
}
uint64_t fq_qc_gate_ticket(fq_qc *q) { std::lock_guard<std::mutex> lk(q->gate_mu); return q->tickets++; }
void fq_qc_gate_enter(fq_qc *q, uint64_t ticket) { std::unique_lock<std::mutex> lk(q->gate_mu); q->gate_cv.wait(lk, [&] { return q->serving == ticket; }); }
void fq_qc_gate_leave(fq_qc *q) { { std::lock_guard<std::mutex> lk(q->gate_mu); ++q->serving; } q->gate_cv.notify_all(); }
// what the device has summed since the last pull, added to the host's tables
int fq_qc::pull() {
  std::lock_guard<std::mutex> lk(dev_mu);
  if (!dev_on) return FQ_OK;
  int rc = bind_own();
  if (rc) return rc;
  const size_t T = table_size, nc = n_contigs;
  std::vector<uint32_t> d(T + 1), a(T + 1), b(T + 1), sc((nc + 1) * 4);
  std::vector<uint64_t> hist(4 * 256), ins((size_t)kInsertLimit), first(nc + 1), est(4 * (size_t)kInsertLimit);
  if (fqdev::d2h(d.data(), d_depth, (T + 1) * 4) || fqdev::d2h(a.data(), d_q20, (T + 1) * 4) || fqdev::d2h(b.data(), d_q30, (T + 1) * 4) || fqdev::d2h(hist.data(), d_hist, hist.size() * 8) ||
      fqdev::d2h(ins.data(), d_insert, ins.size() * 8) || fqdev::d2h(est.data(), d_est, est.size() * 8) ||  fqdev::d2h(sc.data(), d_sex_cnt, sc.size() * 4) || fqdev::d2h(first.data(), d_sex_first, first.size() * 8) || fqdev::sync() ||
      fqdev::dzero(d_depth, (T + 1) * 4) || fqdev::dzero(d_q20, (T + 1) * 4) || fqdev::dzero(d_q30, (T + 1) * 4) || fqdev::dzero(d_hist, hist.size() * 8) || fqdev::dzero(d_insert, ins.size() * 8) || fqdev::dzero(d_est, est.size() * 8) ||
      fqdev::dzero(d_sex_cnt, sc.size() * 4) || fqdev::dfill(d_sex_first, 0xff, first.size() * 8) || fqdev::sync()) {
    err = std::string("QC consumer: reading its tables back failed: ") + fqdev::last_error();
    return FQ_ENODEV;
  }
  {   // the device keeps the depth as a difference table and, for Q20 / Q30, the bases BELOW the threshold per position (fq_qc_base_record): a running sum gives the
      // depth, the depth less those the quality depths
    uint32_t sd = 0;
    for (size_t k = 0; k < T; ++k) { sd += d[k]; depth[k] += sd; q20[k] += sd - a[k]; q30[k] += sd - b[k]; }
  }
  for (int v = 0; v < 256; ++v) { EmpRep[v] += hist[v]; misEmpRep[v] += hist[256 + v]; EmpCycle[v] += hist[512 + v]; misEmpCycle[v] += hist[768 + v]; }
  for (int v = 0; v < kInsertLimit; ++v) InsertDist[v] += ins[v];
  for (size_t v = 0; v < est.size(); ++v) est_hist[v] += est[v];
  std::vector<std::pair<uint64_t, size_t>> order;     // sex-chromosome contigs in the order of their first count
  for (size_t c = 0; c < nc; ++c) if (first[c] != ~0ull) order.emplace_back(first[c], c);
  std::sort(order.begin(), order.end());
  for (const auto &oc : order) {
    ContigStatus &st = cs(ix->contigs[oc.second].name);
    st.overlapped += (int)sc[oc.second * 4]; st.fully += (int)sc[oc.second * 4 + 1]; st.pair_overlapped += (int)sc[oc.second * 4 + 2]; st.fully_paired += (int)sc[oc.second * 4 + 3];
  }
  return FQ_OK;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------------------
extern "C" void fq_qc_default_opts(fq_qc_opts_t *o) {
  memset(o, 0, sizeof *o);
  o->flank_len = 250; o->flank_long_len = 1000; o->read_len = 151; o->cal_dup = 1;   // gap_init_opt, libbwa/bwtaln.c:39-48
}
extern "C" int fq_qc_create(const fq_index_t *ix, const char *ref_prefix, const char *out_prefix, const fq_qc_opts_t *o, fq_qc_t **out) {
  if (!ix || !ref_prefix || !out_prefix || !o || !out) return FQ_EINVAL;
  *out = nullptr;
  fq_qc *q = new fq_qc;
  q->ix = ix; q->o = *o; q->out_prefix = out_prefix;
  const int rc = q->restore(ref_prefix);
  if (rc) { delete q; return rc; }
  q->table.open(q->out_prefix + ".InsertSizeTable");
  if (!q->table.is_open()) { delete q; return FQ_EIO; }
  *out = q;
  return FQ_OK;
}
extern "C" void fq_qc_destroy(fq_qc_t *q) { delete q; }
extern "C" const char *fq_qc_last_error(const fq_qc_t *q) { return q ? q->err.c_str() : "null"; }
extern "C" int fq_qc_begin_file(fq_qc_t *q, const char *fq1, const char *fq2) {
  if (!q || !fq1) return FQ_EINVAL;
  q->cur = FileStat();
  q->cur.f1 = fq1; q->cur.f2 = fq2 ? fq2 : fq1;
  q->file_open = true;
  q->cur_continues = false;
  return FQ_OK;
}
extern "C" int fq_qc_end_file(fq_qc_t *q) {
  if (!q || !q->file_open) return FQ_EINVAL;
  q->files.push_back(q->cur);     // StatCollector::AddFSC
  q->files_continue.push_back(q->cur_continues ? 1 : 0);
  q->file_open = false;
  return FQ_OK;
}

// AddAlignment(p, 0), :950-1000 with q == nullptr
int fq_qc::add_alignment_se(Rec &P, const FqHostReads &hb, long long &failed) {
  int seqid = 0;
  if (P.type != FQ_TYPE_NO_MATCH) {
    const int j = (int)(P.end() - P.r->pos);
    fq_coor_pac2real(ix, P.r->pos, j, &seqid);
    if ((int64_t)P.r->pos + j - ix->contigs[seqid].offset > ix->contigs[seqid].len) P.type = FQ_TYPE_NO_MATCH;
  }
  if (P.type == FQ_TYPE_NO_MATCH) { failed += 2; return 0; }
  auto partial = [](const Rec &R) { for (uint16_t g : R.r->cigar) if ((g >> 14) == FQ_OP_S) return true; return false; };
  const std::string pname = contig_name(seqid);
  if (add_single(P, hb)) {
    if (pname.find('Y') != std::string::npos || pname.find('X') != std::string::npos) { ++cs(pname).overlapped; if (!partial(P)) ++cs(pname).fully; }
    pair_status(&P, nullptr, 0);
    failed += 1;
    return 1;
  }
  failed += 2;
  return 0;
}

// the consumer loop of PairEndMapper over one batch (src/BwtMapper.cpp:2026-2052): counters, then AddAlignment per surviving pair
extern "C" int fq_qc_add_last(fq_qc_t *q, fq_ctx_t *c) {
  if (!q || !c || !q->file_open) return FQ_EINVAL;
  const FqBatchState *S = fq_ctx_state(c);
  const FqHostReads hb = fq_ctx_host_reads(c);
  const fq_opts_t *ao = fq_ctx_opts(c);
  q->o.mode = ao->mode;
  FileStat &F = q->cur;
  F.NumBase += fq_ctx_last_bases(c);
  F.NumRead += (ao->single_end ? 1LL : 2LL) * S->n_pairs;
  F.TotalFiltered += S->n_pairs - S->n_surv;
  if (const FqQcCallOut *D = fq_ctx_qc_out(c)) {
    // the call counted on the device (fq_ctx_attach_qc): the sums are in the consumer's device tables; what depends on the order of the
    // records came back laid out in input order and is appended here
    if (D->owner != q || !D->ready) { q->err = "fq_qc_add_last: the context's last call counted for another consumer, or failed"; return FQ_EINVAL; }
    if (fq_ctx_emit_wait(c)) { q->err = std::string("fq_qc_add_last: ") + fq_ctx_last_error(c); return FQ_ENODEV; }      // (the call only enqueued its kernels)
    F.BwaUnmapped += (long long)D->cnt[FQ_QC_C_UNMAPPED];
    F.TotalRetained += (long long)(D->cnt[FQ_QC_C_RETAINED1] + 2 * D->cnt[FQ_QC_C_RETAINED2]);
    F.TotalMAPQ += (long long)(D->cnt[FQ_QC_C_FAILED1] + 2 * D->cnt[FQ_QC_C_FAILED2]);
    q->NumPairReads += 2 * D->cnt[FQ_QC_C_PROPER];
    q->NumPCRDup += 2 * D->cnt[FQ_QC_C_DUP];
    if (D->ist_bytes && fq_ctx_qc_stream(c, 0, [](void *user, const void *data, int64_t n) -> int { ((fq_qc *)user)->table.write((const char *)data, (std::streamsize)n); return 0; }, q) < 0) {
      q->err = std::string("fq_qc_add_last: fetching the .InsertSizeTable lines failed: ") + fq_ctx_last_error(c);
      return FQ_ENODEV;
    }
    if (D->n_pile && fq_ctx_qc_stream(c, 1, [](void *user, const void *data, int64_t n) -> int {
          fq_qc *Q = (fq_qc *)user;
          const FqPileEntry *e = (const FqPileEntry *)data;
          for (int64_t t = 0; t < n / (int64_t)sizeof(FqPileEntry); ++t, ++e) {
            Q->seq_vec[e->k] += (char)e->base; Q->qual_vec[e->k] += (char)e->qual;
            Q->cycle_vec[e->k].push_back(e->cyc); Q->maq_vec[e->k].push_back(e->maq); Q->strand_vec[e->k].push_back(e->strand != 0);
          }
          return 0;
        }, q) < 0) {
      q->err = std::string("fq_qc_add_last: fetching the pileup entries failed: ") + fq_ctx_last_error(c);
      return FQ_ENODEV;
    }
    if (q->shard && D->dup_key) {
      char key[64];
      for (int sp = 0; sp < D->n_surv; ++sp) {
        const uint64_t k = D->dup_key[sp];
        if (k == FQ_QC_DUP_EMPTY) continue;
        int contig = 0;
        fq_coor_pac2real(q->ix, (int64_t)(k >> 32), 1, &contig);      // (a proper pair's outer ends lie inside its contig)
        snprintf(key, sizeof key, "%d:%d:%d", contig, (int)(uint32_t)(k >> 32), (int)(uint32_t)k);
        q->dup_log.emplace_back(key);
      }
    }
    return FQ_OK;
  }
  if (q->device_adds) { q->err = "fq_qc_add_last: this consumer counts on the device (fq_ctx_attach_qc); a batch of a context without it cannot be mixed in"; return FQ_EINVAL; }
  q->host_adds = true;
  q->est_valid = false;
  if (S->n_surv > 0 && !S->rec) { q->err = "fq_qc_add_last: the call's result arrays were left on the device (FQ_EMIT_DEVICE_ONLY)"; return FQ_EINVAL; }
  if (S->n_surv > 0 && !hb.has_qual()) { q->err = "the batch carries no qualities"; return FQ_EINVAL; }
  // the batch's records in the host's vocabulary, from the C-ABI arrays (all threads); they live until the per-base statistics have run
  std::vector<FqRead> recs((size_t)S->n_surv * 2);
  {
    const size_t n = recs.size();
    const int T = n >= 4096 ? 8 : 1;
    const size_t per = (n + (size_t)T - 1) / (size_t)T;
    auto fill = [&](int t) { const size_t lo = (size_t)t * per, hi = std::min(n, lo + per); for (size_t i = lo; i < hi; ++i) recs[i] = S->read(i); };
    if (T == 1) fill(0); else q->pool.run(T, fill);
  }
  for (int sp = 0; sp < S->n_surv; ++sp) {
    const FqRead &a = recs[2 * (size_t)sp], &b = recs[2 * (size_t)sp + 1];
    if (a.type == FQ_TYPE_NO_MATCH && b.type == FQ_TYPE_NO_MATCH) { ++F.BwaUnmapped; continue; }
    if (ao->single_end) {   // SingleEndMapper's consumer loop, src/BwtMapper.cpp:1355-1370
      Rec P;
      P.r = &a; P.type = a.type; P.name = fq_read_name(&hb, a.r % S->n_pairs, a.r / S->n_pairs, a.revived);
      F.TotalRetained += q->add_alignment_se(P, hb, F.TotalMAPQ);
      continue;
    }
    Rec P, Q;
    P.r = &a; P.type = a.type; P.name = fq_read_name(&hb, a.r % S->n_pairs, a.r / S->n_pairs, a.revived);
    Q.r = &b; Q.type = b.type; Q.name = fq_read_name(&hb, b.r % S->n_pairs, b.r / S->n_pairs, b.revived);
    F.TotalRetained += q->add_alignment(P, Q, hb, F.TotalMAPQ);
  }
  q->run_stat_jobs(hb, 8);
  return FQ_OK;
}

// ProcessCore, :2012-2028
// x + 1.0 + 1.0 + ... (t times), every sum rounded as the loop would round it -- without the loop.  Between two powers of two every x + j (j whole) is a
// multiple of the binade's unit, so those additions are exact and may be done at once; the one addition that crosses into the next binade rounds, and is done
// alone.  (InsertSizeEstimator adds 1.0 per table line to cells that start at 1e-6: tens of millions of dependent additions for a deep run.)
static double add_ones(double x, uint64_t t) {
  while (t) {
    if (!(x >= 1.0) || x >= 4503599627370496.0) { x += 1.0; --t; continue; }      // below 1 (the first addition rounds) or past 2^52 (every one may)
    int e; (void)std::frexp(x, &e);                    // x in [2^(e-1), 2^e)
    const double top = std::ldexp(1.0, e);
    const double room = std::ceil(top - x) - 1.0;      // whole steps that stay below `top` (top - x is exact: same binade)
    const uint64_t j = room < 1.0 ? 0 : (uint64_t)std::min<double>(room, (double)t);
    if (j) { x += (double)j; t -= j; }
    if (t) { x += 1.0; --t; }                          // the crossing
  }
  return x;
}
extern "C" double fq_qc_add_ones(double x, uint64_t t) { return add_ones(x, t); }      // (for the test that holds it to the loop)
namespace {
struct WriteTrace {
  bool on; double t0, t;
  WriteTrace() : on(getenv("FASTQUICK_TRACE") != nullptr), t0(now()), t(t0) {}
  static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return 1e3 * (double)ts.tv_sec + 1e-6 * (double)ts.tv_nsec; }
  void mark(const char *what) { if (!on) return; const double n = now(); fprintf(stderr, "TRACE -   QC files   %8.1f ms  %s\n", n - t, what); t = n; }
};
}  // namespace
extern "C" int fq_qc_write(fq_qc_t *q) {
  if (!q) return FQ_EINVAL;
  WriteTrace wt;
  if (int rc = q->pull()) return rc;
  wt.mark("sums fetched from the device");
  q->table.flush();
  wt.mark(".InsertSizeTable flushed");
  const std::string &pre = q->out_prefix;
  {   // GetDepthDist, :1858-1918
    for (auto &chr : q->flank_idx) {
      for (auto &reg : chr.second)
        for (int pos = reg.first; pos <= reg.second.first; ++pos) {
          const int d = (int)q->depth[reg.second.second + (size_t)(pos - reg.first)];
          if (d == 0) continue;                 // never touched: not in the reference's PositionTable
          q->NumBaseMapped += d;
          ++q->DepthDist[d > 1023 ? 1023 : d];
          const unsigned g = q->gc_flat[reg.second.second + (size_t)(pos - reg.first)];
          q->GCDist[g] += d;
          ++q->PosNum[g];
        }
    }
    for (size_t i = 1; i != q->DepthDist.size(); ++i) {
      q->NumCov += q->DepthDist[i];
      if (i >= 2) q->NumCov2 += q->DepthDist[i];
      if (i >= 5) q->NumCov5 += q->DepthDist[i];
      if (i >= 10) q->NumCov10 += q->DepthDist[i];
    }
    const int chopped = (int)std::floor(q->o.read_len * 0.65f + 0.5);
    q->total_region_size = (uint64_t)(((q->o.flank_len - chopped) * 2 + 1) * (int64_t)q->NumShort + ((q->o.flank_long_len - chopped) * 2 + 1) * (int64_t)q->NumLong +
                                      ((q->o.flank_len - chopped) * 2 + 1) * (int64_t)q->NumXorY);
    std::ofstream f(pre + ".DepthDist");
    q->DepthDist[0] = q->total_region_size - q->NumCov;
    for (uint32_t i = 0; i != q->DepthDist.size(); ++i) f << i << "\t" << q->DepthDist[i] << std::endl;
  }
  {   // GetGCDist, :1920-1937
    std::ofstream f(pre + ".GCDist");
    const double MeanDepth = q->NumBaseMapped / (double)q->NumCov;
    for (uint32_t i = 0; i != 101; ++i) {
      f << i << "\t" << q->GCDist[i] << "\t" << q->PosNum[i] << "\t";
      if (q->PosNum[i] == 0) f << 0; else f << (double(q->GCDist[i]) / q->PosNum[i]) / MeanDepth;
      f << std::endl;
    }
  }
  {   // GetEmpRepDist, :1939-1953
    std::ofstream f(pre + ".EmpRepDist");
    for (uint32_t i = 0; i != q->EmpRep.size(); ++i) {
      f << i << "\t" << q->misEmpRep[i] << "\t" << q->EmpRep[i] << "\t";
      if (q->EmpRep[i] == 0) f << 0; else f << (-10) * log10((double)(q->misEmpRep[i] + 1) / (q->EmpRep[i] + 2));
      f << std::endl;
    }
  }
  {   // GetEmpCycleDist, :1955-1972
    std::ofstream f(pre + ".EmpCycleDist");
    double prevQual = 0;
    for (uint32_t i = 0; i != q->EmpCycle.size(); ++i) {
      const double ph = (-10) * log10((double)(q->misEmpCycle[i] + 1e-6) / (q->EmpCycle[i] + 1e-6));
      f << i + 1 << "\t" << q->misEmpCycle[i] << "\t" << q->EmpCycle[i] << "\t" << (q->misEmpCycle[i] == 0 ? prevQual : ph) << "\t" << q->CycleDist[i] << std::endl;
      if (q->misEmpCycle[i] != 0) prevQual = ph;
    }
  }
  {   // GetInsertSizeDist, :1969-1996: the estimator of src/InsertSizeEstimator.cpp reads the .InsertSizeTable file back, once with the
      // "FwdOnly" pairs left out and once with the "RevOnly" ones (InputInsertSizeTable :43-143), and UpdateWeight (:145-173) turns
      // observed (ObsDist) and censored (MisDist) counts per insert size into a density; the two densities are added.
    const int kLimit = 4096;   // INSERT_LIMIT, InsertSizeEstimator.h
    std::vector<double> fsum(2000, 0.);
    for (int pass = 0; pass < 2; ++pass) {
      const std::string skip = pass == 0 ? "FwdOnly" : "RevOnly";
      std::vector<double> Obs(kLimit, 1e-6), Mis(kLimit, 1e-6);
      int totalPair = 0;
      // every line of the table was decided by the device path, which counted what the reader below would take from it: the same additions of 1.0
      // to the same cells (a cell's value depends on how many there were, not on their order), without reading the table back
      const bool counted = q->est_valid && q->device_adds && !q->shard;
      if (counted) {
        for (int k = 0; k < kLimit; ++k) {
          Obs[k] = add_ones(Obs[k], q->est_hist[k]);
          Mis[k] = add_ones(Mis[k], q->est_hist[(pass == 0 ? 2 : 1) * (size_t)kLimit + k] + q->est_hist[3 * (size_t)kLimit + k]);
          totalPair += (int)(q->est_hist[k] + q->est_hist[(pass == 0 ? 2 : 1) * (size_t)kLimit + k] + q->est_hist[3 * (size_t)kLimit + k]);
        }
      }
      std::ifstream fin(counted ? std::string("/dev/null/none") : pre + ".InsertSizeTable");
      std::string line;
      while (std::getline(fin, line)) {
        std::vector<std::string> v;
        for (size_t start = 0;;) {
          const size_t end = line.find('\t', start);
          v.push_back(line.substr(start, end == std::string::npos ? std::string::npos : end - start));
          if (end == std::string::npos) break;
          start = end + 1;
        }
        if (v.size() < 15) continue;
        int Max = atoi(v[1].c_str()), Max2 = atoi(v[2].c_str()), ObsI = atoi(v[3].c_str());
        const int Flag1 = atoi(v[6].c_str()), Flag2 = atoi(v[11].c_str());
        const std::string &Cigar1 = v[8], &Cigar2 = v[13], &Status = v[14];
        if (Max >= kLimit || Max == -1) Max = kLimit - 1;
        if (Max2 >= kLimit || Max2 == -1) Max2 = kLimit - 1;
        if (ObsI >= kLimit || ObsI == -1) ObsI = kLimit - 1;
        if (Status == "Abnormal" || Status == "LowQual" || Status == "NotPair" || Status == skip) continue;
        else if (Status == "FwdOnly") Mis[Max] += 1.;
        else if (Status == "RevOnly") Mis[Max2] += 1.;
        else if (Status == "PropPair") Obs[ObsI] += 1.;
        else if (Status == "PartialPair") {
          const bool s1 = Cigar1.find('S') != std::string::npos, s2 = Cigar2.find('S') != std::string::npos;
          if (!s1 && s2) Mis[(Flag1 & 16) ? Max2 : Max] += 1.;          // read 1 whole: censored on the side it maps to
          else if (s1 && !s2) Mis[(Flag2 & 16) ? Max2 : Max] += 1.;
          else continue;
        } else continue;   // (the reference exits on an unknown status; its own writer produces none)
        ++totalPair;
      }
      std::vector<double> F(2000, 0.), f(2000, 0.), G(2000, 0.), g(2000, 0.);
      for (int k = 0; k < 2000; ++k) {
        const double m = Mis[k], n = Obs[k];
        if (k != 0) { f[k] = n / (1 - G[k - 1]) * 1 / double(totalPair); F[k] = F[k - 1] + f[k]; }
        else { f[k] = n / double(totalPair); F[k] = f[k]; }
        if (k != 0) { g[k] = m / (1 - F[k]) * 1 / double(totalPair); G[k] = G[k - 1] + g[k]; }
        else { g[k] = m / double(totalPair); G[k] = g[k]; }
      }
      for (int k = 0; k < 2000; ++k) fsum[k] = pass == 0 ? f[k] : (fsum[k] + f[k]);
    }
    std::ofstream f(pre + ".AdjustedInsertSizeDist");
    for (size_t i = 0; i < fsum.size(); ++i) f << i << "\t" << fsum[i] << std::endl;
  }
  wt.mark("depth, GC, cycle tables; insert size estimate");
  {   // GetInsertSizeDist's raw table, :1998-2002
    std::ofstream f(pre + ".RawInsertSizeDist");
    for (uint32_t i = 0; i != q->InsertDist.size(); ++i) f << i << "\t" << q->InsertDist[i] << std::endl;
  }
  {   // GetSexChromInfo, :2004-2015
    std::ofstream f(pre + ".SexChromInfo");
    for (auto &kv : q->contig_status)
      f << kv.first << "\t" << kv.second.overlapped << "\t" << kv.second.fully << "\t" << kv.second.pair_overlapped << "\t" << kv.second.fully_paired << std::endl;
  }
  // the markers in the order the two writers below walk them
  struct Site { const std::string *chrom; int pos; unsigned k; };
  std::vector<Site> sites;
  for (auto &chr : q->vcf_table)
    for (auto &site : chr.second) sites.push_back({&chr.first, (int)site.first, site.second});
  const int WT = (int)std::max<size_t>(1, std::min<size_t>({(size_t)16, (size_t)std::max(1, fq_host_cpus()), sites.size() / 64 + 1}));
  const size_t per_site = (sites.size() + (size_t)WT - 1) / (size_t)WT;
  {   // GetPileup, :2030-2065 (the same bytes; a deep run's file is tens of megabytes of single characters: the markers' lines are put together on several threads,
      // a run of markers each, and written in order)
    FILE *fp = fopen((pre + ".Pileup").c_str(), "wb");
    const int qualoffset = (q->o.mode & FQ_MODE_IL13) ? 64 : 33;
    std::vector<std::string> part((size_t)WT);
    auto work = [&](int t) {
      std::string &ln = part[(size_t)t];
      char num[16];
      const size_t lo = (size_t)t * per_site, hi = std::min(sites.size(), lo + per_site);
      size_t need = 0;
      for (size_t i = lo; i < hi; ++i) need += 48 + q->seq_vec[sites[i].k].size() * 8;
      ln.reserve(need);
      for (size_t i = lo; i < hi; ++i) {
        const unsigned k = sites[i].k;
        if (q->seq_vec[k].empty()) continue;
        ln += *sites[i].chrom; ln += '\t';
        ln += std::to_string(sites[i].pos); ln += "\t.\t";
        ln += std::to_string(q->strand_vec[k].size()); ln += '\t';
        for (uint32_t t2 = 0; t2 != q->strand_vec[k].size(); ++t2) ln += (char)(q->strand_vec[k][t2] ? toupper(q->seq_vec[k][t2]) : tolower(q->seq_vec[k][t2]));
        ln += '\t';
        for (uint32_t t2 = 0; t2 != q->qual_vec[k].size(); ++t2) ln += char(q->qual_vec[k][t2] + qualoffset);
        ln += '\t';
        ln.append((const char *)q->maq_vec[k].data(), q->maq_vec[k].size());
        ln += '\t';
        for (uint32_t t2 = 0; t2 != q->cycle_vec[k].size(); ++t2) {
          int v = q->cycle_vec[k][t2], n = 0;
          if (v < 0) { ln += '-'; v = -v; }
          do { num[n++] = (char)('0' + v % 10); v /= 10; } while (v);
          while (n) ln += num[--n];
          if (t2 != q->cycle_vec[k].size() - 1) ln += ',';
        }
        ln += '\n';
      }
    };
    if (WT == 1) work(0); else q->pool.run(WT, work);
    if (fp) { for (auto &ln : part) if (!ln.empty()) fwrite(ln.data(), 1, ln.size(), fp); fclose(fp); }
  }
  wt.mark(".Pileup");
  {   // SummaryOutput, :2343-2483
    std::ofstream fq(pre + ".FASTQ.csv");
    fq << "FileIndex,PairEnd1,PairEnd2" << std::endl;
    for (size_t i = 0; i != q->files.size(); ++i) {
      for (std::string *s : {&q->files[i].f1, &q->files[i].f2}) { const size_t sl = s->find_last_of("\\/"); if (sl != std::string::npos) s->erase(0, sl + 1); }
      fq << i + 1 << "," << q->files[i].f1 << "," << q->files[i].f2 << "\n";
    }
    fq.close();
    std::ofstream fc(pre + ".Sequence.csv");
    long long total_base = 0, total_reads = 0, total_retained = 0, total_unmapped = 0, total_low = 0;
    fc << "FileIndex,NumOfBases,NumOfReads,NumOfUmappedReads,NumOfLowMAPQReads,NumOfQCPassReads,ReadLength" << std::endl;
    for (size_t i = 0; i != q->files.size(); ++i) {
      const FileStat &F = q->files[i];
      fc << i + 1 << "," << F.NumBase << "," << F.NumRead << "," << F.BwaUnmapped << "," << F.TotalMAPQ << "," << F.TotalRetained << ",";
      fc << ((F.NumRead == 0) ? 0 : (F.NumBase / F.NumRead)) << std::endl;
      total_base += F.NumBase; total_reads += F.NumRead; total_retained += F.TotalRetained; total_unmapped += F.BwaUnmapped; total_low += F.TotalMAPQ;
    }
    const double avgReadLen = std::floor(0.5 + ((total_reads == 0) ? 0 : ((double)total_base / total_reads)));
    fc << "Total," << total_base << "," << total_reads << "," << total_unmapped << "," << total_low << "," << total_retained << ",";
    fc << avgReadLen << std::endl;
    fc.close();
    std::ofstream f(pre + ".Summary");
    f << "Statistics : " << "Value\n";
    const uint64_t ref_genome_size = (uint64_t)q->o.genome_size, ref_N_size = (uint64_t)q->o.genome_n_size;
    const auto report_genome_size = ref_genome_size - ref_N_size;
    const double est = (double)q->NumBaseMapped / avgReadLen * report_genome_size / q->total_region_size;
    f << "Estimated Read Mapping Rate : " << est / total_reads << "\n";
    f << "Estimated Read PCR Duplication Rate : " << q->NumPCRDup / ((double)q->NumPairReads) << "[" << q->NumPCRDup << "/" << (double)q->NumPairReads << "]\n";
    f << "Whole Genome Coverage : " << (double)total_base / ref_genome_size << "[" << total_base << "/" << ref_genome_size << "]\n";
    f << "Expected Read Depth : " << (double)total_base / report_genome_size << "[" << total_base << "/" << report_genome_size << "]\n";
    f << "Estimated Read Depth : ";
    f << ((q->NumCov == 0) ? 0 : q->NumBaseMapped / (double)q->total_region_size) << "[" << q->NumBaseMapped << "/" << q->total_region_size << "]\n";
    f << "Reduced Genome Size : " << q->total_region_size << std::endl;
    f << "Depth 1 or above position fraction : " << q->NumCov / (double)q->total_region_size << std::endl;
    f << "Depth 2 or above position fraction : " << q->NumCov2 / (double)q->total_region_size << std::endl;
    f << "Depth 5 or above position fraction : " << q->NumCov5 / (double)q->total_region_size << std::endl;
    f << "Depth 10 or above position fraction : " << q->NumCov10 / (double)q->total_region_size << std::endl;
    long long s20 = 0, s30 = 0;
    for (uint32_t v : q->q20) s20 += v;
    for (uint32_t v : q->q30) s30 += v;
    f << "Q20 Base Fraction : " << (q->NumBaseMapped == 0 ? 0 : double(s20) / q->NumBaseMapped) << std::endl;
    f << "Q30 Base Fraction : " << (q->NumBaseMapped == 0 ? 0 : double(s30) / q->NumBaseMapped) << std::endl;
    f << "Estimated AvgDepth for Q20 bases : " << double(s20) / q->NumCov << std::endl;
    f << "Estimated AvgDepth for Q30 bases : " << double(s30) / q->NumCov << std::endl;
    auto median_from = [&](size_t lo) -> size_t {
      long long tmp = 0, total = 0;
      for (size_t i = lo; i != q->InsertDist.size(); ++i) total += (long long)q->InsertDist[i];
      for (size_t i = lo; i != q->InsertDist.size(); ++i) { tmp += (long long)q->InsertDist[i]; if (tmp > total / 2) return i; }
      return 0;
    };
    f << "Median Insert Size(>=500bp) : " << median_from(500) << std::endl;
    f << "Median Insert Size(>=300bp) : " << median_from(300) << std::endl;
  }
  {   // GetVCF, :2185-2275 (CalLikelihood :2113-2155).  The arithmetic types are the reference's: float accumulators and priors,
      // pow / the 0.5-minus term in double, every other log10 on a float (std::log10(float)), PHRED = (-10) * log10(x), REV_PHRED =
      // pow(10.0, x / -10.0); the file carries the day it was written in its second line.
    std::ofstream fout(pre + ".vcf");
    std::ofstream &f = fout;
    char day[100] = {0};
    { const time_t now = time(nullptr); strftime(day, sizeof day, "%Y%m%d", localtime(&now)); }
    f << "##fileformat=VCFv4.2\n" << "##fileDate=" << day << "\n" << "##source=VerifyBamID2\n";
    f << "##INFO=<ID=AF,Number=A,Type=Float,Description=\"Allele Frequency, for each ALT allele, in the same order as listed\">\n";
    f << "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n";
    f << "##FORMAT=<ID=GP,Number=1,Type=String,Description=\"Genotype\">\n";
    f << "##FORMAT=<ID=PL,Number=G,Type=Integer,Description=\"Normalized, Phred-scaled likelihoods for genotypes as defined in the VCF specification\">\n";
    f << "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tIntendedSample\n";
    // CalLikelihood's terms per quality value (the pileups keep `char` qualities), evaluated with the reference's types: float seq_error = pow(10.0, q / -10.0);
    // std::log10 of a float expression where the expression is float, of a double where it is double
    struct VcfTerms { float one_minus, third, two_thirds; double half_minus; };     // (0.5 - seq_error / 3 is a double expression: its log10 is added as a double)
    std::vector<VcfTerms> vcf_terms(256);
    for (int v = 0; v < 256; ++v) {
      const char qv = (char)(unsigned char)v;
      float seq_error = pow(10.0, (qv / (-10.0)));
      VcfTerms T;
      T.one_minus = std::log10(1 - seq_error); T.half_minus = std::log10(0.5 - seq_error / 3); T.third = std::log10(seq_error / 3); T.two_thirds = std::log10(2 * seq_error / 3);
      vcf_terms[(size_t)v] = T;
    }
    // (a marker's line depends on its own pileup alone: runs of markers on several threads, each into a stream of its own with the file stream's formatting, written in order)
    std::vector<std::string> part((size_t)WT);
    auto work = [&](int t) {
      std::ostringstream f;
      float alleleFrq = 0.;
      const size_t lo = (size_t)t * per_site, hi = std::min(sites.size(), lo + per_site);
      for (size_t si = lo; si < hi; ++si) {
        const unsigned k = sites[si].k;
        const fq_qc::Marker &m = q->markers[k];
        std::string af;   // value of the INFO key AF
        bool has_af = false;
        for (size_t at = 0; at < m.info.size();) {
          size_t end = m.info.find(';', at);
          if (end == std::string::npos) end = m.info.size();
          if (m.info.compare(at, 3, "AF=") == 0) { af = m.info.substr(at + 3, end - at - 3); has_af = true; break; }
          at = end + 1;
        }
        if (!has_af) continue;
        alleleFrq = (float)atof(af.c_str());
        const std::string &seq = q->seq_vec[k], &qual = q->qual_vec[k];
        if (seq.empty()) continue;
        f << m.chrom_raw << "\t" << m.pos << "\t" << m.id << "\t" << m.ref << "\t" << m.alt << "\t" << m.qual << "\t" << m.filter << "\t";
        f << "AF=" << af << ";AC=" << seq.size() << "\t" << "GT:PL:GP\t";
        const char maj = m.ref[0], mnr = m.alt[0];
        float GL0(0), GL1(0), GL2(0);
        for (uint32_t i = 0; i != seq.size(); ++i) {      // (the four terms depend on the quality value alone: looked up, added in the reference's order)
          const VcfTerms &T = vcf_terms[(unsigned char)qual[i]];
          if (seq[i] == maj) { GL0 += T.one_minus; GL1 += T.half_minus; GL2 += T.third; }
          else if (seq[i] == mnr) { GL0 += T.third; GL1 += T.half_minus; GL2 += T.one_minus; }
          else { GL0 += T.two_thirds; GL1 += T.two_thirds; GL2 += T.two_thirds; }
        }
        float PL[3];
        PL[0] = std::floor(GL0 * (-10) + 0.5); PL[1] = std::floor(GL1 * (-10) + 0.5); PL[2] = std::floor(GL2 * (-10) + 0.5);
        float prior[3], post[3], sum;
        prior[0] = (-10) * std::log10((1 - alleleFrq) * (1 - alleleFrq));
        prior[1] = (-10) * std::log10(2 * alleleFrq * (1 - alleleFrq));
        prior[2] = (-10) * std::log10(alleleFrq * alleleFrq);
        post[0] = prior[0] + PL[0]; post[1] = prior[1] + PL[1]; post[2] = prior[2] + PL[2];
        sum = (-10) * std::log10(pow(10.0, (post[0] / (-10.0))) + pow(10.0, (post[1] / (-10.0))) + pow(10.0, (post[2] / (-10.0))));
        post[0] = std::floor(post[0] - sum + 0.5); post[1] = std::floor(post[1] - sum + 0.5); post[2] = std::floor(post[2] - sum + 0.5);
        const char *gt = post[0] < post[1] ? (post[0] < post[2] ? "0/0:" : "1/1:") : (post[1] < post[2] ? "0/1:" : "1/1:");
        f << gt << PL[0] << "," << PL[1] << "," << PL[2] << ":" << post[0] << "," << post[1] << "," << post[2] << "\n";
      }
      part[(size_t)t] = f.str();
    };
    if (WT == 1) work(0); else q->pool.run(WT, work);
    for (auto &ln : part) fout.write(ln.data(), (std::streamsize)ln.size());
  }
  wt.mark("summaries, .vcf");
  return FQ_OK;
}

// ---- the consumer over several ranks -----------------------------------------------------------------------------------------
// Every rank feeds the records of its shard of the input to a consumer of its own; what those consumers have gathered is put
// together on one of them (fq_qc_merge), in the order of the input, and written there (fq_qc_write).  Most of the state is sums
// (FileStatCollector's counters, src/StatCollector.h:46-62; the depth / quality / cycle / insert-size tables of AddSingleAlignment
// and ProcessPairStatus, src/StatCollector.cpp:424-921): those add.  What depends on the order of the records is kept as a log
// and replayed: the .InsertSizeTable lines and the pileup strings of the markers are appended, the duplicate keys of a shard's
// proper pairs are looked up in the union of all shards seen so far (the count of duplicates among N pairs is N minus the distinct
// keys, whatever the order), and the sex-chromosome contigs are entered in the order they were first counted -- .SexChromInfo walks
// an unordered_map, whose order is that of its insertions.
namespace {
struct Out {
  std::vector<uint8_t> b;
  template <class T> void put(const T &v) { const uint8_t *p = (const uint8_t *)&v; b.insert(b.end(), p, p + sizeof(T)); }
  void bytes(const void *p, size_t n) { put<uint64_t>(n); const uint8_t *q = (const uint8_t *)p; b.insert(b.end(), q, q + n); }
  void str(const std::string &s) { bytes(s.data(), s.size()); }
  template <class T> void vec(const std::vector<T> &v) { bytes(v.data(), v.size() * sizeof(T)); }
};
struct In {
  const uint8_t *p, *end;
  bool ok = true;
  template <class T> T get() { T v{}; if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
  const uint8_t *bytes(size_t *n) { *n = (size_t)get<uint64_t>(); if (!ok || (size_t)(end - p) < *n) { ok = false; *n = 0; return p; } const uint8_t *q = p; p += *n; return q; }
  std::string str() { size_t n; const uint8_t *q = bytes(&n); return std::string((const char *)q, n); }
  template <class T> bool vec(std::vector<T> &v) { size_t n; const uint8_t *q = bytes(&n); if (!ok || n % sizeof(T)) { ok = false; return false; } v.resize(n / sizeof(T)); if (n) memcpy(v.data(), q, n); return true; }
};
const uint64_t kStateMagic = 0x3153435146ull;   // "FQCS1"
void put_file(Out &o, const FileStat &F) { o.put(F.NumRead); o.put(F.NumBase); o.put(F.TotalFiltered); o.put(F.BwaUnmapped); o.put(F.TotalMAPQ); o.put(F.TotalRetained); o.str(F.f1); o.str(F.f2); }
FileStat get_file(In &in) { FileStat F; F.NumRead = in.get<long long>(); F.NumBase = in.get<long long>(); F.TotalFiltered = in.get<long long>(); F.BwaUnmapped = in.get<long long>(); F.TotalMAPQ = in.get<long long>(); F.TotalRetained = in.get<long long>(); F.f1 = in.str(); F.f2 = in.str(); return F; }
void add_file(FileStat &a, const FileStat &b) { a.NumRead += b.NumRead; a.NumBase += b.NumBase; a.TotalFiltered += b.TotalFiltered; a.BwaUnmapped += b.BwaUnmapped; a.TotalMAPQ += b.TotalMAPQ; a.TotalRetained += b.TotalRetained; }
}  // namespace

extern "C" int fq_qc_state_reset(fq_qc_t *q) {
  if (!q) return FQ_EINVAL;
  if (int rc = q->pull()) return rc;       // (what the device holds is dropped with the rest: pulled into the tables that are cleared below)
  q->shard = true;
  std::fill(q->depth.begin(), q->depth.end(), 0u); std::fill(q->q20.begin(), q->q20.end(), 0u); std::fill(q->q30.begin(), q->q30.end(), 0u);
  for (auto *v : {&q->EmpRep, &q->misEmpRep, &q->EmpCycle, &q->misEmpCycle, &q->InsertDist, &q->CycleDist}) std::fill(v->begin(), v->end(), (size_t)0);
  for (size_t k = 0; k < q->seq_vec.size(); ++k) { q->seq_vec[k].clear(); q->qual_vec[k].clear(); q->cycle_vec[k].clear(); q->maq_vec[k].clear(); q->strand_vec[k].clear(); }
  q->NumPCRDup = q->NumPairReads = 0;
  q->dup_log.clear();
  q->contig_status.clear(); q->contig_order.clear();
  q->files.clear(); q->files_continue.clear();
  if (q->file_open) { const std::string f1 = q->cur.f1, f2 = q->cur.f2; q->cur = FileStat(); q->cur.f1 = f1; q->cur.f2 = f2; q->cur_continues = true; }
  q->table.flush();
  q->table_mark = (std::streamoff)q->table.tellp();
  return FQ_OK;
}

extern "C" int64_t fq_qc_state_export(fq_qc_t *q, void *buf, int64_t cap) {
  if (!q) return FQ_EINVAL;
  if (!q->shard) { q->err = "fq_qc_state_export: not a shard consumer (call fq_qc_state_reset before its first batch)"; return FQ_EINVAL; }
  if (int rc = q->pull()) return rc;
  Out o;
  o.put(kStateMagic);
  o.put<int32_t>(q->o.mode);
  o.put<uint64_t>(q->depth.size());
  {   // depth tables: the touched positions (index, depth, Q20, Q30), or the whole tables when most are
    size_t nz = 0;
    for (uint32_t d : q->depth) nz += d != 0;
    const bool dense = nz * 16 > q->depth.size() * 12;
    o.put<uint8_t>(dense ? 1 : 0);
    if (dense) { o.vec(q->depth); o.vec(q->q20); o.vec(q->q30); }
    else {
      std::vector<uint32_t> sp;
      sp.reserve(nz * 4);
      for (size_t k = 0; k < q->depth.size(); ++k) if (q->depth[k]) { sp.push_back((uint32_t)k); sp.push_back(q->depth[k]); sp.push_back(q->q20[k]); sp.push_back(q->q30[k]); }
      o.vec(sp);
    }
  }
  for (auto *v : {&q->EmpRep, &q->misEmpRep, &q->EmpCycle, &q->misEmpCycle, &q->InsertDist, &q->CycleDist}) o.vec(*v);
  {   // pileups: the markers this shard's reads covered
    uint64_t n = 0;
    for (const auto &sq : q->seq_vec) n += !sq.empty();
    o.put(n);
    for (size_t k = 0; k < q->seq_vec.size(); ++k) {
      if (q->seq_vec[k].empty()) continue;
      o.put<uint32_t>((uint32_t)k);
      o.str(q->seq_vec[k]); o.str(q->qual_vec[k]); o.vec(q->cycle_vec[k]); o.vec(q->maq_vec[k]);
      std::vector<uint8_t> st(q->strand_vec[k].begin(), q->strand_vec[k].end());
      o.vec(st);
    }
  }
  o.put<uint64_t>(q->dup_log.size());
  for (const auto &k : q->dup_log) o.str(k);
  o.put<uint64_t>(q->contig_order.size());
  for (const auto &name : q->contig_order) { const ContigStatus &c = q->contig_status[name]; o.str(name); o.put(c.overlapped); o.put(c.fully); o.put(c.pair_overlapped); o.put(c.fully_paired); }
  o.put<uint64_t>(q->files.size());
  for (size_t i = 0; i < q->files.size(); ++i) { o.put<uint8_t>((uint8_t)q->files_continue[i]); put_file(o, q->files[i]); }
  o.put<uint8_t>(q->file_open ? 1 : 0);
  if (q->file_open) { o.put<uint8_t>(q->cur_continues ? 1 : 0); put_file(o, q->cur); }
  {   // the .InsertSizeTable lines of this segment
    q->table.flush();
    std::ifstream fin(q->out_prefix + ".InsertSizeTable", std::ios_base::binary);
    std::string text;
    if (fin.is_open()) { fin.seekg(q->table_mark); text.assign(std::istreambuf_iterator<char>(fin), std::istreambuf_iterator<char>()); }
    o.str(text);
  }
  if (!buf || cap < (int64_t)o.b.size()) return (int64_t)o.b.size();
  memcpy(buf, o.b.data(), o.b.size());
  return (int64_t)o.b.size();
}

extern "C" int fq_qc_merge(fq_qc_t *q, const void *buf, int64_t len) {
  if (!q || !buf || len < 8) return FQ_EINVAL;
  q->est_valid = false;                    // (a merged segment's lines are only in the table file)
  In in{(const uint8_t *)buf, (const uint8_t *)buf + len};
  if (in.get<uint64_t>() != kStateMagic) { q->err = "fq_qc_merge: not an exported consumer state"; return FQ_EINVAL; }
  q->o.mode = in.get<int32_t>();
  if (in.get<uint64_t>() != q->depth.size()) { q->err = "fq_qc_merge: the state belongs to another marker set"; return FQ_EINVAL; }
  if (in.get<uint8_t>()) {
    std::vector<uint32_t> d, a, b;
    if (!in.vec(d) || !in.vec(a) || !in.vec(b) || d.size() != q->depth.size() || a.size() != d.size() || b.size() != d.size()) { q->err = "fq_qc_merge: truncated state"; return FQ_EINVAL; }
    for (size_t k = 0; k < d.size(); ++k) { q->depth[k] += d[k]; q->q20[k] += a[k]; q->q30[k] += b[k]; }
  } else {
    std::vector<uint32_t> sp;
    if (!in.vec(sp) || sp.size() % 4) { q->err = "fq_qc_merge: truncated state"; return FQ_EINVAL; }
    for (size_t t = 0; t < sp.size(); t += 4) { if (sp[t] >= q->depth.size()) { q->err = "fq_qc_merge: corrupt state"; return FQ_EINVAL; } q->depth[sp[t]] += sp[t + 1]; q->q20[sp[t]] += sp[t + 2]; q->q30[sp[t]] += sp[t + 3]; }
  }
  for (auto *v : {&q->EmpRep, &q->misEmpRep, &q->EmpCycle, &q->misEmpCycle, &q->InsertDist, &q->CycleDist}) {
    std::vector<size_t> w;
    if (!in.vec(w) || w.size() != v->size()) { q->err = "fq_qc_merge: truncated state"; return FQ_EINVAL; }
    for (size_t k = 0; k < w.size(); ++k) (*v)[k] += w[k];
  }
  for (uint64_t n = in.get<uint64_t>(); n > 0 && in.ok; --n) {
    const uint32_t k = in.get<uint32_t>();
    const std::string sq = in.str(), ql = in.str();
    std::vector<int> cyc; std::vector<unsigned char> mq; std::vector<uint8_t> st;
    if (!in.vec(cyc) || !in.vec(mq) || !in.vec(st) || k >= q->seq_vec.size()) { q->err = "fq_qc_merge: corrupt state"; return FQ_EINVAL; }
    q->seq_vec[k] += sq; q->qual_vec[k] += ql;
    q->cycle_vec[k].insert(q->cycle_vec[k].end(), cyc.begin(), cyc.end());
    q->maq_vec[k].insert(q->maq_vec[k].end(), mq.begin(), mq.end());
    for (uint8_t v : st) q->strand_vec[k].push_back(v != 0);
  }
  for (uint64_t n = in.get<uint64_t>(); n > 0 && in.ok; --n) {   // the shard's proper pairs against the keys of everything merged so far
    const std::string key = in.str();
    if (!q->dup_table.insert(key).second) q->NumPCRDup += 2;
    q->NumPairReads += 2;
    if (q->shard) q->dup_log.push_back(key);
  }
  for (uint64_t n = in.get<uint64_t>(); n > 0 && in.ok; --n) {
    const std::string name = in.str();
    ContigStatus &c = q->cs(name);
    c.overlapped += in.get<int>(); c.fully += in.get<int>(); c.pair_overlapped += in.get<int>(); c.fully_paired += in.get<int>();
  }
  for (uint64_t n = in.get<uint64_t>(); n > 0 && in.ok; --n) {
    const bool continues = in.get<uint8_t>() != 0;
    const FileStat F = get_file(in);
    if (continues) {
      if (!q->file_open) { q->err = "fq_qc_merge: the state continues a file that is not open here (fq_qc_begin_file)"; return FQ_EINVAL; }
      add_file(q->cur, F);
      q->files.push_back(q->cur); q->files_continue.push_back(q->cur_continues ? 1 : 0);
      q->file_open = false;
    } else { q->files.push_back(F); q->files_continue.push_back(0); }
  }
  if (in.get<uint8_t>()) {
    const bool continues = in.get<uint8_t>() != 0;
    const FileStat F = get_file(in);
    if (continues) {
      if (!q->file_open) { q->err = "fq_qc_merge: the state continues a file that is not open here (fq_qc_begin_file)"; return FQ_EINVAL; }
      add_file(q->cur, F);
    } else { q->cur = F; q->file_open = true; q->cur_continues = false; }
  }
  const std::string text = in.str();
  if (!in.ok) { q->err = "fq_qc_merge: truncated state"; return FQ_EINVAL; }
  q->table.write(text.data(), (std::streamsize)text.size());
  return FQ_OK;
}
