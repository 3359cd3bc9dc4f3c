// fq_cli.cpp -- `FASTQuick_amd align`: the reference's `FASTQuick align --sam_out` command line on top of the C ABI.
//
// Mirrors runAlign (src/FASTQuick.cpp:159-488): same flag names and meanings for the flags the hot path reads, the
// index prefix convention (<index_prefix>.FASTQuick.fa.*), SAM text on stdout in the --sam_out dialect, the summary
// notices on stderr.  The FASTQ tokenizer follows kseq_read3_fpc (libbwa/kseq.h:327-370): name up to the first white
// space, bases = printable characters up to the '+' line, quality = exactly as many characters as bases.
// Not built yet: BAM output with genome-coordinate translation, StatCollector QC files -- asking for them
// is an error, not a silent downgrade.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fastquick_amd.h"

namespace {
[[noreturn]] void die(const std::string &m) { fprintf(stderr, "FATAL ERROR - \n%s\n", m.c_str()); exit(EXIT_FAILURE); }
void notice(const char *fmt, long long a) { fprintf(stderr, "NOTICE - "); fprintf(stderr, fmt, a); fputc('\n', stderr); }

struct FastqReader {
  gzFile fp = nullptr;
  std::vector<unsigned char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  explicit FastqReader(const std::string &path) : buf(1 << 20) {
    fp = gzopen(path.c_str(), "rb");
    if (!fp) die("Open " + path + " failed!");
    gzbuffer(fp, 1 << 20);
  }
  ~FastqReader() { if (fp) gzclose(fp); }
  int getc() {
    if (pos == end) {
      if (eof) return -1;
      const int n = gzread(fp, buf.data(), (unsigned)buf.size());
      if (n <= 0) { eof = true; return -1; }
      pos = 0; end = (size_t)n;
    }
    return buf[pos++];
  }
  // returns false at end of file
  bool next(std::string &name, std::string &seq, std::string &qual) {
    int c;
    while ((c = getc()) != -1 && c != '@' && c != '>') {}
    if (c == -1) return false;
    name.clear(); seq.clear(); qual.clear();
    while ((c = getc()) != -1 && !isspace(c)) name.push_back((char)c);
    if (c != '\n') while ((c = getc()) != -1 && c != '\n') {}
    while ((c = getc()) != -1 && c != '+' && c != '>' && c != '@') if (isgraph(c)) seq.push_back((char)c);
    if (c != '+') die("FASTA input is not supported by align (no quality line for " + name + ")");
    while ((c = getc()) != -1 && c != '\n') {}
    qual.resize(seq.size());
    for (size_t i = 0; i < seq.size(); ++i) { c = getc(); if (c == -1) die("truncated quality string for " + name); qual[i] = (char)c; }
    c = getc();
    if (c != -1 && c != '\n') die("Error:" + name + " this fastq file contains reads with different length");   // kseq.h:362-365
    return true;
  }
};

struct Args {
  std::string fq1, fq2, out_prefix = "Empty", index_prefix = "Empty";
  bool sam_out = false;
  fq_opts_t o;
  int opte = -1;
  long long chunk_pairs = 16LL * 262144;
  int device = 0;
};

int usage() {
  fprintf(stderr, "Usage: FASTQuick_amd align --index_prefix P --fastq_1 R1.fq[.gz] --fastq_2 R2.fq[.gz] --out_prefix O --sam_out\n"
                  "                       [--q INT] [--n FLOAT|INT] [--kmer_thresh INT] [--o INT] [--e INT] [--i INT] [--d INT] [--l INT] [--k INT]\n"
                  "                       [--m INT] [--R INT] [--N] [--L] [--max_isize INT] [--max_occ INT] [--is_sw] [--n_multi INT] [--N_multi INT]\n"
                  "                       [--ap_prior FLOAT] [--force_isize] [--t INT] [--chunk_pairs INT] [--device INT]\n"
                  "       FASTQuick_amd index --ref REDUCED.FASTQuick.fa [--rollhash]\n");
  return 1;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) return usage();
  const std::string cmd = argv[1];
  if (cmd == "index") {
    std::string ref;
    int rollhash = 0;
    for (int i = 2; i < argc; ++i) {
      if (!strcmp(argv[i], "--ref") && i + 1 < argc) ref = argv[++i];
      else if (!strcmp(argv[i], "--rollhash")) rollhash = 1;
      else return usage();
    }
    if (ref.empty()) return usage();
    const int rc = fq_index_build(ref.c_str(), rollhash);
    if (rc) die("fq_index_build failed (" + std::to_string(rc) + ")");
    return 0;
  }
  if (cmd != "align") return usage();
  Args A;
  fq_default_opts(&A.o);
  for (int i = 2; i < argc; ++i) {
    const std::string f = argv[i];
    auto need = [&](const char *) -> const char * { if (i + 1 >= argc) die("missing value for " + f); return argv[++i]; };
    if (f == "--fastq_1") A.fq1 = need("");
    else if (f == "--fastq_2") A.fq2 = need("");
    else if (f == "--out_prefix") A.out_prefix = need("");
    else if (f == "--index_prefix") A.index_prefix = need("");
    else if (f == "--sam_out") A.sam_out = true;
    else if (f == "--kmer_thresh") A.o.filter_thresh = atoi(need(""));
    else if (f == "--n") A.o.fnr = atof(need(""));
    else if (f == "--o") A.o.max_gapo = atoi(need(""));
    else if (f == "--e") A.opte = atoi(need(""));
    else if (f == "--i") A.o.indel_end_skip = atoi(need(""));
    else if (f == "--d") A.o.max_del_occ = atoi(need(""));
    else if (f == "--l") A.o.seed_len = atoi(need(""));
    else if (f == "--k") A.o.max_seed_diff = atoi(need(""));
    else if (f == "--m") A.o.max_entries = atoi(need(""));
    else if (f == "--t") A.o.host_threads = atoi(need(""));   // accepted for command-line compatibility (see fastquick_amd.h)
    else if (f == "--R") A.o.max_top2 = atoi(need(""));
    else if (f == "--q") A.o.trim_qual = atoi(need(""));
    else if (f == "--N") { A.o.mode |= 0x10; A.o.max_top2 = 0x7fffffff; }
    else if (f == "--L") A.o.mode |= 4;
    else if (f == "--max_isize") A.o.max_isize = atoi(need(""));
    else if (f == "--max_occ") A.o.max_occ = (uint32_t)atoi(need(""));
    else if (f == "--is_sw") A.o.is_sw = !A.o.is_sw;          // a bool flag on a default-1 int: it toggles (src/FASTQuick.cpp:278)
    else if (f == "--n_multi") A.o.n_multi = atoi(need(""));
    else if (f == "--N_multi") A.o.N_multi = atoi(need(""));
    else if (f == "--ap_prior") A.o.ap_prior = atof(need(""));
    else if (f == "--force_isize") A.o.force_isize = 1;
    else if (f == "--chunk_pairs") A.chunk_pairs = atoll(need(""));
    else if (f == "--device") A.device = atoi(need(""));
    else if (f == "--RG" || f == "--frac_samp" || f == "--fq_list" || f == "--bam_in" || f == "--cal_dup" || f == "--I") die(f + " is not supported by this build");
    else die("unknown option " + f);
  }
  if (A.o.fnr >= 1.0) { A.o.max_diff = (int)A.o.fnr; A.o.fnr = -1.0; }                  // src/FASTQuick.cpp:312-315
  if (A.opte > 0) { A.o.max_gape = A.opte; A.o.mode &= ~1; }                             // :316-319
  if (A.out_prefix == "Empty") die("--out_prefix is required");
  if (A.index_prefix == "Empty") die("--index_prefix is required");
  if (A.fq1.empty() || A.fq2.empty()) die("--fastq_1 and --fastq_2 are required (paired-end path)");
  if (!A.sam_out) die("BAM output (genome-coordinate translation + BGZF; SURVEY 8f.2) is not built yet: pass --sam_out");
  A.chunk_pairs = std::max<long long>(262144, A.chunk_pairs / 262144 * 262144);

  fq_index_t *ix = nullptr;
  const std::string pre = A.index_prefix + ".FASTQuick.fa";
  int rc = fq_index_load(pre.c_str(), A.device, &ix);
  if (rc) die("cannot load index " + pre + " onto HIP device " + std::to_string(A.device) + " (" + std::to_string(rc) + "); there is no CPU fallback");
  fq_ctx_t *ctx = nullptr;
  rc = fq_ctx_create(ix, &A.o, (int32_t)A.chunk_pairs, &ctx);
  if (rc) die("fq_ctx_create failed (" + std::to_string(rc) + "): option outside the supported range");

  {
    const int64_t n = fq_sam_header(ix, nullptr, 0);
    std::vector<char> h((size_t)n + 1);
    fq_sam_header(ix, h.data(), n + 1);
    fwrite(h.data(), 1, (size_t)n, stdout);
  }
  FastqReader r1(A.fq1), r2(A.fq2);
  std::string n1, s1, q1, n2, s2, q2;
  long long num_read = 0, filtered = 0, unmapped = 0, num_base = 0;
  std::vector<char> sam;
  bool more = true;
  while (more) {
    // one chunk = a whole number of reference batches (READ_BUFFER_SIZE pairs), so batch boundaries fall where the reference's do
    std::vector<std::string> names, seqs[2], quals[2];
    int stride = 16;
    while ((long long)names.size() < A.chunk_pairs) {
      const bool a = r1.next(n1, s1, q1), b = r2.next(n2, s2, q2);
      if (!a || !b) { more = false; break; }
      auto strip = [](std::string &nm) { const size_t t = nm.size(); if (t > 2 && nm[t - 2] == '/' && (nm[t - 1] == '1' || nm[t - 1] == '2')) nm.resize(t - 2); };
      strip(n1); strip(n2);
      if (names.size() % 262144 == 0 && n1 != n2)                                       // src/BwtMapper.cpp:2088-2092
        die("Abort, please make sure input pair of fastq files are in the same order!");
      names.push_back(n1);
      seqs[0].push_back(s1); quals[0].push_back(q1);
      seqs[1].push_back(s2); quals[1].push_back(q2);
      stride = std::max<int>(stride, (int)std::max(s1.size(), s2.size()));
    }
    const int n = (int)names.size();
    if (n == 0) break;
    stride = (stride + 15) & ~15;
    size_t name_stride = 8;
    for (auto &nm : names) name_stride = std::max(name_stride, nm.size() + 1);
    std::vector<uint8_t> seq((size_t)2 * n * stride, 0), qual((size_t)2 * n * stride, 0);
    std::vector<int32_t> len((size_t)2 * n);
    std::vector<char> nm((size_t)n * name_stride, 0);
    for (int e = 0; e < 2; ++e)
      for (int i = 0; i < n; ++i) {
        memcpy(&seq[((size_t)e * n + i) * stride], seqs[e][i].data(), seqs[e][i].size());
        memcpy(&qual[((size_t)e * n + i) * stride], quals[e][i].data(), quals[e][i].size());
        len[(size_t)e * n + i] = (int32_t)seqs[e][i].size();
      }
    for (int i = 0; i < n; ++i) memcpy(&nm[(size_t)i * name_stride], names[i].data(), names[i].size());
    fq_read_batch_t in = {n, stride, seq.data(), qual.data(), len.data(), nm.data(), (int32_t)name_stride};
    fq_result_batch_t res;
    rc = fq_align_batch(ctx, &in, &res);
    if (rc) die(std::string("fq_align_batch failed: ") + fq_ctx_last_error(ctx));
    const int64_t sz = fq_sam_format_last(ctx, nullptr, 0);
    sam.resize((size_t)sz + 1);
    fq_sam_format_last(ctx, sam.data(), sz + 1);
    fwrite(sam.data(), 1, (size_t)sz, stdout);
    num_read += 2LL * n; filtered += res.n_both_filtered; unmapped += res.n_both_unmapped; num_base += res.n_bases;
    fprintf(stderr, "NOTICE - %lld sequences are processed.\n", num_read);
  }
  fflush(stdout);
  notice("%lld sequences are loaded.", num_read);
  notice("%lld sequences are filtered.", filtered * 2);
  notice("%lld sequences are unmapped.", unmapped * 2);
  fq_stats_t st;
  fq_stats_get(ctx, &st);
  fprintf(stderr, "NOTICE - device time (ms): prep %.1f width %.1f gap %.1f sa %.1f sw %.1f refine %.1f ; host %.1f ; wall %.1f\n", st.kernel_ms[0],
          st.kernel_ms[1], st.kernel_ms[2], st.kernel_ms[3], st.kernel_ms[4], st.kernel_ms[5], st.host_ms_total, st.wall_ms_total);
  fq_ctx_destroy(ctx);
  fq_index_destroy(ix);
  return 0;
}
