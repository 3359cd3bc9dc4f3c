// fq_text_batch.h -- a batch whose reads are resident in HBM as FASTQ text: what the device front end (fq_frontend.cpp) produces and
// fq_align_text (fq_align.cpp) consumes.  Opaque in the C ABI (fq_text_batch_t).
#pragma once
#include <vector>

#include "fq_frontend.h"

struct fq_text_batch {
  int n_pairs = 0, single_end = 0, uniform_len = 0, max_len = 0, row_cap = 0, name_stride = 0, batch_pairs = 0, device = 0;
  const uint8_t *d_text[2] = {nullptr, nullptr};
  const FqTextRec *d_rec = nullptr;
  const uint64_t *d_head = nullptr;
  const uint16_t *d_hlen = nullptr;
  const char *d_names = nullptr;
  std::vector<char> first_names;   // [n_sub][2][name_stride]: the names of every reference batch's first pair (src/BwtMapper.cpp:2087-2092 compares them)
  int64_t text_bytes = 0, comp_bytes = 0, members = 0, refused = 0;
  int64_t pairs_behind = 0;        // an estimate of the pairs that follow this batch in the stream (0: unknown or none)
  int slot = -1;
};

