// fq_index.h -- host-side index object (parsed files + device-staged tables)
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "fq_common.h"

struct FqContig {
  std::string name;
  int64_t offset;
  int32_t len;
};
struct FqHole {
  int64_t offset;
  int32_t len;
  char amb;
};

struct fq_index {
  std::string prefix;
  int64_t l_pac = 0;
  uint32_t seed = 11;
  std::vector<FqContig> contigs;
  std::vector<FqHole> holes;
  // host copies kept for the (host-side) consumers
  std::vector<uint8_t> pac;
  // device
  FqDevIndex dev{};
  FqDevContigs dev_contigs{};   // contig table and N holes in HBM (the consumers on the device: fq_emit.h)
  std::vector<void *> load_scratch;   // small device buffers of the load, freed with the index (hipFree waits for the device)
  void *load_state = nullptr;   // the load's device state (fqdev::State): given back with the index -- destroying it frees device memory, which waits for
                                // every stream of the device, and the load runs beside the first chunk's kernels
  void *d_blk[2] = {nullptr, nullptr};
  void *d_sa[2] = {nullptr, nullptr};
  void *d_pac = nullptr;
  void *d_bitmap = nullptr;   // 6 x 2^29 bytes, contiguous
  int device = 0;
};

// bns_coor_pac2real (libbwa/bntseq.c:268-302)
int fq_coor_pac2real(const fq_index *ix, int64_t pos, int len, int *seqid);
