// fq_fastq_internal.h -- the hand-over from the device front end (fq_frontend.cpp) to the host reader (fq_fastq.cpp), inside the library.
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/fastquick_amd.h"

// Puts a freshly opened and configured BGZF reader (fq_fastq_open + fq_fastq_configure, nothing read yet) where a stream stands after
// `records_seen` records: the compressed file is read on from byte `file_offset` (a member boundary), in front of that come `n_text` bytes
// of inflated text that have not been tokenised yet, and the read slots hold what the records so far left in them:
// slot_names [n_slots][304] NUL padded, slot_bases [n_slots][96], slot_lens [n_slots], n_slots = 2 * batch_pairs (NULL: untouched slots).
int fq_fastq_resume(fq_fastq_t *r, int64_t file_offset, const uint8_t *text, size_t n_text, int64_t records_seen,
                    const uint8_t *slot_names, const uint8_t *slot_bases, const uint16_t *slot_lens, int shorter_after_longer);
